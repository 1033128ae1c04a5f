// amvs_pool.h -- cache of device blocks for the post-steps' short-lived buffers (amvs_pool.hip has the ordering rule)
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace amvs {

hipError_t pool_malloc(void **out, size_t bytes);          // like hipMalloc; may return a block released earlier
template <class T> inline hipError_t pool_malloc(T **out, size_t bytes) { return pool_malloc((void **)out, bytes); }
void pool_free(void *p);                                   // keeps the block for the next request (or hipFree)
void pool_trim();                                          // amvs_destroy: every cached block back to the driver

}  // namespace amvs
