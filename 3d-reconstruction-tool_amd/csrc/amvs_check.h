// amvs_check.h -- index-checked build (-DAMVS_CHECK_INDICES; tools/build_variant.sh check -DAMVS_CHECK_INDICES).
//
// The GPU-side substitute for an address sanitizer, which this pool does not offer for device code: every
// data-dependent global-memory index of the sweep, plane-sweep, extended, fusion and neighbour-search kernels is
// compared with the extent of the buffer it addresses BEFORE the access.  A violation is counted, the first one
// is recorded (translation unit, source line, index, extent) in a device-side report, and the access is
// redirected to a safe index, so that the run completes and names the place instead of faulting (or, worse,
// silently reading a neighbouring allocation -- the out-of-bounds read of rounds 1-2 did exactly that for two
// rounds).  The C ABI reports it: every synchronising entry point returns AMVS_EINDEX, amvs_index_check() returns
// the record.  In the shipped build the macros are the identity and cost nothing.
//
// A translation unit defines AMVS_TU_ID (a small integer, see amvs_capi.hip: index_report) before including this
// header and places AMVS_CHECK_TU(name) once at file scope (outside any namespace).
#pragma once
#include <hip/hip_runtime.h>

#ifdef AMVS_CHECK_INDICES
#ifndef AMVS_TU_ID
#error "define AMVS_TU_ID before including amvs_check.h"
#endif
namespace amvs {
namespace chk {
struct Report { unsigned long long count, where, index, extent; };   // where = AMVS_TU_ID << 32 | line
static __device__ Report g_report;
__device__ __forceinline__ long long check(long long i, long long lo, long long hi, int line)
{
    if (i < lo || i >= hi) {
        if (atomicAdd(&g_report.count, 1ull) == 0ull) {
            g_report.where = ((unsigned long long)(AMVS_TU_ID) << 32) | (unsigned)line;
            g_report.index = (unsigned long long)i;
            g_report.extent = (unsigned long long)hi;
        }
        return lo;
    }
    return i;
}
}  // namespace chk
}  // namespace amvs
// index i of a buffer of n elements / of the index range [lo, hi)
#define AMVS_IDX(i, n) ((decltype(i))amvs::chk::check((long long)(i), 0ll, (long long)(n), __LINE__))
#define AMVS_IDX_LOHI(i, lo, hi) ((decltype(i))amvs::chk::check((long long)(i), (long long)(lo), (long long)(hi), __LINE__))
#define AMVS_CHECK_TU(name)                                                                              \
    namespace amvs {                                                                                     \
    void check_fetch_##name(unsigned long long out[4], bool reset)                                       \
    {                                                                                                    \
        chk::Report r{};                                                                                 \
        (void)hipMemcpyFromSymbol(&r, HIP_SYMBOL(chk::g_report), sizeof(r));                             \
        out[0] = r.count; out[1] = r.where; out[2] = r.index; out[3] = r.extent;                         \
        if (reset) { chk::Report z{}; (void)hipMemcpyToSymbol(HIP_SYMBOL(chk::g_report), &z, sizeof(z)); } \
    }                                                                                                    \
    }
#else
#define AMVS_IDX(i, n) (i)
#define AMVS_IDX_LOHI(i, lo, hi) (i)
#define AMVS_CHECK_TU(name)
#endif
