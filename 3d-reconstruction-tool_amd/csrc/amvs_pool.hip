// amvs_pool.hip -- a process-wide cache of device blocks for the SHORT-LIVED buffers of the post-steps (the
// neighbour statistic of the stereo outlier filter, the scratch arrays of fusion / filter / voxel grid): hipMalloc +
// hipFree cost 0.1-0.2 ms a pair and a post-step takes a dozen pairs per call (measured, round 4: 2.7 of the 12.6 ms
// of amvs_knn_mean_distance on a 500 000-point cloud).  pool_malloc serves a request from a block released earlier
// when one of at most twice the size is cached, pool_free keeps the block instead of returning it to the driver,
// pool_trim (amvs_destroy) returns everything.
//
// Ordering: a cached block may be handed out again while work that used it is still queued.  That is safe here
// because every user enqueues on its context's one stream (in order) and every post-step synchronises that stream
// before it returns to the caller -- a block never changes streams with work in flight.
#include "amvs_pool.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace amvs {

namespace {

struct Live { size_t size; int device; };

std::mutex g_mutex;
std::unordered_map<void *, Live> g_live;                       // blocks handed out
std::map<std::pair<int, size_t>, std::vector<void *>> g_free;  // (device, size) -> cached blocks
size_t g_cached_bytes = 0;
constexpr size_t POOL_MAX_CACHED = size_t(4) << 30;            // beyond this, blocks go back to the driver

void trim_locked()
{
    for (auto &kv : g_free)
        for (void *p : kv.second) (void)hipFree(p);
    g_free.clear();
    g_cached_bytes = 0;
}

}  // namespace

hipError_t pool_malloc(void **out, size_t bytes)
{
    *out = nullptr;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const size_t size = (std::max<size_t>(bytes, 1) + 255) & ~size_t(255);
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_free.lower_bound({dev, size});
    if (it != g_free.end() && it->first.first == dev && it->first.second <= 2 * size + 4096) {
        void *p = it->second.back();
        it->second.pop_back();
        g_cached_bytes -= it->first.second;
        g_live[p] = Live{it->first.second, dev};
        if (it->second.empty()) g_free.erase(it);
        *out = p;
        return hipSuccess;
    }
    void *p = nullptr;
    e = hipMalloc(&p, size);
    if (e != hipSuccess) {                     // make room: give the cache back, try once more
        (void)hipGetLastError();
        trim_locked();
        e = hipMalloc(&p, size);
        if (e != hipSuccess) return e;
    }
    g_live[p] = Live{size, dev};
    *out = p;
    return hipSuccess;
}

void pool_free(void *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_live.find(p);
    if (it == g_live.end()) { (void)hipFree(p); return; }      // not ours
    const Live b = it->second;
    g_live.erase(it);
    if (g_cached_bytes + b.size > POOL_MAX_CACHED) { (void)hipFree(p); return; }
    g_free[{b.device, b.size}].push_back(p);
    g_cached_bytes += b.size;
}

void pool_trim()
{
    std::lock_guard<std::mutex> lock(g_mutex);
    trim_locked();
}

}  // namespace amvs
