// Spread accumulators for the parameter-gradient reductions of the training step.
//
// Measured on MI355X: fp32 atomics that land on the same 128-byte line are served one REQUEST at a time at ~27 ns each
// (device-scope read-modify-write), independent of how many lanes the request carries.  A reduction kernel with thousands of
// workgroups that all add into the same 64..1088 floats is therefore bound by that chain (4064 workgroups x 4 requests per
// line = 443 us for a kernel that streams its 530 MB in 70 us).  Cure: every workgroup adds its partials - one coalesced
// request per line - into copy (workgroup % kSpread) of a library-owned scratch buffer; a small finish kernel sums the
// copies into the caller's accumulator (single writer, plain add) and re-zeroes the scratch for the next user.
//
// One scratch buffer per device; users are serialised by stream order (every training-step launch is on one stream).
#include "common.h"

#include <mutex>

namespace rtfs {

__global__ __launch_bounds__(256) void spread_finish_kernel(float* __restrict__ scr, SpreadOut o) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    int j = 0, base = 0;
#pragma unroll
    for (int k = 0; k < kSpreadSlots - 1; ++k) {
        const int span = (o.n[j] + 31) & ~31;
        if (j < kSpreadSlots - 1 && e >= base + span) base += span, ++j;
    }
    const int i = e - base;
    if (i >= o.n[j]) return;
    float s = 0.f;
#pragma unroll 8
    for (int c = 0; c < kSpread; ++c) {
        float* p = scr + (size_t)c * kSpreadCap + e;
        s += *p;
        *p = 0.f;
    }
    if (o.dst[j]) o.dst[j][i] += s;  // a region without destination (optional output the kernel still adds into) is only cleared
}

static float* g_scratch[16] = {};
static std::mutex g_mu;

float* spread_scratch() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_scratch[dev]) {
        float* p = nullptr;
        const size_t bytes = (size_t)kSpread * kSpreadCap * sizeof(float);
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, bytes) != hipSuccess) return nullptr;
        g_scratch[dev] = p;
    }
    return g_scratch[dev];
}

int spread_finish(float* scr, const SpreadOut& o, hipStream_t st) {
    int total = 0;
    for (int j = 0; j < kSpreadSlots; ++j) total += (o.n[j] + 31) & ~31;
    if (total <= 0) return RTFS_OK;
    if (total > kSpreadCap) return RTFS_EINVAL;
    hipLaunchKernelGGL(spread_finish_kernel, dim3((total + 255) / 256), dim3(256), 0, st, scr, o);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

}  // namespace rtfs
