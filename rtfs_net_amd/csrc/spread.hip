// Spread accumulators for the parameter-gradient reductions of the training step.
//
// Measured on MI355X: fp32 atomics that land on the same 128-byte line are served one REQUEST at a time at ~27 ns each
// (device-scope read-modify-write), independent of how many lanes the request carries.  A reduction kernel with thousands of
// workgroups that all add into the same 64..1088 floats is therefore bound by that chain (4064 workgroups x 4 requests per
// line = 443 us for a kernel that streams its 530 MB in 70 us).  Cure: every workgroup adds its partials - one coalesced
// request per line - into copy (workgroup % kSpread) of a library-owned scratch buffer; a small finish kernel sums the
// copies into the caller's accumulator (single writer, plain add) and re-zeroes the scratch for the next user.
//
// One scratch buffer per device AND LANE; the users of a lane are serialised by stream order.  Lane 0 is the stream the training step runs on;
// lane 1 (rtfs_spread_lane) belongs to the side stream that carries the step's weight-gradient launches underneath the bandwidth-bound adjoint
// chain (models/hip_train.py): two producers that run concurrently must not share a scratch region.
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

// All pending regions of a deferred section in one launch.  Two regions may name the same destination (the R RTFS blocks share their weights, and a
// block's adjoint adds into one gradient buffer per parameter): the sums are ADDED with fp32 atomics.
constexpr int kMaxPend = 24;
constexpr int kMaxRegion = 12352;  // largest region any producer asks for (attention QKV norm adjoint: 4 x 1024 + 2 x 4096 + 12, 32-float aligned)
struct FlushArgs {
    int n;
    int off[kMaxPend], total[kMaxPend];
    SpreadOut o[kMaxPend];
};
__global__ __launch_bounds__(256) void spread_flush_kernel(float* __restrict__ scr, FlushArgs fa) {
    int e = blockIdx.x * 256 + threadIdx.x;
    int p = 0;
    while (p < fa.n && e >= fa.total[p]) e -= fa.total[p], ++p;
    if (p >= fa.n) return;
    const SpreadOut& o = fa.o[p];
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
        float* q = scr + (size_t)c * kSpreadCap + fa.off[p] + e;
        s += *q;
        *q = 0.f;
    }
    if (o.dst[j]) atomicAdd(o.dst[j] + i, s);
}

namespace {
struct DevState {
    float* scr = nullptr;
    int cursor = 0;        // floats; start of the next free region (0 outside deferred sections)
    bool deferred = false;
    FlushArgs pend{};      // pend.n regions wait for the flush
};
constexpr int kLanes = 2;
DevState g_dev[16][kLanes];
thread_local int g_lane = 0;  // the lane the CALLING host thread currently issues for (per thread: one host thread per device may drive its own step)
std::mutex g_mu;

DevState* dev_state() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    DevState& d = g_dev[dev][g_lane];
    if (!d.scr) {
        float* p = nullptr;
        const size_t bytes = (size_t)kSpread * kSpreadCap * sizeof(float);
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        // the fill runs on the NULL stream and returns before it has run; the first producer may be on a non-blocking stream (the training step's
        // weight-gradient side stream, torch.cuda.Stream) that the null stream does not order: without the wait its first partial sums can land before
        // the fill and be zeroed (seen once in ~25 runs of the GPU suite: one parameter gradient of the first training step of the process missing)
        if (hipMemset(p, 0, bytes) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return nullptr;
        d.scr = p;
    }
    return &d;
}

int flush_locked(DevState& d, hipStream_t st) {
    if (d.pend.n > 0) {
        int total = 0;
        for (int p = 0; p < d.pend.n; ++p) total += d.pend.total[p];
        hipLaunchKernelGGL(spread_flush_kernel, dim3((total + 255) / 256), dim3(256), 0, st, d.scr, d.pend);
        d.pend.n = 0;
        RTFS_LAUNCH_CHECK();
    }
    d.cursor = 0;
    return RTFS_OK;
}
}  // namespace

float* spread_scratch() {
    std::lock_guard<std::mutex> lk(g_mu);
    DevState* d = dev_state();
    return d ? d->scr + d->cursor : nullptr;
}

int spread_finish(float* scr, const SpreadOut& o, hipStream_t st, bool consumed_now) {
    int total = 0;
    for (int j = 0; j < kSpreadSlots; ++j) total += (o.n[j] + 31) & ~31;
    if (total <= 0) return RTFS_OK;
    if (total > kSpreadCap) return RTFS_EINVAL;
    std::lock_guard<std::mutex> lk(g_mu);
    DevState* d = dev_state();
    if (!d) return RTFS_ELAUNCH;
    if (!d->deferred) {  // immediate mode: the region starts at the scratch base, any size up to a copy's capacity
        hipLaunchKernelGGL(spread_finish_kernel, dim3((total + 255) / 256), dim3(256), 0, st, scr, o);
        RTFS_LAUNCH_CHECK();
        return RTFS_OK;
    }
    // deferred sections pack regions behind one another and keep kMaxRegion floats free behind the cursor: a producer that outgrows it must
    // raise the constant (the request fails here, loudly, instead of running past the copy)
    if (total > kMaxRegion) return RTFS_EINVAL;
    const int p = d->pend.n++;
    d->pend.off[p] = (int)(scr - d->scr), d->pend.total[p] = total, d->pend.o[p] = o;
    d->cursor = d->pend.off[p] + total;
    if (consumed_now || d->pend.n == kMaxPend || d->cursor + kMaxRegion > kSpreadCap) return flush_locked(*d, st);
    return RTFS_OK;
}

}  // namespace rtfs

extern "C" {
// Deferred mode of the parameter-gradient reducers (one stream per device, as for the scratch itself): between rtfs_spread_defer(1) and
// rtfs_spread_defer(0) the per-kernel finish launches (219 per training step, ~5 us each + a launch gap) are recorded and applied by one launch per
// ~20 producers; the destinations are complete after rtfs_spread_flush / rtfs_spread_defer(0), both stream-ordered.  Reference: the accumulation of
// parameter gradients over the shared RTFS block, autograd of separators/tdanet.py:106-133.
int rtfs_spread_defer(int on, void* stream) {
    std::lock_guard<std::mutex> lk(rtfs::g_mu);
    rtfs::DevState* d = rtfs::dev_state();
    if (!d) return RTFS_ELAUNCH;
    const int rc = rtfs::flush_locked(*d, (hipStream_t)stream);
    d->deferred = on != 0;
    return rc;
}
// Select the scratch lane (0 or 1) that the following reducer launches, rtfs_spread_defer and rtfs_spread_flush of THIS host thread use: lane 1 for
// launches issued on a second stream that may run concurrently with lane 0's.  Deferred sections are per (device, lane); the selection is per host
// thread (thread_local), so a second host thread driving another device is not redirected by this thread's lane window.  Two host threads that
// drive the SAME device concurrently are outside the contract (as for the scratch itself: its users are serialised by stream order).
int rtfs_spread_lane(int lane) {
    if (lane < 0 || lane >= rtfs::kLanes) return RTFS_EINVAL;
    std::lock_guard<std::mutex> lk(rtfs::g_mu);
    rtfs::g_lane = lane;
    return RTFS_OK;
}
int rtfs_spread_flush(void* stream) {
    std::lock_guard<std::mutex> lk(rtfs::g_mu);
    rtfs::DevState* d = rtfs::dev_state();
    if (!d) return RTFS_ELAUNCH;
    return rtfs::flush_locked(*d, (hipStream_t)stream);
}
}
