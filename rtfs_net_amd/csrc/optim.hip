// Optimizer step of the training step (BASELINE config 3 / 4: AdamW(lr 1e-3, weight_decay 0.1) behind gradient clipping at 5.0 - the reference's
// config yaml:117-120 and train.py:135-146 hand both to PyTorch / Lightning).  torch.optim.AdamW (foreach) + clip_grad_norm_ over RTFS-Net's 403
// parameter tensors (740 k numbers) is ~15 multi-tensor launches and a host-bound 2.0 ms at the end of every step with the GPU idle (round 5,
// tools/opt_step_ab.py; fused=True: 1.26 ms).  Here: TWO launches over a chunk map of all tensors -
//   rtfs_grad_sqnorm     sum of squares of every gradient element into one double (fp64 atomics, one per workgroup);
//   rtfs_adamw_clip_step clip coefficient min(1, max_norm / (norm + 1e-6)) from that double ON THE DEVICE (no host round trip), the gradients scaled
//                        in place (as clip_grad_norm_ leaves them), then torch.optim.AdamW's update in its own operation order:
//                        p *= 1 - lr wd;  m = lerp(m, g, 1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps).
// The pointer tables (parameter, gradient, exp_avg, exp_avg_sq per tensor) and the chunk map live in device memory; the host side
// (rtfs_net_amd/optim.py, a torch.optim.Optimizer whose state_dict is interchangeable with torch.optim.AdamW's) refreshes the gradient pointers per step.
#include "common.h"

namespace rtfs {

constexpr int kOptChunk = 1024;  // elements per workgroup

struct OptTables {
    const long long* p;     // [n] parameter pointers
    const long long* g;     // [n] gradient pointers
    const long long* m;     // [n] exp_avg
    const long long* v;     // [n] exp_avg_sq
    const long long* size;  // [n] elements per tensor
    const int* chunk;       // [nchunks][2] = (tensor, first element)
};

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(OptTables t, double* __restrict__ out) {
    __shared__ float red[4];
    const int ti = t.chunk[2 * blockIdx.x], start = t.chunk[2 * blockIdx.x + 1];
    const float* g = reinterpret_cast<const float*>(t.g[ti]);
    const long long n = t.size[ti];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kOptChunk / 256; ++k) {
        const long long i = (long long)start + threadIdx.x + 256 * k;
        const float x = i < n ? g[i] : 0.f;
        s = fmaf(x, x, s);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3]);
}

// decay = 1 - lr wd, w1 = 1 - beta1, w2 = 1 - beta2, step_size = lr / bias_correction1: formed in double on the host as torch.optim.AdamW forms its Python
// scalars (1 - beta2 in fp32 is 1.3e-5 off the fp32 value of the double 0.001: exp_avg_sq would drift from torch's by that much)
__global__ __launch_bounds__(256) void adamw_clip_kernel(OptTables t, const double* __restrict__ sqnorm, float max_norm, float decay, float w1, float beta2,
                                                         float w2, float eps, float step_size, float bc2_sqrt) {
    const int ti = t.chunk[2 * blockIdx.x], start = t.chunk[2 * blockIdx.x + 1];
    float* p = reinterpret_cast<float*>(t.p[ti]);
    float* g = reinterpret_cast<float*>(t.g[ti]);
    float* m = reinterpret_cast<float*>(t.m[ti]);
    float* v = reinterpret_cast<float*>(t.v[ti]);
    const long long n = t.size[ti];
    float coef = 1.f;
    if (max_norm > 0.f) {  // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
        const float total = (float)sqrt(*sqnorm);
        coef = fminf(max_norm / (total + 1e-6f), 1.f);
    }
#pragma unroll
    for (int k = 0; k < kOptChunk / 256; ++k) {
        const long long i = (long long)start + threadIdx.x + 256 * k;
        if (i < n) {
            const float gi = g[i] * coef;
            g[i] = gi;
            const float pi = p[i] * decay;
            const float mi = fmaf(w1, gi - m[i], m[i]);            // exp_avg.lerp_(grad, 1 - beta1)
            const float vi = fmaf(w2 * gi, gi, v[i] * beta2);      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
            m[i] = mi, v[i] = vi;
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            p[i] = pi - step_size * (mi / denom);                  // param.addcdiv_(exp_avg, denom, value = -step_size)
        }
    }
}

}  // namespace rtfs

using namespace rtfs;

extern "C" {

// tables (all DEVICE memory): params / grads / exp_avg / exp_avg_sq = [n_tensors] int64 device pointers to contiguous fp32 tensors, sizes = [n_tensors] int64
// element counts, chunks = [n_chunks][2] int32 (tensor index, first element; kOptChunk = 1024 elements per chunk).  sqnorm: one double (zeroed here).
int rtfs_grad_sqnorm(const long long* params, const long long* grads, const long long* exp_avg, const long long* exp_avg_sq, const long long* sizes,
                     const int* chunks, int n_chunks, double* sqnorm, void* stream) {
    if (n_chunks <= 0) return RTFS_EINVAL;
    if (hipMemsetAsync(sqnorm, 0, sizeof(double), (hipStream_t)stream) != hipSuccess) return RTFS_ELAUNCH;
    const OptTables t{params, grads, exp_avg, exp_avg_sq, sizes, chunks};
    hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, t, sqnorm);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

// max_norm <= 0: no clipping (sqnorm is not read).  bias_correction1 = 1 - beta1^step, bias_correction2_sqrt = sqrt(1 - beta2^step) of THIS step.
// The hyper-parameters are doubles: torch.optim.AdamW holds them as Python floats and forms 1 - beta, lr / bias_correction1, ... in double before they
// become fp32 kernel scalars (1 - 0.999f is 1.3e-5 away from fp32(0.001)).
int rtfs_adamw_clip_step(const long long* params, const long long* grads, const long long* exp_avg, const long long* exp_avg_sq, const long long* sizes,
                         const int* chunks, int n_chunks, const double* sqnorm, double max_norm, double lr, double beta1, double beta2, double eps,
                         double weight_decay, double bias_correction1, double bias_correction2_sqrt, void* stream) {
    if (n_chunks <= 0 || !(bias_correction1 > 0.0) || !(bias_correction2_sqrt > 0.0)) return RTFS_EINVAL;
    const OptTables t{params, grads, exp_avg, exp_avg_sq, sizes, chunks};
    hipLaunchKernelGGL(adamw_clip_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, t, sqnorm, (float)max_norm, (float)(1.0 - lr * weight_decay),
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)(lr / bias_correction1), (float)bias_correction2_sqrt);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

}  // extern "C"
