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
        // a NaN norm goes through to every gradient and parameter as torch.clamp does (fminf alone would return 1 and hide a diverged step from NaN-skip logic)
        coef = (total == total) ? fminf(max_norm / (total + 1e-6f), 1.f) : total;
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

// ---- BatchNorm2d of the CAF cell's key / value embeddings in the training step (fusion.py:249-253: depth-wise 1x1 conv u = dw x -> BatchNorm2d) -------------
// The per-channel arithmetic between the statistics kernel and the cell (forward) and between the reduction and the apply kernel (adjoint) was ~47 + ~71 tiny
// torch launches on the step's critical path (0.69 ms per step, tools/train_glue_clusters.py).  One launch each, one thread per channel, fp64 where the torch
// code was fp64 (the statistics).  `glob`: sums over ALL ranks (== `loc` without SyncBatchNorm); *n: positions behind `glob`.
struct CafTag {
    const float *dw, *g, *be;      // depth-wise weight [C], BatchNorm weight / bias [C]
    float *run_mean, *run_var;     // running statistics (updated in place)
    long long* batches;            // num_batches_tracked
    float *inv, *mean_u, *s, *b;   // outputs [C]: 1 / sqrt(var_u + eps), mean of u, folded scale dw g inv, folded shift be - mean_u g inv
};

__global__ __launch_bounds__(256) void caf_bn_prepare_kernel(const double* __restrict__ glob, const double* __restrict__ loc, const double* __restrict__ n_ptr,
                                                             CafTag k, CafTag v, float momentum, float eps, float* __restrict__ mean_x,
                                                             float* __restrict__ var_x, float* __restrict__ lsum, float* __restrict__ lcov) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= kC) return;
    const double n = *n_ptr;
    const double mean = glob[c] / n;
    double var = glob[kC + c] / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float mx = (float)mean, vx = (float)var;
    mean_x[c] = mx, var_x[c] = vx;
    lsum[c] = (float)loc[c];
    lcov[c] = (float)(loc[kC + c] - mean * loc[c]);  // sum_local (x - mean) x, differenced in fp64
    const float unbias = (float)(n / (n - 1.0 > 1.0 ? n - 1.0 : 1.0));
    const CafTag* tags[2] = {&k, &v};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const CafTag& t = *tags[j];
        const float dw = t.dw[c], g = t.g[c];
        const float mean_u = dw * mx, var_u = dw * dw * vx;
        t.run_mean[c] = t.run_mean[c] * (1.f - momentum) + momentum * mean_u;
        t.run_var[c] = t.run_var[c] * (1.f - momentum) + momentum * var_u * unbias;
        const float inv = 1.0f / sqrtf(var_u + eps);
        t.inv[c] = inv, t.mean_u[c] = mean_u;
        t.s[c] = dw * g * inv;
        t.b[c] = t.be[c] - mean_u * g * inv;
        if (c == 0) *t.batches += 1;
    }
}

struct CafAdj {
    const float *dw, *g, *inv;  // [C]
    float *d_dw, *d_g, *d_be;   // parameter gradients of this rank [C] (written)
};

// Rloc / Rglob: [4][C] = (A, Bx) of key, (A, Bx) of value - A = sum dy, Bx = sum dy x; this rank's / all ranks'.  coef: [6][C] = (c1, c2, c3) per tag for
// rtfs_caf_bwd_apply (dx = dy c1 + c2 + c3 x).  The arithmetic of models/hip_train.py caf_bn_adjoint, unchanged.
__global__ __launch_bounds__(256) void caf_bn_adjoint_kernel(const float* __restrict__ Rloc, const float* __restrict__ Rglob, const double* __restrict__ n_ptr,
                                                             const float* __restrict__ mean_x, const float* __restrict__ lsum,
                                                             const float* __restrict__ lcov, CafAdj k, CafAdj v, float* __restrict__ coef) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= kC) return;
    const float n = (float)*n_ptr, mx = mean_x[c];
    const CafAdj* tags[2] = {&k, &v};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const CafAdj& t = *tags[j];
        const float dw = t.dw[c], gm = t.g[c], inv = t.inv[c];
        const float A = Rloc[(2 * j) * kC + c], Bx = Rloc[(2 * j + 1) * kC + c];
        const float Ag = Rglob[(2 * j) * kC + c], Bxg = Rglob[(2 * j + 1) * kC + c];
        const float Qg = dw * inv * (Bxg - mx * Ag), Ql = dw * inv * (Bx - mx * A);
        const float c1 = dw * gm * inv;
        const float c3 = -c1 * (Qg / n) * (dw * inv);
        const float c2 = -c1 * Ag / n - c3 * mx;
        coef[(3 * j) * kC + c] = c1, coef[(3 * j + 1) * kC + c] = c2, coef[(3 * j + 2) * kC + c] = c3;
        t.d_dw[c] = gm * inv * (Bx - (Ag / n) * lsum[c] - (Qg / n) * dw * inv * lcov[c]);
        t.d_g[c] = Ql;
        t.d_be[c] = A;
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

// sums_glob / sums_loc: [2][C] fp64 (sum x, sum x^2) over all ranks / this rank (the same pointer without SyncBatchNorm); n: ONE double on the device, the
// number of positions behind sums_glob.  Per tag (key, value): dw, g, be [C] in; run_mean, run_var [C] and num_batches_tracked (int64) updated in place;
// inv, mean_u, s, b [C] out.  mean_x, var_x, lsum, lcov [C]: kept for rtfs_caf_bn_adjoint.
int rtfs_caf_bn_prepare(const double* sums_glob, const double* sums_loc, const double* n, const float* k_dw, const float* k_g, const float* k_be,
                        float* k_run_mean, float* k_run_var, long long* k_batches, float* k_inv, float* k_mean_u, float* k_s, float* k_b, const float* v_dw,
                        const float* v_g, const float* v_be, float* v_run_mean, float* v_run_var, long long* v_batches, float* v_inv, float* v_mean_u, float* v_s,
                        float* v_b, float momentum, float eps, float* mean_x, float* var_x, float* lsum, float* lcov, void* stream) {
    const CafTag k{k_dw, k_g, k_be, k_run_mean, k_run_var, k_batches, k_inv, k_mean_u, k_s, k_b};
    const CafTag v{v_dw, v_g, v_be, v_run_mean, v_run_var, v_batches, v_inv, v_mean_u, v_s, v_b};
    hipLaunchKernelGGL(caf_bn_prepare_kernel, dim3((kC + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums_glob, sums_loc, n, k, v, momentum, eps, mean_x,
                       var_x, lsum, lcov);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_caf_bn_adjoint(const float* R_loc, const float* R_glob, const double* n, const float* mean_x, const float* lsum, const float* lcov, const float* k_dw,
                        const float* k_g, const float* k_inv, float* k_d_dw, float* k_d_g, float* k_d_be, const float* v_dw, const float* v_g,
                        const float* v_inv, float* v_d_dw, float* v_d_g, float* v_d_be, float* coef, void* stream) {
    const CafAdj k{k_dw, k_g, k_inv, k_d_dw, k_d_g, k_d_be};
    const CafAdj v{v_dw, v_g, v_inv, v_d_dw, v_d_g, v_d_be};
    hipLaunchKernelGGL(caf_bn_adjoint_kernel, dim3((kC + 255) / 256), dim3(256), 0, (hipStream_t)stream, R_loc, R_glob, n, mean_x, lsum, lcov, k, v, coef);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

}  // extern "C"
