// Loss head behind the separation path: PairwiseNegSDR (src/losses/matrix.py:13-53) for
// sdr_type snr | sisdr | sdsdr with its exact adjoint.  Every quantity of the reference's formula is a function of five
// sums per (utterance b, estimate i, target j) pair - S_e, S_t, S_ee, S_tt, S_et and S_dd = sum (e-t)^2 (kept so that
// the snr / sdsdr noise energy is not a difference of large numbers) - so the waveforms are read once:
//   rtfs_neg_sdr_sums    the six sums, accumulated in fp64 from the first product on (one fp64 atomic per workgroup and sum)
//   rtfs_neg_sdr_finish  pair loss  -10 log10(num / (den + EPS) + EPS)  and the coefficients of its gradient
//   rtfs_neg_sdr_grad    d(est_i)[t] = sum_j G[b][i][j] (ce (e_i[t] - mean e_i) + ct (t_j[t] - mean t_j))
// With e' = e - mean(e), t' = t - mean(t) (zero_mean), a = S_e't' / (S_t't' + EPS):
//   snr:   num = S_t't',      den = S_e'e' - 2 S_e't' + S_t't'
//   sisdr: num = a^2 S_t't',  den = S_e'e' - 2 a S_e't' + a^2 S_t't'
//   sdsdr: num = a^2 S_t't',  den = S_e'e' - 2 S_e't' + S_t't'
#include "common.h"

namespace rtfs {

constexpr double kLossEps = 1e-8;

// est, tgt: [B][n][T].  sums: [B][n][n][6] doubles (zeroed by the caller).  grid (chunks, B*n*n)
__global__ __launch_bounds__(256) void neg_sdr_sums_kernel(const float* __restrict__ est, const float* __restrict__ tgt, double* __restrict__ sums, int n,
                                                           int T, int per_wg) {
    __shared__ double red[6][4];
    const int pair = blockIdx.y, b = pair / (n * n), i = (pair / n) % n, j = pair % n;
    const float* e = est + ((size_t)b * n + i) * T;
    const float* t = tgt + ((size_t)b * n + j) * T;
    const int t0 = blockIdx.x * per_wg, t1 = min(T, t0 + per_wg);
    double s[6] = {0, 0, 0, 0, 0, 0};  // fp64 throughout: sisdr's noise energy S_ee - 2a S_et + a^2 S_tt cancels to 1e-6 of its terms at 60 dB
    for (int k = t0 + threadIdx.x; k < t1; k += 256) {
        const double ev = e[k], tv = t[k], dv = ev - tv;
        s[0] += ev, s[1] += tv, s[2] = fma(ev, ev, s[2]), s[3] = fma(tv, tv, s[3]), s[4] = fma(ev, tv, s[4]), s[5] = fma(dv, dv, s[5]);
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        double v = s[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int q = threadIdx.x;
        atomicAdd(sums + (size_t)pair * 6 + q, red[q][0] + red[q][1] + red[q][2] + red[q][3]);
    }
}

// kind: 0 snr, 1 sisdr, 2 sdsdr.  pw: [B][n][n] pair losses; coef: [B][n][n][4] = (ce, ct, mean e, mean t) of d(pw)/d(est_i)
__global__ void neg_sdr_finish_kernel(const double* __restrict__ sums, int kind, int zero_mean, int take_log, float* __restrict__ pw,
                                      float* __restrict__ coef, int npairs, int T) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    const double* s = sums + (size_t)p * 6;
    const double me = zero_mean ? s[0] / T : 0.0, mt = zero_mean ? s[1] / T : 0.0;
    const double See = s[2] - T * me * me, Stt = s[3] - T * mt * mt, Set = s[4] - T * me * mt;
    const double Sdd = s[5] - T * (me - mt) * (me - mt);  // sum (e' - t')^2
    const double a = Set / (Stt + kLossEps);
    double num, den, dnum_e = 0.0, dnum_t = 0.0, dden_e, dden_t;  // d/d(e'_k) = x_e * e'_k + x_t * t'_k
    if (kind == 0) {
        num = Stt, den = Sdd;
        dden_e = 2.0, dden_t = -2.0;
    } else {
        const double da_t = 1.0 / (Stt + kLossEps);  // d a / d e'_k = t'_k * da_t
        num = a * a * Stt;
        dnum_t = 2.0 * a * Stt * da_t;
        if (kind == 1) {
            den = See - 2.0 * a * Set + a * a * Stt;
            dden_e = 2.0, dden_t = -2.0 * a + (-2.0 * Set + 2.0 * a * Stt) * da_t;
        } else {
            den = Sdd;
            dden_e = 2.0, dden_t = -2.0;
        }
    }
    const double r = num / (den + kLossEps);
    double val, dr;  // loss = -val; d loss / d r = -dr
    if (take_log) {
        val = 10.0 * log10(r + kLossEps);
        dr = 10.0 / (log(10.0) * (r + kLossEps));
    } else {
        val = r, dr = 1.0;
    }
    pw[p] = (float)(-val);
    // d r = d num / (den+EPS) - num d den / (den+EPS)^2
    const double inv = 1.0 / (den + kLossEps);
    coef[(size_t)p * 4 + 0] = (float)(-dr * (dnum_e * inv - num * dden_e * inv * inv));
    coef[(size_t)p * 4 + 1] = (float)(-dr * (dnum_t * inv - num * dden_t * inv * inv));
    coef[(size_t)p * 4 + 2] = (float)me;
    coef[(size_t)p * 4 + 3] = (float)mt;
}

// dest[b][i][t] = sum_j G[b][i][j] * (ce * (e_i[t] - me) + ct * (t_j[t] - mt)).  grid (ceil(T/1024), B*n)
__global__ __launch_bounds__(256) void neg_sdr_grad_kernel(const float* __restrict__ est, const float* __restrict__ tgt, const float* __restrict__ coef,
                                                           const float* __restrict__ G, float* __restrict__ dest, int n, int T) {
    const int bi = blockIdx.y, b = bi / n;
    const float* e = est + (size_t)bi * T;
    for (int k = blockIdx.x * 1024 + threadIdx.x; k < min(T, (int)(blockIdx.x + 1) * 1024); k += 256) {
        const float ev = e[k];
        float acc = 0.f;
        for (int j = 0; j < n; ++j) {
            const float* c = coef + ((size_t)bi * n + j) * 4;
            const float g = G[(size_t)bi * n + j];
            acc += g * (c[0] * (ev - c[2]) + c[1] * (tgt[((size_t)b * n + j) * T + k] - c[3]));
        }
        dest[(size_t)bi * T + k] = acc;
    }
}

}  // namespace rtfs

using namespace rtfs;

extern "C" {

int rtfs_neg_sdr_sums(const float* est, const float* tgt, double* sums, int B, int n_src, int T, void* stream) {
    if (B <= 0 || n_src <= 0 || T <= 0) return RTFS_EINVAL;
    const int per = 8192;
    hipLaunchKernelGGL(neg_sdr_sums_kernel, dim3((T + per - 1) / per, B * n_src * n_src), dim3(256), 0, (hipStream_t)stream, est, tgt, sums, n_src, T, per);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_neg_sdr_finish(const double* sums, int kind, int zero_mean, int take_log, float* pw, float* coef, int B, int n_src, int T, void* stream) {
    if (B <= 0 || n_src <= 0 || T <= 0 || kind < 0 || kind > 2) return RTFS_EINVAL;
    const int np = B * n_src * n_src;
    hipLaunchKernelGGL(neg_sdr_finish_kernel, dim3((np + 63) / 64), dim3(64), 0, (hipStream_t)stream, sums, kind, zero_mean, take_log, pw, coef, np, T);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_neg_sdr_grad(const float* est, const float* tgt, const float* coef, const float* G, float* dest, int B, int n_src, int T, void* stream) {
    if (B <= 0 || n_src <= 0 || T <= 0) return RTFS_EINVAL;
    hipLaunchKernelGGL(neg_sdr_grad_kernel, dim3((T + 1023) / 1024, B * n_src), dim3(256), 0, (hipStream_t)stream, est, tgt, coef, G, dest, n_src, T);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

}  // extern "C"
