// Backward of the dual-path SRU stage (training step).  Forward lives in dualpath.hip; the recurrence and its adjoint
// follow oracle/sru_ref.py / SURVEY.md §8a ("Backward" paragraph).
//
//   rtfs_sru_scan_train_fwd   forward recurrence that also stores the cell state C[s][l][64] (needed by the adjoint)
//   rtfs_sru_scan_bwd         reverse-time adjoint: dH -> dU (same layout as U), dX (skip input, layers 1-3),
//                             weight_c / bias gradients accumulated with one atomic per lane per sequence
//   rtfs_ln4d_c_bwd           adjoint of LayerNormalization4D over the 64 channels of each position (normalizations.py:33-37)
//   rtfs_seq_gather           G layout -> sequence-major [S][npos][64] copy, raw or LN4D-normalised (operands of rtfs_wgrad)
#include "common.h"

namespace rtfs {

struct SeqMapS {
    int seq_div;
    long long stride_hi, stride_lo, pos_stride;
    int npos, L;
    __device__ __forceinline__ size_t base(int s) const { return (size_t)(s / seq_div) * stride_hi + (size_t)(s % seq_div) * stride_lo; }
};

// same lane mapping as sru_scan_kernel (dualpath.hip): one wave per sequence, lane = dir*32 + j
template <int KM>
__global__ __launch_bounds__(256) void sru_scan_train_kernel(const float* __restrict__ U, const float* __restrict__ X, const float* __restrict__ wc,
                                                             const float* __restrict__ bias, float scale_x, float* __restrict__ Hout,
                                                             float* __restrict__ Cout, int S, int L) {
    constexpr int UNR = 8;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= S) return;
    const int lane = threadIdx.x & 63;
    const bool rev = lane >= 32;
    const float wf = wc[lane], wr = wc[64 + lane], bf = bias[lane], br = bias[64 + lane];
    const float* u = U + (size_t)s * L * 64 * KM + (KM == 4 ? lane * 4 : lane);
    const float* x = X + (size_t)s * L * 64 + lane;
    float* h = Hout + (size_t)s * L * 64 + lane;
    float* cs = Cout + (size_t)s * L * 64 + lane;
    float c = 0.f;
    // two register sets: batch i + 1 is fetched before the recurrence of batch i (as in sru_scan_kernel, dualpath.hip)
    float va[UNR][4], vb[UNR][4];
    auto load = [&](int t0, float (&v)[UNR][4]) {
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            const int t = min(t0 + i, L - 1);
            const int l = rev ? L - 1 - t : t;
            if (KM == 4) {
                const float4 q = ld4(u + (size_t)l * 256);
                v[i][0] = q.x, v[i][1] = q.y, v[i][2] = q.z, v[i][3] = q.w;
            } else {
                const float* p = u + (size_t)l * 192;
                v[i][0] = p[0], v[i][1] = p[64], v[i][2] = p[128];
                v[i][3] = x[(size_t)l * 64] * scale_x;
            }
        }
    };
    auto run = [&](int t0, const float (&v)[UNR][4]) {
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            const int t = t0 + i;
            if (t < L) {
                const int l = rev ? L - 1 - t : t;
                const float f = sigmoidf_fast(v[i][1] + bf + wf * c);
                const float r = sigmoidf_fast(v[i][2] + br + wr * c);
                c = v[i][0] + (c - v[i][0]) * f;
                cs[(size_t)l * 64] = c;
                h[(size_t)l * 64] = v[i][3] + (c - v[i][3]) * r;
            }
        }
    };
    load(0, va);
#pragma unroll 1
    for (int t0 = 0; t0 < L; t0 += 2 * UNR) {
        load(t0 + UNR, vb);
        run(t0, va);
        load(t0 + 2 * UNR, va);
        run(t0 + UNR, vb);
    }
}

// adjoint, walking the forward processing order backwards and carrying dc:
//   gh = dH_t; dx' = gh(1-r); dr = gh(c_t - x'); du2 = dr r(1-r); dct = gh r + dc; du0 = dct(1-f); df = dct(c_prev - u0);
//   du1 = df f(1-f); dc <- dct f + du1 wf + du2 wr; dwf += du1 c_prev; dwr += du2 c_prev; dbf += du1; dbr += du2
template <int KM>
__global__ __launch_bounds__(256) void sru_scan_bwd_kernel(const float* __restrict__ U, const float* __restrict__ X, const float* __restrict__ Cst,
                                                           const float* __restrict__ wc, const float* __restrict__ bias, float scale_x,
                                                           const float* __restrict__ dH, float* __restrict__ dU, float* __restrict__ dX,
                                                           float* __restrict__ scr, int S, int L) {
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= S) return;
    const int lane = threadIdx.x & 63;
    const bool rev = lane >= 32;
    const float wf = wc[lane], wr = wc[64 + lane], bf = bias[lane], br = bias[64 + lane];
    const float* u = U + (size_t)s * L * 64 * KM + (KM == 4 ? lane * 4 : lane);
    float* du = dU + (size_t)s * L * 64 * KM + (KM == 4 ? lane * 4 : lane);
    const size_t o64 = (size_t)s * L * 64 + lane;
    float dc = 0.f, awf = 0.f, awr = 0.f, abf = 0.f, abr = 0.f;
    // the loads of a step do not depend on the carried state: 8 steps are fetched (clamped, unconditional) ahead of the dependent
    // chain, as in the forward scan (a per-step load pays one memory latency per step)
    constexpr int UNR = 8;
#pragma unroll 1
    for (int i0 = 0; i0 < L; i0 += UNR) {
        float cpv[UNR], cv[UNR], ghv[UNR], u0v[UNR], u1v[UNR], u2v[UNR], xpv[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            // forward processed l_k = rev ? L-1-k : k for k = 0..L-1; the adjoint visits k = L-1-i
            const int k = max(L - 1 - (i0 + j), 0);
            const int l = rev ? L - 1 - k : k;
            const int lp = min(max(rev ? l + 1 : l - 1, 0), L - 1);  // position processed just before l (masked below when k == 0)
            cpv[j] = Cst[o64 + (size_t)lp * 64];
            cv[j] = Cst[o64 + (size_t)l * 64];
            ghv[j] = dH[o64 + (size_t)l * 64];
            if (KM == 4) {
                const float4 v = ld4(u + (size_t)l * 256);
                u0v[j] = v.x, u1v[j] = v.y, u2v[j] = v.z, xpv[j] = v.w;
            } else {
                const float* p = u + (size_t)l * 192;
                u0v[j] = p[0], u1v[j] = p[64], u2v[j] = p[128];
                xpv[j] = X[o64 + (size_t)l * 64] * scale_x;
            }
        }
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const int i = i0 + j;
            if (i < L) {  // wave-uniform
                const int k = L - 1 - i;
                const int l = rev ? L - 1 - k : k;
                const float cprev = k > 0 ? cpv[j] : 0.f;
                const float c = cv[j], gh = ghv[j], u0 = u0v[j], u1 = u1v[j], u2 = u2v[j], xp = xpv[j];
                const float f = sigmoidf_fast(u1 + bf + wf * cprev);
                const float r = sigmoidf_fast(u2 + br + wr * cprev);
                const float dxp = gh * (1.f - r);
                const float dr = gh * (c - xp);
                const float du2 = dr * r * (1.f - r);
                const float dct = gh * r + dc;
                const float du0 = dct * (1.f - f);
                const float df = dct * (cprev - u0);
                const float du1 = df * f * (1.f - f);
                dc = dct * f + du1 * wf + du2 * wr;
                awf = fmaf(du1, cprev, awf);
                awr = fmaf(du2, cprev, awr);
                abf += du1;
                abr += du2;
                if (KM == 4) {
                    st4(du + (size_t)l * 256, f4(du0, du1, du2, dxp));
                } else {
                    float* q = du + (size_t)l * 192;
                    q[0] = du0, q[64] = du1, q[128] = du2;
                    dX[o64 + (size_t)l * 64] = dxp * scale_x;
                }
            }
        }
    }
    float* mine = spread_copy(scr, blockIdx.x);  // [dwc 128 | dbias 128]
    atomicAdd(mine + lane, awf);
    atomicAdd(mine + 64 + lane, awr);
    atomicAdd(mine + 128 + lane, abf);
    atomicAdd(mine + 192 + lane, abr);
}

// LN4D over channels, adjoint.  dxn, G, dG in G layout [rows][64]; dG += dx; dgamma/dbeta += per-channel sums.
__global__ __launch_bounds__(256) void ln4d_c_bwd_kernel(const float* __restrict__ dxn, const float* __restrict__ G, const float* __restrict__ gamma,
                                                         float* __restrict__ dG, float* __restrict__ scr, long long rows, int rows_per_wg) {
    __shared__ __attribute__((aligned(16))) float lds[1024];
    const int c4 = (threadIdx.x & 15) * 4, rsub = threadIdx.x >> 4;
    const float4 g4 = ld4(gamma + c4);
    const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
    float4 ag = f4(0, 0, 0, 0), ab = f4(0, 0, 0, 0);
    for (long long rb = r0; rb < r1; rb += 16) {
        const long long r = rb + rsub;
        const bool ok = r < r1;
        const size_t o = (size_t)(ok ? r : r0) * 64 + c4;
        const float4 x = ld4(G + o);
        float4 g = ld4(dxn + o);
        if (!ok) g = f4(0, 0, 0, 0);
        float sum = x.x + x.y + x.z + x.w;
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) sum += __shfl_xor(sum, m, 64);
        const float mean = sum * (1.f / 64.f);
        const float4 d = f4(x.x - mean, x.y - mean, x.z - mean, x.w - mean);
        float sq = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) sq += __shfl_xor(sq, m, 64);
        const float rstd = 1.0f / sqrtf(sq * (1.f / 64.f) + kEps);
        const float4 xh = d * rstd;
        const float4 a = g * g4;
        float s1 = a.x + a.y + a.z + a.w, s2 = a.x * xh.x + a.y * xh.y + a.z * xh.z + a.w * xh.w;
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) {
            s1 += __shfl_xor(s1, m, 64);
            s2 += __shfl_xor(s2, m, 64);
        }
        s1 *= (1.f / 64.f), s2 *= (1.f / 64.f);
        if (ok) {
            const float4 dx = f4((a.x - s1 - xh.x * s2) * rstd, (a.y - s1 - xh.y * s2) * rstd, (a.z - s1 - xh.z * s2) * rstd, (a.w - s1 - xh.w * s2) * rstd);
            st4(dG + o, dx + ld4(dG + o));
        }
        ag = fma4(g, xh, ag);
        ab = ab + g;
    }
    // threads (rsub, quad) -> per channel, one coalesced atomic request per array into this workgroup's scratch copy [dgamma 64 | dbeta 64]
    float* mine = spread_copy(scr, blockIdx.x);
    st4(lds + threadIdx.x * 4, ag);
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll 4
        for (int r = 0; r < 16; ++r) t += lds[r * 64 + threadIdx.x];
        atomicAdd(mine + threadIdx.x, t);
    }
    __syncthreads();
    st4(lds + threadIdx.x * 4, ab);
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll 4
        for (int r = 0; r < 16; ++r) t += lds[r * 64 + threadIdx.x];
        atomicAdd(mine + 64 + threadIdx.x, t);
    }
}

// out[s][pos][64] = (LN ? LN4D_c(G[map(s,pos)]) : G[map(s,pos)])
template <bool LN>
__global__ __launch_bounds__(256) void seq_gather_kernel(SeqMapS map, const float* __restrict__ G, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ out, int S) {
    const long long idx = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int c4 = (threadIdx.x & 15) * 4;
    const bool ok = idx < (long long)S * map.npos;
    const int s = ok ? (int)(idx / map.npos) : 0, pos = ok ? (int)(idx % map.npos) : 0;
    float4 v = ld4(G + map.base(s) + (size_t)pos * map.pos_stride + c4);
    if (LN) {
        float sum = v.x + v.y + v.z + v.w;
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) sum += __shfl_xor(sum, m, 64);
        const float mean = sum * (1.f / 64.f);
        const float4 d = f4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
        float sq = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) sq += __shfl_xor(sq, m, 64);
        const float rstd = 1.0f / sqrtf(sq * (1.f / 64.f) + kEps);
        v = fma4(d * rstd, ld4(gamma + c4), ld4(beta + c4));
    }
    if (ok) st4(out + (size_t)idx * 64 + c4, v);
}

}  // namespace rtfs

using namespace rtfs;

static SeqMapS make_map(int dim, int T2) {
    SeqMapS m;
    if (dim == 4) {
        m.seq_div = 1, m.stride_hi = (long long)kF2 * kH, m.stride_lo = 0, m.pos_stride = kH, m.npos = kF2;
    } else {
        m.seq_div = kF2, m.stride_hi = (long long)T2 * kF2 * kH, m.stride_lo = kH, m.pos_stride = (long long)kF2 * kH, m.npos = T2;
    }
    m.L = m.npos - 7;
    return m;
}

extern "C" {

int rtfs_sru_scan_train_fwd(const float* U, const float* X, const float* wc, const float* bias, float scale_x, float* H, float* C, int S, int L,
                            int km, void* stream) {
    if (S <= 0 || L <= 0) return RTFS_EINVAL;
    dim3 grid((S + 3) / 4);
    if (km == 4)
        hipLaunchKernelGGL((sru_scan_train_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, U, X, wc, bias, scale_x, H, C, S, L);
    else if (km == 3)
        hipLaunchKernelGGL((sru_scan_train_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, U, X, wc, bias, scale_x, H, C, S, L);
    else
        return RTFS_EINVAL;
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

// dwc, dbias: [2][64] accumulated into.  km = 3 also writes dX [S][L][64] (gradient w.r.t. the skip input, already * scale_x).
int rtfs_sru_scan_bwd(const float* U, const float* X, const float* C, const float* wc, const float* bias, float scale_x, const float* dH, float* dU,
                      float* dX, float* dwc, float* dbias, int S, int L, int km, void* stream) {
    if (S <= 0 || L <= 0) return RTFS_EINVAL;
    float* scr = spread_scratch();
    if (!scr) return RTFS_ELAUNCH;
    dim3 grid((S + 3) / 4);
    if (km == 4)
        hipLaunchKernelGGL((sru_scan_bwd_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, U, X, C, wc, bias, scale_x, dH, dU, dX, scr, S, L);
    else if (km == 3)
        hipLaunchKernelGGL((sru_scan_bwd_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, U, X, C, wc, bias, scale_x, dH, dU, dX, scr, S, L);
    else
        return RTFS_EINVAL;
    RTFS_LAUNCH_CHECK();
    return spread_finish(scr, SpreadOut{{dwc, dbias}, {128, 128}}, (hipStream_t)stream);
}

int rtfs_ln4d_c_bwd(const float* dxn, const float* G, const float* gamma, float* dG, float* dgamma, float* dbeta, long long rows, void* stream) {
    if (rows <= 0) return RTFS_EINVAL;
    float* scr = spread_scratch();
    if (!scr) return RTFS_ELAUNCH;
    const int per = 256;
    hipLaunchKernelGGL(ln4d_c_bwd_kernel, dim3((unsigned)((rows + per - 1) / per)), dim3(256), 0, (hipStream_t)stream, dxn, G, gamma, dG, scr, rows, per);
    RTFS_LAUNCH_CHECK();
    return spread_finish(scr, SpreadOut{{dgamma, dbeta}, {64, 64}}, (hipStream_t)stream);
}

// ln != 0: LN4D-normalise (gamma, beta) while gathering.  out: [S][npos][64]
int rtfs_seq_gather(const float* G, const float* gamma, const float* beta, int ln, float* out, int B, int T2, int dim, void* stream) {
    if ((dim != 3 && dim != 4) || B <= 0) return RTFS_EINVAL;
    SeqMapS m = make_map(dim, T2);
    const int S = dim == 4 ? B * T2 : B * kF2;
    const long long n = (long long)S * m.npos;
    dim3 grid((unsigned)((n + 15) / 16));
    if (ln)
        hipLaunchKernelGGL(seq_gather_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, m, G, gamma, beta, out, S);
    else
        hipLaunchKernelGGL(seq_gather_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, m, G, gamma, beta, out, S);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

}  // extern "C"
