// Backward of the S3 mask, the CAF cell (training-mode BatchNorm), the iSTFT decoder and the encoder convolution.
//
//   rtfs_mask_bwd_elem     adjoint of relu + complex multiply (mask_generator.py:70-82): d(masked) -> dz (pre-ReLU), da_emb +=
//   rtfs_prelu_bwd         dx = dy * prelu'(x), dslope += sum dy*x*[x<=0]                   (mask_generator.0)
//   rtfs_chan_stats        per-channel (sum, sumsq) over all rows of [rows][256]            (training-mode BatchNorm2d of the CAF cell)
//   rtfs_caf_bwd_reduce    per-channel / per-(b,tv,c) reductions of the CAF adjoint
//   rtfs_caf_bwd_apply     dx of the CAF cell from those reductions
//   rtfs_istft_bwd         adjoint of overlap-add + irfft + 9-tap gather: dout [B][L] -> dtaps [B][T][129][32]
//   rtfs_spec_patches      im2col of the spectrogram for the encoder-conv weight gradient: [B][T][129][32] (18 used)
#include "common.h"

namespace rtfs {

__device__ __forceinline__ float hann256b(int i) { return 0.5f - 0.5f * cospif((float)i * (1.0f / 128.0f)); }

// m: saved post-ReLU mask [rows][256] (re | im halves), e: a_emb, dO: grad of masked.  dz = dm * [m > 0]; de accumulated.
__global__ __launch_bounds__(256) void mask_bwd_elem_kernel(const float* __restrict__ dO, const float* __restrict__ e, const float* __restrict__ m,
                                                            float* __restrict__ dz, float* __restrict__ de, long long rows) {
    const long long r = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (r >= rows) return;
    const int c4 = (threadIdx.x & 31) * 4;
    const size_t o = (size_t)r * kC + c4;
    const float4 dor = ld4(dO + o), doi = ld4(dO + o + 128), er = ld4(e + o), ei = ld4(e + o + 128), mr = ld4(m + o), mi = ld4(m + o + 128);
    auto gate = [](float g, float mm) { return mm > 0.f ? g : 0.f; };
    const float4 dmr = f4(dor.x * er.x + doi.x * ei.x, dor.y * er.y + doi.y * ei.y, dor.z * er.z + doi.z * ei.z, dor.w * er.w + doi.w * ei.w);
    const float4 dmi = f4(doi.x * er.x - dor.x * ei.x, doi.y * er.y - dor.y * ei.y, doi.z * er.z - dor.z * ei.z, doi.w * er.w - dor.w * ei.w);
    st4(dz + o, f4(gate(dmr.x, mr.x), gate(dmr.y, mr.y), gate(dmr.z, mr.z), gate(dmr.w, mr.w)));
    st4(dz + o + 128, f4(gate(dmi.x, mi.x), gate(dmi.y, mi.y), gate(dmi.z, mi.z), gate(dmi.w, mi.w)));
    const float4 der = f4(dor.x * mr.x + doi.x * mi.x, dor.y * mr.y + doi.y * mi.y, dor.z * mr.z + doi.z * mi.z, dor.w * mr.w + doi.w * mi.w);
    const float4 dei = f4(doi.x * mr.x - dor.x * mi.x, doi.y * mr.y - dor.y * mi.y, doi.z * mr.z - dor.z * mi.z, doi.w * mr.w - dor.w * mi.w);
    st4(de + o, der);  // (plain store since round 4: this is the FIRST contribution to d(a_emb) - the += form cost a 1 GB memset and a 1 GB read per step)
    st4(de + o + 128, dei);
}

template <bool ACCUM>
// (dy and dx may be the same buffer - rtfs_gemm_prelu_bwd's small-map form runs this in place: every element is read once, by the thread that writes it)
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* dy, const float* __restrict__ x, float slope, float* dx, float* __restrict__ scr, long long n4) {
    __shared__ float red[4];
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 g = ld4(dy + i * 4), v = ld4(x + i * 4);
        acc += (v.x > 0.f ? 0.f : g.x * v.x) + (v.y > 0.f ? 0.f : g.y * v.y) + (v.z > 0.f ? 0.f : g.z * v.z) + (v.w > 0.f ? 0.f : g.w * v.w);
        float4 d = f4(v.x > 0.f ? g.x : g.x * slope, v.y > 0.f ? g.y : g.y * slope, v.z > 0.f ? g.z : g.z * slope, v.w > 0.f ? g.w : g.w * slope);
        if (ACCUM) d = d + ld4(dx + i * 4);
        st4(dx + i * 4, d);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(spread_copy(scr, blockIdx.x), red[0] + red[1] + red[2] + red[3]);
}

// sum[c], sumsq[c] over rows of x [rows][256]  (fp64 accumulation across workgroups)
__global__ __launch_bounds__(256) void chan_stats_kernel(const float* __restrict__ x, double* __restrict__ sum, double* __restrict__ sumsq, long long rows,
                                                         int rows_per_wg) {
    __shared__ __attribute__((aligned(16))) float lds[2048];
    const int c4 = (threadIdx.x & 63) * 4, rsub = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
    float4 s = f4(0, 0, 0, 0), q = f4(0, 0, 0, 0);
    for (long long r = r0 + rsub; r < r1; r += 4) {
        const float4 v = ld4(x + (size_t)r * kC + c4);
        s = s + v;
        q = fma4(v, v, q);
    }
    st4(lds + threadIdx.x * 4, s);
    st4(lds + 1024 + threadIdx.x * 4, q);
    __syncthreads();
    // thread = channel: one coalesced fp64 atomic request per line (same-line requests are served serially, ~27 ns each)
    const int c = threadIdx.x;
    const float ts = lds[c] + lds[256 + c] + lds[512 + c] + lds[768 + c];
    const float tq = lds[1024 + c] + lds[1280 + c] + lds[1536 + c] + lds[1792 + c];
    atomicAdd(sum + c, (double)ts);
    atomicAdd(sumsq + c, (double)tq);
}

// CAF forward (fusion.py:259-272) with folded BatchNorm: key = relu(x*ks+kb), val = x*vs+vb, out = key*rsz^ + att^*val.
// Reductions of the adjoint (dOut = gradient of out):
//   drsz[b][tv][c] += sum_{t->tv, f} dOut*key        datt[b][tv][c] += sum dOut*val
//   per channel (over all b,t,f):  R[0] = sum dk, R[1] = sum dk*x   (dk = dOut*rsz^*[key>0])
//                                  R[2] = sum dv, R[3] = sum dv*x   (dv = dOut*att^)
// grid: (T, B): one workgroup per time frame (all f), thread = channel quad x 4 sub-rows.
__global__ __launch_bounds__(256) void caf_bwd_reduce_kernel(const float* __restrict__ dOut, const float* __restrict__ x, const float* __restrict__ ks,
                                                             const float* __restrict__ kb, const float* __restrict__ vs, const float* __restrict__ vb,
                                                             const float* __restrict__ att, const float* __restrict__ rsz, float* __restrict__ datt,
                                                             float* __restrict__ drsz, float* __restrict__ scr /* R [4][256] spread */, int T, int Tv) {
    __shared__ __attribute__((aligned(16))) float lds[1024];
    const int t = blockIdx.x, b = blockIdx.y;
    const int c4 = (threadIdx.x & 63) * 4, fsub = threadIdx.x >> 6;
    const int tv = nearest_src(t, Tv, T);
    const size_t ov = ((size_t)b * Tv + tv) * kC + c4;
    const float4 ks4 = ld4(ks + c4), kb4 = ld4(kb + c4), vs4 = ld4(vs + c4), vb4 = ld4(vb + c4), a4 = ld4(att + ov), r4 = ld4(rsz + ov);
    float4 arsz = f4(0, 0, 0, 0), aatt = f4(0, 0, 0, 0), r0 = f4(0, 0, 0, 0), r1 = f4(0, 0, 0, 0), r2 = f4(0, 0, 0, 0), r3 = f4(0, 0, 0, 0);
    for (int f = fsub; f < kF; f += 4) {
        const size_t o = (((size_t)b * T + t) * kF + f) * kC + c4;
        const float4 g = ld4(dOut + o), xv = ld4(x + o);
        const float4 kt = fma4(xv, ks4, kb4), val = fma4(xv, vs4, vb4);
        const float4 key = relu4(kt);
        arsz = fma4(g, key, arsz);
        aatt = fma4(g, val, aatt);
        const float4 dk = f4(kt.x > 0.f ? g.x * r4.x : 0.f, kt.y > 0.f ? g.y * r4.y : 0.f, kt.z > 0.f ? g.z * r4.z : 0.f, kt.w > 0.f ? g.w * r4.w : 0.f);
        const float4 dv = g * a4;
        r0 = r0 + dk, r1 = fma4(dk, xv, r1), r2 = r2 + dv, r3 = fma4(dv, xv, r3);
    }
    auto commit = [&](float4 v, float* out) {  // thread = channel after the LDS transpose: one coalesced request per line
        st4(lds + threadIdx.x * 4, v);
        __syncthreads();
        const int c = threadIdx.x;
        atomicAdd(out + c, lds[c] + lds[256 + c] + lds[512 + c] + lds[768 + c]);
        __syncthreads();
    };
    commit(arsz, drsz + ((size_t)b * Tv + tv) * kC);
    commit(aatt, datt + ((size_t)b * Tv + tv) * kC);
    float* R = spread_copy(scr, blockIdx.x + blockIdx.y);
    commit(r0, R), commit(r1, R + 256), commit(r2, R + 512), commit(r3, R + 768);
}

// dx = dk*ck1 + ck2 + ck3*x  +  dv*cv1 + cv2 + cv3*x, with per-channel coefficient vectors prepared on the host from the
// BatchNorm adjoint (training: batch statistics; eval: ck2 = ck3 = cv2 = cv3 = 0): coef [6][256] = ck1,ck2,ck3,cv1,cv2,cv3.
template <bool ACCUM>
__global__ __launch_bounds__(256) void caf_bwd_apply_kernel(const float* __restrict__ dOut, const float* __restrict__ x, const float* __restrict__ ks,
                                                            const float* __restrict__ kb, const float* __restrict__ att, const float* __restrict__ rsz,
                                                            const float* __restrict__ coef, float* __restrict__ dx, int T, int Tv) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= T * kF) return;
    const int c4 = (threadIdx.x & 63) * 4;
    const int t = p / kF, tv = nearest_src(t, Tv, T);
    const size_t o = ((size_t)b * T * kF + p) * kC + c4, ov = ((size_t)b * Tv + tv) * kC + c4;
    const float4 g = ld4(dOut + o), xv = ld4(x + o), r4 = ld4(rsz + ov), a4 = ld4(att + ov);
    const float4 kt = fma4(xv, ld4(ks + c4), ld4(kb + c4));
    const float4 dk = f4(kt.x > 0.f ? g.x * r4.x : 0.f, kt.y > 0.f ? g.y * r4.y : 0.f, kt.z > 0.f ? g.z * r4.z : 0.f, kt.w > 0.f ? g.w * r4.w : 0.f);
    const float4 dv = g * a4;
    float4 d = fma4(dk, ld4(coef + c4), ld4(coef + 256 + c4));
    d = fma4(xv, ld4(coef + 512 + c4), d);
    d = fma4(dv, ld4(coef + 768 + c4), d);
    d = d + ld4(coef + 1024 + c4);
    d = fma4(xv, ld4(coef + 1280 + c4), d);
    if (ACCUM) d = d + ld4(dx + o);
    st4(dx + o, d);
}

// ---- iSTFT adjoint ---------------------------------------------------------------------------------------------------------
template <bool INV>
__device__ __forceinline__ float2* fft256b(float2* a, float2* b, int j) {
    constexpr float sgn = INV ? 1.f : -1.f;
#pragma unroll
    for (int Ns = 1; Ns < 256; Ns *= 4) {
        const int k = j & (Ns - 1);
        float2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = a[j + r * 64];
        if (Ns > 1) {
#pragma unroll
            for (int r = 1; r < 4; ++r) {
                float sn, cs;
                sincospif(sgn * 2.0f * (float)(r * k) / (float)(Ns * 4), &sn, &cs);
                v[r] = make_float2(v[r].x * cs - v[r].y * sn, v[r].x * sn + v[r].y * cs);
            }
        }
        const float2 a0 = make_float2(v[0].x + v[2].x, v[0].y + v[2].y), a1 = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
        const float2 a2 = make_float2(v[1].x + v[3].x, v[1].y + v[3].y), d = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
        const float2 a3 = make_float2(-sgn * d.y, sgn * d.x);
        const int j0 = ((j - k) << 2) + k;
        b[j0] = make_float2(a0.x + a2.x, a0.y + a2.y);
        b[j0 + Ns] = make_float2(a1.x + a3.x, a1.y + a3.y);
        b[j0 + 2 * Ns] = make_float2(a0.x - a2.x, a0.y - a2.y);
        b[j0 + 3 * Ns] = make_float2(a1.x - a3.x, a1.y - a3.y);
        __syncthreads();
        float2* t = a;
        a = b;
        b = t;
    }
    return a;
}

// dspec[b][t][k][2] = (c_k/256) * FFT(dframe)[k], dframe[i] = hann[i] * dout[n]/env[n], n = t*128 + i - 128 (imag of k = 0, 128 is 0).
__global__ __launch_bounds__(256) void istft_bwd_spec_kernel(const float* __restrict__ dout, float* __restrict__ dspec, int L, int T) {
    __shared__ float2 buf[4][2][256];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = min(blockIdx.x * 4 + w, T - 1), b = blockIdx.y;
    const bool valid = blockIdx.x * 4 + w < T;
    float2* A = buf[w][0];
    float2* Bf = buf[w][1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = lane + r * 64;
        const int n = t * kHop + i - kWin / 2;
        float v = 0.f;
        if (n >= 0 && n < L) {
            const int m = n + kWin / 2, t1 = m / kHop;
            float env = 0.f;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int tt = t1 - d, ii = m - tt * kHop;
                if (tt >= 0 && tt < T && ii < kWin) {
                    const float wv = hann256b(ii);
                    env = fmaf(wv, wv, env);
                }
            }
            if (env > 1e-11f) v = hann256b(i) * dout[(size_t)b * L + n] / env;
        }
        A[i] = make_float2(v, 0.f);
    }
    __syncthreads();
    float2* R = fft256b<false>(A, Bf, lane);
    float2* out = reinterpret_cast<float2*>(dspec) + ((size_t)b * T + t) * kF;
    if (valid)
        for (int k = lane; k < kF; k += 64) {
            const float c = (k == 0 || k == 128) ? (1.0f / 256.0f) : (2.0f / 256.0f);
            out[k] = make_float2(R[k].x * c, (k == 0 || k == 128) ? 0.f : R[k].y * c);
        }
}

// dtaps[b][tt][ff][o*9+kt*3+kf] = dspec[b][tt+kt-1][ff+kf-1][o]  (0 outside, columns 18..31 = 0)
__global__ __launch_bounds__(256) void istft_bwd_taps_kernel(const float* __restrict__ dspec, float* __restrict__ dtaps, int T) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (p >= T * kF) return;
    const int col = threadIdx.x & 31;
    const int tt = p / kF, ff = p - tt * kF;
    float v = 0.f;
    if (col < 18) {
        const int o = col / 9, kt = (col % 9) / 3, kf = col % 3;
        const int t = tt + kt - 1, f = ff + kf - 1;
        if (t >= 0 && t < T && f >= 0 && f < kF) v = dspec[(((size_t)b * T + t) * kF + f) * 2 + o];
    }
    dtaps[((size_t)b * T * kF + p) * 32 + col] = v;
}

// patches[b][t][f][ci*9+dt*3+df] = spec[b][t+dt-1][f+df-1][ci]  (0 outside; columns 18..31 = 0)
__global__ __launch_bounds__(256) void spec_patches_kernel(const float* __restrict__ spec, float* __restrict__ patches, int T) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (p >= T * kF) return;
    const int col = threadIdx.x & 31;
    const int t = p / kF, f = p - t * kF;
    float v = 0.f;
    if (col < 18) {
        const int ci = col / 9, dt = (col % 9) / 3, df = col % 3;
        const int tt = t + dt - 1, ff = f + df - 1;
        if (tt >= 0 && tt < T && ff >= 0 && ff < kF) v = spec[(((size_t)b * T + tt) * kF + ff) * 2 + ci];
    }
    patches[((size_t)b * T * kF + p) * 32 + col] = v;
}


// out = x[0] + ... + x[n-1] (n <= 8): the running d(a0) sum of the training step formed ONCE from the input gradients of blocks R-1 .. 1 (each kept as the next
// block's dx anyway) instead of a read-modify-write of a [B][T][F][256] tensor inside every block's last kernel (round 6)
struct SumArgs {
    const float* x[8];
    int n;
};
__global__ __launch_bounds__(256) void sum_n_kernel(SumArgs a, float* __restrict__ out, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = ld4(a.x[k < a.n ? k : 0] + i * 4);  // (all loads in flight; slots past n re-read x[0]: an L1 hit)
    float4 s = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k)
        if (k < a.n) s = s + v[k];
    st4(out + i * 4, s);
}

}  // namespace rtfs

using namespace rtfs;

#define LAUNCH(kernel, grid, ...)                                                      \
    hipLaunchKernelGGL(kernel, grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
    RTFS_LAUNCH_CHECK();

extern "C" {

int rtfs_mask_bwd_elem(const float* dmasked, const float* a_emb, const float* m, float* dz, float* da_emb, long long rows, void* stream) {
    if (rows <= 0) return RTFS_EINVAL;
    LAUNCH(mask_bwd_elem_kernel, dim3((unsigned)((rows + 7) / 8)), dmasked, a_emb, m, dz, da_emb, rows);
    return RTFS_OK;
}

int rtfs_prelu_bwd(const float* dy, const float* x, float slope, float* dx, int accumulate, float* dslope, long long n, void* stream) {
    if (n <= 0 || (n & 3)) return RTFS_EINVAL;
    const long long n4 = n / 4;
    dim3 grid((unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096));
    float* scr = spread_scratch();
    if (!scr) return RTFS_ELAUNCH;
    if (accumulate) { LAUNCH(prelu_bwd_kernel<true>, grid, dy, x, slope, dx, scr, n4); }
    else { LAUNCH(prelu_bwd_kernel<false>, grid, dy, x, slope, dx, scr, n4); }
    return spread_finish(scr, SpreadOut{{dslope}, {1}}, (hipStream_t)stream);
}

int rtfs_chan_stats(const float* x, double* sum, double* sumsq, long long rows, void* stream) {
    if (rows <= 0) return RTFS_EINVAL;
    const int per = 2048;
    LAUNCH(chan_stats_kernel, dim3((unsigned)((rows + per - 1) / per)), x, sum, sumsq, rows, per);
    return RTFS_OK;
}

int rtfs_caf_bwd_reduce(const float* dOut, const float* x, const float* ks, const float* kb, const float* vs, const float* vb, const float* att,
                        const float* rsz, float* datt, float* drsz, float* R, int B, int T, int Tv, void* stream) {
    if (B <= 0 || T <= 0 || Tv <= 0) return RTFS_EINVAL;
    float* scr = spread_scratch();
    if (!scr) return RTFS_ELAUNCH;
    LAUNCH(caf_bwd_reduce_kernel, dim3(T, B), dOut, x, ks, kb, vs, vb, att, rsz, datt, drsz, scr, T, Tv);
    return spread_finish(scr, SpreadOut{{R}, {1024}}, (hipStream_t)stream, /*consumed_now=*/true);  // (the host side folds R into the BatchNorm adjoint next)
}

int rtfs_caf_bwd_apply(const float* dOut, const float* x, const float* ks, const float* kb, const float* att, const float* rsz, const float* coef,
                       float* dx, int accumulate, int B, int T, int Tv, void* stream) {
    if (B <= 0 || T <= 0 || Tv <= 0) return RTFS_EINVAL;
    dim3 grid((T * kF + 3) / 4, B);
    if (accumulate) { LAUNCH(caf_bwd_apply_kernel<true>, grid, dOut, x, ks, kb, att, rsz, coef, dx, T, Tv); }
    else { LAUNCH(caf_bwd_apply_kernel<false>, grid, dOut, x, ks, kb, att, rsz, coef, dx, T, Tv); }
    return RTFS_OK;
}

// dspec: workspace [B][T][129][2]; dtaps: [B][T][129][32]
int rtfs_istft_bwd(const float* dout, float* dspec, float* dtaps, int B, int L, void* stream) {
    if (B <= 0 || L < kWin / 2 + 1) return RTFS_EINVAL;
    const int T = 1 + L / kHop;
    LAUNCH(istft_bwd_spec_kernel, dim3((T + 3) / 4, B), dout, dspec, L, T);
    LAUNCH(istft_bwd_taps_kernel, dim3((T * kF + 7) / 8, B), dspec, dtaps, T);
    return RTFS_OK;
}

int rtfs_spec_patches(const float* spec, float* patches, int B, int T, void* stream) {
    if (B <= 0 || T <= 0) return RTFS_EINVAL;
    LAUNCH(spec_patches_kernel, dim3((T * kF + 7) / 8, B), spec, patches, T);
    return RTFS_OK;
}

int rtfs_sum_n(const float* const* xs, int n, float* out, long long count, void* stream) {
    if (!xs || n < 1 || n > 8 || count <= 0 || (count & 3)) return RTFS_EINVAL;
    SumArgs a{};
    a.n = n;
    for (int k = 0; k < n; ++k) {
        if (!xs[k]) return RTFS_EINVAL;
        a.x[k] = xs[k];
    }
    LAUNCH(sum_n_kernel, dim3((unsigned)((count / 4 + 255) / 256)), a, out, count / 4);
    return RTFS_OK;
}

}  // extern "C"
