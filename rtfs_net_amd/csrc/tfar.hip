// Bandwidth-bound stages of the RTFS block and the CAF fusion, channels-last, 16-byte vector access.
//
//   rtfs_dwconv_fwd     depth-wise 4x4 convolutions (+bias) with gLN partial sums; input normalised on read
//                       downsample_layers (tdanet.py:61-76,112-114), InjectionMultiSum embeddings (fusion.py:25-52)
//   rtfs_pool_fwd       adaptive_avg_pool2d(D0) + D1 (tdanet.py:117-118)
//   rtfs_tfar_mix_fwd   local * sigmoid(gate)^ + global^   (fusion.py:67)  (^ = nearest up-sampling, fusion.py:59-61)
//   rtfs_caf_video_fwd  video side of ATTNFusionCell (fusion.py:255,262-265): grouped 1x1 convs + gLN, head mean, softmax over Tv
//   rtfs_caf_fuse_fwd   audio side: key/value depth-wise 1x1 + BatchNorm(eval) folded, k1 + k2 (fusion.py:259-272) [+ a0]
//
// Thread mapping for all H=64 tensors: 16 consecutive lanes own the 64 channels of one pixel (float4 each).
#include <type_traits>
#include "common.h"
#include "intdiv.h"

#include <algorithm>

namespace rtfs {

constexpr int kMaxConv = 4;

struct NormRefLite {
    const float* x;
    const double* slot;
    double inv_n;
    const float *gamma, *beta;
};

__device__ __forceinline__ void norm_coef(const NormRefLite& r, int b, int c4, float4& sc, float4& sh) {
    float mean, rstd;
    stats_finalize(r.slot, b, r.inv_n, mean, rstd);
    const float4 g = ld4(r.gamma + c4), be = ld4(r.beta + c4);
    sc = g * rstd;
    sh = f4(be.x - mean * sc.x, be.y - mean * sc.y, be.z - mean * sc.z, be.w - mean * sc.w);
}

struct DwArgs {
    const float* in;      // [B][Tin][Fin][64]
    const double* slot;   // gLN stats of `in` (mode >= 1)
    double inv_n;
    const float *gamma, *beta;
    float slope;          // PReLU slope (mode 2)
    int Tin, Fin, Tout, Fout;
    int nconv;
    const float* w[kMaxConv];     // [16 taps][64], tap = dt*4 + df
    const float* bias[kMaxConv];  // [64] or null
    float* out[kMaxConv];         // [B][Tout][Fout][64]
    double* stats[kMaxConv];      // [B][2]
    // MODE 3 (stride 1 only): the input is the TFAR mix  gLN(in) * sigmoid(gLN(gate)^) + gLN(glob)^  (InjectionMultiSum, fusion.py:59-67;
    // ^ = nearest up-sampling from (Tg, Fg)), formed on the way into LDS - the mixed tensor never exists in HBM
    NormRefLite gate, glob;
    int Tg, Fg;
    unsigned mt, mf;  // ceil(2^32 / Tin), ceil(2^32 / Fin): the nearest source index floor(i * in / out) of the mix without a division (csrc/intdiv.h, round 6)
    // GADD (dwconv_s1_kernel<1, 1, 4, true>): addout[pixel] = addsrc[pixel] + gLN(in)[pixel] rides along (tdanet.py:117-118: G = pooled + gLN(D1))
    const float* addsrc;
    float* addout;
};

// Depth-wise 4x4 convolution, sliding-window form.
//   MODE 0: raw input; 1: gLN(input); 2: PReLU(gLN(input)); 3 (dwconv_s1_kernel only): TFAR mix of three gLN'd tensors.
//   STRIDE 1: 'same' padding (1 before, 2 after); 2: padding 1.
//   Zero padding applies to the transformed input (conv_layers.py:104-113): out-of-range taps are masked to 0.
// Thread = (output time row, channel quad); it walks a frequency segment keeping a 4-column x 4-row register
// window of the (normalised) input, so every input element is loaded once per overlapping time row (4x for
// stride 1, 2x for stride 2; L1 hits) instead of 16x.  Loads use clamped addresses + a select instead of
// branches so the compiler issues a whole step's loads before the first use.  Tap weights sit in LDS
// ([conv][tap][64]); a workgroup (16 rows x 16 channel quads) commits its gLN partial sums once.
// Column c of the input lives in window slot (c+1)&3.   grid: (ceil(Tout/16), B, nseg); fseg % 4 == 0.
template <int STRIDE, int NCONV, int MODE>
__global__ __launch_bounds__(256, 2) void dwconv_kernel(DwArgs a, int fseg) {
    __shared__ __attribute__((aligned(16))) float ws[NCONV][16 * 64];
    __shared__ float red[8];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < NCONV * 256; i += 256) {
        const int j = i >> 8, o = (i & 255) * 4;
        st4(&ws[j][o], ld4(a.w[j] + o));
    }
    const int c4 = (threadIdx.x & 15) * 4;
    const int to = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int Tin = a.Tin, Fin = a.Fin, Tout = a.Tout, Fout = a.Fout;
    const int f0 = blockIdx.z * fseg, f1 = min(Fout, f0 + fseg);
    float mean = 0.f, rstd = 1.f;
    if (MODE >= 1) stats_finalize(a.slot, b, a.inv_n, mean, rstd);
    float4 sc = f4(1, 1, 1, 1), sh = f4(0, 0, 0, 0);
    if (MODE >= 1) {
        const float4 g = ld4(a.gamma + c4), be = ld4(a.beta + c4);
        sc = g * rstd;
        sh = f4(be.x - mean * sc.x, be.y - mean * sc.y, be.z - mean * sc.z, be.w - mean * sc.w);
    }
    const bool tvalid = to < Tout;
    const float* rowp[4];
    float rmask[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ti = to * STRIDE - 1 + r;
        const bool ok = tvalid && ti >= 0 && ti < Tin;
        rmask[r] = ok ? 1.f : 0.f;
        rowp[r] = a.in + (((size_t)b * Tin + min(max(ti, 0), Tin - 1)) * Fin) * kH + c4;
    }
    auto load_col = [&](int c, float4(&col)[4]) {
        const float cm = (c >= 0 && c < Fin) ? 1.f : 0.f;
        const size_t off = (size_t)min(max(c, 0), Fin - 1) * kH;
        float4 x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = ld4(rowp[r] + off);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float4 v = x[r];
            if (MODE >= 1) v = fma4(v, sc, sh);
            if (MODE == 2) v = prelu4(v, a.slope);
            col[r] = v * (cm * rmask[r]);
        }
    };
    __syncthreads();
    float4 win[4][4];  // [column slot][time row]
    if (STRIDE == 1) {
        load_col(f0 - 1, win[0]);
        load_col(f0, win[1]);
        load_col(f0 + 1, win[2]);
    } else {
        load_col(2 * f0 - 1, win[0]);
        load_col(2 * f0, win[1]);
    }
    float s[NCONV], q[NCONV];
#pragma unroll
    for (int j = 0; j < NCONV; ++j) s[j] = q[j] = 0.f;
    // a single convolution keeps its 16 tap vectors in registers; several share the LDS copy
    float4 wreg[NCONV == 1 ? 16 : 1];
    if (NCONV == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) wreg[i] = ld4(&ws[0][i * 64 + c4]);
    }
    const size_t orow = (((size_t)b * Tout + (tvalid ? to : 0)) * Fout) * kH + c4;
    constexpr int STEPS = 4 / STRIDE;
#pragma unroll 1
    for (int f = f0; f < f1; f += STEPS) {
#pragma unroll
        for (int j = 0; j < STEPS; ++j) {
            const int fo = f + j;
            int base;  // window slot of tap column df = 0
            if (STRIDE == 1) {
                load_col(fo + 2, win[(j + 3) & 3]);
                base = j;
            } else {
                load_col(2 * fo + 1, win[(2 * j + 2) & 3]);
                load_col(2 * fo + 2, win[(2 * j + 3) & 3]);
                base = 2 * j;
            }
            float4 acc[NCONV];
#pragma unroll
            for (int k = 0; k < NCONV; ++k) acc[k] = a.bias[k] ? ld4(a.bias[k] + c4) : f4(0, 0, 0, 0);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int df = 0; df < 4; ++df) {
                    const float4 x = win[(base + df) & 3][dt];
                    if (NCONV == 1) {
                        acc[0] = fma4(wreg[dt * 4 + df], x, acc[0]);
                    } else {
#pragma unroll
                        for (int k = 0; k < NCONV; ++k) acc[k] = fma4(ld4(&ws[k][(dt * 4 + df) * 64 + c4]), x, acc[k]);
                    }
                }
            if (tvalid && fo < f1) {
#pragma unroll
                for (int k = 0; k < NCONV; ++k) {
                    st4(a.out[k] + orow + (size_t)fo * kH, acc[k]);
                    s[k] += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
                    q[k] += acc[k].x * acc[k].x + acc[k].y * acc[k].y + acc[k].z * acc[k].z + acc[k].w * acc[k].w;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NCONV; ++k) {
        if (k) __syncthreads();
        block_stats_commit(s[k], q[k], red, a.stats[k], b);
    }
}

// Stride-1 depth-wise 4x4 convolution, LDS-staged.  The register-window kernel above issues one column's four loads per output
// step and needs them in that same step, so every step pays a full memory latency (measured: 150-190 us for a 530 MB stream).
// Here a workgroup stages the (16+3) x (8+3) pixel x 64 channel input block of 16 x 8 outputs with ALL loads in flight at once
// (13 x 16 B per thread), applies the gLN / PReLU transform and the zero padding ONCE per element on the way into LDS, and the
// sliding 4x4 window then walks LDS (100-cycle latency) instead of HBM.  53.5 KB of LDS: two workgroups per CU alternate
// between their load and compute phases.   grid: (ceil(T/16), B, nseg); a workgroup walks its f segment in blocks of 8.
// Round 4: one or two convolutions take 4-column blocks (38-43 KB of LDS, <= 168 VGPRs: THREE workgroups per CU instead of two).  The kernel is bound by
// the bytes it keeps in flight - a workgroup only has loads outstanding during its staging phase - and a third workgroup per CU measured -10 ... -12 %
// on the full-resolution launches (same box: 149 -> 134 us, 186 -> 164 us); four workgroups (<= 128 VGPRs) spill the mix form and gain nothing more.
// Four convolutions keep the 8-column block (their accumulators do not fit 168 registers).
template <int NCONV, int MODE, int TC = (NCONV <= 2 ? 4 : 8), bool GADD = false>
__global__ __launch_bounds__(256, (NCONV <= 2 ? 3 : 2)) void dwconv_s1_kernel(DwArgs a, int fseg) {
    constexpr int TR = 16, R = TR + 3, CB = TC + 3, RPI = 16 / TC;  // RPI: tile rows covered by one 256-thread pass over the new columns
    constexpr int RS = CB * 64;  // unpadded: ds_read_b128 serves lanes {0-3,12-15,20-27 | ...}, for which 256-byte rows at a multiple of
                                 // 64 floats are already conflict-free (a +16 pad was measured 20 % slower)
    __shared__ __attribute__((aligned(16))) float tile[R * RS];
    __shared__ __attribute__((aligned(16))) float ws[NCONV][16 * 64];
    __shared__ float red[8];
    const int b = blockIdx.y, bx = blockIdx.x, bz = blockIdx.z;
    for (int i = threadIdx.x; i < NCONV * 256; i += 256) {
        const int j = i >> 8, o = (i & 255) * 4;
        st4(&ws[j][o], ld4(a.w[j] + o));
    }
    const int c4 = (threadIdx.x & 15) * 4, tr = threadIdx.x >> 4;
    const int T = a.Tin, F = a.Fin;  // stride 1: output size == input size
    const int t0 = bx * TR, to = t0 + tr;
    const int f0 = bz * fseg, f1 = min(F, f0 + fseg);
    float4 sc = f4(1, 1, 1, 1), sh = f4(0, 0, 0, 0);
    if (MODE >= 1) {
        float mean, rstd;
        stats_finalize(a.slot, b, a.inv_n, mean, rstd);
        const float4 g = ld4(a.gamma + c4), be = ld4(a.beta + c4);
        sc = g * rstd;
        sh = f4(be.x - mean * sc.x, be.y - mean * sc.y, be.z - mean * sc.z, be.w - mean * sc.w);
    }
    const float* inb = a.in + (size_t)b * T * F * kH;
    // MODE 3: gate / glob tensors at (Tg, Fg) and their folded gLN coefficients
    // (kept in LDS and re-read per staging group: four more float4 of per-channel constants in registers spill the compute phase)
    __shared__ __attribute__((aligned(16))) float mixc[MODE == 3 ? 4 * 64 : 4];
    const float *gateb = nullptr, *globb = nullptr;
    if (MODE == 3) {
        if (threadIdx.x < 16) {
            float4 scg, shg, sce, she;
            norm_coef(a.gate, b, c4, scg, shg);
            norm_coef(a.glob, b, c4, sce, she);
            st4(mixc + c4, scg), st4(mixc + 64 + c4, shg), st4(mixc + 128 + c4, sce), st4(mixc + 192 + c4, she);
        }
        gateb = a.gate.x + (size_t)b * a.Tg * a.Fg * kH;
        globb = a.glob.x + (size_t)b * a.Tg * a.Fg * kH;
    }
    const bool tvalid = to < T;
    float s[NCONV], q[NCONV];
#pragma unroll
    for (int j = 0; j < NCONV; ++j) s[j] = q[j] = 0.f;
    float4 bias4[NCONV];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NCONV; ++k) bias4[k] = a.bias[k] ? ld4(a.bias[k] + c4) : f4(0, 0, 0, 0);
    const size_t orow = (((size_t)b * T + (tvalid ? to : 0)) * F) * kH + c4;

    constexpr int NN = (R * TC * 16 + 255) / 256, NH = (R * 3 * 16 + 255) / 256;
    float4 vn[NN];  // the block's new columns, raw.  Modes 0-2: requested one block AHEAD (under the previous block's window pass)
    auto is_interior = [&](int fb) { return t0 >= 1 && t0 + TR + 1 < T && fb >= 1 && fb + TC + 1 < F; };  // (workgroup-uniform)
    auto issue_new = [&](int fb) {
        // Blocks whose (16+3) x (TC+3) input window lies inside the tensor (3 of 4 at the headline shape) take a lean path: no clamping of the
        // load addresses, no per-element zero-padding select, the transform in native 4-vectors (common.h) - the staging code was ~45 % of this
        // kernel's VALU instructions (PMC, round 3: 44 M VALU per launch against 16.6 M for the convolution's FMAs; VALU 50 % busy).
        if (is_interior(fb)) {
            const unsigned boff = (((unsigned)(t0 - 1) * F + (fb - 1 + 3)) * kH) * 4u;
#pragma unroll
            for (int i = 0; i < NN; ++i) {
                const int r = i == NN - 1 ? min((int)(threadIdx.x / (TC * 16)) + RPI * i, R - 1) : (int)(threadIdx.x / (TC * 16)) + RPI * i;
                vn[i] = ld4_off(inb, boff + (((unsigned)r * F + ((threadIdx.x >> 4) % TC)) * kH + c4) * 4u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NN; ++i) {
                const int idx = threadIdx.x + i * 256, r = min(idx / (TC * 16), R - 1), c = 3 + ((idx >> 4) % TC);
                const int ti = t0 - 1 + r, fi = fb - 1 + c;
                vn[i] = ld4_off(inb, (((unsigned)min(max(ti, 0), T - 1) * F + min(max(fi, 0), F - 1)) * kH + c4) * 4u);
            }
        }
    };
    if (MODE != 3) issue_new(f0);
#pragma unroll 1
    for (int fb = f0; fb < f1; fb += TC) {
        // ---- stage input rows t0-1 .. t0+17, columns fb-1 .. fb+9 (transformed, zero outside the tensor).  Only the first block of
        // the segment fetches all 11 columns: afterwards the 3 halo columns fb-1 .. fb+1 are the previous block's last three, moved
        // inside LDS (read before the barrier, written after it), and 8 new columns come from HBM - 1.19x read amplification
        // (the t halo) instead of 1.63x.
        const bool first = fb == f0;
        float4 vh[NH];
        int moff = c4;
        asm volatile("" : "+v"(moff));  // opaque per block: the constants are re-read from LDS instead of living in 16 VGPRs
        auto xform = [&](float4 x, float4 g, float4 e, int ti, int fi) {
            if (MODE >= 1) x = fma4(x, sc, sh);
            if (MODE == 2) x = prelu4_minfma(x, a.slope - 1.0f);  // (the same form as the interior path below)
            if (MODE == 3) x = fma4(x, sigmoid4(fma4(g, ld4(mixc + moff), ld4(mixc + 64 + moff))), fma4(e, ld4(mixc + 128 + moff), ld4(mixc + 192 + moff)));
            if (!(ti >= 0 && ti < T && fi >= 0 && fi < F)) x = f4(0, 0, 0, 0);
            return x;
        };
        // MODE 3 fetches three tensors per element: after the barrier the new columns stream through registers in groups of NG elements
        // (3 NG loads in flight), each mixed and stored as soon as it arrives - the register file cannot hold all 3 x NN loads
        constexpr int NG = 4;
        auto fetch_new = [&](int i0, int i1) {
            float4 gn[MODE == 3 ? NN : 1], en[MODE == 3 ? NN : 1];
#pragma unroll
            for (int i = i0; i < i1; ++i) {  // new columns: block columns 3 .. 10
                const int idx = threadIdx.x + i * 256, r = min(idx / (TC * 16), R - 1), c = 3 + ((idx >> 4) % TC);
                const int ti = min(max(t0 - 1 + r, 0), T - 1), fi = min(max(fb - 1 + c, 0), F - 1);
                vn[i] = ld4_off(inb, (((unsigned)ti * F + fi) * kH + c4) * 4u);  // saddr + 32-bit offset
                if (MODE == 3) {
                    const unsigned og = (((unsigned)div_magic((unsigned)ti * a.Tg, T, a.mt) * a.Fg + div_magic((unsigned)fi * a.Fg, F, a.mf)) * kH + c4) * 4u;
                    gn[i] = ld4_off(gateb, og), en[i] = ld4_off(globb, og);
                }
            }
#pragma unroll
            for (int i = i0; i < i1; ++i) {
                const int idx = threadIdx.x + i * 256, r = idx / (TC * 16), c = 3 + ((idx >> 4) % TC);
                vn[i] = xform(vn[i], gn[MODE == 3 ? i : 0], en[MODE == 3 ? i : 0], t0 - 1 + r, fb - 1 + c);
            }
        };
        auto store_new = [&](int i0, int i1) {
#pragma unroll
            for (int i = i0; i < i1; ++i) {
                const int idx = threadIdx.x + i * 256, r = idx / (TC * 16), c = 3 + ((idx >> 4) % TC);
                if (idx < R * TC * 16) st4(tile + r * RS + c * 64 + c4, vn[i]);
            }
        };
        if (first) {
            float4 gh[MODE == 3 ? NH : 1], eh[MODE == 3 ? NH : 1];
#pragma unroll
            for (int i = 0; i < NH; ++i) {  // halo columns 0 .. 2 from memory
                const int idx = threadIdx.x + i * 256, r = min(idx / 48, R - 1), c = (idx % 48) >> 4;
                const int ti = min(max(t0 - 1 + r, 0), T - 1), fi = min(max(fb - 1 + c, 0), F - 1);
                vh[i] = ld4_off(inb, (((unsigned)ti * F + fi) * kH + c4) * 4u);
                if (MODE == 3) {
                    const unsigned og = (((unsigned)div_magic((unsigned)ti * a.Tg, T, a.mt) * a.Fg + div_magic((unsigned)fi * a.Fg, F, a.mf)) * kH + c4) * 4u;
                    gh[i] = ld4_off(gateb, og), eh[i] = ld4_off(globb, og);
                }
            }
            if (MODE == 3) {  // (modes 0-2 transform after the barrier, below: their loads stay in flight across it)
#pragma unroll
                for (int i = 0; i < NH; ++i) {
                    const int idx = threadIdx.x + i * 256, r = idx / 48, c = (idx % 48) >> 4;
                    vh[i] = xform(vh[i], gh[i], eh[i], t0 - 1 + r, fb - 1 + c);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NH; ++i) {  // ... or from the previous block's columns 8 .. 10 (already transformed / zero-padded)
                const int idx = threadIdx.x + i * 256, r = min(idx / 48, R - 1), c = (idx % 48) >> 4;
                vh[i] = ld4(tile + r * RS + (c + TC) * 64 + c4);
            }
        }
        const bool interior = is_interior(fb);
        __syncthreads();  // previous block's window reads are done
        if (MODE != 3) {
            if (interior) {
                const float4v scv = to_v4(sc), shv = to_v4(sh);
                const float am1 = a.slope - 1.0f;
#pragma unroll
                for (int i = 0; i < NN; ++i) {
                    float4v x = to_v4(vn[i]);
                    if (MODE >= 1) x = x * scv + shv;
                    if (MODE == 2) x = __builtin_elementwise_min(x, float4v{0.f, 0.f, 0.f, 0.f}) * am1 + x;  // prelu(x) = x + (a - 1) min(x, 0)
                    vn[i] = to_f4(x);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NN; ++i) {
                    const int idx = threadIdx.x + i * 256, r = idx / (TC * 16), c = 3 + ((idx >> 4) % TC);
                    vn[i] = xform(vn[i], f4(0, 0, 0, 0), f4(0, 0, 0, 0), t0 - 1 + r, fb - 1 + c);
                }
            }
            store_new(0, NN);
        } else {
#pragma unroll
            for (int i0 = 0; i0 < NN; i0 += NG) {
                fetch_new(i0, i0 + NG < NN ? i0 + NG : NN);
                store_new(i0, i0 + NG < NN ? i0 + NG : NN);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int idx = threadIdx.x + i * 256, r = idx / 48, c = (idx % 48) >> 4;
            if (idx < R * 3 * 16)
                st4(tile + r * RS + c * 64 + c4, (first && MODE != 3) ? xform(vh[i], f4(0, 0, 0, 0), f4(0, 0, 0, 0), t0 - 1 + r, fb - 1 + c) : vh[i]);
        }
        __syncthreads();
        if (MODE != 3 && fb + TC < f1) issue_new(fb + TC);  // the next block's new columns travel while this block's window pass runs
        // ---- 8 output columns, OPS = 4 per step (round 3): for each of the four tap rows the step reads the OPS + 3 window columns of that row
        // once and every tap once, and feeds OPS outputs from them - 11 ds_read_b128 per output quad instead of 20 (16 taps + 4 new window
        // entries per single output before).  With one output per step the kernel read 80 bytes of LDS per output float against 8 bytes of
        // HBM traffic: LDS issue, not HBM, set the pace of the compute phase.  Row-at-a-time keeps the window at 7 registers quads.
        constexpr int OPS = 4;
        const float* trow = tile + tr * RS + c4;
#pragma unroll 1
        for (int jb = 0; jb < TC; jb += OPS) {
            int woff = c4;
            asm volatile("" : "+v"(woff));  // re-read the taps from LDS here: hoisted out of the loops they pin 64 VGPRs per convolution (spills)
            float4v accv[NCONV][OPS];
#pragma unroll
            for (int k = 0; k < NCONV; ++k)
#pragma unroll
                for (int jj = 0; jj < OPS; ++jj) accv[k][jj] = to_v4(bias4[k]);
#pragma unroll 1
            for (int dt = 0; dt < 4; ++dt) {  // (rolled: unrolled, hipcc issues all 28 window and 16 NCONV tap reads first and spills)
                float4v wr[OPS + 3];
#pragma unroll
                for (int c = 0; c < OPS + 3; ++c) wr[c] = ld4v(trow + dt * RS + (jb + c) * 64);
#pragma unroll
                for (int df = 0; df < 4; ++df)
#pragma unroll
                    for (int k = 0; k < NCONV; ++k) {
                        const float4v tap = ld4v(&ws[k][(dt * 4 + df) * 64 + woff]);  // quad-broadcast LDS read, shared by the OPS outputs
#pragma unroll
                        for (int jj = 0; jj < OPS; ++jj) accv[k][jj] = tap * wr[jj + df] + accv[k][jj];  // native vectors: v_pk_fma_f32 without operand shuffles (common.h)
                    }
            }
            float4 acc[NCONV][OPS];
#pragma unroll
            for (int k = 0; k < NCONV; ++k)
#pragma unroll
                for (int jj = 0; jj < OPS; ++jj) acc[k][jj] = to_f4(accv[k][jj]);
#pragma unroll
            for (int jj = 0; jj < OPS; ++jj) {
                const int fo = fb + jb + jj;
                if (tvalid && fo < f1) {
#pragma unroll
                    for (int k = 0; k < NCONV; ++k) {
                        st4(a.out[k] + orow + (size_t)fo * kH, acc[k][jj]);
                        s[k] += acc[k][jj].x + acc[k][jj].y + acc[k][jj].z + acc[k][jj].w;
                        q[k] += acc[k][jj].x * acc[k][jj].x + acc[k][jj].y * acc[k][jj].y + acc[k][jj].z * acc[k][jj].z + acc[k][jj].w * acc[k][jj].w;
                    }
                    if constexpr (GADD)  // the transformed input pixel itself (tile row tr + 1, column jb + jj + 1) + the pooled term
                        st4(a.addout + orow + (size_t)fo * kH, ld4(trow + RS + (jb + jj + 1) * 64) + ld4(a.addsrc + orow + (size_t)fo * kH));
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NCONV; ++k) {
        __syncthreads();
        block_stats_commit(s[k], q[k], red, a.stats[k], b);
    }
}

// The three readers of gLN(D0) in one pass (tdanet.py:112-118, fusion.py:25-52 of fusion_layers[0]): the stride-1 'same' convolution that
// makes the TFAR local embedding l0, the stride-2 convolution that makes the next pyramid level D1, and the adaptive average pooling of
// gLN(D0) onto D1's grid.  Staging is dwconv_s1_kernel<1, 1>'s (16 x 8 outputs from a (16+3) x (8+3) pixel LDS tile, gLN and the zero padding
// applied once on the way in); after the sliding-window pass a second pass reads the SAME tile: the 8 x 4 stride-2 outputs whose 4 x 4 windows
// (rows 2 t2 - 1 .. 2 t2 + 2, columns 2 f2 - 1 .. 2 f2 + 2: padding 1 = the tile's zeros) and pooling windows (rows 2 t2 .. 2 t2 + nt - 1,
// nt = 3 for odd T and 2 for even T; columns 2 f2 .. 2 f2 + 2 for 129 -> 64 bins) lie inside it because tiles start on even rows / columns.
// D0 is read once (1.19x with the t halo) instead of three times (stride-2 kernel 2x through L1, pooling 1x, stride-1 1.19x).
struct TrioArgs {
    const float* in;     // [B][T][129][64]
    const double* slot;  // gLN statistics of `in`
    double inv_n;
    const float *gamma, *beta;
    int T, T2;
    const float* w1;  // stride 1, no bias
    float* out1;      // [B][T][129][64]
    double* stats1;
    const float *w2, *bias2;  // stride 2
    float* out2;              // [B][T2][64][64]
    double* stats2;
    float* pooled;  // [B][T2][64][64]
};

__global__ __launch_bounds__(256, 3) void dwconv_trio_kernel(TrioArgs a, int fseg) {  // (4-column blocks, three workgroups per CU: see dwconv_s1_kernel)
    constexpr int TR = 16, TC = 4, R = TR + 3, CB = TC + 3;
    constexpr int RS = CB * 64;
    __shared__ __attribute__((aligned(16))) float tile[R * RS];
    __shared__ __attribute__((aligned(16))) float ws[2][16 * 64];
    __shared__ float red[8];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < 2 * 256; i += 256) {
        const int j = i >> 8, o = (i & 255) * 4;
        st4(&ws[j][o], ld4((j ? a.w2 : a.w1) + o));
    }
    const int c4 = (threadIdx.x & 15) * 4, tr = threadIdx.x >> 4;
    const int T = a.T, T2 = a.T2;
    constexpr int F = kF;
    const int t0 = blockIdx.x * TR, to = t0 + tr;
    const int f0 = blockIdx.z * fseg, f1 = min(F, f0 + fseg);
    float4 sc, sh;
    {
        float mean, rstd;
        stats_finalize(a.slot, b, a.inv_n, mean, rstd);
        const float4 g = ld4(a.gamma + c4), be = ld4(a.beta + c4);
        sc = g * rstd;
        sh = f4(be.x - mean * sc.x, be.y - mean * sc.y, be.z - mean * sc.z, be.w - mean * sc.w);
    }
    const float* inb = a.in + (size_t)b * T * F * kH;
    const bool tvalid = to < T;
    float s1 = 0.f, q1 = 0.f, s2 = 0.f, q2 = 0.f;
    __syncthreads();
    const float4 bias2 = ld4(a.bias2 + c4);
    const size_t orow = (((size_t)b * T + (tvalid ? to : 0)) * F) * kH + c4;
    // second pass: this thread's stride-2 outputs (lt, 2 lp) and (lt, 2 lp + 1) of the tile
    const int lt = tr >> 1, lp = tr & 1;
    const int t2 = (t0 >> 1) + lt;
    const bool t2valid = t2 < T2;
    const float mt3 = (T & 1) ? 1.f : 0.f;                    // pooling window rows: 2 t2 .. 2 t2 + 1 (+ 2 t2 + 2 for odd T)
    const float pinv = 1.0f / (((T & 1) ? 3.f : 2.f) * 3.f);  // x 3 columns
    const size_t orow2 = (((size_t)b * T2 + (t2valid ? t2 : 0)) * kF2) * kH + c4;

#pragma unroll 1
    for (int fb = f0; fb < f1; fb += TC) {
        constexpr int NN = (R * TC * 16 + 255) / 256, NH = (R * 3 * 16 + 255) / 256;
        const bool first = fb == f0;
        float4 vn[NN], vh[NH];
        auto xform = [&](float4 x, int ti, int fi) {
            x = fma4(x, sc, sh);
            if (!(ti >= 0 && ti < T && fi >= 0 && fi < F)) x = f4(0, 0, 0, 0);
            return x;
        };
        if (first) {
#pragma unroll
            for (int i = 0; i < NH; ++i) {  // halo columns 0 .. 2 from memory
                const int idx = threadIdx.x + i * 256, r = min(idx / 48, R - 1), c = (idx % 48) >> 4;
                const int ti = min(max(t0 - 1 + r, 0), T - 1), fi = min(max(fb - 1 + c, 0), F - 1);
                vh[i] = ld4_off(inb, (((unsigned)ti * F + fi) * kH + c4) * 4u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NH; ++i) {  // ... or from the previous block's columns 8 .. 10 (already transformed / zero-padded)
                const int idx = threadIdx.x + i * 256, r = min(idx / 48, R - 1), c = (idx % 48) >> 4;
                vh[i] = ld4(tile + r * RS + (c + TC) * 64 + c4);
            }
        }
#pragma unroll
        for (int i = 0; i < NN; ++i) {
            const int idx = threadIdx.x + i * 256, r = min(idx / (TC * 16), R - 1), c = 3 + ((idx >> 4) % TC);
            const int ti = t0 - 1 + r, fi = fb - 1 + c;
            vn[i] = ld4_off(inb, (((unsigned)min(max(ti, 0), T - 1) * F + min(max(fi, 0), F - 1)) * kH + c4) * 4u);
        }
        __syncthreads();  // previous block's window reads are done
#pragma unroll
        for (int i = 0; i < NN; ++i) {
            const int idx = threadIdx.x + i * 256, r = idx / (TC * 16), c = 3 + ((idx >> 4) % TC);
            if (idx < R * TC * 16) st4(tile + r * RS + c * 64 + c4, xform(vn[i], t0 - 1 + r, fb - 1 + c));
        }
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int idx = threadIdx.x + i * 256, r = idx / 48, c = (idx % 48) >> 4;
            if (idx < R * 3 * 16) st4(tile + r * RS + c * 64 + c4, first ? xform(vh[i], t0 - 1 + r, fb - 1 + c) : vh[i]);
        }
        __syncthreads();
        // ---- pass 1: 8 stride-1 output columns, 4 per step: per tap row the 7 window columns and the 4 taps are read once and feed 4 outputs
        // (11 ds_read_b128 per output quad instead of 20; see dwconv_s1_kernel)
        {
            const float* trow = tile + tr * RS + c4;
#pragma unroll 1
            for (int jb = 0; jb < TC; jb += 4) {
                int woff = c4;
                asm volatile("" : "+v"(woff));  // taps re-read from LDS per step (hoisted they pin 64 VGPRs)
                float4v accv[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) accv[jj] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int dt = 0; dt < 4; ++dt) {
                    float4v wr[7];
#pragma unroll
                    for (int c = 0; c < 7; ++c) wr[c] = ld4v(trow + dt * RS + (jb + c) * 64);
#pragma unroll
                    for (int df = 0; df < 4; ++df) {
                        const float4v tap = ld4v(&ws[0][(dt * 4 + df) * 64 + woff]);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) accv[jj] = tap * wr[jj + df] + accv[jj];  // native vectors: v_pk_fma_f32 without shuffles (common.h)
                    }
                }
                float4 acc[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc[jj] = to_f4(accv[jj]);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int fo = fb + jb + jj;
                    if (tvalid && fo < f1) {
                        st4(a.out1 + orow + (size_t)fo * kH, acc[jj]);
                        s1 += acc[jj].x + acc[jj].y + acc[jj].z + acc[jj].w;
                        q1 += acc[jj].x * acc[jj].x + acc[jj].y * acc[jj].y + acc[jj].z * acc[jj].z + acc[jj].w * acc[jj].w;
                    }
                }
            }
        }
        // ---- pass 2: stride-2 convolution + pooling at (t2, f2), f2 = fb / 2 + 2 lp + {0, 1}: tile rows 2 lt .. 2 lt + 3, columns 4 lp .. 4 lp + 5
        {
            int woff = c4;
            asm volatile("" : "+v"(woff));
            const float* base = tile + (2 * lt) * RS + ((TC / 2) * lp) * 64 + c4;
#pragma unroll 1
            for (int o = 0; o < TC / 4; ++o) {  // (rolled: one 4 x 4 window of registers at a time)
                float4v x[4][4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[c][r] = ld4v(base + r * RS + (2 * o + c) * 64);
                float4v accv = to_v4(bias2);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int df = 0; df < 4; ++df) accv = ld4v(&ws[1][(dt * 4 + df) * 64 + woff]) * x[df][dt] + accv;
                // pooling window: rows dt = 1, 2 (, 3), columns df = 1, 2, 3 of the same 4 x 4 window
                float4v rs[3];
#pragma unroll
                for (int dt = 1; dt < 4; ++dt) rs[dt - 1] = x[1][dt] + x[2][dt] + x[3][dt];
                const float4 acc = to_f4(accv), pool = to_f4((rs[0] + rs[1] + rs[2] * mt3) * pinv);
                const int f2 = (fb >> 1) + (TC / 4) * lp + o;
                if (t2valid && f2 < kF2 && 2 * f2 < f1) {
                    st4(a.out2 + orow2 + (size_t)f2 * kH, acc);
                    st4(a.pooled + orow2 + (size_t)f2 * kH, pool);
                    s2 += acc.x + acc.y + acc.z + acc.w;
                    q2 += acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
                }
            }
        }
    }
    __syncthreads();
    block_stats_commit(s1, q1, red, a.stats1, b);
    __syncthreads();
    block_stats_commit(s2, q2, red, a.stats2, b);
}

// G = pooled + gLN(D1)  (the second half of rtfs_pool_fwd when dwconv_trio_kernel has produced the pooled term)
__global__ __launch_bounds__(256) void pool_add_kernel(const float* __restrict__ pooled, NormRefLite d1, float* __restrict__ G, int n_pix) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (p >= n_pix) return;
    const int c4 = (threadIdx.x & 15) * 4;
    float4 sc1, sh1;
    norm_coef(d1, b, c4, sc1, sh1);
    const size_t o = ((size_t)b * n_pix + p) * kH + c4;
    st4(G + o, ld4(pooled + o) + fma4(ld4(d1.x + o), sc1, sh1));
}

// G = adaptive_avg_pool2d(gLN(D0p) -> (T2,F2)) + gLN(D1p); window [floor(i*in/out), ceil((i+1)*in/out)).
__global__ __launch_bounds__(256) void pool_kernel(NormRefLite d0, NormRefLite d1, float* __restrict__ G, int T, int T2) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (p >= T2 * kF2) return;
    const int c4 = (threadIdx.x & 15) * 4;
    const int t2 = p / kF2, f2 = p - t2 * kF2;
    float4 sc0, sh0, sc1, sh1;
    norm_coef(d0, b, c4, sc0, sh0);
    norm_coef(d1, b, c4, sc1, sh1);
    const int ts = (t2 * T) / T2, te = ((t2 + 1) * T + T2 - 1) / T2;
    const int fs = (f2 * kF) / kF2, fe = ((f2 + 1) * kF + kF2 - 1) / kF2;
    float4 s = f4(0, 0, 0, 0);
    for (int t = ts; t < te; ++t)
        for (int f = fs; f < fe; ++f) s = s + ld4(d0.x + (((size_t)b * T + t) * kF + f) * kH + c4);
    const float inv = 1.0f / (float)((te - ts) * (fe - fs));
    const size_t o = ((size_t)b * T2 * kF2 + p) * kH + c4;
    st4(G + o, fma4(s * inv, sc0, sh0) + fma4(ld4(d1.x + o), sc1, sh1));
}

// out[p] = gLN(loc)[p] * sigmoid(gLN(gate)[up(p)]) + gLN(glob)[up(p)]; loc at (T,F), gate/glob at (Tg,Fg).
__global__ __launch_bounds__(256) void tfar_mix_kernel(NormRefLite loc, NormRefLite gate, NormRefLite glob, float* __restrict__ out, int T, int F,
                                                       int Tg, int Fg) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (p >= T * F) return;
    const int c4 = (threadIdx.x & 15) * 4;
    const int t = p / F, f = p - t * F;
    const int tg = nearest_src(t, Tg, T), fg = nearest_src(f, Fg, F);
    float4 scl, shl, scg, shg, sce, she;
    norm_coef(loc, b, c4, scl, shl);
    norm_coef(gate, b, c4, scg, shg);
    norm_coef(glob, b, c4, sce, she);
    const size_t o = ((size_t)b * T * F + p) * kH + c4;
    const size_t og = (((size_t)b * Tg + tg) * Fg + fg) * kH + c4;
    const float4 l = fma4(ld4(loc.x + o), scl, shl);
    const float4 g = sigmoid4(fma4(ld4(gate.x + og), scg, shg));
    const float4 e = fma4(ld4(glob.x + og), sce, she);
    st4(out + o, fma4(l, g, e));
}

// One pass over the positions of an utterance's video tensor vb [512][Tv] (contiguous) for the CAF video kernels: thread c receives
// body(t, vb[2c][t], vb[2c + 1][t]) for t = 0 .. Tv - 1 in order.  The rows are staged through LDS in nh groups of R = 512 / nh rows: coalesced
// global reads (the group is one contiguous R x Tv block), transposed into X[t][row] (row stride R + 2 floats: 8-byte aligned pairs, 2-way
// conflicts on the staging writes only), then the 256 / nh threads that own the group's rows walk t.  Dynamic LDS: Tv (R + 2) floats.
template <class Body>
__device__ __forceinline__ void caf_staged_sweep(const float* __restrict__ vb, int Tv, int nh, int c, Body&& body) {
    extern __shared__ __attribute__((aligned(16))) float caf_x[];
    const int R = 512 / nh, LDX = R + 2, per = 256 / nh;
    for (int h = 0; h < nh; ++h) {
        __syncthreads();  // the previous group / pass has been consumed
        const float* src = vb + (size_t)h * R * Tv;
        for (int idx = threadIdx.x; idx < R * Tv; idx += 256) {
            const int ch = idx / Tv, t = idx - ch * Tv;
            caf_x[t * LDX + ch] = src[idx];
        }
        __syncthreads();
        if (c / per == h) {
            const float* xp = caf_x + 2 * (c - h * per);
            for (int t = 0; t < Tv; ++t) {
                const float2 x = *reinterpret_cast<const float2*>(xp + t * LDX);
                body(t, x.x, x.y);
            }
        }
    }
}
// number of channel groups for caf_staged_sweep (0: the tensor does not fit in <= 8 groups - direct reads); LDS budget 150 KB
static int caf_groups(int Tv) {
    for (int nh = 1; nh <= 8; nh *= 2)
        if ((size_t)Tv * (512 / nh + 2) * sizeof(float) <= 150 * 1024) return nh;
    return 0;
}

// Video side of the CAF cell, one workgroup per utterance, thread = audio channel c (256).
//   v: [B][512][Tv] (NCW, as produced by the VP block).  att_w [1024][2], att_b/att_g/att_be [1024]; rs_w [256][2], rs_b/rs_g/rs_be [256].
//   att_out, rsz_out: [B][Tv][256]
__global__ __launch_bounds__(256) void caf_video_kernel(const float* __restrict__ v, const float* __restrict__ att_w, const float* __restrict__ att_b,
                                                        const float* __restrict__ att_g, const float* __restrict__ att_be,
                                                        const float* __restrict__ rs_w, const float* __restrict__ rs_b, const float* __restrict__ rs_g,
                                                        const float* __restrict__ rs_be, float* __restrict__ att_out, float* __restrict__ rsz_out,
                                                        int Tv, int nh) {
    __shared__ float red[16];
    const int b = blockIdx.x, c = threadIdx.x, lane = c & 63, w = c >> 6;
    const float* v0 = v + ((size_t)b * 512 + 2 * c) * Tv;
    const float* v1 = v0 + Tv;
    float aw0[4], aw1[4], ab[4], ag[4], abe[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int o = c * 4 + h;
        aw0[h] = att_w[2 * o], aw1[h] = att_w[2 * o + 1], ab[h] = att_b[o], ag[h] = att_g[o], abe[h] = att_be[o];
    }
    const float rw0 = rs_w[2 * c], rw1 = rs_w[2 * c + 1], rb = rs_b[c], rg = rs_g[c], rbe = rs_be[c];

    // Every pass walks the thread's two input rows.  Thread c's rows 2c, 2c + 1 are Tv floats apart from its neighbours': read directly, every
    // wave-load touches 64 cache lines and the kernel is bound by the texture-address unit (540 us; batching 8 positions per latency: 487 us).
    // Round 3: caf_staged_sweep - the utterance's [512][Tv] block is contiguous, so it is copied into LDS with perfectly coalesced loads,
    // transposed ([t][row]), in nh channel groups that fit the LDS, and the passes read their (x0, x1) pairs as one conflict-free 8-byte LDS
    // read.  nh = 0 (the block would need more than 8 groups: Tv > ~580): the direct reads, 8 positions per batch with the loads first.
    auto sweep = [&](auto&& body) {
        if (nh > 0) {
            caf_staged_sweep(v + (size_t)b * 512 * Tv, Tv, nh, c, [&](int t, float x0, float x1) { body(t, x0, x1); });
            return;
        }
#pragma unroll 1
        for (int t0 = 0; t0 < Tv; t0 += 8) {
            float a0[8], a1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = min(t0 + j, Tv - 1);
                a0[j] = v0[t], a1[j] = v1[t];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (t0 + j < Tv) body(t0 + j, a0[j], a1[j]);
        }
    };
    // pass 1: gLN statistics of both pre-norm tensors (two-pass: mean first, then centred squares)
    float s_att = 0.f, s_rs = 0.f;
    sweep([&](int, float x0, float x1) {
#pragma unroll
        for (int h = 0; h < 4; ++h) s_att += fmaf(aw0[h], x0, fmaf(aw1[h], x1, ab[h]));
        s_rs += fmaf(rw0, x0, fmaf(rw1, x1, rb));
    });
    s_att = wave_sum(s_att), s_rs = wave_sum(s_rs);
    if (lane == 0) red[w] = s_att, red[4 + w] = s_rs;
    __syncthreads();
    const float m_att = (red[0] + red[1] + red[2] + red[3]) / (1024.f * Tv);
    const float m_rs = (red[4] + red[5] + red[6] + red[7]) / (256.f * Tv);
    float q_att = 0.f, q_rs = 0.f;
    sweep([&](int, float x0, float x1) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const float d = fmaf(aw0[h], x0, fmaf(aw1[h], x1, ab[h])) - m_att;
            q_att = fmaf(d, d, q_att);
        }
        const float d = fmaf(rw0, x0, fmaf(rw1, x1, rb)) - m_rs;
        q_rs = fmaf(d, d, q_rs);
    });
    q_att = wave_sum(q_att), q_rs = wave_sum(q_rs);
    if (lane == 0) red[8 + w] = q_att, red[12 + w] = q_rs;
    __syncthreads();
    const float r_att = 1.0f / sqrtf((red[8] + red[9] + red[10] + red[11]) / (1024.f * Tv) + kEps);
    const float r_rs = 1.0f / sqrtf((red[12] + red[13] + red[14] + red[15]) / (256.f * Tv) + kEps);

    // pass 2: head mean -> softmax over Tv (max, sum, write); resize branch written directly
    auto att_of = [&](float x0, float x1) {
        float m = 0.f;
#pragma unroll
        for (int h = 0; h < 4; ++h) m += fmaf((fmaf(aw0[h], x0, fmaf(aw1[h], x1, ab[h])) - m_att) * r_att, ag[h], abe[h]);
        return m * 0.25f;
    };
    float mx = -3.0e38f;
    sweep([&](int, float x0, float x1) { mx = fmaxf(mx, att_of(x0, x1)); });
    float sum = 0.f;
    sweep([&](int, float x0, float x1) { sum += __expf(att_of(x0, x1) - mx); });
    const float inv = 1.0f / sum;
    sweep([&](int t, float x0, float x1) {
        const size_t o = ((size_t)b * Tv + t) * 256 + c;
        att_out[o] = __expf(att_of(x0, x1) - mx) * inv;
        rsz_out[o] = fmaf((fmaf(rw0, x0, fmaf(rw1, x1, rb)) - m_rs) * r_rs, rg, rbe);
    });
}

// Adjoint of caf_video_kernel (training step; the forward kernel is mode-independent: gLN only, no BatchNorm on the video side).
// One workgroup per utterance, thread = audio channel c = convolution group c: it owns input rows 2c, 2c + 1, the four attention-embedding
// channels 4c .. 4c + 3 and resize channel c, recomputes their forward values from v (two block-wide gLN statistics, one softmax row) and
// walks its Tv positions three times: softmax dot, gLN adjoint sums, input / parameter gradients.  Parameter gradients of the B workgroups
// are added into the caller's (zeroed) buffers with one atomic per value.
//   datt, drsz: [B][Tv][256];  dv: [B][512][Tv] (written);  d_att_w [1024][2], d_att_b / d_att_g / d_att_be [1024], d_rs_w [256][2], d_rs_b / d_rs_g / d_rs_be [256]
__global__ __launch_bounds__(256) void caf_video_bwd_kernel(const float* __restrict__ v, const float* __restrict__ att_w, const float* __restrict__ att_b,
                                                            const float* __restrict__ att_g, const float* __restrict__ att_be,
                                                            const float* __restrict__ rs_w, const float* __restrict__ rs_b, const float* __restrict__ rs_g,
                                                            const float* __restrict__ rs_be, const float* __restrict__ datt, const float* __restrict__ drsz,
                                                            float* __restrict__ dv, float* __restrict__ d_att_w, float* __restrict__ d_att_b,
                                                            float* __restrict__ d_att_g, float* __restrict__ d_att_be, float* __restrict__ d_rs_w,
                                                            float* __restrict__ d_rs_b, float* __restrict__ d_rs_g, float* __restrict__ d_rs_be, int Tv, int nh) {
    __shared__ float red[16];
    const int b = blockIdx.x, c = threadIdx.x, lane = c & 63, w = c >> 6;
    const float* v0 = v + ((size_t)b * 512 + 2 * c) * Tv;
    const float* v1 = v0 + Tv;
    float aw0[4], aw1[4], ab[4], ag[4], abe[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int o = c * 4 + h;
        aw0[h] = att_w[2 * o], aw1[h] = att_w[2 * o + 1], ab[h] = att_b[o], ag[h] = att_g[o], abe[h] = att_be[o];
    }
    const float rw0 = rs_w[2 * c], rw1 = rs_w[2 * c + 1], rb = rs_b[c], rg = rs_g[c];
    auto block4 = [&](float a, float b_, float& ra, float& rb_) {  // two block-wide sums
        a = wave_sum(a), b_ = wave_sum(b_);
        __syncthreads();
        if (lane == 0) red[w] = a, red[4 + w] = b_;
        __syncthreads();
        ra = red[0] + red[1] + red[2] + red[3], rb_ = red[4] + red[5] + red[6] + red[7];
    };
    const float* dab = datt + (size_t)b * Tv * 256 + c;
    const float* drb = drsz + (size_t)b * Tv * 256 + c;
    // batches of 8 positions, all loads of a batch first (see caf_video_kernel); GRADS: the two upstream gradients ride along
    auto sweep = [&](auto grads, auto&& body) {
        constexpr bool G = decltype(grads)::value;
        if (nh > 0) {  // (the upstream gradients are read [t][c]: coalesced across the threads as they are)
            caf_staged_sweep(v + (size_t)b * 512 * Tv, Tv, nh, c, [&](int t, float x0, float x1) {
                body(t, x0, x1, G ? dab[(size_t)t * 256] : 0.f, G ? drb[(size_t)t * 256] : 0.f);
            });
            return;
        }
#pragma unroll 1
        for (int t0 = 0; t0 < Tv; t0 += 8) {
            float a0[8], a1[8], ga[G ? 8 : 1], gr[G ? 8 : 1];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = min(t0 + j, Tv - 1);
                a0[j] = v0[t], a1[j] = v1[t];
                if constexpr (G) ga[j] = dab[(size_t)t * 256], gr[j] = drb[(size_t)t * 256];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (t0 + j < Tv) body(t0 + j, a0[j], a1[j], ga[G ? j : 0], gr[G ? j : 0]);
        }
    };
    using NoG = std::integral_constant<bool, false>;
    using WithG = std::integral_constant<bool, true>;
    // forward statistics, as the forward kernel computes them (mean first, then centred squares)
    float s_att = 0.f, s_rs = 0.f;
    sweep(NoG{}, [&](int, float x0, float x1, float, float) {
#pragma unroll
        for (int h = 0; h < 4; ++h) s_att += fmaf(aw0[h], x0, fmaf(aw1[h], x1, ab[h]));
        s_rs += fmaf(rw0, x0, fmaf(rw1, x1, rb));
    });
    float m_att, m_rs;
    block4(s_att, s_rs, m_att, m_rs);
    const float n_att = 1024.f * Tv, n_rs = 256.f * Tv;
    m_att /= n_att, m_rs /= n_rs;
    float q_att = 0.f, q_rs = 0.f;
    sweep(NoG{}, [&](int, float x0, float x1, float, float) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const float d = fmaf(aw0[h], x0, fmaf(aw1[h], x1, ab[h])) - m_att;
            q_att = fmaf(d, d, q_att);
        }
        const float d = fmaf(rw0, x0, fmaf(rw1, x1, rb)) - m_rs;
        q_rs = fmaf(d, d, q_rs);
    });
    float r_att, r_rs;
    block4(q_att, q_rs, r_att, r_rs);
    r_att = 1.0f / sqrtf(r_att / n_att + kEps), r_rs = 1.0f / sqrtf(r_rs / n_rs + kEps);
    auto att_of = [&](float x0, float x1) {
        float m = 0.f;
#pragma unroll
        for (int h = 0; h < 4; ++h) m += fmaf((fmaf(aw0[h], x0, fmaf(aw1[h], x1, ab[h])) - m_att) * r_att, ag[h], abe[h]);
        return m * 0.25f;
    };
    float mx = -3.0e38f;
    sweep(NoG{}, [&](int, float x0, float x1, float, float) { mx = fmaxf(mx, att_of(x0, x1)); });
    float sum = 0.f;
    sweep(NoG{}, [&](int, float x0, float x1, float, float) { sum += __expf(att_of(x0, x1) - mx); });
    const float inv = 1.0f / sum;
    // softmax adjoint: d(head mean)_t = p_t (dp_t - sum_t' p_t' dp_t')
    float sdot = 0.f;
    sweep(WithG{}, [&](int, float x0, float x1, float da, float) { sdot = fmaf(__expf(att_of(x0, x1) - mx) * inv, da, sdot); });
    // gLN adjoint sums: S1 = sum u, S2 = sum u xhat (u = dy gamma) over the whole utterance; d gamma / d beta of this thread's channels
    float s1a = 0.f, s2a = 0.f, s1r = 0.f, s2r = 0.f, dga[4] = {0.f, 0.f, 0.f, 0.f}, dba = 0.f, dgr = 0.f, dbr = 0.f;
    sweep(WithG{}, [&](int, float x0, float x1, float da, float dr) {
        const float dy = __expf(att_of(x0, x1) - mx) * inv * (da - sdot) * 0.25f;  // the same for the four heads (mean over them)
        dba += dy;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const float xh = (fmaf(aw0[h], x0, fmaf(aw1[h], x1, ab[h])) - m_att) * r_att;
            dga[h] = fmaf(dy, xh, dga[h]);
            s1a = fmaf(dy, ag[h], s1a), s2a = fmaf(dy * ag[h], xh, s2a);
        }
        const float xr = (fmaf(rw0, x0, fmaf(rw1, x1, rb)) - m_rs) * r_rs;
        dgr = fmaf(dr, xr, dgr), dbr += dr;
        s1r = fmaf(dr, rg, s1r), s2r = fmaf(dr * rg, xr, s2r);
    });
    float S1a, S2a, S1r, S2r;
    block4(s1a, s2a, S1a, S2a);
    block4(s1r, s2r, S1r, S2r);
    S1a /= n_att, S2a /= n_att, S1r /= n_rs, S2r /= n_rs;
    // input and convolution-parameter gradients
    float dw0[4] = {0.f, 0.f, 0.f, 0.f}, dw1[4] = {0.f, 0.f, 0.f, 0.f}, dbb[4] = {0.f, 0.f, 0.f, 0.f}, drw0 = 0.f, drw1 = 0.f, drb_ = 0.f;
    float* dv0 = dv + ((size_t)b * 512 + 2 * c) * Tv;
    float* dv1 = dv0 + Tv;
    sweep(WithG{}, [&](int t, float x0, float x1, float da, float dr) {
        const float dy = __expf(att_of(x0, x1) - mx) * inv * (da - sdot) * 0.25f;
        float g0 = 0.f, g1 = 0.f;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const float xh = (fmaf(aw0[h], x0, fmaf(aw1[h], x1, ab[h])) - m_att) * r_att;
            const float draw = r_att * (dy * ag[h] - S1a - xh * S2a);
            dw0[h] = fmaf(draw, x0, dw0[h]), dw1[h] = fmaf(draw, x1, dw1[h]), dbb[h] += draw;
            g0 = fmaf(aw0[h], draw, g0), g1 = fmaf(aw1[h], draw, g1);
        }
        const float xr = (fmaf(rw0, x0, fmaf(rw1, x1, rb)) - m_rs) * r_rs;
        const float draw = r_rs * (dr * rg - S1r - xr * S2r);
        drw0 = fmaf(draw, x0, drw0), drw1 = fmaf(draw, x1, drw1), drb_ += draw;
        dv0[t] = fmaf(rw0, draw, g0);
        dv1[t] = fmaf(rw1, draw, g1);
    });
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int o = c * 4 + h;
        atomicAdd(d_att_w + 2 * o, dw0[h]), atomicAdd(d_att_w + 2 * o + 1, dw1[h]), atomicAdd(d_att_b + o, dbb[h]);
        atomicAdd(d_att_g + o, dga[h]), atomicAdd(d_att_be + o, dba);
    }
    atomicAdd(d_rs_w + 2 * c, drw0), atomicAdd(d_rs_w + 2 * c + 1, drw1), atomicAdd(d_rs_b + c, drb_);
    atomicAdd(d_rs_g + c, dgr), atomicAdd(d_rs_be + c, dbr);
}

// out = relu(x*ks+kb) * rsz[b][tv(t)] + att[b][tv(t)] * (x*vs+vb) [+ a0];  x,out,a0: [B][T][F][256]; tv(t) = floor(t*Tv/T)
__global__ __launch_bounds__(256) void caf_fuse_kernel(const float* __restrict__ x, const float* __restrict__ ks, const float* __restrict__ kb,
                                                       const float* __restrict__ vs, const float* __restrict__ vb, const float* __restrict__ att,
                                                       const float* __restrict__ rsz, const float* __restrict__ a0, float* __restrict__ out, int T,
                                                       int Tv) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= T * kF) return;
    const int c4 = (threadIdx.x & 63) * 4;
    const int t = p / kF;
    const int tv = nearest_src(t, Tv, T);
    const size_t o = ((size_t)b * T * kF + p) * kC + c4;
    const size_t ov = ((size_t)b * Tv + tv) * kC + c4;
    const float4 xv = ld4(x + o);
    const float4 k1 = relu4(fma4(xv, ld4(ks + c4), ld4(kb + c4))) * ld4(rsz + ov);
    const float4 vv = fma4(xv, ld4(vs + c4), ld4(vb + c4));
    float4 r = fma4(ld4(att + ov), vv, k1);
    if (a0) r = r + ld4(a0 + o);
    st4(out + o, r);
}

}  // namespace rtfs

using namespace rtfs;

static int caf_set_lds(const void* fn, bool* flags) {  // dynamic LDS beyond 64 KB needs the attribute, once per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return RTFS_ELAUNCH;
    if (!flags[dev]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess) return RTFS_ELAUNCH;
        flags[dev] = true;
    }
    return RTFS_OK;
}

template <int STRIDE, int MODE>
static int launch_dw(const DwArgs& a, int B, hipStream_t st) {
    if constexpr (STRIDE == 1) {  // LDS-staged kernel, f segments in multiples of its 8-column block
        // f segments per row tile: 4 at full resolution (2048 workgroups, 2.7 rounds of the 768 resident ones); at the compressed resolution 3 for the
        // 4-column kernels (8 x 32 x 3 = 768 workgroups = one full round of three per CU; 2 left a third of the slots empty: 79 -> 71 us, 35 -> 31 us)
        int nseg = a.Fout >= 96 ? 4 : (a.nconv <= 2 ? 3 : 2);
        // small batches: a workgroup walks its segment's column blocks one after the other (a latency chain per block); with fewer than ~512
        // workgroups the segments shrink to one 8-column piece at the least, which puts the chain's links side by side on idle CUs instead
        const long long wg0 = (long long)((a.Tout + 15) / 16) * B;
        if (wg0 * nseg < 512) nseg = (int)std::min<long long>((a.Fout + 7) / 8, (512 + wg0 - 1) / wg0);
        const int fseg = (((a.Fout + nseg - 1) / nseg) + 7) / 8 * 8;  // (a multiple of the kernel's column block, 8 or 4)
        dim3 grid((a.Tout + 15) / 16, B, (a.Fout + fseg - 1) / fseg);
        switch (a.nconv) {
            case 1: hipLaunchKernelGGL((dwconv_s1_kernel<1, MODE>), grid, dim3(256), 0, st, a, fseg); break;
            case 2: hipLaunchKernelGGL((dwconv_s1_kernel<2, MODE>), grid, dim3(256), 0, st, a, fseg); break;
            case 4: hipLaunchKernelGGL((dwconv_s1_kernel<4, MODE>), grid, dim3(256), 0, st, a, fseg); break;
            default: return RTFS_EINVAL;
        }
        RTFS_LAUNCH_CHECK();
        return RTFS_OK;
    } else {  // stride 2: downsample_layers[1] alone (tdanet.py:112-114) - one convolution per launch
        if (a.nconv != 1) return RTFS_EINVAL;
        const int nseg = a.Fout >= 96 ? 4 : 2;
        const int fseg = (((a.Fout + nseg - 1) / nseg) + 3) / 4 * 4;
        dim3 grid((a.Tout + 15) / 16, B, (a.Fout + fseg - 1) / fseg);
        hipLaunchKernelGGL((dwconv_kernel<STRIDE, 1, MODE>), grid, dim3(256), 0, st, a, fseg);
        RTFS_LAUNCH_CHECK();
        return RTFS_OK;
    }
}

extern "C" {

// nconv in {1,2,4} depth-wise 4x4 convolutions of the same (optionally normalised) input.
// mode 0: raw input, 1: gLN(in), 2: PReLU(gLN(in)).  stride 1 -> output (Tin,Fin); stride 2 -> ((Tin-2)/2+1, (Fin-2)/2+1).
// w[j]: [16][64] (tap-major), bias[j]: [64] or NULL, out[j]: [B][Tout][Fout][64], stats_out[j]: [B][2] (accumulated into).
int rtfs_dwconv_fwd(const float* in, const double* stats_in, const float* gamma, const float* beta, float slope, int mode, int stride, int nconv,
                    const float* const* w, const float* const* bias, float* const* out, double* const* stats_out, int B, int Tin, int Fin,
                    void* stream) {
    if (B <= 0 || nconv < 1 || nconv > kMaxConv || (stride != 1 && stride != 2) || mode < 0 || mode > 2) return RTFS_EINVAL;
    DwArgs a;
    a.in = in, a.slot = stats_in, a.inv_n = 1.0 / ((double)Tin * Fin * kH), a.gamma = gamma, a.beta = beta, a.slope = slope;
    a.Tin = Tin, a.Fin = Fin;
    a.Tout = stride == 1 ? Tin : (Tin - 2) / 2 + 1;
    a.Fout = stride == 1 ? Fin : (Fin - 2) / 2 + 1;
    a.nconv = nconv;
    for (int j = 0; j < kMaxConv; ++j) {
        a.w[j] = j < nconv ? w[j] : nullptr;
        a.bias[j] = j < nconv ? bias[j] : nullptr;
        a.out[j] = j < nconv ? out[j] : nullptr;
        a.stats[j] = j < nconv ? stats_out[j] : nullptr;
    }
    a.gate = a.glob = NormRefLite{nullptr, nullptr, 0.0, nullptr, nullptr};
    a.Tg = a.Fg = 0, a.mt = a.mf = 0;
    a.addsrc = nullptr, a.addout = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (stride == 1) {
        if (mode == 0) return launch_dw<1, 0>(a, B, st);
        if (mode == 1) return launch_dw<1, 1>(a, B, st);
        return launch_dw<1, 2>(a, B, st);
    }
    if (mode == 0) return launch_dw<2, 0>(a, B, st);
    if (mode == 1) return launch_dw<2, 1>(a, B, st);
    return launch_dw<2, 2>(a, B, st);
}

// The same stride-1 convolutions applied to the TFAR mix  gLN(loc) * sigmoid(gLN(gate)^) + gLN(glob)^  (rtfs_tfar_mix_fwd's output) WITHOUT
// materialising it: loc [B][T][F][64], gate / glob [B][Tg][Fg][64] (nearest up-sampling), all passed pre-gLN with their statistics.
int rtfs_dwconv_mix_fwd(const float* loc, const double* loc_stats, const float* loc_g, const float* loc_b, const float* gate,
                        const double* gate_stats, const float* gate_g, const float* gate_b, const float* glob, const double* glob_stats,
                        const float* glob_g, const float* glob_b, int nconv, const float* const* w, const float* const* bias, float* const* out,
                        double* const* stats_out, int B, int T, int F, int Tg, int Fg, void* stream) {
    if (B <= 0 || (nconv != 1 && nconv != 2) || T <= 0 || F <= 0 || Tg <= 0 || Fg <= 0) return RTFS_EINVAL;
    DwArgs a;
    a.in = loc, a.slot = loc_stats, a.inv_n = 1.0 / ((double)T * F * kH), a.gamma = loc_g, a.beta = loc_b, a.slope = 0.f;
    a.Tin = a.Tout = T, a.Fin = a.Fout = F;
    a.nconv = nconv;
    for (int j = 0; j < kMaxConv; ++j) {
        a.w[j] = j < nconv ? w[j] : nullptr;
        a.bias[j] = j < nconv ? bias[j] : nullptr;
        a.out[j] = j < nconv ? out[j] : nullptr;
        a.stats[j] = j < nconv ? stats_out[j] : nullptr;
    }
    const double ng = 1.0 / ((double)Tg * Fg * kH);
    a.gate = NormRefLite{gate, gate_stats, ng, gate_g, gate_b};
    a.glob = NormRefLite{glob, glob_stats, ng, glob_g, glob_b};
    a.Tg = Tg, a.Fg = Fg;
    a.mt = div_magic_of(T), a.mf = div_magic_of(F);
    a.addsrc = nullptr, a.addout = nullptr;
    return launch_dw<1, 3>(a, B, (hipStream_t)stream);
}

// rtfs_pool_add_fwd + rtfs_dwconv_fwd(mode 1, one convolution) in one pass over `in` (= D1, compressed resolution): out = conv(gLN(in)) with its gLN partial
// sums (fusion_layers[1].local_embedding, fusion.py:25-52) AND G = pooled + gLN(in) (tdanet.py:117-118) - the normalised pixel is in the staged tile anyway.
int rtfs_dwconv_gadd_fwd(const float* in, const double* stats_in, const float* gamma, const float* beta, const float* w, float* out, double* stats_out,
                         const float* pooled, float* G, int B, int T, int F, void* stream) {
    if (B <= 0 || T <= 0 || F <= 0 || !pooled || !G || !stats_in) return RTFS_EINVAL;
    DwArgs a;
    a.in = in, a.slot = stats_in, a.inv_n = 1.0 / ((double)T * F * kH), a.gamma = gamma, a.beta = beta, a.slope = 0.f;
    a.Tin = a.Tout = T, a.Fin = a.Fout = F;
    a.nconv = 1;
    for (int j = 0; j < kMaxConv; ++j) a.w[j] = nullptr, a.bias[j] = nullptr, a.out[j] = nullptr, a.stats[j] = nullptr;
    a.w[0] = w, a.out[0] = out, a.stats[0] = stats_out;
    a.gate = a.glob = NormRefLite{nullptr, nullptr, 0.0, nullptr, nullptr};
    a.Tg = a.Fg = 0, a.mt = a.mf = 0;
    a.addsrc = pooled, a.addout = G;
    int nseg = F >= 96 ? 4 : 3;
    const long long wg0 = (long long)((T + 15) / 16) * B;
    if (wg0 * nseg < 512) nseg = (int)std::min<long long>((F + 7) / 8, (512 + wg0 - 1) / wg0);
    const int fseg = (((F + nseg - 1) / nseg) + 7) / 8 * 8;
    hipLaunchKernelGGL((dwconv_s1_kernel<1, 1, 4, true>), dim3((T + 15) / 16, B, (F + fseg - 1) / fseg), dim3(256), 0, (hipStream_t)stream, a, fseg);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_pool_fwd(const float* d0, const double* d0_stats, const float* d0_g, const float* d0_b, const float* d1, const double* d1_stats,
                  const float* d1_g, const float* d1_b, float* G, int B, int T, int T2, void* stream) {
    if (B <= 0 || T <= 0 || T2 <= 0) return RTFS_EINVAL;
    NormRefLite r0{d0, d0_stats, 1.0 / ((double)T * kF * kH), d0_g, d0_b};
    NormRefLite r1{d1, d1_stats, 1.0 / ((double)T2 * kF2 * kH), d1_g, d1_b};
    hipLaunchKernelGGL(pool_kernel, dim3((T2 * kF2 + 15) / 16, B), dim3(256), 0, (hipStream_t)stream, r0, r1, G, T, T2);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

// rtfs_dwconv_fwd(D0, mode 1, stride 1, w1) + rtfs_dwconv_fwd(D0, mode 1, stride 2, w2 + bias2) + the pooling half of rtfs_pool_fwd in ONE pass
// over D0 [B][T][129][64] (dwconv_trio_kernel); rtfs_pool_add_fwd then finishes G = pooled + gLN(D1).  T2 must be (T - 2) / 2 + 1.
int rtfs_dwconv_trio_fwd(const float* d0, const double* d0_stats, const float* d0_g, const float* d0_b, const float* w1, float* out1, double* stats1,
                         const float* w2, const float* bias2, float* out2, double* stats2, float* pooled, int B, int T, int T2, void* stream) {
    if (B <= 0 || T < 2 || T2 != (T - 2) / 2 + 1 || !bias2) return RTFS_EINVAL;
    TrioArgs a{d0, d0_stats, 1.0 / ((double)T * kF * kH), d0_g, d0_b, T, T2, w1, out1, stats1, w2, bias2, out2, stats2, pooled};
    int nseg = 4;
    const long long wg0 = (long long)((T + 15) / 16) * B;
    if (wg0 * nseg < 512) nseg = (int)std::min<long long>((kF + 7) / 8, (512 + wg0 - 1) / wg0);  // (small batches: see launch_dw)
    const int fseg = (((kF + nseg - 1) / nseg) + 7) / 8 * 8;
    hipLaunchKernelGGL(dwconv_trio_kernel, dim3((T + 15) / 16, B, (kF + fseg - 1) / fseg), dim3(256), 0, (hipStream_t)stream, a, fseg);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_pool_add_fwd(const float* pooled, const float* d1, const double* d1_stats, const float* d1_g, const float* d1_b, float* G, int B, int T2,
                      void* stream) {
    if (B <= 0 || T2 <= 0) return RTFS_EINVAL;
    NormRefLite r1{d1, d1_stats, 1.0 / ((double)T2 * kF2 * kH), d1_g, d1_b};
    hipLaunchKernelGGL(pool_add_kernel, dim3((T2 * kF2 + 15) / 16, B), dim3(256), 0, (hipStream_t)stream, pooled, r1, G, T2 * kF2);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_tfar_mix_fwd(const float* loc, const double* loc_stats, const float* loc_g, const float* loc_b, const float* gate,
                      const double* gate_stats, const float* gate_g, const float* gate_b, const float* glob, const double* glob_stats,
                      const float* glob_g, const float* glob_b, float* out, int B, int T, int F, int Tg, int Fg, void* stream) {
    if (B <= 0 || T <= 0 || F <= 0 || Tg <= 0 || Fg <= 0) return RTFS_EINVAL;
    const double nl = 1.0 / ((double)T * F * kH), ng = 1.0 / ((double)Tg * Fg * kH);
    NormRefLite l{loc, loc_stats, nl, loc_g, loc_b}, ga{gate, gate_stats, ng, gate_g, gate_b}, gl{glob, glob_stats, ng, glob_g, glob_b};
    hipLaunchKernelGGL(tfar_mix_kernel, dim3((T * F + 15) / 16, B), dim3(256), 0, (hipStream_t)stream, l, ga, gl, out, T, F, Tg, Fg);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_caf_video_fwd(const float* v, const float* att_w, const float* att_b, const float* att_g, const float* att_be, const float* rs_w,
                       const float* rs_b, const float* rs_g, const float* rs_be, float* att_out, float* rsz_out, int B, int Tv, void* stream) {
    if (B <= 0 || Tv <= 0) return RTFS_EINVAL;
    const int nh = caf_groups(Tv);
    const size_t lds = nh ? (size_t)Tv * (512 / nh + 2) * sizeof(float) : 0;
    static bool set[16] = {};
    if (caf_set_lds(reinterpret_cast<const void*>(caf_video_kernel), set) != RTFS_OK) return RTFS_ELAUNCH;
    hipLaunchKernelGGL(caf_video_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, v, att_w, att_b, att_g, att_be, rs_w, rs_b, rs_g, rs_be, att_out,
                       rsz_out, Tv, nh);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_caf_video_bwd(const float* v, const float* att_w, const float* att_b, const float* att_g, const float* att_be, const float* rs_w,
                       const float* rs_b, const float* rs_g, const float* rs_be, const float* datt, const float* drsz, float* dv, float* d_att_w,
                       float* d_att_b, float* d_att_g, float* d_att_be, float* d_rs_w, float* d_rs_b, float* d_rs_g, float* d_rs_be, int B, int Tv,
                       void* stream) {
    if (B <= 0 || Tv <= 0) return RTFS_EINVAL;
    const int nh = caf_groups(Tv);
    const size_t lds = nh ? (size_t)Tv * (512 / nh + 2) * sizeof(float) : 0;
    static bool set[16] = {};
    if (caf_set_lds(reinterpret_cast<const void*>(caf_video_bwd_kernel), set) != RTFS_OK) return RTFS_ELAUNCH;
    hipLaunchKernelGGL(caf_video_bwd_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, v, att_w, att_b, att_g, att_be, rs_w, rs_b, rs_g, rs_be, datt, drsz,
                       dv, d_att_w, d_att_b, d_att_g, d_att_be, d_rs_w, d_rs_b, d_rs_g, d_rs_be, Tv, nh);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_caf_fuse_fwd(const float* x, const float* ks, const float* kb, const float* vs, const float* vb, const float* att, const float* rsz,
                      const float* a0_or_null, float* out, int B, int T, int Tv, void* stream) {
    if (B <= 0 || T <= 0 || Tv <= 0) return RTFS_EINVAL;
    hipLaunchKernelGGL(caf_fuse_kernel, dim3((T * kF + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, x, ks, kb, vs, vb, att, rsz, a0_or_null, out, T, Tv);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

}  // extern "C"
