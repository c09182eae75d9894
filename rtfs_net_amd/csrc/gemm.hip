// Pixel GEMMs: every 1x1 convolution of the path as a channels-last row-major GEMM on the fp32 MFMA
// pipe, with the surrounding norm / activation / residual work fused into the operand load
// ("prologue") or the accumulator write-back ("epilogue").
//
//   rtfs_bottleneck_fwd   audio_bottleneck  = ConvNormAct(pre gLN, pre ReLU, 1x1 256->256)   tdavnet.py:59,89
//   rtfs_proj_fwd         gateway (dw1x1+PReLU) -> projection 1x1 256->64 (+gLN partial sums) tdanet.py:34-49,108-109
//   rtfs_resid_fwd        TFAR tail -> residual_conv 1x1 64->256 + gateway residual           tdanet.py:54-59,127-131
//   rtfs_mask_fwd         S3: PReLU -> 1x1 256->256 -> ReLU -> complex multiply with a_emb     mask_generator.py:67-99
//   rtfs_gemm_rows_fwd    plain  Y[M][N] = X[M][K] W^T  (SRU layers 1-3 input projections; decoder taps)
//
// Tiling: 256 threads = 4 waves; workgroup tile BM x N (full N), K streamed in 32-deep chunks through
// double-buffered LDS; each wave owns WM x WN tiles of 32x32 (v_mfma_f32_32x32x2_f32).  Weights are
// pre-transposed on the host to Wt[N][K] so both operands are k-contiguous in LDS (common.h).
#include <type_traits>
#include "common.h"

namespace rtfs {

// ------------------------------------------------------------------------------------------------
// prologues: produce A[row][k..k+3] for row < Mb of utterance b
// ------------------------------------------------------------------------------------------------
// A prologue supplies   raw(b, Mb, row, k)  -- the global load of A[row][k..k+3], no arithmetic, so it can be issued
// a whole k-chunk ahead of its use;   init(b, tab) -- fills a 512-float LDS table with per-channel constants once;
// xform(raw, k, tab) -- the arithmetic, run when the chunk is written to LDS (after the MFMA block it overlaps).
struct ProPlain {
    const float* __restrict__ x;
    int K;
    __device__ void init(int, float*) {}
    __device__ float4 raw(int b, int Mb, int row, int k) const { return ld4(x + ((size_t)b * Mb + row) * K + k); }
    __device__ float4 xform(float4 v, int, const float*) const { return v; }
};

// relu(gLN(x))  -- audio_bottleneck pre_norm / pre_act (config yaml:15-20); tab = [scale(256) | shift(256)]
struct ProGlnRelu {
    const float* __restrict__ x;
    const double* slot;
    double inv_n;
    const float *__restrict__ gamma, *__restrict__ beta;
    __device__ void init(int b, float* tab) const {
        float mean, rstd;
        stats_finalize(slot, b, inv_n, mean, rstd);
        const float sc = gamma[threadIdx.x] * rstd;
        tab[threadIdx.x] = sc;
        tab[kC + threadIdx.x] = beta[threadIdx.x] - mean * sc;
    }
    __device__ float4 raw(int b, int Mb, int row, int k) const { return ld4(x + ((size_t)b * Mb + row) * kC + k); }
    __device__ float4 xform(float4 v, int k, const float* tab) const { return relu4(fma4(v, ld4(tab + k), ld4(tab + kC + k))); }
};

// gateway parameters of the RTFS block (argument pack of proj_kernel)
struct ProGateway {
    const float* __restrict__ x;
    const float *__restrict__ gw, *__restrict__ gb;
    float slope;
};

// prelu(x)  -- mask_generator.0 (mask_generator.py:47)
struct ProPrelu {
    const float* __restrict__ x;
    float slope;
    __device__ void init(int, float*) {}
    __device__ float4 raw(int b, int Mb, int row, int k) const { return ld4(x + ((size_t)b * Mb + row) * kC + k); }
    __device__ float4 xform(float4 v, int, const float*) const { return prelu4(v, slope); }
};

// One gLN'd tensor read "normalise on read": slot -> (mean, rstd), gamma/beta per channel.
struct NormRef {
    const float* x;
    const double* slot;
    double inv_n;
    const float *gamma, *beta;
};

// TFAR tail (tdanet.py:127 with upsampling_depth 2; InjectionMultiSum.forward fusion.py:54-69):
//   expanded = gLN(cl) * sigmoid(gLN(cgate))^ + gLN(cg)^ + gLN(d0)        (^ = nearest up-sampling)
struct ProExpanded {
    NormRef cl, d0;      // full resolution [B][T][F][64]
    NormRef cg, cgate;   // compressed      [B][T2][F2][64]
    int T, T2;
    float m[4], r[4];  // (host side leaves these zero; kept for the launchers' aggregate initialisers)
    // (scale, shift) of channel c of gLN j (0 cl, 1 d0, 2 cg, 3 cgate) of utterance b, folded from (mean, rstd, gamma, beta).  Written without
    // touching *this (const, selects on j): modifying this by-value kernel argument, or indexing its members with a run-time j through an array
    // of pointers, forces the whole struct into SCRATCH, and every later read of T / T2 in the tile loop then is a scratch load followed by
    // `s_waitcnt vmcnt(0)` - a full drain of the software-pipelined global loads, once per phase (found in round 3: 208 bytes of scratch per lane).
    __device__ void fold(int b, int j, int c, float& sc, float& sh) const {
        const double* slot = j == 0 ? cl.slot : (j == 1 ? d0.slot : (j == 2 ? cg.slot : cgate.slot));
        const double inv_n = j == 0 ? cl.inv_n : (j == 1 ? d0.inv_n : (j == 2 ? cg.inv_n : cgate.inv_n));
        const float* gp = j == 0 ? cl.gamma : (j == 1 ? d0.gamma : (j == 2 ? cg.gamma : cgate.gamma));
        const float* bp = j == 0 ? cl.beta : (j == 1 ? d0.beta : (j == 2 ? cg.beta : cgate.beta));
        float mean, rstd;
        stats_finalize(slot, b, inv_n, mean, rstd);
        sc = gp[c] * rstd;
        sh = bp[c] - mean * sc;
    }
};

// ------------------------------------------------------------------------------------------------
// epilogues
// ------------------------------------------------------------------------------------------------
// Epilogues work on 16-byte vectors: the kernel multiplies WEIGHTS x PIXELS (see mma_block), so a lane owns one
// pixel row and, per accumulator register group, 4 consecutive output channels `col..col+3`.
template <bool HAS_BIAS, bool ACCUM = false>
struct EpiBias {  // y (= or +=) acc (+ bias)
    static constexpr bool kAccum = ACCUM;
    static constexpr int kSide = 0;
    float* __restrict__ y;
    const float* __restrict__ bias;
    int N;
    __device__ float4 colconst(int col) const { return HAS_BIAS ? ld4(bias + col) : f4(0, 0, 0, 0); }
    __device__ void store(int b, int Mb, int row, int col, float4 v, float4 cc) const {
        float* o = y + ((size_t)b * Mb + row) * N + col;
        v = v + cc;
        if (ACCUM) v = v + ld4(o);
        st4(o, v);
    }
};

// argument pack of proj_kernel: y = acc + bias, plus the gLN partial sums of y
struct EpiBiasStats {
    float* __restrict__ y;
    const float* __restrict__ bias;
    int N;
    double* slot;
};

// argument pack of resid_kernel: out = acc + bias + gateway(s_in) [+ a0]   (tdanet.py:131; refinement_module.py:60)
struct EpiResidual {
    float* __restrict__ y;
    const float* __restrict__ bias;
    const float* __restrict__ s_in;
    const float *__restrict__ gw, *__restrict__ gb;
    float slope;
    const float* __restrict__ a0;  // may be null
    // PROJ variant: the NEXT block's gateway + projection (tdanet.py:108-109, shared weights) computed from the output tile while it
    // is still in LDS: py = Wp . prelu(out * gw + gb) + pbias (pre-gLN, [B][TF][64]) and its gLN partial sums
    const float* __restrict__ pw;     // [64][256]
    const float* __restrict__ pbias;  // [64]
    float* __restrict__ py;
    double* pslot;
    // CAF variant (block 0): the audio side of ATTNFusionCell (fusion.py:259-272; key / value depth-wise 1x1 + BatchNorm(eval) folded into
    // ks/kb, vs/vb) applied to the block output x while it is in the epilogue registers:
    //   out = relu(x*ks+kb) * rsz[b][tv(t)] + att[b][tv(t)] * (x*vs+vb) [+ s_in]      (block 0's input s_in IS a0, refinement_module.py:55-60)
    const float *__restrict__ caf_ks, *__restrict__ caf_kb, *__restrict__ caf_vs, *__restrict__ caf_vb;  // [256]
    const float *__restrict__ att, *__restrict__ rsz;                                                    // [B][Tv][256]
    int Tv;
};

// S3 complex mask (mask_generator.py:70-82): m = relu(acc + bias); channels [0,128) real, [128,256) imaginary
struct EpiMask {
    static constexpr bool kAccum = false;
    static constexpr int kSide = 0;
    float* __restrict__ y;
    const float* __restrict__ bias;
    const float* __restrict__ emb;
    float* __restrict__ m_out;  // training: the post-ReLU mask itself (needed by the adjoint), or null
    __device__ void store2(int b, int Mb, int row, int col, float4 vr, float4 vi) const {
        const size_t o = ((size_t)b * Mb + row) * kC + col;
        store2e(b, Mb, row, col, vr, vi, ld4(emb + o), ld4(emb + o + 128));
    }
    __device__ void store2e(int b, int Mb, int row, int col, float4 vr, float4 vi, float4 er, float4 ei) const {
        store2eb(b, Mb, row, col, vr, vi, er, ei, ld4(bias + col), ld4(bias + col + 128));
    }
    // (bias quads handed in: a caller that keeps them in registers has no load - and no vmcnt(0) - inside its store loop)
    __device__ void store2eb(int b, int Mb, int row, int col, float4 vr, float4 vi, float4 er, float4 ei, float4 br, float4 bi) const {
        const size_t o = ((size_t)b * Mb + row) * kC + col;
        const float4 mr = relu4(vr + br), mi = relu4(vi + bi);
        if (m_out) {
            st4(m_out + o, mr);
            st4(m_out + o + 128, mi);
        }
        st4(y + o, f4(er.x * mr.x - ei.x * mi.x, er.y * mr.y - ei.y * mi.y, er.z * mr.z - ei.z * mi.z, er.w * mr.w - ei.w * mi.w));
        st4(y + o + 128, f4(fmaf(er.x, mi.x, ei.x * mr.x), fmaf(er.y, mi.y, ei.y * mr.y), fmaf(er.z, mi.z, ei.z * mr.z), fmaf(er.w, mi.w, ei.w * mr.w)));
    }
};

// Adjoint of the S3 complex product + ReLU gate (mask_generator.py:70-82), applied to the decoder's input gradient d(masked) while its row is in the
// epilogue registers (training step): dz = gradient w.r.t. the mask pre-activation, de = the part of d(a_emb) that arrives through the product.
// Same arithmetic as mask_bwd_elem_kernel (bwd_misc.hip); d(masked) itself never exists in HBM.
struct EpiMaskBwd {
    static constexpr bool kAccum = false;
    static constexpr int kSide = 0;
    const float* __restrict__ emb;  // a_emb [rows][256]
    const float* __restrict__ m;    // post-ReLU mask [rows][256]
    float* __restrict__ dz;
    float* __restrict__ de;
    __device__ void store2(int b, int Mb, int row, int col, float4 vr, float4 vi) const {
        const size_t o = ((size_t)b * Mb + row) * kC + col;
        store2e(b, Mb, row, col, vr, vi, ld4(emb + o), ld4(emb + o + 128));
    }
    __device__ void store2e(int b, int Mb, int row, int col, float4 dor, float4 doi, float4 er, float4 ei) const {
        const size_t o = ((size_t)b * Mb + row) * kC + col;
        const float4 mr = ld4(m + o), mi = ld4(m + o + 128);
        auto gate = [](float g, float mm) { return mm > 0.f ? g : 0.f; };
        const float4 dmr = f4(dor.x * er.x + doi.x * ei.x, dor.y * er.y + doi.y * ei.y, dor.z * er.z + doi.z * ei.z, dor.w * er.w + doi.w * ei.w);
        const float4 dmi = f4(doi.x * er.x - dor.x * ei.x, doi.y * er.y - dor.y * ei.y, doi.z * er.z - dor.z * ei.z, doi.w * er.w - dor.w * ei.w);
        st4(dz + o, f4(gate(dmr.x, mr.x), gate(dmr.y, mr.y), gate(dmr.z, mr.z), gate(dmr.w, mr.w)));
        st4(dz + o + 128, f4(gate(dmi.x, mi.x), gate(dmi.y, mi.y), gate(dmi.z, mi.z), gate(dmi.w, mi.w)));
        st4(de + o, f4(dor.x * mr.x + doi.x * mi.x, dor.y * mr.y + doi.y * mi.y, dor.z * mr.z + doi.z * mi.z, dor.w * mr.w + doi.w * mi.w));
        st4(de + o + 128, f4(doi.x * mr.x - dor.x * mi.x, doi.y * mr.y - dor.y * mi.y, doi.z * mr.z - dor.z * mi.z, doi.w * mr.w - dor.w * mi.w));
    }
};

// Input-gradient GEMMs of the training step whose consumer is an activation's adjoint (ws256_kernel only, round 6): the rows of the activation's INPUT x are
// fetched next to the output rows and the element-wise adjoint runs in the epilogue registers, so the GEMM's output is not written and re-read for it.
//   SIDE 1  y = prelu'(x) * acc,  dslope += sum acc * x * [x <= 0]                     (mask_generator.py:47-48: PReLU -> Conv2d; prelu_bwd_kernel's arithmetic)
//   SIDE 2  y = acc (unchanged) and the REDUCE pass of relu(gLN(x))'s adjoint with g = acc * [gLN(x) > 0]: S1 = sum g gamma, S2 = sum g gamma xhat per
//           utterance, dgamma_c += sum g xhat, dbeta_c += sum g    (tdavnet.py:59,89: audio_bottleneck's pre-norm + pre-act; gln_bwd_reduce_kernel<256, 2>'s arithmetic)
template <int SIDE>
struct EpiAdjoint {
    static constexpr bool kAccum = false;
    static constexpr int kSide = SIDE;
    float* __restrict__ y;
    const float* __restrict__ x;  // the activation's input [B][Mb][256]
    float slope;                  // SIDE 1
    const double* slot;           // SIDE 2: gLN statistics of x
    double inv_n;
    const float *__restrict__ gamma, *__restrict__ beta;
    double* red;  // SIDE 2: [B][kStatStride] (S1, S2)
    float* scr;   // spread scratch: SIDE 1 [dslope], SIDE 2 [dgamma 256 | dbeta 256]
    __device__ float4 colconst(int) const { return f4(0, 0, 0, 0); }
};

// sum and sum of squares of a float4 in scalar VALU instructions.  Left to hipcc, the SLP vectoriser turned the gLN partial sums of the MFMA waves of
// resid_ws_kernel into v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[0,1] - the packed form that returned wrong low halves next to bf16 MFMA traffic
// (DESIGN.md rule 10), here issued BETWEEN the bf16 MFMAs of the same wave.
__device__ __forceinline__ void stat_sums(float4 o, float& s, float& q) {
    asm("v_add_f32 %0, %2, %3\n\tv_mul_f32 %1, %2, %2\n\tv_add_f32 %0, %0, %4\n\tv_fmac_f32 %1, %3, %3\n\tv_add_f32 %0, %0, %5\n\tv_fmac_f32 %1, %4, %4\n\tv_fmac_f32 %1, %5, %5"
        : "=&v"(s), "=&v"(q)
        : "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
}

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
// NT: precision of the contraction (common.h): 0 fp32, 1 bf16, 3 split-bf16.  NT != 0: `Wt` is the host-PACKED weight (same indexing).
template <int K, int N, int BM, int WM, int WN, bool PAIRED, int BK, class Pro, class Epi, int NT = 0>
__global__ __launch_bounds__(256, (WM * WN >= 8 ? 2 : 1)) void pixel_gemm_kernel(Pro pro, Epi epi, const float* __restrict__ Wt, int Mb) {
    static_assert(NT == 0 || BK % 16 == 0, "the bf16 instruction consumes 16 k per step");
    constexpr int LD = BK + 4;
    constexpr int WGN = N / (32 * WN), WGM = 4 / WGN;
    static_assert(WGM * WGN == 4 && BM == WGM * WM * 32, "wave tiling must cover the workgroup tile");
    static_assert(!PAIRED || (N == 256 && (WN == 2 || WN == 4)), "paired epilogue needs N=256, WN in {2,4}");
    __shared__ __attribute__((aligned(16))) float As[2][BM * LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][N * LD];
    const int b = blockIdx.y, m0 = blockIdx.x * BM;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = w / WGN, wn = w % WGN;
    __shared__ __attribute__((aligned(16))) float ptab[2 * kC];
    pro.init(b, ptab);
    __syncthreads();

    constexpr int A_PER = BM * (BK / 4) / 256;
    static_assert(256 % (BK / 4) == 0, "a thread keeps the same channel quad for all its rows");
    float4 areg[A_PER];
    int ak0 = 0;
    ChunkRegs<N, BK> breg;
    const int ac4 = (threadIdx.x % (BK / 4)) * 4, arow0 = threadIdx.x / (BK / 4);

    // rows past the end of the utterance are clamped (their accumulator rows are never stored): no branch, so the
    // loads of chunk kc+1 are issued before the MFMA block of chunk kc and first touched after it
    auto load_a = [&](int k0) {
        ak0 = k0;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) areg[i] = pro.raw(b, Mb, min(m0 + arow0 + i * (1024 / BK), Mb - 1), k0 + ac4);
    };
    auto store_a = [&](float* dst) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) st4(dst + (arow0 + i * (1024 / BK)) * LD + ac4, pack4<NT>(pro.xform(areg[i], ak0 + ac4, ptab)));
    };

    floatx16 acc[WN][WM];  // [weight tile][pixel tile]: rows = output channels, lanes = pixels
    acc_zero(acc);

    load_a(0);
    breg.load(Wt, K, 0);
    store_a(As[0]);
    breg.store(Bs[0], LD);
    __syncthreads();

    constexpr int NK = K / BK;
    constexpr int BT = PAIRED ? (WN == 4 ? -4 : 128) : 32;
    const int bcol0 = PAIRED ? wn * (WN * 16) : wn * WN * 32;  // paired: this wave owns WN/2 real tiles and their imaginary partners
#pragma unroll 1
    for (int kc = 0; kc < NK; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < NK) {
            load_a((kc + 1) * BK);
            breg.load(Wt, K, (kc + 1) * BK);
        }
        mma_block_nt<NT, WN, WM, 32, BT>(acc, Bs[cur] + bcol0 * LD, LD, As[cur] + wm * WM * 32 * LD, LD, BK);
        if (kc + 1 < NK) {
            store_a(As[cur ^ 1]);
            breg.store(Bs[cur ^ 1], LD);
        }
        __syncthreads();
    }

    const int i = lane & 31, kh = lane >> 5;
    if constexpr (N == 256 && WGM == 1) {
        // 256-wide outputs: accumulator-direct stores touch 32 pixel rows x 32 B per instruction (see resid_kernel).  Each 32-row
        // tile is transposed through LDS (aliased on the weight stages, free after the last k-chunk) and leaves / meets its
        // epilogue operands as whole 1 KB rows.
        constexpr int LDO = 260;
        static_assert(sizeof(Bs) >= 32 * LDO * sizeof(float), "output tile aliases the weight stages");
        float* Ot = &Bs[0][0];
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            const int rbase = m0 + m * 32;
            // S3 mask: the embedding rows this thread will combine with are fetched (clamped, unconditional) before the transposition
            float4 er[4], ei[4];
            if constexpr (PAIRED) {
                const int q4 = (threadIdx.x & 31) * 4;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const size_t o = ((size_t)b * Mb + min(rbase + (int)(threadIdx.x >> 5) + 8 * it, Mb - 1)) * kC + q4;
                    er[it] = ld4(epi.emb + o), ei[it] = ld4(epi.emb + o + 128);
                }
            }
            __syncthreads();  // m = 0: every wave left the k loop; m > 0: the previous tile has been read
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = PAIRED ? bcol0 + (n % (WN / 2)) * 32 + (n >= WN / 2 ? 128 : 0) + 8 * g + 4 * kh : bcol0 + n * 32 + 8 * g + 4 * kh;
                    st4(Ot + i * LDO + col, acc_group(acc[n][m], g));
                }
            __syncthreads();
            if constexpr (PAIRED) {
                const int q4 = (threadIdx.x & 31) * 4;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int r = (threadIdx.x >> 5) + 8 * it;
                    if (rbase + r < Mb) epi.store2e(b, Mb, rbase + r, q4, ld4(Ot + r * LDO + q4), ld4(Ot + r * LDO + 128 + q4), er[it], ei[it]);
                }
            } else {
                const int cq = (threadIdx.x & 63) * 4;
                const float4 cc = epi.colconst(cq);
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int r = (threadIdx.x >> 6) + 4 * it;
                    if (rbase + r < Mb) epi.store(b, Mb, rbase + r, cq, ld4(Ot + r * LDO + cq), cc);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int row = m0 + (wm * WM + m) * 32 + i;
        if (row < Mb) {
            if constexpr (PAIRED) {
#pragma unroll
                for (int n = 0; n < WN / 2; ++n)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        epi.store2(b, Mb, row, bcol0 + n * 32 + 8 * g + 4 * kh, acc_group(acc[n][m], g), acc_group(acc[n + WN / 2][m], g));
            } else {
#pragma unroll
                for (int n = 0; n < WN; ++n)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = bcol0 + n * 32 + 8 * g + 4 * kh;
                        epi.store(b, Mb, row, col, acc_group(acc[n][m], g), epi.colconst(col));
                    }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// residual_conv (K = 64 -> N = 256) with the operands SWAPPED: out^T[n][p] = W[n][k] . E^T[k][p].
//  * the weight is the MFMA A operand and lives in registers for the whole workgroup (64 VGPRs per
//    wave: wave w owns output channels [64w, 64w+64)), so nothing but the 64-pixel E tile goes through LDS;
//  * the accumulator (lane = pixel, 4-register group = 4 consecutive channels) is transposed through LDS one 32-pixel
//    half at a time, so the epilogue (gateway residual, +a0, store) moves whole 1 KB pixel rows per wave-instruction;
//  * 52 KB of LDS -> 2-3 workgroups per CU overlap one's MFMA phase with the others' memory phases.
// Each workgroup walks `tiles_per_wg` consecutive 64-pixel tiles of one utterance.
// ------------------------------------------------------------------------------------------------
// CAF (rtfs_resid_caf_fwd): the CAF cell's audio side rides in the epilogue; HAS_A0 then means "+ s_in" (no separate a0 stream).
// DEEP: ONE workgroup per CU with the whole 512-entry register file (256 VGPR + 256 AGPR per lane): the projection weights stay resident
// next to the residual-conv weights, both 32-pixel halves' residual operands are fetched before the MFMA phase and the next tile's E
// operands right after it - every load is issued a full phase before its first use, so a tile costs its arithmetic (fp32 MFMA + the
// VALU it does not overlap with) instead of arithmetic + three exposed load -> use latencies that two co-resident workgroups only
// partly cover for each other.  Per element the same arithmetic in the same order as the two-workgroup form (the fp32
// partial sums of the projection's gLN statistics cover 4x more tiles per workgroup: 1e-7-level differences downstream).
// DEEP = 1: both halves' residual operands are fetched at the top of their tile; DEEP = 2: each half's register set is re-filled for the NEXT tile as
// soon as its epilogue has consumed it (reads in flight through every phase).  Measured at B = 32 (us, normal / 1 / 2): residual + projection 1053 / 1047 / 956,
// CAF + projection 1037 / 980 / 1051, plain 555 / 545 / 623 - the launcher picks per variant.
template <bool HAS_A0, bool PROJ = false, int NT = 0, bool CAF = false, int DEEP = 0>  // NT != 0: Wt and epi.pw are host-PACKED (common.h)
__global__ __launch_bounds__(256, (DEEP ? 1 : 2)) void resid_kernel(ProExpanded pro, EpiResidual epi, const float* __restrict__ Wt, int Mb, int tiles_per_wg) {
    constexpr int LDE = 68;
    constexpr int LDO = 260;
    __shared__ __attribute__((aligned(16))) float Es[64 * LDE];
    __shared__ __attribute__((aligned(16))) float Ot[32 * LDO];  // one 32-pixel half of the output tile, pixel-major
    __shared__ float pred[8];
    float ps = 0.f, pq = 0.f;  // PROJ: gLN partial sums of the projection output
    const int b = blockIdx.y;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const int cq = (threadIdx.x & 63) * 4;  // this thread's channel quad in the coalesced epilogue
    const float4 cbias = ld4(epi.bias + cq), cgw = ld4(epi.gw + cq), cgb = ld4(epi.gb + cq);
    __shared__ __attribute__((aligned(16))) float cafc[CAF ? 4 * kC : 4];  // CAF: ks | kb | vs | vb, re-read per epilogue row (16 VGPRs otherwise)
    if (CAF) {
        const float* src[4] = {epi.caf_ks, epi.caf_kb, epi.caf_vs, epi.caf_vb};
#pragma unroll
        for (int j = 0; j < 4; ++j) cafc[j * kC + threadIdx.x] = src[j][threadIdx.x];
    }
    // fold (mean, rstd, gamma, beta) of the four gLNs into scale/shift tables once: Ns[tensor][scale|shift][channel]
    __shared__ __attribute__((aligned(16))) float Ns[4][2][kH];
    const int c4 = (threadIdx.x & 15) * 4;
    {
        const int j = threadIdx.x >> 6, c = threadIdx.x & 63;
        float sc, sh;
        pro.fold(b, j, c, sc, sh);
        Ns[j][0][c] = sc;
        Ns[j][1][c] = sh;
    }

    float4 wf[2][8];  // W fragments: rows n = 64w + 32nt + i, k = 8q + 4kh .. +3
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 8; ++q) wf[nt][q] = ld4(Wt + (size_t)(64 * w + 32 * nt + i) * 64 + 8 * q + 4 * kh);
    __syncthreads();  // Ns tables

    // per-utterance base pointers (wave-uniform): everything below addresses with 32-bit offsets (an utterance is < 2^31 bytes)
    const float* cl_b = pro.cl.x + (size_t)b * Mb * kH;
    const float* d0_b = pro.d0.x + (size_t)b * Mb * kH;
    const float* cg_b = pro.cg.x + (size_t)b * pro.T2 * kF2 * kH;
    const float* cgate_b = pro.cgate.x + (size_t)b * pro.T2 * kF2 * kH;
    const float* s_b = epi.s_in + (size_t)b * Mb * kC;
    const float* a0_b = (HAS_A0 && !CAF) ? epi.a0 + (size_t)b * Mb * kC : nullptr;
    const float* att_b = CAF ? epi.att + (size_t)b * epi.Tv * kC : nullptr;
    const float* rsz_b = CAF ? epi.rsz + (size_t)b * epi.Tv * kC : nullptr;
    float* y_b = epi.y + (size_t)b * Mb * kC;
    const int tile0 = blockIdx.x * tiles_per_wg;
    // E tile operands: 64 pixels x 64 channels, 4 float4 per thread and tensor (rows past the end are clamped: their columns are never
    // stored).  (Measured, round 2, in the two-workgroup form: issuing tile i + 1's loads in the second half of tile i's epilogue - into
    // the registers the accumulators have just left - made every variant SLOWER: 1065 -> 1163 us with the fused projection, 563 -> 576 us
    // without; single early loads only queue in front of the epilogue's own operands.  The DEEP form, with the register file to keep a whole
    // tile's operands in flight, does prefetch them - see the template comment.)
    float4 xa[4], xd[4], xg[4], xs[4];
    auto load_e = [&](int m0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {  // all 16 loads first
            const int row = min(m0 + (int)(threadIdx.x >> 4) + it * 16, Mb - 1);
            const int t = row / kF, f = row - t * kF;
            const int t2 = nearest_src(t, pro.T2, pro.T), f2 = nearest_src(f, kF2, kF);
            const unsigned hi = ((unsigned)row * kH + c4) * 4u, lo = (((unsigned)t2 * kF2 + f2) * kH + c4) * 4u;  // byte offsets in the utterance
            xa[it] = ld4_off(cl_b, hi), xd[it] = ld4_off(d0_b, hi), xg[it] = ld4_off(cg_b, lo), xs[it] = ld4_off(cgate_b, lo);
        }
    };
    // DEEP: projection weights resident: W_p[16w + (lane & 15)][64 (lane >> 4) + 4t .. +3], t = 0..15
    float4 wpr[(DEEP && PROJ) ? 16 : 1];
    if constexpr (DEEP && PROJ) {
        const float* wpp0 = epi.pw + (size_t)(16 * w + (lane & 15)) * 256 + 64 * (lane >> 4);
#pragma unroll
        for (int t = 0; t < 16; ++t) wpr[t] = ld4(wpp0 + 4 * t);
    }
    constexpr int NH = DEEP ? 2 : 1;  // register sets of residual operands: DEEP keeps one per 32-pixel half, each re-filled for the NEXT tile as soon as
                                      // its half's epilogue has consumed it - some 64 KB of reads are in flight per CU through every phase of a tile
    float4 sv[NH][8], av[NH][8];
    // CAF: a 32-pixel half spans at most two STFT frames t, hence at most two video frames tv(t) = floor(t Tv / T): both (att, rsz) row
    // pairs are fetched with the residual operands (the registers of the a0 stream, which this variant does not have)
    float4 catt[NH][2], crsz[NH][2];
    int caf_split[NH];  // first pixel row of the half that belongs to the second frame
    auto load_sv = [&](int m0, int pt) {
        const int hs = DEEP ? pt : 0;
        const int prow = m0 + pt * 32 + (threadIdx.x >> 6);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const unsigned o = ((unsigned)min(prow + 4 * it, Mb - 1) * kC + cq) * 4u;
            sv[hs][it] = ld4_off(s_b, o);
            if (HAS_A0 && !CAF) av[hs][it] = ld4_off(a0_b, o);
        }
        if (CAF) {
            const int p0 = min(m0 + pt * 32, Mb - 1), p1 = min(m0 + pt * 32 + 31, Mb - 1);
            const int t0 = p0 / kF, t1 = p1 / kF;
            caf_split[hs] = (t0 + 1) * kF;
            const unsigned o0 = ((unsigned)nearest_src(t0, epi.Tv, pro.T) * kC + cq) * 4u, o1 = ((unsigned)nearest_src(t1, epi.Tv, pro.T) * kC + cq) * 4u;
            catt[hs][0] = ld4_off(att_b, o0), crsz[hs][0] = ld4_off(rsz_b, o0), catt[hs][1] = ld4_off(att_b, o1), crsz[hs][1] = ld4_off(rsz_b, o1);
        }
    };
    const int tile_last = min(tile0 + tiles_per_wg, (Mb + 63) / 64) - 1;
    if constexpr (DEEP != 0) load_e(tile0 * 64);
    if constexpr (DEEP == 2) {
        load_sv(tile0 * 64, 0);
        load_sv(tile0 * 64, 1);
    }
#pragma unroll 1
    for (int tl = 0; tl < tiles_per_wg; ++tl) {
        const int m0 = (tile0 + tl) * 64;
        if (m0 >= Mb) break;
        if constexpr (!DEEP) load_e(m0);
        {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float4 a = fma4(xa[it], ld4(&Ns[0][0][c4]), ld4(&Ns[0][1][c4])), d = fma4(xd[it], ld4(&Ns[1][0][c4]), ld4(&Ns[1][1][c4]));
                const float4 g = fma4(xg[it], ld4(&Ns[2][0][c4]), ld4(&Ns[2][1][c4])), sg = sigmoid4(fma4(xs[it], ld4(&Ns[3][0][c4]), ld4(&Ns[3][1][c4])));
                st4(Es + ((threadIdx.x >> 4) + it * 16) * LDE + c4, pack4<NT>(fma4(a, sg, g) + d));
            }
        }
        __syncthreads();
        // residual-side operands of the first 32-pixel half: issued BEFORE the MFMA phase so that their HBM latency runs under it
        // (the kernel is latency-bound once the projection rides in the epilogue)
        if constexpr (DEEP != 2) {
            load_sv(m0, 0);
            if constexpr (DEEP == 1) load_sv(m0, 1);
            __builtin_amdgcn_sched_barrier(0);  // (left alone, hipcc sinks these loads below the MFMAs to save registers)
        }
        floatx16 acc[2][2];
        acc_zero(acc);
        if constexpr (NT == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 e0 = ld4(Es + i * LDE + 8 * q + 4 * kh);
                const float4 e1 = ld4(Es + (32 + i) * LDE + 8 * q + 4 * kh);
    #pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].x, e0.x, acc[nt][0], 0, 0, 0);
                    acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].x, e1.x, acc[nt][1], 0, 0, 0);
                    acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].y, e0.y, acc[nt][0], 0, 0, 0);
                    acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].y, e1.y, acc[nt][1], 0, 0, 0);
                    acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].z, e0.z, acc[nt][0], 0, 0, 0);
                    acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].z, e1.z, acc[nt][1], 0, 0, 0);
                    acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].w, e0.w, acc[nt][0], 0, 0, 0);
                    acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].w, e1.w, acc[nt][1], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2) {  // 16 k per step: packed slots 2 q2 and 2 q2 + 1 of every row
                const Frag e0 = frag_lds<NT>(ld4(Es + i * LDE + 16 * q2 + 4 * kh), ld4(Es + i * LDE + 16 * q2 + 8 + 4 * kh));
                const Frag e1 = frag_lds<NT>(ld4(Es + (32 + i) * LDE + 16 * q2 + 4 * kh), ld4(Es + (32 + i) * LDE + 16 * q2 + 8 + 4 * kh));
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const Frag wq = frag_lds<NT>(wf[nt][2 * q2], wf[nt][2 * q2 + 1]);
                    mma32<NT>(acc[nt][0], wq, e0);
                    mma32<NT>(acc[nt][1], wq, e1);
                }
            }
        }
        const int m0_next = min(tile0 + tl + 1, tile_last) * 64;  // (the last tile re-fetches itself: L2 hits, no branch around the loads)
        if constexpr (DEEP) {  // next tile's E operands
            __builtin_amdgcn_sched_barrier(0);
            load_e(m0_next);
            __builtin_amdgcn_sched_barrier(0);
        }
        // Epilogue through LDS.  In the accumulator a lane owns ONE pixel and 16 B of channels, so a direct global epilogue makes
        // every b128 access touch 32 pixel rows x 32 B (measured: 50 L1 tag accesses per wave-instruction, TA 60 % busy, the
        // kernel issue-stalled behind it).  Transposed through LDS, thread = (channel quad, pixel row): one wave-instruction is
        // one pixel's whole 1 KB row - 8 lines - for s_in, a0 and the store; the per-channel constants sit in registers.
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            float4 wps[DEEP ? 1 : 16];  // PROJ: W_p[16w + (lane & 15)][64 (lane >> 4) + 4t .. +3], t = 0..15 (streamed per half unless DEEP)
            const float* wpp = PROJ ? epi.pw + (size_t)(16 * w + (lane & 15)) * 256 + 64 * (lane >> 4) : nullptr;
            const int prow = m0 + pt * 32 + (threadIdx.x >> 6);
            constexpr int hs_ = 0;
            const int hs = DEEP ? pt : hs_;
            if (!DEEP && pt == 1) load_sv(m0, 1);
            int coff = cq;
            if (CAF) asm volatile("" : "+v"(coff));  // opaque: the table reads stay inside the epilogue loop
            __syncthreads();  // pt = 0: every wave is done with Es / pt = 1: Ot of the previous half has been read
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    st4(Ot + i * LDO + 64 * w + 32 * nt + 8 * g + 4 * kh,
                        f4(acc[nt][pt][4 * g], acc[nt][pt][4 * g + 1], acc[nt][pt][4 * g + 2], acc[nt][pt][4 * g + 3]));
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = (threadIdx.x >> 6) + 4 * it;
                float4 v = ld4(Ot + r * LDO + cq) + cbias + prelu4_minfma(fma4(sv[hs][it], cgw, cgb), epi.slope - 1.0f);
                if (CAF) {
                    const bool second = prow + 4 * it >= caf_split[hs];
                    const float4 at = second ? catt[hs][1] : catt[hs][0], rz = second ? crsz[hs][1] : crsz[hs][0];
                    v = fma4(at, fma4(v, ld4(cafc + 2 * kC + coff), ld4(cafc + 3 * kC + coff)), relu4(fma4(v, ld4(cafc + coff), ld4(cafc + kC + coff))) * rz);
                    if (HAS_A0) v = v + sv[hs][it];
                } else if (HAS_A0) {
                    v = v + av[hs][it];
                }
                if (prow + 4 * it < Mb) st4_off(y_b, ((unsigned)(prow + 4 * it) * kC + cq) * 4u, v);
                if (PROJ) {
                    st4(Ot + r * LDO + cq, pack4<NT>(prelu4(fma4(v, cgw, cgb), epi.slope)));  // the next block's gateway, in place
                    // two of the 16 projection-weight fragments per iteration, into the registers sv / av just left (L2 latency
                    // hidden behind the rest of this loop)
                    if constexpr (!DEEP) wps[2 * it] = ld4(wpp + 8 * it), wps[2 * it + 1] = ld4(wpp + 8 * it + 4);
                }
            }
            if constexpr (DEEP == 2) {  // this half's register set is free: the same half of the NEXT tile, in flight under the projection / the other half
                __builtin_amdgcn_sched_barrier(0);
                load_sv(m0_next, pt);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (PROJ) {
                // projection of the 32 gated pixels now in Ot, as proj_kernel does it: 16x16x4 MFMA, wave w owns output channels
                // 16w .. 16w+15 for both 16-pixel sub-tiles and the whole K = 256; lane group kk = lane >> 4 takes k = 64kk + s.
                // The weight fragments (64 VGPRs) are streamed from L2 into the registers the epilogue loads just left.
                const int j = lane & 15, kk = lane >> 4;
                __syncthreads();  // Ot holds the gated tile
                // four independent accumulator chains (2 pixel sub-tiles x even / odd k quads): the 40-cycle dependent latency of
                // the 32-cycle instruction never shows
                floatx4 pa[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) pa[c] = floatx4{0.f, 0.f, 0.f, 0.f};
                const float* ap = Ot + j * LDO + 64 * kk;
                auto wp = [&](int t) -> const float4& {
                    if constexpr (DEEP) return wpr[t];
                    else return wps[t];
                };
                if constexpr (NT == 0) {
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const float4 e0 = ld4(ap + 4 * t), e1 = ld4(ap + 16 * LDO + 4 * t);
                        const int c = 2 * (t & 1);
                        pa[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp(t).x, e0.x, pa[c], 0, 0, 0);
                        pa[c + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp(t).x, e1.x, pa[c + 1], 0, 0, 0);
                        pa[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp(t).y, e0.y, pa[c], 0, 0, 0);
                        pa[c + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp(t).y, e1.y, pa[c + 1], 0, 0, 0);
                        pa[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp(t).z, e0.z, pa[c], 0, 0, 0);
                        pa[c + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp(t).z, e1.z, pa[c + 1], 0, 0, 0);
                        pa[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp(t).w, e0.w, pa[c], 0, 0, 0);
                        pa[c + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp(t).w, e1.w, pa[c + 1], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int t2 = 0; t2 < 8; ++t2) {  // 16x16x32: lane group kk supplies the 8 k values of packed slots 2 t2, 2 t2 + 1
                        const Frag e0 = frag_lds<NT>(ld4(ap + 8 * t2), ld4(ap + 8 * t2 + 4));
                        const Frag e1 = frag_lds<NT>(ld4(ap + 16 * LDO + 8 * t2), ld4(ap + 16 * LDO + 8 * t2 + 4));
                        const Frag wq = frag_lds<NT>(wp(2 * t2), wp(2 * t2 + 1));
                        const int c = 2 * (t2 & 1);
                        mma16<NT>(pa[c], wq, e0);
                        mma16<NT>(pa[c + 1], wq, e1);
                    }
                }
                const float4 pb4 = ld4(epi.pbias + 16 * w + 4 * kk);
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const int p = m0 + pt * 32 + 16 * sub + j;
                    if (p < Mb) {
                        const float4 o = f4(pa[sub][0] + pa[sub + 2][0], pa[sub][1] + pa[sub + 2][1], pa[sub][2] + pa[sub + 2][2],
                                            pa[sub][3] + pa[sub + 2][3]) + pb4;
                        st4(epi.py + ((size_t)b * Mb + p) * kH + 16 * w + 4 * kk, o);
                        float s4, q4;
                        stat_sums(o, s4, q4);  // (scalar VALU: no packed op_sel instruction next to the bf16 MFMAs)
                        ps += s4;
                        pq += q4;
                    }
                }
            }
        }
    }
    if (PROJ) {
        __syncthreads();
        block_stats_commit(ps, pq, pred, epi.pslot, b);
    }
}

// ------------------------------------------------------------------------------------------------
// Warp-specialised form of the projection-carrying residual kernels at large batch (round 3): ONE 8-wave workgroup per CU, waves 0-3
// ("M") issue nothing but MFMAs and their LDS fragment traffic, waves 4-7 ("X") own every global access and all element-wise work.  Each
// SIMD then holds one M and one X wave: while the X wave waits for HBM, LDS or the barrier, the M wave keeps the matrix pipe busy - in the
// one-workgroup DEEP form above every wave paid the LDS / barrier / load -> use latencies of its epilogue with the matrix pipe idle
// (15 us per 64-pixel tile against 9.3 us of arithmetic and 10.6 us of HBM time at 6 TB/s).
// The tile loop is a three-stage software pipeline over 32-pixel HALVES h, one barrier per phase:
//   phase h   M:  residual conv of half h (Es[h & 1] -> accumulators -> Ot[h % 3], pixel-major);  projection of half h - 2 (gated Ot[(h - 2) % 3])
//             X:  epilogue of half h - 1 on Ot[(h - 1) % 3] (+ bias + gateway(s_in) + a0 | CAF cell, store, next block's gateway in place);
//                 re-fill that half's residual-operand registers for half h + 1;  TFAR tail of half h + 1 -> Es[(h + 1) & 1];  E operands of half h + 2
// so a residual operand is requested two phases and an E operand one phase (>= 4 us) before its first use: 128 KB of s_in / a0 and 64 KB of E
// reads are in flight per CU at any time, none of them in M's way.  Per element the same arithmetic in the same order as resid_kernel.
// ------------------------------------------------------------------------------------------------
template <int NT, bool CAF>
__global__ __launch_bounds__(512, 2) void resid_ws_kernel(ProExpanded pro, EpiResidual epi, const float* __restrict__ Wt, int Mb, int tiles_per_wg) {
    constexpr int LDE = 68, LDO = 260;
    __shared__ __attribute__((aligned(16))) float Es[2][32 * LDE];
    __shared__ __attribute__((aligned(16))) float Ot[3][32 * LDO];
    __shared__ __attribute__((aligned(16))) float Ns[4][2][kH];
    __shared__ __attribute__((aligned(16))) float cafc[CAF ? 4 * kC : 4];
    __shared__ float pred[8];
    const int b = blockIdx.y;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave index in an SGPR: the role branches below are scalar
    const bool mrole = w < 4;
    const int lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const int tiles = (Mb + 63) / 64, tile0 = blockIdx.x * tiles_per_wg;
    const int ntl = min(tiles_per_wg, tiles - tile0);
    if (ntl <= 0) return;
    const int H = 2 * ntl, row0 = tile0 * 64;  // halves of this workgroup; half h covers pixel rows row0 + 32 h .. + 31
    if (threadIdx.x < 256) {
        if (CAF) {
            const float* src[4] = {epi.caf_ks, epi.caf_kb, epi.caf_vs, epi.caf_vb};
#pragma unroll
            for (int j = 0; j < 4; ++j) cafc[j * kC + threadIdx.x] = src[j][threadIdx.x];
        }
        const int j = threadIdx.x >> 6, c = threadIdx.x & 63;
        float sc, sh;
        pro.fold(b, j, c, sc, sh);
        Ns[j][0][c] = sc;
        Ns[j][1][c] = sh;
    }
    __syncthreads();
    float ps = 0.f, pq = 0.f;  // M: gLN partial sums of the projection output
#ifdef RESID_TIMING
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define RT_MARK(j) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[j] += t_ - tlast; tlast = t_; } while (0)
#else
#define RT_MARK(j) do { } while (0)
#endif

    if (mrole) {
        // ================================================ M waves ================================================
        float4 wf[2][8];  // residual-conv weight fragments: rows n = 64w + 32nt + i, k = 8q + 4kh .. +3
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 8; ++q) wf[nt][q] = ld4(Wt + (size_t)(64 * w + 32 * nt + i) * 64 + 8 * q + 4 * kh);
        float4 wpr[16];   // projection weight fragments: W_p[16w + (lane & 15)][64 (lane >> 4) + 4t .. +3]
        const int j = lane & 15, kk = lane >> 4;
        {
            const float* wpp0 = epi.pw + (size_t)(16 * w + j) * 256 + 64 * kk;
#pragma unroll
            for (int t = 0; t < 16; ++t) wpr[t] = ld4(wpp0 + 4 * t);
        }
        const float4 pb4 = ld4(epi.pbias + 16 * w + 4 * kk);
        const __amdgpu_buffer_rsrc_t rpy = __builtin_amdgcn_make_buffer_rsrc(epi.py + (size_t)b * Mb * kH, 0, (int)((unsigned)Mb * kH * 4u), 0x00020000);
        __syncthreads();  // (pairs with the X waves' prologue barrier: Es[0] holds half 0)
        int hb = 0;       // h % 3
#pragma unroll 1
        for (int h = 0; h < H + 2; ++h) {
            RT_MARK(5);
            if (h < H) {  // residual conv of half h
                const float* E = Es[h & 1];
                floatx16 acc[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
                if constexpr (NT == 0) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float4 e0 = ld4(E + i * LDE + 8 * q + 4 * kh);
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].x, e0.x, acc[nt], 0, 0, 0);
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].y, e0.y, acc[nt], 0, 0, 0);
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].z, e0.z, acc[nt], 0, 0, 0);
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].w, e0.w, acc[nt], 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {
                        const Frag e0 = frag_lds<NT>(ld4(E + i * LDE + 16 * q2 + 4 * kh), ld4(E + i * LDE + 16 * q2 + 8 + 4 * kh));
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) mma32<NT>(acc[nt], frag_lds<NT>(wf[nt][2 * q2], wf[nt][2 * q2 + 1]), e0);
                    }
                }
                float* O = Ot[hb];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) st4(O + i * LDO + 64 * w + 32 * nt + 8 * g + 4 * kh, acc_group(acc[nt], g));
            }
            RT_MARK(0);
            if (h >= 2) {  // projection of half h - 2 (its gated tile was written by the X waves in phase h - 1)
                const float* ap = Ot[hb == 2 ? 0 : hb + 1] + j * LDO + 64 * kk;
                floatx4 pa[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) pa[c] = floatx4{0.f, 0.f, 0.f, 0.f};
                if constexpr (NT == 0) {
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const float4 e0 = ld4(ap + 4 * t), e1 = ld4(ap + 16 * LDO + 4 * t);
                        const int c = 2 * (t & 1);
                        pa[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpr[t].x, e0.x, pa[c], 0, 0, 0);
                        pa[c + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpr[t].x, e1.x, pa[c + 1], 0, 0, 0);
                        pa[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpr[t].y, e0.y, pa[c], 0, 0, 0);
                        pa[c + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpr[t].y, e1.y, pa[c + 1], 0, 0, 0);
                        pa[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpr[t].z, e0.z, pa[c], 0, 0, 0);
                        pa[c + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpr[t].z, e1.z, pa[c + 1], 0, 0, 0);
                        pa[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpr[t].w, e0.w, pa[c], 0, 0, 0);
                        pa[c + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpr[t].w, e1.w, pa[c + 1], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int t2 = 0; t2 < 8; ++t2) {
                        const Frag e0 = frag_lds<NT>(ld4(ap + 8 * t2), ld4(ap + 8 * t2 + 4));
                        const Frag e1 = frag_lds<NT>(ld4(ap + 16 * LDO + 8 * t2), ld4(ap + 16 * LDO + 8 * t2 + 4));
                        const Frag wq = frag_lds<NT>(wpr[2 * t2], wpr[2 * t2 + 1]);
                        const int c = 2 * (t2 & 1);
                        mma16<NT>(pa[c], wq, e0);
                        mma16<NT>(pa[c + 1], wq, e1);
                    }
                }
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const int p = row0 + (h - 2) * 32 + 16 * sub + j;
                    const float4 o = f4(pa[sub][0] + pa[sub + 2][0], pa[sub][1] + pa[sub + 2][1], pa[sub][2] + pa[sub + 2][2],
                                        pa[sub][3] + pa[sub + 2][3]) + pb4;
                    __builtin_amdgcn_raw_buffer_store_b128(uint4v{__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)}, rpy,
                                                           (int)(((unsigned)p * kH + 16 * w + 4 * kk) * 4u), 0, 0);
                    if (p < Mb) {  // (scalar VALU through inline asm: see stat_sums)
                        float s4, q4;
                        stat_sums(o, s4, q4);
                        ps += s4;
                        pq += q4;
                    }
                }
            }
            hb = hb == 2 ? 0 : hb + 1;
            RT_MARK(1);
            __syncthreads();
            RT_MARK(4);
        }
    } else {
        // ================================================ X waves ================================================
        const int tx = threadIdx.x - 256;
        const int cq = (tx & 63) * 4, c4 = (tx & 15) * 4, er = tx >> 4, xr = tx >> 6;
        const float4 cbias = ld4(epi.bias + cq), cgw = ld4(epi.gw + cq), cgb = ld4(epi.gb + cq);
        const float* cl_b = pro.cl.x + (size_t)b * Mb * kH;
        const float* d0_b = pro.d0.x + (size_t)b * Mb * kH;
        const float* cg_b = pro.cg.x + (size_t)b * pro.T2 * kF2 * kH;
        const float* cgate_b = pro.cgate.x + (size_t)b * pro.T2 * kF2 * kH;
        const float* s_b = epi.s_in + (size_t)b * Mb * kC;
        const float* a0_b = CAF ? nullptr : epi.a0 + (size_t)b * Mb * kC;
        const float* att_b = CAF ? epi.att + (size_t)b * epi.Tv * kC : nullptr;
        const float* rsz_b = CAF ? epi.rsz + (size_t)b * epi.Tv * kC : nullptr;
        // (stores through a buffer descriptor: rows past the end are dropped by its range check.  A store under a branch makes hipcc turn every
        // later wait for an older load into a wait for the store acknowledgements as well - see ws256_kernel)
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(epi.y + (size_t)b * Mb * kC, 0, (int)((unsigned)Mb * kC * 4u), 0x00020000);
        float4 xa[2][2], xd[2][2], xg[2][2], xs[2][2];  // E operands, [register stage][row]
        float4 sv[2][8], av[CAF ? 1 : 2][8];            // residual operands, [register stage][row]
        float4 catt[2][2], crsz[2][2];                  // CAF: (att, rsz) rows of the <= 2 video frames a half spans
        int caf_split[2] = {0, 0};
        auto load_e = [&](auto st_, int h) {
            constexpr int st = decltype(st_)::value;
            const int m0 = row0 + 32 * min(h, H - 1);  // (past the last half: re-fetch it - L2 hits, no branch around the loads)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = min(m0 + er + 16 * it, Mb - 1);
                const int t = row / kF, f = row - t * kF;
                const int t2 = nearest_src(t, pro.T2, pro.T), f2 = nearest_src(f, kF2, kF);
                const unsigned hi = ((unsigned)row * kH + c4) * 4u, lo = (((unsigned)t2 * kF2 + f2) * kH + c4) * 4u;
                xa[st][it] = ld4_off(cl_b, hi), xd[st][it] = ld4_off(d0_b, hi), xg[st][it] = ld4_off(cg_b, lo), xs[st][it] = ld4_off(cgate_b, lo);
            }
        };
        auto xform_e = [&](auto st_, float* E) {
            constexpr int st = decltype(st_)::value;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const float4 a = fma4(xa[st][it], ld4(&Ns[0][0][c4]), ld4(&Ns[0][1][c4])), d = fma4(xd[st][it], ld4(&Ns[1][0][c4]), ld4(&Ns[1][1][c4]));
                const float4 g = fma4(xg[st][it], ld4(&Ns[2][0][c4]), ld4(&Ns[2][1][c4])), sg = sigmoid4(fma4(xs[st][it], ld4(&Ns[3][0][c4]), ld4(&Ns[3][1][c4])));
                st4(E + (er + 16 * it) * LDE + c4, pack4<NT>(fma4(a, sg, g) + d));
            }
        };
        auto load_sv = [&](auto st_, int h) {
            constexpr int st = decltype(st_)::value;
            const int m0 = row0 + 32 * min(h, H - 1);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const unsigned o = ((unsigned)min(m0 + xr + 4 * it, Mb - 1) * kC + cq) * 4u;
                sv[st][it] = ld4_off(s_b, o);
                if constexpr (!CAF) av[st][it] = ld4_off(a0_b, o);
            }
            if constexpr (CAF) {
                const int p0 = min(m0, Mb - 1), p1 = min(m0 + 31, Mb - 1);
                const int t0 = p0 / kF, t1 = p1 / kF;
                caf_split[st] = (t0 + 1) * kF;
                const unsigned o0 = ((unsigned)nearest_src(t0, epi.Tv, pro.T) * kC + cq) * 4u, o1 = ((unsigned)nearest_src(t1, epi.Tv, pro.T) * kC + cq) * 4u;
                catt[st][0] = ld4_off(att_b, o0), crsz[st][0] = ld4_off(rsz_b, o0), catt[st][1] = ld4_off(att_b, o1), crsz[st][1] = ld4_off(rsz_b, o1);
            }
        };
        auto epilogue = [&](auto st_, int h, float* O) {  // half h from O (pixel-major accumulators) with the residual operands of stage st
            constexpr int st = decltype(st_)::value;
            const int prow = row0 + 32 * h + xr;
            int coff = cq;
            if (CAF) asm volatile("" : "+v"(coff));  // opaque: the table reads stay inside the loop
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = xr + 4 * it;
                float4 v = ld4(O + r * LDO + cq) + cbias + prelu4_minfma(fma4(sv[st][it], cgw, cgb), epi.slope - 1.0f);
                if constexpr (CAF) {
                    const bool second = prow + 4 * it >= caf_split[st];
                    const float4 at = second ? catt[st][1] : catt[st][0], rz = second ? crsz[st][1] : crsz[st][0];
                    v = fma4(at, fma4(v, ld4(cafc + 2 * kC + coff), ld4(cafc + 3 * kC + coff)), relu4(fma4(v, ld4(cafc + coff), ld4(cafc + kC + coff))) * rz);
                    v = v + sv[st][it];
                } else {
                    v = v + av[st][it];
                }
                __builtin_amdgcn_raw_buffer_store_b128(uint4v{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, ry,
                                                       (int)(((unsigned)(prow + 4 * it) * kC + cq) * 4u), 0, 0);
                st4(O + r * LDO + cq, pack4<NT>(prelu4(fma4(v, cgw, cgb), epi.slope)));  // the next block's gateway, in place
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        // the X waves' instructions go first whenever both waves of a SIMD are ready: their loads / stores / LDS traffic are the long-latency
        // chain of a phase, the M wave's MFMAs fill whatever issue slots are left (same-box A/B: 977-993 -> 960-968 us per launch)
        __builtin_amdgcn_s_setprio(3);
        load_e(S0{}, 0);
        load_e(S1{}, 1);
        load_sv(S0{}, 0);
        xform_e(S0{}, Es[0]);
        __syncthreads();
        int hb = 0;
        // phase h, stage p = h & 1: epilogue(h - 1) with stage p ^ 1, refill it for half h + 1, E(h + 1) from stage p ^ 1 -> Es[p ^ 1], fetch E(h + 2) -> stage p
        auto phase = [&](auto p_, int h) {
            constexpr int p = decltype(p_)::value;
            using Q = std::integral_constant<int, p ^ 1>;
            RT_MARK(5);
            if (h >= 1 && h <= H) epilogue(Q{}, h - 1, Ot[hb == 0 ? 2 : hb - 1]);
            __builtin_amdgcn_sched_barrier(0);
            RT_MARK(0);
            load_sv(Q{}, h + 1);
            __builtin_amdgcn_sched_barrier(0);
            RT_MARK(1);
            xform_e(Q{}, Es[p ^ 1]);
            RT_MARK(2);
            load_e(p_, h + 2);
            __builtin_amdgcn_sched_barrier(0);
            hb = hb == 2 ? 0 : hb + 1;
            RT_MARK(3);
            __syncthreads();
            RT_MARK(4);
        };
#pragma unroll 1
        for (int h = 0; h < H + 2; h += 2) {
            phase(S0{}, h);
            phase(S1{}, h + 1);
        }
    }
    // gLN partial sums of the projection output: M waves only
    ps = wave_sum(ps);
    pq = wave_sum(pq);
    if (mrole && lane == 0) {
        pred[w] = ps;
        pred[4 + w] = pq;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(epi.pslot + kStatStride * b, (double)pred[0] + (double)pred[1] + (double)pred[2] + (double)pred[3]);
        atomicAdd(epi.pslot + kStatStride * b + 1, (double)pred[4] + (double)pred[5] + (double)pred[6] + (double)pred[7]);
    }
#ifdef RESID_TIMING
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {  // this wave's segment sums over the workgroup's first output rows (the output is garbage in this build)
        unsigned long long* o = reinterpret_cast<unsigned long long*>(epi.y + ((size_t)b * Mb + row0 + w) * kC);
        for (int j = 0; j < 6; ++j) o[j] = tacc[j];
        o[6] = (unsigned long long)H;
    }
#endif
#undef RT_MARK
}

// ------------------------------------------------------------------------------------------------
// gateway + projection (K = 256 -> N = 64) on v_mfma_f32_16x16x4_f32, operands swapped like resid_kernel:
// y^T[n][p] = Wp[n][k] . A^T[k][p].  64-pixel tile; wave w owns output channels 16w..16w+15 for ALL 64 pixels and the
// WHOLE K, so there is no cross-wave reduction and every wave has the same epilogue.  The MFMA's K index is free as long
// as both operands agree: lane group kk = lane>>4 takes k = 64kk + s at step s, which makes a lane's weights 64
// consecutive floats (16 x float4, resident in VGPRs for the life of the workgroup) and its pixel operand 4 consecutive
// floats per ds_read_b128 (feeds 4 MFMAs).  4 independent accumulators (the 4 pixel sub-tiles) are interleaved, so the
// 40-cycle dependent latency of the 32-cycle instruction never shows.  The A tile (gateway applied on its way in) is the
// only thing in LDS (66.6 KB -> 2 workgroups per CU: one in its 256-MFMA phase while the other stores / fetches /
// writes back).  ~180 VGPRs, no spills (the 32x32x2 variants of this kernel needed 128 VGPRs of weights and spilled).
// ------------------------------------------------------------------------------------------------
template <int NT = 0>  // NT != 0: Wt is host-PACKED (common.h)
__global__ __launch_bounds__(256, 2) void proj_kernel(ProGateway pro, EpiBiasStats epi, const float* __restrict__ Wt, int Mb, int tiles_per_wg) {
    constexpr int LDA = 260, TM = 64;
    __shared__ __attribute__((aligned(16))) float As[TM * LDA];
    __shared__ float red[8];
    const int b = blockIdx.y;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, kk = lane >> 4;

    float4 wf[16];  // W[16w + j][64kk .. 64kk+63]
#pragma unroll
    for (int t = 0; t < 16; ++t) wf[t] = ld4(Wt + (size_t)(16 * w + j) * 256 + 64 * kk + 4 * t);
    const float4 bias4 = ld4(epi.bias + 16 * w + 4 * kk);
    const int c4 = (threadIdx.x & 63) * 4;  // this thread's channel quad of the A tile (constant across rows)
    const float4 gw4 = ld4(pro.gw + c4), gb4 = ld4(pro.gb + c4);

    float s = 0.f, qq = 0.f;
    const int tile0 = blockIdx.x * tiles_per_wg;
    float4 araw[16];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int it = 0; it < 16; ++it) araw[it] = ld4(pro.x + ((size_t)b * Mb + min(m0 + w + it * 4, Mb - 1)) * kC + c4);
    };
    fetch(tile0 * TM);
#pragma unroll 1
    for (int tl = 0; tl < tiles_per_wg; ++tl) {
        const int m0 = (tile0 + tl) * TM;
        if (m0 >= Mb) break;
#pragma unroll
        for (int it = 0; it < 16; ++it) st4(As + (w + it * 4) * LDA + c4, pack4<NT>(prelu4(fma4(araw[it], gw4, gb4), pro.slope)));
        if (tl + 1 < tiles_per_wg) fetch(min(m0 + TM, Mb - 1));  // flies under this tile's MFMAs
        __syncthreads();
        floatx4 acc[4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) acc[pt] = floatx4{0.f, 0.f, 0.f, 0.f};
        const float* ap = As + j * LDA + 64 * kk;
        if constexpr (NT == 0) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                float4 e[4];
    #pragma unroll
                for (int pt = 0; pt < 4; ++pt) e[pt] = ld4(ap + pt * 16 * LDA + 4 * t);
    #pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t].x, e[pt].x, acc[pt], 0, 0, 0);
    #pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t].y, e[pt].y, acc[pt], 0, 0, 0);
    #pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t].z, e[pt].z, acc[pt], 0, 0, 0);
    #pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t].w, e[pt].w, acc[pt], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int t2 = 0; t2 < 8; ++t2) {
                const Frag wq = frag_lds<NT>(wf[2 * t2], wf[2 * t2 + 1]);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
                    mma16<NT>(acc[pt], wq, frag_lds<NT>(ld4(ap + pt * 16 * LDA + 8 * t2), ld4(ap + pt * 16 * LDA + 8 * t2 + 4)));
            }
        }
        __syncthreads();  // As consumed by every wave: the next tile may overwrite it while the epilogues run
        // accumulator of sub-tile pt: channels 16w + 4kk .. +3 (registers 0..3) of pixel m0 + 16pt + j
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int p = m0 + 16 * pt + j;
            if (p < Mb) {
                const float4 v = f4(acc[pt][0], acc[pt][1], acc[pt][2], acc[pt][3]) + bias4;
                st4(epi.y + ((size_t)b * Mb + p) * kH + 16 * w + 4 * kk, v);
                float s4, q4;
                stat_sums(v, s4, q4);  // (scalar VALU: no packed op_sel instruction next to the bf16 MFMAs)
                s += s4;
                qq += q4;
            }
        }
    }
    __syncthreads();
    block_stats_commit(s, qq, red, epi.slot, b);
}

// ------------------------------------------------------------------------------------------------
// Weight-stationary form of the 256 -> 256 pixel GEMMs at large batch (audio bottleneck tdavnet.py:59,89; S3 mask
// mask_generator.py:67-99; their input-gradient GEMMs in the training step), fp32, round 3.
// pixel_gemm_kernel<256, 256, 64, ...> re-stages the whole 256 KB weight through LDS for every 64-pixel tile (16 chunks, one
// barrier each: 4 GB of L2 -> LDS weight traffic per launch at B = 32, matrix pipe 67-71 % busy).  Here ONE 4-wave workgroup
// per CU keeps the weight in registers for its whole life - wave w owns output channels [64 w, 64 w + 64) x all 256 k =
// 256 registers per lane, the MFMA A operand - and walks `tiles_per_wg` consecutive 32-pixel tiles of one utterance.  Only the
// pixel tile goes through LDS (the B operand: one ds_read_b128 per 8 MFMAs, requested one step ahead), and the K loop of a tile -
// 256 MFMAs per wave - never stops for a tile hand-over: everything else rides between its steps,
//   steps 0-7    previous tile's accumulators (read out of the matrix registers at the end of their tile) -> LDS, pixel-major;
//   step  8      barrier;   steps 9-17  previous tile's epilogue: whole 1 KB pixel rows LDS -> registers -> epi -> HBM;
//   steps 18-25  next tile's rows (requested one tile ago, in registers since) -> pro.xform -> the other LDS tile;
//   steps 26-27  global loads of the tile after next (and of the S3 embedding rows this tile's epilogue will need);
// one more barrier at the end of the tile (this tile's fragments all read, next tile written).  What the matrix pipe still waits for
// is the VALU share of that work (rule 9 of DESIGN.md: fp32 MFMA and VALU do not overlap) - ~200 instructions per 16384 pipe cycles.
// Same products, same k order per accumulator as pixel_gemm_kernel (k ascending in 8-quads) => bit-identical output.
// ------------------------------------------------------------------------------------------------
template <class Pro, class Epi, bool PAIRED, int NT = 0>  // NT (common.h): 0 fp32; 1 / 3 bf16 / split-bf16 MFMA, Wt host-PACKED, the pixel tile packed on store
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void ws256_kernel(Pro pro, Epi epi, const float* __restrict__ Wt, int Mb, int tiles_per_wg) {
    constexpr int LD = 260, TP = 32;
    __shared__ __attribute__((aligned(16))) float As[2][TP * LD];  // pixel tiles [pixel][k], double-buffered
    __shared__ __attribute__((aligned(16))) float Ot[TP * LD];     // previous tile's output [pixel][channel]
    __shared__ __attribute__((aligned(16))) float ptab[2 * kC];
    const int b = blockIdx.y;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    pro.init(b, ptab);
    if constexpr (Epi::kSide == 2) ptab[threadIdx.x] = epi.gamma[threadIdx.x], ptab[kC + threadIdx.x] = epi.beta[threadIdx.x];

    float4 wf[2][32];  // W fragments: rows n = 64 w + 32 nt + i, k = 8 q + 4 kh .. +3
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 32; ++q) wf[nt][q] = ld4(Wt + (size_t)(64 * w + 32 * nt + i) * kC + 8 * q + 4 * kh);
    // half of the weight lives in the accumulation registers (256 + 256 per lane) and is read by the MFMAs from there: bound to that class
    // here, or hipcc treats those registers as spill slots and copies every value back to a VGPR before its MFMA (~100 VALU per tile)
    // bf16 modes: the host-packed slots [hi(k0 k1) hi(k2 k3) lo(k0 k1) lo(k2 k3)] of step q2 = quads 2 q2, 2 q2 + 1 regrouped ONCE into the 4-register
    // operand tuples of v_mfma_f32_32x32x16_bf16 (hi and lo planes of the 8 k values 16 q2 + 4 kh + {0..3, 8..11}; see unfold_ws_kernel, dualpath.hip)
    bf16x8 whi[2][NT ? 16 : 1], wlo[2][NT ? 16 : 1];
    if constexpr (NT == 0) {
#pragma unroll
        for (int q = 0; q < 32; ++q) asm volatile("" : "+a"(wf[0][q].x), "+a"(wf[0][q].y), "+a"(wf[0][q].z), "+a"(wf[0][q].w));
    } else {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q2 = 0; q2 < 16; ++q2) {
                const float4 s0 = wf[nt][2 * q2], s1 = wf[nt][2 * q2 + 1];
                whi[nt][q2] = __builtin_bit_cast(bf16x8, uint4v{__float_as_uint(s0.x), __float_as_uint(s0.y), __float_as_uint(s1.x), __float_as_uint(s1.y)});
                wlo[nt][q2] = __builtin_bit_cast(bf16x8, uint4v{__float_as_uint(s0.z), __float_as_uint(s0.w), __float_as_uint(s1.z), __float_as_uint(s1.w)});
            }
#pragma unroll
        for (int q2 = 0; q2 < 16; ++q2) asm volatile("" : "+a"(whi[0][q2]), "+a"(whi[1][q2]));
    }
    __syncthreads();  // ptab

    const int ntile = (Mb + TP - 1) / TP;
    const int tile0 = blockIdx.x * tiles_per_wg, tile_end = min(tile0 + tiles_per_wg, ntile);
    if (tile0 >= ntile) return;
    // Every global access of the tile loop goes through a buffer descriptor of this utterance's [Mb][256] slab: byte offset = per-thread
    // constant + wave-uniform tile offset (one VALU add per access), rows past the end are answered with zeros / dropped by the range
    // check - no clamps, no branches around loads or stores (a store under a branch makes every later vmcnt wait conservative).
    const unsigned slab = (unsigned)Mb * kC * 4u;
    const size_t uoff = (size_t)b * Mb * kC;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pro.x) + uoff, 0, (int)slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(epi.y + uoff, 0, (int)slab, 0x00020000);
    auto bld = [](const __amdgpu_buffer_rsrc_t& r, unsigned off) {
        const uint4v v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
        return f4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    };
    auto bst = [](const __amdgpu_buffer_rsrc_t& r, unsigned off, float4 v) {
        __builtin_amdgcn_raw_buffer_store_b128(uint4v{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, r, (int)off, 0, 0);
    };
    constexpr unsigned kNowhere = 0x40000000u;  // past the end of any slab (the launcher keeps slabs below 2^30 bytes)
    // staging and plain epilogue: thread = (channel quad cq, rows w + 4 it): one wave-instruction moves one pixel's whole 1 KB row
    const int cq = lane * 4;
    const unsigned voff = (unsigned)(w * kC + cq) * 4u;
    float4 raw[8];
    auto load_a = [&](int tile) {  // (tiles past the end re-fetch the last one: L2 hits)
        const unsigned base = (unsigned)(min(tile, tile_end - 1) * TP) * kC * 4u;
#pragma unroll
        for (int it = 0; it < 8; ++it) raw[it] = bld(rx, voff + (base + it * 4u * kC * 4u));
    };
    // fp32: the 4 channels as they are.  bf16 modes: a pixel row is 16 groups of 16 channels, each stored as [hi plane of the kh = 0 fragment half
    // (channels 0-3, 8-11) | hi, kh = 1 | lo, kh = 0 | lo, kh = 1], 16 bytes each: a lane's ds_read_b128 is a whole MFMA operand tuple
    const int cst = NT == 0 ? cq : (cq >> 4) * 16 + ((cq >> 2) & 1) * 4 + ((cq >> 3) & 1) * 2;
    auto store_a1 = [&](float* dst, int it) {
        const float4 y = pro.xform(raw[it], cq, ptab);
        if constexpr (NT == 0) {
            st4(dst + (w + 4 * it) * LD + cst, y);
        } else {
            const float4 pk = pack4<NT>(y);
            *reinterpret_cast<float2*>(dst + (w + 4 * it) * LD + cst) = make_float2(pk.x, pk.y);
            if constexpr (NT == 3) *reinterpret_cast<float2*>(dst + (w + 4 * it) * LD + cst + 8) = make_float2(pk.z, pk.w);
        }
    };
    // S3 mask epilogue: thread = (channel quad q4 of the real half, rows pr + 8 it)
    const int q4 = (threadIdx.x & 31) * 4, pr = threadIdx.x >> 5;
    const unsigned voffe = (unsigned)(pr * kC + q4) * 4u;
    float4 er[4], ei[4];
    __amdgpu_buffer_rsrc_t re = ry, rm = ry;
    if constexpr (PAIRED) {
        re = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(epi.emb) + uoff, 0, (int)slab, 0x00020000);
        rm = __builtin_amdgcn_make_buffer_rsrc(epi.m_out ? epi.m_out + uoff : epi.y, 0, epi.m_out ? (int)slab : 0, 0x00020000);  // no mask output: every store dropped
    }
    auto load_emb = [&](unsigned base) {
        if constexpr (PAIRED) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const unsigned o = voffe + (base + it * 8u * kC * 4u);
                er[it] = bld(re, o), ei[it] = bld(re, o + 512u);
            }
        }
    };
    // activation adjoint in the epilogue (EpiAdjoint): the activation's input rows of the tile whose epilogue runs next, fetched with the plain epilogue's
    // thread map (whole 1 KB rows, channel quad cq fixed per thread => the per-channel sums stay in registers for the workgroup's whole life)
    constexpr int SIDE = Epi::kSide;
    __shared__ float redl[8];
    float4 xs[8];
    __amdgpu_buffer_rsrc_t rs = ry;
    float4 dgam = f4(0, 0, 0, 0), dbet = f4(0, 0, 0, 0);
    float smean = 0.f, srstd = 0.f, sum1 = 0.f;  // SIDE 1: sum1 = dslope
    if constexpr (SIDE != 0) rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(epi.x) + uoff, 0, (int)slab, 0x00020000);
    if constexpr (SIDE == 2) {  // (gamma | beta wait in ptab - ProPlain leaves it free; mean / rstd are wave-uniform: scalar registers)
        stats_finalize(epi.slot, b, epi.inv_n, smean, srstd);
        smean = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, smean)));
        srstd = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, srstd)));
    }
    auto load_side = [&](unsigned base) {
        if constexpr (SIDE != 0) {
#pragma unroll
            for (int it = 0; it < 8; ++it) xs[it] = bld(rs, voff + (base + it * 4u * kC * 4u));
        }
    };
    float4 hold[8];  // previous tile's accumulators [nt][g]
    float4 orow[8];  // previous tile's output rows on their way out
    if constexpr (SIDE != 0) {  // (the first tile's hand-over runs a dropped epilogue over these: its sums must see zeros, not whatever the registers held)
#pragma unroll
        for (int it = 0; it < 8; ++it) hold[it] = f4(0, 0, 0, 0);
    }
    auto ot_write = [&](int it) { st4(Ot + i * LD + 64 * w + 32 * (it >> 2) + 8 * (it & 3) + 4 * kh, hold[it]); };
    auto ot_read = [&]() {
#pragma unroll
        for (int it = 0; it < 8; ++it)
            orow[it] = PAIRED ? ld4(Ot + (pr + 8 * (it & 3)) * LD + q4 + 128 * (it >> 2)) : ld4(Ot + (w + 4 * it) * LD + cq);
    };
    float4 cc = f4(0, 0, 0, 0), cci = f4(0, 0, 0, 0);  // per-thread column constants (bias quads), loaded once
    if constexpr (PAIRED)
        cc = ld4(epi.bias + q4), cci = ld4(epi.bias + q4 + 128);
    else
        cc = epi.colconst(cq);
    auto epi_out = [&](int it, unsigned base) {
        if constexpr (PAIRED) {  // EpiMask::store2e (mask_generator.py:70-82)
            if (it < 4) {
                const unsigned o = voffe + (base + it * 8u * kC * 4u);
                const float4 mr = relu4(orow[it] + cc), mi = relu4(orow[it + 4] + cci), xr = er[it], xi = ei[it];
                bst(rm, o, mr);
                bst(rm, o + 512u, mi);
                bst(ry, o, f4(xr.x * mr.x - xi.x * mi.x, xr.y * mr.y - xi.y * mi.y, xr.z * mr.z - xi.z * mi.z, xr.w * mr.w - xi.w * mi.w));
                bst(ry, o + 512u, f4(fmaf(xr.x, mi.x, xi.x * mr.x), fmaf(xr.y, mi.y, xi.y * mr.y), fmaf(xr.z, mi.z, xi.z * mr.z), fmaf(xr.w, mi.w, xi.w * mr.w)));
            }
        } else if constexpr (SIDE == 1) {  // prelu_bwd_kernel's arithmetic (bwd_misc.hip); rows past the end: x reads as 0, the accumulator is 0
            const float4 g = orow[it], v = xs[it];
            sum1 += (v.x > 0.f ? 0.f : g.x * v.x) + (v.y > 0.f ? 0.f : g.y * v.y) + (v.z > 0.f ? 0.f : g.z * v.z) + (v.w > 0.f ? 0.f : g.w * v.w);
            bst(ry, voff + (base + it * 4u * kC * 4u),
                f4(v.x > 0.f ? g.x : g.x * epi.slope, v.y > 0.f ? g.y : g.y * epi.slope, v.z > 0.f ? g.z : g.z * epi.slope, v.w > 0.f ? g.w : g.w * epi.slope));
        } else if constexpr (SIDE == 2) {  // gln_bwd_reduce_kernel<256, 2>'s arithmetic (bwd_elem.hip)
            // (S1 = sum g gamma and S2 = sum g gamma xhat of this utterance are sum_c gamma_c dbeta_c and sum_c gamma_c dgamma_c: formed once, at the end)
            const float4 xh = sub4(xs[it], smean) * srstd;
            const float4 yy = fma4(xh, ld4(ptab + cq), ld4(ptab + kC + cq));
            float4 g = orow[it];
            bst(ry, voff + (base + it * 4u * kC * 4u), g);
            g = f4(yy.x > 0.f ? g.x : 0.f, yy.y > 0.f ? g.y : 0.f, yy.z > 0.f ? g.z : 0.f, yy.w > 0.f ? g.w : 0.f);
            dbet = dbet + g;
            dgam = fma4(g, xh, dgam);
        } else {  // EpiBias<HAS_BIAS, false>::store
            bst(ry, voff + (base + it * 4u * kC * 4u), orow[it] + cc);
        }
    };

    load_a(tile0);
#pragma unroll
    for (int it = 0; it < 8; ++it) store_a1(As[0], it);
    load_emb(kNowhere);  // (not used: the same sequence of memory operations as at the end of a tile)
    load_side(kNowhere);
    load_a(tile0 + 1);
    __syncthreads();
    unsigned prev_base = kNowhere;  // no previous tile yet: every store of its epilogue is dropped

#pragma unroll 1
    for (int tile = tile0; tile < tile_end; ++tile) {
        const int cur = (tile - tile0) & 1;
        const float* ep = As[cur] + i * LD + 4 * kh;
        float* An = As[cur ^ 1];
        const unsigned tile_base = (unsigned)(tile * TP) * kC * 4u;
        floatx16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        // one wave per SIMD: nobody else covers the LDS latency, so the fragment of step q + 1 is read before the MFMAs of step q
        // (every step pinned with sched_barrier: left alone, hipcc reads each fragment right before its first MFMA and waits for it).
        // The tile's hand-over work as 32 slots, one per step of the fp32 K loop, two per 16-k step of the bf16 loops, each BETWEEN two MFMAs
        // (its LDS / buffer instructions then issue in the shadow of a running MFMA).  Order of the memory operations inside a tile: stores,
        // then loads, and every load is consumed in the NEXT tile before that tile's stores - vmcnt counts loads and stores alike, and a wait
        // for a load that has younger stores behind it waits for their acknowledgements as well.
        auto piece = [&](int q) {
            if (q < 8) ot_write(q);
            if (q == 8) __syncthreads();
            if (q == 9) ot_read();
            if (q >= 10 && q < 18) epi_out(q - 10, prev_base);
            if (q == 14) load_emb(tile_base);  // this tile's embedding rows: its epilogue runs inside the next tile
            if (q == 18) load_side(tile_base);  // after the epilogue's last store; consumed by this tile's epilogue inside the next tile
            if (q >= 18 && q < 26) store_a1(An, q - 18);
            if (q == 26) load_a(tile + 2);
        };
        if constexpr (NT == 0) {
            float4 eb[2];
            eb[0] = ld4(ep);
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                if (q + 1 < 32) eb[(q + 1) & 1] = ld4(ep + 8 * (q + 1));
                __builtin_amdgcn_sched_barrier(0);
                const float4 e = eb[q & 1];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[0][q].x, e.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[1][q].x, e.x, acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                piece(q);
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[0][q].y, e.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[1][q].y, e.y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[0][q].z, e.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[1][q].z, e.z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[0][q].w, e.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[1][q].w, e.w, acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // 16 k per step; (hi, lo) operand tuples of the pixel row, the two accumulator chains alternating (order per accumulator as mma32<NT>)
            auto tuple = [](float4 v) { return __builtin_bit_cast(bf16x8, uint4v{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}); };
            float4 ebp[2][2];
            ebp[0][0] = ld4(ep), ebp[0][1] = ld4(ep + 8);
#pragma unroll
            for (int q2 = 0; q2 < 16; ++q2) {
                if (q2 + 1 < 16) {
                    ebp[(q2 + 1) & 1][0] = ld4(ep + 16 * (q2 + 1));
                    if constexpr (NT == 3) ebp[(q2 + 1) & 1][1] = ld4(ep + 16 * (q2 + 1) + 8);
                }
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 fh = tuple(ebp[q2 & 1][0]), fl = tuple(ebp[q2 & 1][1]);
                if constexpr (NT == 3) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[0][q2], fh, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[1][q2], fh, acc[1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    piece(2 * q2);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[0][q2], fl, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[1][q2], fl, acc[1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    piece(2 * q2 + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[0][q2], fh, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[1][q2], fh, acc[1], 0, 0, 0);
                } else {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[0][q2], fh, acc[0], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    piece(2 * q2);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[1][q2], fh, acc[1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    piece(2 * q2 + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) hold[it] = acc_group(acc[it >> 2], it & 3);
        prev_base = tile_base;
        __syncthreads();  // every wave has read its last fragment of this tile and the previous output tile; the next tile is in LDS
    }
    // the last tile's epilogue
#pragma unroll
    for (int it = 0; it < 8; ++it) ot_write(it);
    __syncthreads();
    ot_read();
#pragma unroll
    for (int it = 0; it < 8; ++it) epi_out(it, prev_base);
    if constexpr (SIDE == 1) {
        sum1 = wave_sum(sum1);
        if (lane == 0) redl[w] = sum1;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(spread_copy(epi.scr, blockIdx.x + blockIdx.y), redl[0] + redl[1] + redl[2] + redl[3]);
    }
    if constexpr (SIDE == 2) {
        __syncthreads();  // Ot: every wave has read the last tile's rows
        st4(Ot + w * 512 + cq, dgam);
        st4(Ot + w * 512 + 256 + cq, dbet);
        __syncthreads();
        float* mine = spread_copy(epi.scr, blockIdx.x + blockIdx.y);  // [dgamma 256 | dbeta 256]; thread = channel: one coalesced atomic request per line
        const int c = threadIdx.x;
        atomicAdd(mine + c, Ot[c] + Ot[512 + c] + Ot[1024 + c] + Ot[1536 + c]);
        atomicAdd(mine + 256 + c, Ot[256 + c] + Ot[768 + c] + Ot[1280 + c] + Ot[1792 + c]);
        const float4 sg4 = ld4(ptab + cq);
        block_stats_commit(dot4(dbet, sg4), dot4(dgam, sg4), redl, epi.red, b);
    }
}

// the weight-stationary form pays one 256 KB weight read per workgroup: worth it from ~32 tiles per workgroup on
static bool ws256_applies(int B, int Mb) { return (long long)((Mb + 31) / 32) * B >= 32 * 256 && (long long)Mb * kC * 4 < (1ll << 30); }

template <bool PAIRED, int NT, class Pro, class Epi>
static int launch_ws256(const Pro& pro, const Epi& epi, const float* Wt, int B, int Mb, hipStream_t st) {
    if (B <= 0 || Mb <= 0) return RTFS_EINVAL;
    // one workgroup per CU: workgroups per utterance = ceil(256 / B), consecutive tiles of one utterance each
    const int tiles = (Mb + 31) / 32;
    const int per_utt = (256 + B - 1) / B;
    const int per = (tiles + per_utt - 1) / per_utt;
    hipLaunchKernelGGL((ws256_kernel<Pro, Epi, PAIRED, NT>), dim3((tiles + per - 1) / per, B), dim3(256), 0, st, pro, epi, Wt, Mb, per);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

template <int K, int N, int BM, int WM, int WN, bool PAIRED, int BK = 32, int NT = 0, class Pro, class Epi>
static int launch(const Pro& pro, const Epi& epi, const float* Wt, int B, int Mb, hipStream_t st) {
    if (B <= 0 || Mb <= 0) return RTFS_EINVAL;
    if constexpr (K == 256 && N == 256 && (NT == 0 || NT == 1 || NT == 3) && !Epi::kAccum) {
        if (ws256_applies(B, Mb)) return launch_ws256<PAIRED, NT>(pro, epi, Wt, B, Mb, st);
    }
    dim3 grid((Mb + BM - 1) / BM, B);
    hipLaunchKernelGGL((pixel_gemm_kernel<K, N, BM, WM, WN, PAIRED, BK, Pro, Epi, NT>), grid, dim3(256), 0, st, pro, epi, Wt, Mb);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}


// Row GEMM onto 64 output columns with the weight in registers (round 5; training step, fp32, large M): Y[M][64] (= or +=) X[M][K] . Wt[64][K]^T.
// The generic pixel_gemm_kernel re-stages the weight through LDS for every row tile and streams these narrow maps - dx += dU . W of the SRU layers (K = 192),
// the residual conv's input gradient (K = 256) - at 3.8 TB/s.  Here wave w keeps its 16 output columns x all K (K / 4 registers per lane, v_mfma_f32_16x16x4_f32:
// lane (n, kg) holds W[16 w + n][16 Q + 4 kg .. + 3]) for a contiguous range of 32-row tiles; only the rows go through LDS (double-buffered, the next tile's rows in
// flight under the MFMAs), a lane ends with 4 consecutive columns of one row (16-byte accumulate / store).  Two to three workgroups per CU overlap each other's phases.
template <int K, bool ACC>
__global__ __launch_bounds__(256, 2) void rows_ws64_kernel(const float* __restrict__ X, const float* __restrict__ Wt, float* __restrict__ Y, int M, int tiles_per_wg) {
    constexpr int BM = 32, LDX = K + 4, NQ = K / 16, XIT = BM * (K / 4) / 256;
    __shared__ __attribute__((aligned(16))) float Xs[2][BM * LDX];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, kg = lane >> 4;
    float4 wq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) wq[q] = ld4(Wt + (size_t)(16 * w + j) * K + 16 * q + 4 * kg);
    const int ntiles = (M + BM - 1) / BM;
    const int t0 = blockIdx.x * tiles_per_wg, t1 = min(ntiles, t0 + tiles_per_wg);
    if (t0 >= t1) return;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (int)((long long)M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(Y, 0, (int)((long long)M * 256), 0x00020000);
    float4 xr[XIT];
    auto fetch = [&](int t) {  // rows past M come back as zeros from the range check
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int idx = threadIdx.x + it * 256, r = idx / (K / 4), c4 = (idx % (K / 4)) * 4;
            const uint4v v = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(((unsigned)(t * BM + r) * (unsigned)K + (unsigned)c4) * 4u), 0, 0);
            xr[it] = f4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int idx = threadIdx.x + it * 256, r = idx / (K / 4), c4 = (idx % (K / 4)) * 4;
            st4(&Xs[buf][r * LDX + c4], xr[it]);
        }
    };
    fetch(t0);
    stage(0);
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (int t = t0; t < t1; ++t) {
        if (t + 1 < t1) fetch(t + 1);
        const unsigned yoff = ((unsigned)(t * BM + j) * 64u + (unsigned)(16 * w + 4 * kg)) * 4u;  // this lane's row j of row tile 0, its 4 columns
        floatx4 yold[2];
        if constexpr (ACC) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const uint4v v = __builtin_amdgcn_raw_buffer_load_b128(ry, (int)(yoff + (unsigned)rt * 16u * 256u), 0, 0);
                yold[rt] = floatx4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
            }
        }
        const float* xp = &Xs[cur][j * LDX + 4 * kg];
        floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float4 e0 = ld4(xp + 16 * q), e1 = ld4(xp + 16 * LDX + 16 * q);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[q].x, e0.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[q].x, e1.x, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[q].y, e0.y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[q].y, e1.y, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[q].z, e0.z, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[q].z, e1.z, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[q].w, e0.w, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[q].w, e1.w, acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            floatx4 v = acc[rt];
            if constexpr (ACC) v = v + yold[rt];
            __builtin_amdgcn_raw_buffer_store_b128(uint4v{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, ry,
                                                   (int)(yoff + (unsigned)rt * 16u * 256u), 0, 0);  // (rows past M are dropped by the range check)
        }
        if (t + 1 < t1) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
}

template <int K>
static int rows_ws64(const float* X, const float* Wt, float* Y, int M, int accumulate, hipStream_t st) {
    const int ntiles = (M + 31) / 32;
    const int per = (ntiles + 767) / 768 < 4 ? 4 : (ntiles + 767) / 768;  // ~768 workgroups (three per CU), at least 4 tiles each to pay for the weight read
    const dim3 grid((ntiles + per - 1) / per);
    if constexpr (K == 192 || K == 96) {  // (`Y +=` exists for the two residual-gradient maps only, see rows_gemm)
        if (accumulate) hipLaunchKernelGGL((rows_ws64_kernel<K, true>), grid, dim3(256), 0, st, X, Wt, Y, M, per);
    }
    if (accumulate && !(K == 192 || K == 96)) return RTFS_EINVAL;
    if (!accumulate) hipLaunchKernelGGL((rows_ws64_kernel<K, false>), grid, dim3(256), 0, st, X, Wt, Y, M, per);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

}  // namespace rtfs

using namespace rtfs;

template <int K, int N, int BM, int WM, int WN, int NT = 0>
static int rows_gemm(const float* X, const float* Wt, const float* bias, float* Y, int M, int accumulate, hipStream_t st) {
    // Forms that exist (round 6: 112 of the 160 bias x accumulate x precision instantiations of this launcher were never launched - tools/kernel_coverage.py): no row
    // GEMM of the path carries a bias, and `Y +=` is the input-gradient form of the two maps onto 64 columns that add to a residual gradient (dx += dU . W of the SRU
    // layers, K = 192; dG += dY96 . Wqkv of the attention, K = 96)
    ProPlain pro{X, K};
    if (bias) return RTFS_EINVAL;
    if constexpr (N == 64 && (K == 192 || K == 96)) {
        if (accumulate) return launch<K, N, BM, WM, WN, false, 32, NT>(pro, EpiBias<false, true>{Y, nullptr, N}, Wt, 1, M, st);
    }
    if (accumulate) return RTFS_EINVAL;
    return launch<K, N, BM, WM, WN, false, 32, NT>(pro, EpiBias<false, false>{Y, nullptr, N}, Wt, 1, M, st);
}

template <int NT>
static int bottleneck_impl(const float* a_emb, const double* stats, const float* gamma, const float* beta, const float* Wt, const float* bias, float* a0,
                           int B, int TF, hipStream_t st) {
    ProGlnRelu pro{a_emb, stats, 1.0 / ((double)TF * kC), gamma, beta};
    EpiBias<true> epi{a0, bias, kC};
    return launch<256, 256, 64, 2, 2, false, 16, NT>(pro, epi, Wt, B, TF, st);
}

template <int NT>
static int proj_impl(const float* s, const float* gw, const float* gb, float gslope, const float* Wt, const float* bias, float* y, double* stats_out,
                     int B, int TF, hipStream_t st) {
    ProGateway pro{s, gw, gb, gslope};
    EpiBiasStats epi{y, bias, kH, stats_out};
    if (B <= 0 || TF <= 0) return RTFS_EINVAL;
    // tiles per workgroup: as many as still leave ~512 workgroups (one round at 2 per CU) - the weight fragments are loaded once per
    // workgroup and the tile loop pipelines itself.  Swept at B = 32: 2 -> 418 us, 8 -> 382, 32 -> 359, 64 -> 405 (too few workgroups).
    const int tiles = (TF + 63) / 64;
    const long long want = ((long long)tiles * B + 511) / 512;
    const int per = (int)(want < 2 ? 2 : (want > 32 ? 32 : want));
    hipLaunchKernelGGL(proj_kernel<NT>, dim3((tiles + per - 1) / per, B), dim3(256), 0, st, pro, epi, Wt, TF, per);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

struct CafArgs {  // the CAF cell's audio side for the block-0 variant (rtfs_resid_caf_fwd)
    const float *ks, *kb, *vs, *vb, *att, *rsz;
    int Tv;
};

// shared by rtfs_resid_fwd / rtfs_resid_proj_fwd / rtfs_resid_caf_fwd and their bf16 siblings (Wp == nullptr: no fused projection)
template <int NT>
static int resid_impl(const float* cl, const double* cl_stats, const float* cl_g, const float* cl_b, const float* d0, const double* d0_stats,
                      const float* d0_g, const float* d0_b, const float* cg, const double* cg_stats, const float* cg_g, const float* cg_b,
                      const float* cgate, const double* cgate_stats, const float* cgate_g, const float* cgate_b, const float* Wt, const float* bias,
                      const float* s_in, const float* gw, const float* gb, float gslope, const float* a0_or_null, float* out, const float* Wp,
                      const float* pbias, float* py, double* pstats, int B, int T, int T2, hipStream_t st, const CafArgs* caf = nullptr, int variant = 0) {
    const double nf = 1.0 / ((double)T * kF * kH), nl = 1.0 / ((double)T2 * kF2 * kH);
    ProExpanded pro{{cl, cl_stats, nf, cl_g, cl_b}, {d0, d0_stats, nf, d0_g, d0_b}, {cg, cg_stats, nl, cg_g, cg_b},
                    {cgate, cgate_stats, nl, cgate_g, cgate_b}, T, T2, {0, 0, 0, 0}, {0, 0, 0, 0}};
    EpiResidual epi{out, bias, s_in, gw, gb, gslope, a0_or_null, Wp, pbias, py, pstats, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (B <= 0 || T <= 0) return RTFS_EINVAL;
    if (Wp && (!a0_or_null || !py || !pstats)) return RTFS_EINVAL;
    if (caf) {
        if (!caf->ks || !caf->kb || !caf->vs || !caf->vb || !caf->att || !caf->rsz || caf->Tv <= 0 || caf->Tv > T) return RTFS_EINVAL;
        if (a0_or_null && a0_or_null != s_in) return RTFS_EINVAL;  // block 0: the "+ a0" of the cell is "+ the block input"
        epi.caf_ks = caf->ks, epi.caf_kb = caf->kb, epi.caf_vs = caf->vs, epi.caf_vb = caf->vb, epi.att = caf->att, epi.rsz = caf->rsz, epi.Tv = caf->Tv;
    }
    // tiles per workgroup: ~1024 workgroups (two rounds at 2 per CU), capped at 16.  Swept at B = 32: 4 -> 786 us, 16 -> 744, 32 -> 743,
    // 64 -> 804; small batches get more, smaller workgroups.
    const int Mb = T * kF, tiles = (Mb + 63) / 64;
    // Projection-carrying variants at large batch: one workgroup per CU.  `variant` (include/rtfs_hip.h): 0 = this launcher's choice,
    // 1 = two 4-wave workgroups per CU (the small-batch form), 2 = one 4-wave workgroup with the whole register file (resid_kernel, DEEP),
    // 3 = one 8-wave workgroup, MFMA waves + memory / element-wise waves (resid_ws_kernel).
    if (variant < 0 || variant > 3) return RTFS_EINVAL;
    const bool large = Wp && (long long)tiles * B >= 2048;
    const int form = !large ? 1 : (variant ? variant : 3);
    if (form >= 2) {
        const long long wantd = ((long long)tiles * B + 255) / 256;
        const int perd = (int)(wantd > 128 ? 128 : wantd);
        const dim3 gridd((tiles + perd - 1) / perd, B);
        if (form == 3) {
            if (caf)
                hipLaunchKernelGGL((resid_ws_kernel<NT, true>), gridd, dim3(512), 0, st, pro, epi, Wt, Mb, perd);
            else
                hipLaunchKernelGGL((resid_ws_kernel<NT, false>), gridd, dim3(512), 0, st, pro, epi, Wt, Mb, perd);
        } else if (caf)
            hipLaunchKernelGGL((resid_kernel<true, true, NT, true, 1>), gridd, dim3(256), 0, st, pro, epi, Wt, Mb, perd);
        else
            hipLaunchKernelGGL((resid_kernel<true, true, NT, false, 2>), gridd, dim3(256), 0, st, pro, epi, Wt, Mb, perd);
        RTFS_LAUNCH_CHECK();
        return RTFS_OK;
    }
    const long long want = ((long long)tiles * B + 1023) / 1024;
    const int per = (int)(want < 2 ? 2 : (want > 16 ? 16 : want));
    const dim3 grid((tiles + per - 1) / per, B);
    if (caf) {
        if (Wp)
            hipLaunchKernelGGL((resid_kernel<true, true, NT, true>), grid, dim3(256), 0, st, pro, epi, Wt, Mb, per);
        else if (a0_or_null)
            hipLaunchKernelGGL((resid_kernel<true, false, NT, true>), grid, dim3(256), 0, st, pro, epi, Wt, Mb, per);
        else
            hipLaunchKernelGGL((resid_kernel<false, false, NT, true>), grid, dim3(256), 0, st, pro, epi, Wt, Mb, per);
    } else if (Wp)
        hipLaunchKernelGGL((resid_kernel<true, true, NT>), grid, dim3(256), 0, st, pro, epi, Wt, Mb, per);
    else if (a0_or_null)
        hipLaunchKernelGGL((resid_kernel<true, false, NT>), grid, dim3(256), 0, st, pro, epi, Wt, Mb, per);
    else
        hipLaunchKernelGGL((resid_kernel<false, false, NT>), grid, dim3(256), 0, st, pro, epi, Wt, Mb, per);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

template <int NT>
static int mask_impl(const float* x, float slope, const float* Wt, const float* bias, const float* a_emb, float* masked, float* m_or_null, int B, int TF,
                     hipStream_t st) {
    ProPrelu pro{x, slope};
    EpiMask epi{masked, bias, a_emb, m_or_null};
    return launch<256, 256, 64, 2, 2, true, 16, NT>(pro, epi, Wt, B, TF, st);
}

// d(masked) = dtaps [rows][32] . dec_wT [256][32]^T (the decoder ConvTranspose's input gradient, rtfs_gemm_rows K = 32, N = 256) with the S3 mask's
// element-wise adjoint (rtfs_mask_bwd_elem) in the epilogue: dz, da_emb written, d(masked) never stored.
template <int NT>
static int decoder_mask_bwd_impl(const float* dtaps, const float* dec_wT, const float* a_emb, const float* m, float* dz, float* da_emb, long long rows,
                                 hipStream_t st) {
    if (rows <= 0 || rows * 1024 >= (1ll << 40)) return RTFS_EINVAL;
    ProPlain pro{dtaps, 32};
    EpiMaskBwd epi{a_emb, m, dz, da_emb};
    return launch<32, 256, 64, 2, 2, true, 32, NT>(pro, epi, dec_wT, 1, (int)rows, st);
}

extern "C" {

// a0 = Wt . relu(gLN(a_emb)) + bias.  a_emb, a0: [B][TF][256]; stats: [B][2] (sum, sumsq of a_emb).
int rtfs_bottleneck_fwd(const float* a_emb, const double* stats, const float* gamma, const float* beta, const float* Wt,
                        const float* bias, float* a0, int B, int TF, void* stream) {
    return bottleneck_impl<0>(a_emb, stats, gamma, beta, Wt, bias, a0, B, TF, (hipStream_t)stream);
}
int rtfs_bottleneck_fwd_bf16(const float* a_emb, const double* stats, const float* gamma, const float* beta, const void* Wpk, const float* bias,
                             float* a0, int B, int TF, int terms, void* stream) {
    const float* W = (const float*)Wpk;
    RTFS_TERMS_DISPATCH(terms, bottleneck_impl<1>(a_emb, stats, gamma, beta, W, bias, a0, B, TF, (hipStream_t)stream),
                        bottleneck_impl<3>(a_emb, stats, gamma, beta, W, bias, a0, B, TF, (hipStream_t)stream),
                        bottleneck_impl<6>(a_emb, stats, gamma, beta, W, bias, a0, B, TF, (hipStream_t)stream));
}

// y = Wp . prelu(s*gw+gb) + bias (pre-gLN projection output, [B][TF][64]) and its gLN partial sums.
int rtfs_proj_fwd(const float* s, const float* gw, const float* gb, float gslope, const float* Wt, const float* bias, float* y,
                  double* stats_out, int B, int TF, void* stream) {
    return proj_impl<0>(s, gw, gb, gslope, Wt, bias, y, stats_out, B, TF, (hipStream_t)stream);
}
// (terms = 6 runs the fp32 kernel: the projection / residual kernels are bound by their 256-channel streams, and splitting operands in registers
// costs them more VALU than the fp32 MFMAs it replaces - measured 546 vs 372 us and 1126 vs 1056 us; both are fp32-accurate)
int rtfs_proj_fwd_bf16(const float* s, const float* gw, const float* gb, float gslope, const void* Wpk, const float* bias, float* y, double* stats_out,
                       int B, int TF, int terms, void* stream) {
    const float* W = (const float*)Wpk;
    RTFS_TERMS_DISPATCH(terms, proj_impl<1>(s, gw, gb, gslope, W, bias, y, stats_out, B, TF, (hipStream_t)stream),
                        proj_impl<3>(s, gw, gb, gslope, W, bias, y, stats_out, B, TF, (hipStream_t)stream),
                        proj_impl<0>(s, gw, gb, gslope, W, bias, y, stats_out, B, TF, (hipStream_t)stream));
}

// out = Wr . expanded + bias + prelu(s*gw+gb) [+ a0]; the four tensors of `expanded` are passed pre-gLN with their stats.
int rtfs_resid_fwd(const float* cl, const double* cl_stats, const float* cl_g, const float* cl_b,      //
                   const float* d0, const double* d0_stats, const float* d0_g, const float* d0_b,      //
                   const float* cg, const double* cg_stats, const float* cg_g, const float* cg_b,      //
                   const float* cgate, const double* cgate_stats, const float* cgate_g, const float* cgate_b,
                   const float* Wt, const float* bias, const float* s_in, const float* gw, const float* gb, float gslope,
                   const float* a0_or_null, float* out, int B, int T, int T2, void* stream) {
    return resid_impl<0>(cl, cl_stats, cl_g, cl_b, d0, d0_stats, d0_g, d0_b, cg, cg_stats, cg_g, cg_b, cgate, cgate_stats, cgate_g, cgate_b, Wt, bias, s_in,
                         gw, gb, gslope, a0_or_null, out, nullptr, nullptr, nullptr, nullptr, B, T, T2, (hipStream_t)stream);
}
int rtfs_resid_fwd_bf16(const float* cl, const double* cl_stats, const float* cl_g, const float* cl_b,      //
                        const float* d0, const double* d0_stats, const float* d0_g, const float* d0_b,      //
                        const float* cg, const double* cg_stats, const float* cg_g, const float* cg_b,      //
                        const float* cgate, const double* cgate_stats, const float* cgate_g, const float* cgate_b,
                        const void* Wpk, const float* bias, const float* s_in, const float* gw, const float* gb, float gslope,
                        const float* a0_or_null, float* out, int B, int T, int T2, int terms, void* stream) {
    const float* W = (const float*)Wpk;
#define RESID_NT(NTV)                                                                                                                                  \
    resid_impl<NTV>(cl, cl_stats, cl_g, cl_b, d0, d0_stats, d0_g, d0_b, cg, cg_stats, cg_g, cg_b, cgate, cgate_stats, cgate_g, cgate_b, W, bias, s_in, gw, \
                    gb, gslope, a0_or_null, out, nullptr, nullptr, nullptr, nullptr, B, T, T2, (hipStream_t)stream)
    RTFS_TERMS_DISPATCH(terms, RESID_NT(1), RESID_NT(3), RESID_NT(0));
#undef RESID_NT
}

// rtfs_resid_fwd (with a0) fused with the NEXT block's rtfs_proj_fwd (shared block weights): out as above, plus
// py = Wp . prelu(out*gw+gb) + pbias ([B][TF][64], pre-gLN) and its gLN partial sums in pstats - the projection reads the output tile
// from LDS instead of re-reading 1 GB of `out` from HBM.
int rtfs_resid_proj_fwd(const float* cl, const double* cl_stats, const float* cl_g, const float* cl_b,      //
                        const float* d0, const double* d0_stats, const float* d0_g, const float* d0_b,      //
                        const float* cg, const double* cg_stats, const float* cg_g, const float* cg_b,      //
                        const float* cgate, const double* cgate_stats, const float* cgate_g, const float* cgate_b,
                        const float* Wt, const float* bias, const float* s_in, const float* gw, const float* gb, float gslope,
                        const float* a0, float* out, const float* Wp, const float* pbias, float* py, double* pstats, int B, int T, int T2,
                        int variant, void* stream) {
    if (!a0 || !Wp || !py || !pstats) return RTFS_EINVAL;
    return resid_impl<0>(cl, cl_stats, cl_g, cl_b, d0, d0_stats, d0_g, d0_b, cg, cg_stats, cg_g, cg_b, cgate, cgate_stats, cgate_g, cgate_b, Wt, bias, s_in,
                         gw, gb, gslope, a0, out, Wp, pbias, py, pstats, B, T, T2, (hipStream_t)stream, nullptr, variant);
}
int rtfs_resid_proj_fwd_bf16(const float* cl, const double* cl_stats, const float* cl_g, const float* cl_b,      //
                             const float* d0, const double* d0_stats, const float* d0_g, const float* d0_b,      //
                             const float* cg, const double* cg_stats, const float* cg_g, const float* cg_b,      //
                             const float* cgate, const double* cgate_stats, const float* cgate_g, const float* cgate_b,
                             const void* Wpk, const float* bias, const float* s_in, const float* gw, const float* gb, float gslope,
                             const float* a0, float* out, const void* Wp_pk, const float* pbias, float* py, double* pstats, int B, int T, int T2,
                             int variant, int terms, void* stream) {
    if (!a0 || !Wp_pk || !py || !pstats) return RTFS_EINVAL;
    const float *W = (const float*)Wpk, *Wp = (const float*)Wp_pk;
#define RESID_NT(NTV)                                                                                                                                  \
    resid_impl<NTV>(cl, cl_stats, cl_g, cl_b, d0, d0_stats, d0_g, d0_b, cg, cg_stats, cg_g, cg_b, cgate, cgate_stats, cgate_g, cgate_b, W, bias, s_in, gw, \
                    gb, gslope, a0, out, Wp, pbias, py, pstats, B, T, T2, (hipStream_t)stream, nullptr, variant)
    RTFS_TERMS_DISPATCH(terms, RESID_NT(1), RESID_NT(3), RESID_NT(0));
#undef RESID_NT
}

// Block 0: rtfs_resid_fwd (no a0: the block runs on a0 itself) + rtfs_caf_fuse_fwd (+ a0 = + s_in when add_input) [+ block 1's rtfs_proj_fwd
// when Wp != NULL, which needs add_input]: the block output never reaches HBM, and its residual stream doubles as the cell's a0 stream.
int rtfs_resid_caf_fwd(const float* cl, const double* cl_stats, const float* cl_g, const float* cl_b,      //
                       const float* d0, const double* d0_stats, const float* d0_g, const float* d0_b,      //
                       const float* cg, const double* cg_stats, const float* cg_g, const float* cg_b,      //
                       const float* cgate, const double* cgate_stats, const float* cgate_g, const float* cgate_b,
                       const float* Wt, const float* bias, const float* s_in, const float* gw, const float* gb, float gslope,
                       const float* ks, const float* kb, const float* vs, const float* vb, const float* att, const float* rsz, int Tv, int add_input,
                       float* out, const float* Wp_or_null, const float* pbias, float* py, double* pstats, int B, int T, int T2, int variant,
                       void* stream) {
    const CafArgs caf{ks, kb, vs, vb, att, rsz, Tv};
    if (Wp_or_null && !add_input) return RTFS_EINVAL;
    return resid_impl<0>(cl, cl_stats, cl_g, cl_b, d0, d0_stats, d0_g, d0_b, cg, cg_stats, cg_g, cg_b, cgate, cgate_stats, cgate_g, cgate_b, Wt, bias, s_in,
                         gw, gb, gslope, add_input ? s_in : nullptr, out, Wp_or_null, pbias, py, pstats, B, T, T2, (hipStream_t)stream, &caf, variant);
}
int rtfs_resid_caf_fwd_bf16(const float* cl, const double* cl_stats, const float* cl_g, const float* cl_b,      //
                            const float* d0, const double* d0_stats, const float* d0_g, const float* d0_b,      //
                            const float* cg, const double* cg_stats, const float* cg_g, const float* cg_b,      //
                            const float* cgate, const double* cgate_stats, const float* cgate_g, const float* cgate_b,
                            const void* Wpk, const float* bias, const float* s_in, const float* gw, const float* gb, float gslope,
                            const float* ks, const float* kb, const float* vs, const float* vb, const float* att, const float* rsz, int Tv,
                            int add_input, float* out, const void* Wp_pk_or_null, const float* pbias, float* py, double* pstats, int B, int T, int T2,
                            int variant, int terms, void* stream) {
    const CafArgs caf{ks, kb, vs, vb, att, rsz, Tv};
    if (Wp_pk_or_null && !add_input) return RTFS_EINVAL;
    const float *W = (const float*)Wpk, *Wp = (const float*)Wp_pk_or_null;
#define RESID_NT(NTV)                                                                                                                                  \
    resid_impl<NTV>(cl, cl_stats, cl_g, cl_b, d0, d0_stats, d0_g, d0_b, cg, cg_stats, cg_g, cg_b, cgate, cgate_stats, cgate_g, cgate_b, W, bias, s_in, gw, \
                    gb, gslope, add_input ? s_in : nullptr, out, Wp, pbias, py, pstats, B, T, T2, (hipStream_t)stream, &caf, variant)
    RTFS_TERMS_DISPATCH(terms, RESID_NT(1), RESID_NT(3), RESID_NT(0));
#undef RESID_NT
}

// masked = complex_mul(relu(Wm . prelu(x) + bias), a_emb)   all [B][TF][256]
int rtfs_mask_fwd(const float* x, float slope, const float* Wt, const float* bias, const float* a_emb, float* masked, float* m_or_null, int B,
                  int TF, void* stream) {
    return mask_impl<0>(x, slope, Wt, bias, a_emb, masked, m_or_null, B, TF, (hipStream_t)stream);
}
int rtfs_mask_fwd_bf16(const float* x, float slope, const void* Wpk, const float* bias, const float* a_emb, float* masked, float* m_or_null, int B,
                       int TF, int terms, void* stream) {
    const float* W = (const float*)Wpk;
    RTFS_TERMS_DISPATCH(terms, mask_impl<1>(x, slope, W, bias, a_emb, masked, m_or_null, B, TF, (hipStream_t)stream),
                        mask_impl<3>(x, slope, W, bias, a_emb, masked, m_or_null, B, TF, (hipStream_t)stream),
                        mask_impl<6>(x, slope, W, bias, a_emb, masked, m_or_null, B, TF, (hipStream_t)stream));
}

// Y[M][N] (= or +=) X[M][K] . Wt[N][K]^T (+ bias), row-major.  Used for the SRU layer 1-3 projections, the decoder taps,
// the attention projections in training mode and every input-gradient GEMM of the backward pass (Wt = transposed weight).
int rtfs_gemm_rows(const float* X, const float* Wt, const float* bias_or_null, float* Y, int M, int K, int N, int accumulate, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    // narrow maps at large M (training step): the weight-stationary 64-column kernel (32-bit byte offsets: M K 4 < 2^31)
    if (N == 64 && bias_or_null == nullptr && M >= 65536 && (long long)M * K * 4 < (1LL << 31)) {
        if (K == 192) return rows_ws64<192>(X, Wt, Y, M, accumulate, st);
        if (K == 96) return rows_ws64<96>(X, Wt, Y, M, accumulate, st);
        if (K == 64) return rows_ws64<64>(X, Wt, Y, M, accumulate, st);
        if (K == 256 && !accumulate) return rows_ws64<256>(X, Wt, Y, M, accumulate, st);  // (accumulating, the generic kernel measured faster: 431 vs 453 us)
    }
#define RG(KK, NN, BM, WM, WN) \
    if (K == KK && N == NN) return rows_gemm<KK, NN, BM, WM, WN>(X, Wt, bias_or_null, Y, M, accumulate, st);
    RG(64, 192, 64, 1, 3)
    RG(256, 32, 128, 1, 1)
    RG(192, 64, 128, 2, 1)
    RG(256, 64, 128, 2, 1)
    RG(64, 256, 64, 2, 2)
    RG(32, 256, 64, 2, 2)
    RG(256, 256, 64, 2, 2)
    RG(64, 64, 128, 2, 1)
    RG(64, 96, 128, 1, 3)
    RG(96, 64, 128, 2, 1)
#undef RG
    return RTFS_EINVAL;
}

// ---- input-gradient GEMMs of the training step with the consumer activation's adjoint in the epilogue (round 6) ----------------------------------
// Large maps (the weight-stationary kernel applies): one launch, the GEMM's output rows meet the activation's input rows in the epilogue registers.
// Small maps: the launches they replace, in order (same results; the PReLU adjoint then runs in place on dx).
int rtfs_gemm_prelu_bwd(const float* dz, const float* Wt, const float* x, float slope, float* dx, float* dslope, int B, int rows, void* stream) {
    if (!dz || !Wt || !x || !dx || !dslope || B <= 0 || rows <= 0 || (long long)B * rows > 0x7fffffffLL) return RTFS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (!ws256_applies(B, rows)) {
        const int rc = rtfs_gemm_rows(dz, Wt, nullptr, dx, B * rows, kC, kC, 0, stream);
        if (rc != RTFS_OK) return rc;
        return rtfs_prelu_bwd(dx, x, slope, dx, 0, dslope, (long long)B * rows * kC, stream);
    }
    float* scr = spread_scratch();
    if (!scr) return RTFS_ELAUNCH;
    const int rc = launch_ws256<false, 0>(ProPlain{dz, kC}, EpiAdjoint<1>{dx, x, slope, nullptr, 0.0, nullptr, nullptr, nullptr, scr}, Wt, B, rows, st);
    if (rc != RTFS_OK) return rc;
    return spread_finish(scr, SpreadOut{{dslope}, {1}}, st);
}

int rtfs_gemm_gln_relu_bwd_reduce(const float* dy, const float* Wt, const float* x, const double* stats, const float* gamma, const float* beta, float* dR,
                                  double* red, float* dgamma, float* dbeta, int B, int rows, void* stream) {
    if (!dy || !Wt || !x || !stats || !gamma || !beta || !dR || !red || !dgamma || !dbeta || B <= 0 || rows <= 0 || (long long)B * rows > 0x7fffffffLL)
        return RTFS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (!ws256_applies(B, rows)) {
        const int rc = rtfs_gemm_rows(dy, Wt, nullptr, dR, B * rows, kC, kC, 0, stream);
        if (rc != RTFS_OK) return rc;
        return rtfs_gln_bwd_reduce(dR, x, stats, gamma, beta, 2, 0.f, red, dgamma, dbeta, nullptr, B, rows, kC, stream);
    }
    float* scr = spread_scratch();
    if (!scr) return RTFS_ELAUNCH;
    const int rc = launch_ws256<false, 0>(ProPlain{dy, kC}, EpiAdjoint<2>{dR, x, 0.f, stats, 1.0 / ((double)rows * kC), gamma, beta, red, scr}, Wt, B, rows, st);
    if (rc != RTFS_OK) return rc;
    return spread_finish(scr, SpreadOut{{dgamma, dbeta}, {kC, kC}}, st);
}

int rtfs_decoder_mask_bwd(const float* dtaps, const float* dec_wT, const float* a_emb, const float* m, float* dz, float* da_emb, long long rows,
                          void* stream) {
    return decoder_mask_bwd_impl<0>(dtaps, dec_wT, a_emb, m, dz, da_emb, rows, (hipStream_t)stream);
}
int rtfs_decoder_mask_bwd_bf16(const float* dtaps, const void* dec_wT_pk, const float* a_emb, const float* m, float* dz, float* da_emb, long long rows,
                               int terms, void* stream) {
    const float* W = (const float*)dec_wT_pk;
    RTFS_TERMS_DISPATCH(terms, decoder_mask_bwd_impl<1>(dtaps, W, a_emb, m, dz, da_emb, rows, (hipStream_t)stream),
                        decoder_mask_bwd_impl<3>(dtaps, W, a_emb, m, dz, da_emb, rows, (hipStream_t)stream),
                        decoder_mask_bwd_impl<6>(dtaps, W, a_emb, m, dz, da_emb, rows, (hipStream_t)stream));
}

int rtfs_gemm_rows_fwd(const float* X, const float* Wt, const float* bias_or_null, float* Y, int M, int K, int N, void* stream) {
    return rtfs_gemm_rows(X, Wt, bias_or_null, Y, M, K, N, 0, stream);
}

// bf16 / split-bf16 row GEMMs (Wpk = host-packed weight [N][K]): every shape of rtfs_gemm_rows
int rtfs_gemm_rows_bf16(const float* X, const void* Wpk, const float* bias_or_null, float* Y, int M, int K, int N, int accumulate, int terms, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const float* W = (const float*)Wpk;
    // terms 6 (plain fp32 operands): the narrow maps at large M are stream-bound and have a weight-stationary fp32 kernel (rows_ws64_kernel: 60 us against 115
    // for the six-term LDS-staged form at K = 192) - fp32-equivalent either way
    if (terms == 6 && N == 64 && bias_or_null == nullptr && M >= 65536 && (long long)M * K * 4 < (1LL << 31) && (K == 192 || K == 96 || K == 64 || (K == 256 && !accumulate)))
        return rtfs_gemm_rows(X, W, bias_or_null, Y, M, K, N, accumulate, stream);
    // ... and the 256 x 256 input-gradient GEMMs of the training step: fp32 has the weight-stationary ws256_kernel (1.03 ms), the six-term form here is the LDS-staged one (2.2 ms)
    if (terms == 6 && K == 256 && N == 256 && !accumulate && M >= 65536) return rtfs_gemm_rows(X, W, bias_or_null, Y, M, K, N, accumulate, stream);
#define RG(KK, NN, BM, WM, WN)  \
    if (K == KK && N == NN)     \
        RTFS_TERMS_DISPATCH(terms, (rows_gemm<KK, NN, BM, WM, WN, 1>(X, W, bias_or_null, Y, M, accumulate, st)), (rows_gemm<KK, NN, BM, WM, WN, 3>(X, W, bias_or_null, Y, M, accumulate, st)), (rows_gemm<KK, NN, BM, WM, WN, 6>(X, W, bias_or_null, Y, M, accumulate, st)));
    RG(64, 192, 64, 1, 3)
    RG(256, 32, 128, 1, 1)
    RG(192, 64, 128, 2, 1)
    RG(256, 64, 128, 2, 1)
    RG(64, 256, 64, 2, 2)
    RG(32, 256, 64, 2, 2)
    RG(256, 256, 64, 2, 2)
    RG(64, 64, 128, 2, 1)
    RG(64, 96, 128, 1, 3)
    RG(96, 64, 128, 2, 1)
#undef RG
    return RTFS_EINVAL;
}
int rtfs_gemm_rows_fwd_bf16(const float* X, const void* Wpk, const float* bias_or_null, float* Y, int M, int K, int N, int terms, void* stream) {
    return rtfs_gemm_rows_bf16(X, Wpk, bias_or_null, Y, M, K, N, 0, terms, stream);
}

}  // extern "C"
