// TF-domain multi-head self-attention of the RTFS block: MultiHeadSelfAttention2D.forward
// (/root/reference/src/models/layers/attention.py:149-189), 4 heads, tokens = compressed time frames.
//
//   rtfs_attn_qkv_fwd    12 x ConvActNorm(1x1 conv -> PReLU -> LN4D over (c,F))  -> Q,K [B][4][T2][256], V [B][4][T2][1024]
//   rtfs_attn_core_fwd   softmax(Q K^T / 16) V per (b, head, 32-query tile)       -> O [B][T2][64 ch][64 f]
//   rtfs_attn_out_fwd    attn_concat_proj (1x1 64->64 -> PReLU -> LN4D(64,F)) + residual, in place on G
//
// Feature order inside a head is e = c*64 + f as in the reference (attention.py:164-168); heads are kept as
// an explicit axis instead of being concatenated on the batch axis (attention.py:160-162).
// QK^T and PV run on v_mfma_f32_32x32x2_f32; softmax is a wave-level reduction over the key axis.
#include "common.h"

namespace rtfs {

constexpr int kHeads = 4;
constexpr int kQkvN = 96;  // 4 heads x (4 q + 4 k + 16 v) output channels

// ------------------------------------------------------------------------------------------------
// QKV: two tokens (b,t) per workgroup: rows = 2 x 64 frequency bins, K = 64 channels, N = 96.
// Column order n: [0,16) Q (h*4+e), [16,32) K (h*4+e), [32,96) V (h*16+c).
// gamma/beta are host-permuted to the output order: gq,gk [4][256], gv [4][1024] (index e*64+f / c*64+f).
// ------------------------------------------------------------------------------------------------
// LN4D apply for one element as three scalar VALU instructions that the compiler cannot pair into v_pk_*_f32.  The packed forms it chose
// for the scatter below (v_pk_mul_f32 with op_sel:[0,1] on the (mean, rstd) register pair) sporadically produced rstd = 0 for the low
// element of lanes 48..63 - the output was beta for 16 consecutive float4 stores, ~100 such rows per launch in the bf16 mode, a few
// per forward in the split-bf16 mode - whenever two workgroups shared a CU; one workgroup per CU or scalar arithmetic never showed it
// (tools/qkv_det.py reproduces it in seconds).  Same rounding as fmaf((y - mean) * rstd, g, b).
__device__ __forceinline__ float ln_apply(float y, float mean, float rstd, float g, float b) {
    float o;
    asm("v_sub_f32 %0, %1, %2\n\tv_mul_f32 %0, %0, %3\n\tv_fma_f32 %0, %0, %4, %5" : "=&v"(o) : "v"(y), "v"(mean), "v"(rstd), "v"(g), "v"(b));
    return o;
}
__device__ __forceinline__ float4 ln_apply4(float4 y, const float* st2, float4 g, float4 b) {
    const float mean = st2[0], rstd = st2[1];
    return f4(ln_apply(y.x, mean, rstd, g.x, b.x), ln_apply(y.y, mean, rstd, g.y, b.y), ln_apply(y.z, mean, rstd, g.z, b.z), ln_apply(y.w, mean, rstd, g.w, b.w));
}

template <int NT = 0>  // precision (common.h); NT != 0: Wt host-PACKED
__global__ __launch_bounds__(256, 2) void attn_qkv_kernel(const float* __restrict__ G, const float* __restrict__ Wt, const float* __restrict__ bias,
                                                       const float* __restrict__ slope, const float* __restrict__ gq, const float* __restrict__ bq,
                                                       const float* __restrict__ gk, const float* __restrict__ bk, const float* __restrict__ gv,
                                                       const float* __restrict__ bv, float* __restrict__ Q, float* __restrict__ Kx,
                                                       float* __restrict__ V, float* __restrict__ Ypre, int BT, int T2) {
    constexpr int LDA = 68, LDY = 97;
    // one LDS arena: [X tile | W] during the GEMM, then re-used as the post-PReLU tile Ys (61 KB -> two workgroups per CU)
    __shared__ __attribute__((aligned(16))) float arena[(128 + kQkvN) * LDA];
    static_assert(128 * LDY <= (128 + kQkvN) * LDA, "Ys must fit in the arena");
    float* As = arena;
    float* Bs = arena + 128 * LDA;
    float* Ys = arena;
    __shared__ float st[24][2];

    const int tok0 = blockIdx.x * 2;
    const int ntok = min(2, BT - tok0);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;

    // stage X (2 tokens x 64 f x 64 c = 2048 float4) and W (96 x 64 = 1536 float4)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int row = idx >> 4, c4 = idx & 15;
        float4 v = ld4(G + ((size_t)tok0 * 64 + min(row, ntok * 64 - 1)) * 64 + c4 * 4);  // clamped: loads stay unconditional and batched
        if (row >= ntok * 64) v = f4(0, 0, 0, 0);
        st4(As + row * LDA + c4 * 4, pack4<NT>(v));
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int row = idx >> 4, c4 = idx & 15;
        st4(Bs + row * LDA + c4 * 4, ld4(Wt + row * 64 + c4 * 4));
    }
    __syncthreads();

    floatx16 acc[1][3];
    acc_zero(acc);
    mma_block_nt<NT, 1, 3>(acc, As + w * 32 * LDA, LDA, Bs, LDA, 64);
    __syncthreads();  // every wave is done reading As / Bs before Ys overwrites them

    // post-PReLU tile into LDS, and the LN4D statistics over (c, F) per token and module straight from the accumulators: a wave owns 32
    // frequency bins of ONE token, a lane 16 of them for one output column per column tile; the 12 groups of a token are runs of 4 (Q, K)
    // or 16 (V) consecutive columns = lanes.  Sum and sum of squares per lane, xor-shuffles inside the run and across the two lane halves,
    // one partial per (wave, group) through LDS.  (The first version walked the LDS tile group by group, two passes of dependent 4-byte
    // reads, 6 groups per wave: 6.6 us of latency chain per workgroup.)
    __shared__ float part[4][12][2];
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        const int col = n * 32 + (lane & 31);
        const float cb = bias[col], cs = slope[col];
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = w * 32 + acc_row(r);
            const float pre = acc[0][n][r] + cb;
            if (Ypre && row < ntok * 64) Ypre[((size_t)tok0 * 64 + row) * kQkvN + col] = pre;  // training: pre-activation for the adjoint
            const float y = prelu(pre, cs);
            Ys[row * LDY + col] = y;
            s += y;
            q = fmaf(y, y, q);
        }
        const int run = n == 0 ? 4 : 16;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
            if (o < run) s += __shfl_xor(s, o, 64), q += __shfl_xor(q, o, 64);
        s += __shfl_xor(s, 32, 64), q += __shfl_xor(q, 32, 64);
        if (lane < 32 && (lane & (run - 1)) == 0) {
            const int g = n == 0 ? (lane >> 2) : 8 + 2 * (n - 1) + (lane >> 4);
            part[w][g][0] = s, part[w][g][1] = q;
        }
    }
    __syncthreads();
    if (threadIdx.x < 24) {
        const int tok = threadIdx.x / 12, g = threadIdx.x % 12;
        const float cnt = g < 8 ? 256.f : 1024.f;
        const float mean = (part[2 * tok][g][0] + part[2 * tok + 1][g][0]) / cnt;
        const float var = fmaxf((part[2 * tok][g][1] + part[2 * tok + 1][g][1]) / cnt - mean * mean, 0.f);
        st[threadIdx.x][0] = mean;
        st[threadIdx.x][1] = 1.0f / sqrtf(var + kEps);
    }
    __syncthreads();

    // normalise and scatter into the per-head layouts: a thread owns 4 consecutive frequency bins of one (head, channel) - 16-byte
    // parameter loads and stores (the scalar version issued 4x the instructions for the same 24 KB per token)
    for (int tok = 0; tok < ntok; ++tok) {
        const int bt = tok0 + tok, b = bt / T2, t = bt % T2;
        {  // Q and K: [h][e*64+f], 1024 elements per token = one float4 per thread
            const int i = threadIdx.x * 4;
            const int h = i >> 8, ef = i & 255, e = ef >> 6, f = ef & 63;
            const size_t o = (((size_t)b * kHeads + h) * T2 + t) * 256 + ef;
            const float* yr = Ys + (tok * 64 + f) * LDY + h * 4 + e;
            const float* sq = st[tok * 12 + h];
            const float* sk = st[tok * 12 + 4 + h];
            const float4 yq = f4(yr[0], yr[LDY], yr[2 * LDY], yr[3 * LDY]), yk = f4(yr[16], yr[LDY + 16], yr[2 * LDY + 16], yr[3 * LDY + 16]);
            const float4 gq4 = ld4(gq + i), bq4 = ld4(bq + i), gk4 = ld4(gk + i), bk4 = ld4(bk + i);
            st4(Q + o, ln_apply4(yq, sq, gq4, bq4));
            st4(Kx + o, ln_apply4(yk, sk, gk4, bk4));
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {  // V: [h][c*64+f], 4096 elements per token
            const int i = (threadIdx.x + it * 256) * 4;
            const int h = i >> 10, cf = i & 1023, c = cf >> 6, f = cf & 63;
            const size_t o = (((size_t)b * kHeads + h) * T2 + t) * 1024 + cf;
            const float* yr = Ys + (tok * 64 + f) * LDY + 32 + h * 16 + c;
            const float* sv = st[tok * 12 + 8 + h];
            const float4 y = f4(yr[0], yr[LDY], yr[2 * LDY], yr[3 * LDY]);
            const float4 g4 = ld4(gv + i), b4 = ld4(bv + i);
            st4(V + o, ln_apply4(y, sv, g4, b4));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// core: grid (ceil(T2/32), 4, B): one 32-query tile of one head.  MAXKT = compile-time bound on key tiles of 32.
//  * S = Q K^T / 16: wave w owns key tiles w, w+4, ...; BOTH operands are read straight from global memory in
//    MFMA fragment shape (each lane streams 16-byte pieces of its own Q / K row; rows are 1 KB and fully consumed
//    by the wave, so L1 absorbs the partial-line accesses).  No LDS, no barrier.
//  * softmax over the key axis through a [32][keys] LDS tile (the only LDS use; two barriers).
//  * O = P V: wave w owns output features [256w, 256w+256); P fragments come from the LDS tile, V fragments are
//    coalesced 128-byte row segments read directly from global memory -- every V element is fetched once per
//    workgroup and there is no barrier in this phase either.
// ------------------------------------------------------------------------------------------------
template <int MAXKT, int PREC = 0>  // PREC = the NT of common.h (0 fp32, 1 bf16, 3 split-bf16); Q, K, V, P are packed in registers
__global__ __launch_bounds__(256, 2) void attn_core_kernel(const float* __restrict__ Q, const float* __restrict__ Kx, const float* __restrict__ V,
                                                           float* __restrict__ O, float* __restrict__ LSE, int T2, int fsplit) {
    constexpr int LDS_S = MAXKT * 32 + 4;
    __shared__ __attribute__((aligned(16))) float Ss[32 * LDS_S];
    // fsplit = 2 (small batches, round 5: 16 workgroups per utterance leave a batch-1 launch on 16 of 256 CUs, 41 us): two workgroups per query tile, each
    // recomputes the tile's scores and softmax (a fifth of the products) and takes one half of the wave's 256 output features - the same arithmetic per
    // output, so the same bits; gridDim.x = 2 x query tiles, the two halves next to each other on one XCD.
    // Workgroup -> (query tile, head, utterance).  The query tiles of one (head, utterance) pair all read that pair's K and V (640 KB at 2 s): in
    // launch order they would sit on different XCDs (workgroup l runs on XCD l % 8 - observed placement, a speed heuristic only) and every XCD would
    // fetch K / V for itself (round 3: 344 MB fetched per launch against 96 MB of Q / K / V).  Consecutive workgroups OF ONE XCD take the query tiles
    // of one pair instead: pair = (slot / nq) * 8 + xcd with slot = l / 8, so K / V come from HBM once per pair and from that XCD's L2 afterwards.
    int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    {
        const int nq = gridDim.x, npair = gridDim.y * gridDim.z;  // (nq counts the halves of fsplit = 2 as tiles of their own)
        if ((npair & 7) == 0) {
            const int l = blockIdx.x + nq * (blockIdx.y + gridDim.y * blockIdx.z), slot = l >> 3;
            const int pair = (slot / nq) * 8 + (l & 7);
            qt = slot % nq, h = pair % gridDim.y, b = pair / gridDim.y;
        }
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const int pass0 = fsplit == 2 ? (qt & 1) : 0, pass1 = fsplit == 2 ? pass0 + 1 : 2;
    if (fsplit == 2) qt >>= 1;
    const int q0 = qt * 32;
    const int NT = (T2 + 31) / 32;
    const size_t headoff = ((size_t)b * kHeads + h) * T2;
    const float* Qg = Q + headoff * 256;
    const float* Kg = Kx + headoff * 256;
    const float* Vg = V + headoff * 1024;

    // ---- S = Q K^T / 16 ----
    const float* qrow = Qg + (size_t)min(q0 + i, T2 - 1) * 256 + 4 * kh;  // clamped rows are never stored
    for (int kt = w; kt < NT; kt += 4) {
        const float* krow = Kg + (size_t)min(kt * 32 + i, T2 - 1) * 256 + 4 * kh;
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if constexpr (PREC == 0) {
#pragma unroll 8
            for (int q = 0; q < 32; ++q) {
                const float4 a = ld4(qrow + 8 * q), kb = ld4(krow + 8 * q);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, kb.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, kb.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, kb.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, kb.w, acc, 0, 0, 0);
            }
        } else {
#pragma unroll 4
            for (int q2 = 0; q2 < 16; ++q2)
                mma32<PREC>(acc, frag_f32<PREC>(ld4(qrow + 16 * q2), ld4(qrow + 16 * q2 + 8)), frag_f32<PREC>(ld4(krow + 16 * q2), ld4(krow + 16 * q2 + 8)));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Ss[acc_row(r) * LDS_S + kt * 32 + i] = acc[r] * 0.0625f;
    }
    __syncthreads();

    // ---- softmax over keys, 8 rows per wave ----
    for (int rr = 0; rr < 8; ++rr) {
        float* row = Ss + (w * 8 + rr) * LDS_S;
        float v[MAXKT / 2];
        float mx = -3.0e38f;
#pragma unroll
        for (int j = 0; j < MAXKT / 2; ++j) {
            const int col = lane + j * 64;
            v[j] = (col < T2) ? row[col] : -3.0e38f;
            mx = fmaxf(mx, v[j]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < MAXKT / 2; ++j) {
            const int col = lane + j * 64;
            v[j] = (col < T2) ? __expf(v[j] - mx) : 0.f;
            sum += v[j];
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        if (LSE && lane == 0 && pass0 == 0 && q0 + w * 8 + rr < T2) LSE[headoff + q0 + w * 8 + rr] = mx + __logf(sum);  // training: log-sum-exp of the scaled scores
#pragma unroll
        for (int j = 0; j < MAXKT / 2; ++j) {
            const int col = lane + j * 64;
            if (col < NT * 32) row[col] = v[j] * inv;  // zero weight for the padded keys
        }
    }
    __syncthreads();

    // ---- O = P V: two passes of 4 feature tiles (128 features) per wave ----
#pragma unroll 1
    for (int pass = pass0; pass < pass1; ++pass) {
        const int n0 = w * 256 + pass * 128;
        floatx16 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        const float* pa = Ss + i * LDS_S + 4 * kh;
        if constexpr (PREC == 0) {
            // 8 keys per step: this lane's keys are 8kq + 4kh .. +3.  The V fragments of step kq + 1 are requested before the 16 MFMAs of step kq
            // (round 4: with the loads issued right in front of their MFMAs every step paid an L2 round trip with the matrix pipe idle)
            float vb[2][4][4];
            auto load_v = [&](float(&dst)[4][4], int kq) {
                const int key = 8 * kq + 4 * kh;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* vr = Vg + (size_t)min(key + r, T2 - 1) * 1024 + n0 + i;  // padded keys carry zero weight
#pragma unroll
                    for (int n = 0; n < 4; ++n) dst[n][r] = vr[n * 32];
                }
            };
            auto step = [&](const float(&v)[4][4], int kq) {
                const float4 p = ld4(pa + 8 * kq);
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.x, v[n][0], acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.y, v[n][1], acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.z, v[n][2], acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.w, v[n][3], acc[n], 0, 0, 0);
                }
            };
            load_v(vb[0], 0);
            for (int kq = 0; kq < NT * 4; kq += 2) {  // (NT * 4 is even)
                load_v(vb[1], kq + 1);
                step(vb[0], kq);
                load_v(vb[0], min(kq + 2, NT * 4 - 1));
                step(vb[1], kq + 1);
            }
        } else {
            for (int kq2 = 0; kq2 < NT * 2; ++kq2) {  // 16 keys per step: this lane's keys are 16kq2 + 4kh .. +3 and 16kq2 + 8 + 4kh .. +3
                const Frag fp = frag_f32<PREC>(ld4(pa + 16 * kq2), ld4(pa + 16 * kq2 + 8));
                const int key = 16 * kq2 + 4 * kh;
                float vb[4][8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float* vr = Vg + (size_t)min(key + (r & 3) + 2 * (r & 4), T2 - 1) * 1024 + n0 + i;  // padded keys carry zero weight
#pragma unroll
                    for (int n = 0; n < 4; ++n) vb[n][r] = vr[n * 32];
                }
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    mma32<PREC>(acc[n], fp, frag_f32<PREC>(f4(vb[n][0], vb[n][1], vb[n][2], vb[n][3]), f4(vb[n][4], vb[n][5], vb[n][6], vb[n][7])));
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = q0 + acc_row(r);
                const int e = n0 + n * 32 + i;
                const int c = h * 16 + (e >> 6), f = e & 63;
                if (t < T2) O[(((size_t)b * T2 + t) * 64 + c) * 64 + f] = acc[n][r];
            }
    }
}

// ------------------------------------------------------------------------------------------------
// core for MORE than 1024 compressed frames (utterances longer than 16.4 s; the reference has no length limit): the keys are walked
// in blocks of 1024 through the same [32][1024] LDS score tile, in two sweeps - (1) running row maximum / sum of exponentials over
// all blocks (the online-softmax recurrence), (2) scores recomputed block by block, normalised with the FINAL statistics and
// accumulated into O = P V (all 8 feature tiles of the wave stay in registers across the blocks; no rescaling of accumulators).
// QK^T is computed twice (16 of the core's 80 MMAC per key-query pair).  Training: the final statistics give the log-sum-exp the adjoint needs.
// ------------------------------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(256) void attn_core_long_kernel(const float* __restrict__ Q, const float* __restrict__ Kx, const float* __restrict__ V,
                                                             float* __restrict__ O, float* __restrict__ LSE, int T2) {
    constexpr int MAXKT = 32, LDS_S = MAXKT * 32 + 4;
    __shared__ __attribute__((aligned(16))) float Ss[32 * LDS_S];
    __shared__ float rmax[32], rsum[32];
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const int q0 = qt * 32;
    const int NTall = (T2 + 31) / 32, nblk = (NTall + MAXKT - 1) / MAXKT;
    const size_t headoff = ((size_t)b * kHeads + h) * T2;
    const float* Qg = Q + headoff * 256;
    const float* Kg = Kx + headoff * 256;
    const float* Vg = V + headoff * 1024;
    const float* qrow = Qg + (size_t)min(q0 + i, T2 - 1) * 256 + 4 * kh;

    auto scores = [&](int kb) {  // scaled scores of key block kb -> Ss[query][key - 1024 kb]
        const int kt0 = kb * MAXKT, ktn = min(MAXKT, NTall - kt0);
        for (int kt = w; kt < ktn; kt += 4) {
            const float* krow = Kg + (size_t)min((kt0 + kt) * 32 + i, T2 - 1) * 256 + 4 * kh;
            floatx16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if constexpr (PREC == 0) {
#pragma unroll 8
                for (int q = 0; q < 32; ++q) {
                    const float4 a = ld4(qrow + 8 * q), kb4 = ld4(krow + 8 * q);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, kb4.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, kb4.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, kb4.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, kb4.w, acc, 0, 0, 0);
                }
            } else {
#pragma unroll 4
                for (int q2 = 0; q2 < 16; ++q2)
                    mma32<PREC>(acc, frag_f32<PREC>(ld4(qrow + 16 * q2), ld4(qrow + 16 * q2 + 8)), frag_f32<PREC>(ld4(krow + 16 * q2), ld4(krow + 16 * q2 + 8)));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) Ss[acc_row(r) * LDS_S + kt * 32 + i] = acc[r] * 0.0625f;
        }
    };

    if (threadIdx.x < 32) rmax[threadIdx.x] = -3.0e38f, rsum[threadIdx.x] = 0.f;
    // ---- sweep 1: row statistics over all key blocks ----
    for (int kb = 0; kb < nblk; ++kb) {
        __syncthreads();  // the statistics / previous block's scores are settled
        scores(kb);
        __syncthreads();
        const int nkey = min(MAXKT * 32, T2 - kb * MAXKT * 32);
        for (int rr = 0; rr < 8; ++rr) {
            const int rowi = w * 8 + rr;
            const float* row = Ss + rowi * LDS_S;
            float mx = -3.0e38f;
            for (int col = lane; col < nkey; col += 64) mx = fmaxf(mx, row[col]);
            mx = wave_max(mx);
            const float mold = rmax[rowi], mnew = fmaxf(mold, mx);
            float sum = 0.f;
            for (int col = lane; col < nkey; col += 64) sum += __expf(row[col] - mnew);
            sum = wave_sum(sum);
            if (lane == 0) {
                rsum[rowi] = rsum[rowi] * __expf(mold - mnew) + sum;
                rmax[rowi] = mnew;
            }
        }
    }
    __syncthreads();
    if (LSE && threadIdx.x < 32 && q0 + threadIdx.x < T2) LSE[headoff + q0 + threadIdx.x] = rmax[threadIdx.x] + __logf(rsum[threadIdx.x]);
    // ---- sweep 2: P = exp(S - max) / sum per block, O += P V ----
    floatx16 acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    for (int kb = 0; kb < nblk; ++kb) {
        __syncthreads();
        scores(kb);
        __syncthreads();
        const int key0 = kb * MAXKT * 32;
        const int nkey = min(MAXKT * 32, T2 - key0), ktn = min(MAXKT, NTall - kb * MAXKT);
        for (int rr = 0; rr < 8; ++rr) {
            const int rowi = w * 8 + rr;
            float* row = Ss + rowi * LDS_S;
            const float mx = rmax[rowi], inv = 1.0f / rsum[rowi];
            for (int col = lane; col < ktn * 32; col += 64) row[col] = col < nkey ? __expf(row[col] - mx) * inv : 0.f;  // zero weight for the padded keys
        }
        __syncthreads();
        const float* pa = Ss + i * LDS_S + 4 * kh;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int n0 = w * 256 + pass * 128;
            if constexpr (PREC == 0) {
                for (int kq = 0; kq < ktn * 4; ++kq) {
                    const float4 p = ld4(pa + 8 * kq);
                    const int key = key0 + 8 * kq + 4 * kh;
                    float vb[4][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float* vr = Vg + (size_t)min(key + r, T2 - 1) * 1024 + n0 + i;
#pragma unroll
                        for (int n = 0; n < 4; ++n) vb[n][r] = vr[n * 32];
                    }
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        floatx16& a = acc[pass * 4 + n];
                        a = __builtin_amdgcn_mfma_f32_32x32x2f32(p.x, vb[n][0], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x2f32(p.y, vb[n][1], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x2f32(p.z, vb[n][2], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x2f32(p.w, vb[n][3], a, 0, 0, 0);
                    }
                }
            } else {
                for (int kq2 = 0; kq2 < ktn * 2; ++kq2) {
                    const Frag fp = frag_f32<PREC>(ld4(pa + 16 * kq2), ld4(pa + 16 * kq2 + 8));
                    const int key = key0 + 16 * kq2 + 4 * kh;
                    float vb[4][8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const float* vr = Vg + (size_t)min(key + (r & 3) + 2 * (r & 4), T2 - 1) * 1024 + n0 + i;
#pragma unroll
                        for (int n = 0; n < 4; ++n) vb[n][r] = vr[n * 32];
                    }
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        mma32<PREC>(acc[pass * 4 + n], fp, frag_f32<PREC>(f4(vb[n][0], vb[n][1], vb[n][2], vb[n][3]), f4(vb[n][4], vb[n][5], vb[n][6], vb[n][7])));
                }
            }
        }
    }
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = q0 + acc_row(r);
            const int e = w * 256 + n * 32 + i;  // pass * 128 + (n & 3) * 32 = n * 32
            const int c = h * 16 + (e >> 6), f = e & 63;
            if (t < T2) O[(((size_t)b * T2 + t) * 64 + c) * 64 + f] = acc[n][r];
        }
}

// ------------------------------------------------------------------------------------------------
// out-projection + PReLU + LN4D over (64, F) + residual, one token per workgroup: G = Gres + LN(...) (Gres == G: in place; the two may alias exactly, hence
// no __restrict__ on them).
// Computes Y^T[co][f] = W[co][c] . X[c][f] so the O layout [c][f] is consumed without a transpose.
// gamma/beta are host-permuted to [f][c].
// ------------------------------------------------------------------------------------------------
template <int NT = 0>  // NT != 0: W host-PACKED
__global__ __launch_bounds__(256) void attn_out_kernel(const float* __restrict__ O, const float* __restrict__ W, const float* __restrict__ bias,
                                                       float slope, const float* __restrict__ gamma_fc, const float* __restrict__ beta_fc,
                                                       const float* Gres, float* G, float* __restrict__ Ypre) {
    constexpr int LD = 68, LDY = 65;
    __shared__ __attribute__((aligned(16))) float Ws[64 * LD];
    __shared__ __attribute__((aligned(16))) float Xs[64 * LD];
    __shared__ float Ys[64 * LDY];
    __shared__ float red[8];
    __shared__ float bc[2];

    const size_t tok = blockIdx.x;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = w >> 1, wn = w & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int row = idx >> 4, c4 = idx & 15;
        st4(Ws + row * LD + c4 * 4, ld4(W + row * 64 + c4 * 4));
        st4(Xs + row * LD + c4 * 4, ld4(O + tok * 4096 + row * 64 + c4 * 4));
    }
    __syncthreads();
    floatx16 acc[1][1];
    acc_zero(acc);
    if constexpr (NT == 0)
        mma_block_bn<1, 1>(acc, Ws + wm * 32 * LD, LD, Xs + wn * 32, LD, 64);
    else
        mma_block_bn_p<NT, 1, 1>(acc, Ws + wm * 32 * LD, LD, Xs + wn * 32, LD, 64);
    float s = 0.f;
    float yv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = wm * 32 + acc_row(r), f = wn * 32 + (lane & 31);
        const float pre = acc[0][0][r] + bias[co];
        if (Ypre) Ypre[tok * 4096 + f * 64 + co] = pre;  // training: pre-activation, channels-last [f][co]
        const float y = prelu(pre, slope);
        Ys[co * LDY + f] = y;
        yv[r] = y;
        s += y;
    }
    // mean, then the centred sum of squares from the values still in registers (two-pass, no LDS walk)
    s = wave_sum(s);
    if (lane == 0) red[w] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) * (1.f / 4096.f);
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float d = yv[r] - mean;
        q = fmaf(d, d, q);
    }
    q = wave_sum(q);
    if (lane == 0) red[4 + w] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((red[4] + red[5] + red[6] + red[7]) * (1.f / 4096.f) + kEps);
    float* g = G + tok * 4096;
    const float* gr = Gres + tok * 4096;
    // residual update in 16-byte pieces: a thread owns 4 consecutive channels of one frequency bin (gamma / beta / G are [f][c])
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = (threadIdx.x + it * 256) * 4;
        const int f = i >> 6, c = i & 63;
        const float4 y = f4(Ys[c * LDY + f], Ys[(c + 1) * LDY + f], Ys[(c + 2) * LDY + f], Ys[(c + 3) * LDY + f]);
        const float4 ga = ld4(gamma_fc + i), be = ld4(beta_fc + i), g0 = ld4(gr + i);
        st4(g + i, f4(fmaf((y.x - mean) * rstd, ga.x, be.x) + g0.x, fmaf((y.y - mean) * rstd, ga.y, be.y) + g0.y, fmaf((y.z - mean) * rstd, ga.z, be.z) + g0.z,
                      fmaf((y.w - mean) * rstd, ga.w, be.w) + g0.w));
    }
}

}  // namespace rtfs

using namespace rtfs;

template <int NT>
static int attn_qkv_impl(const float* G, const float* Wt, const float* bias, const float* slope, const float* gq, const float* bq, const float* gk,
                         const float* bk, const float* gv, const float* bv, float* Q, float* K, float* V, float* Ypre_or_null, int B, int T2, void* stream) {
    if (B <= 0 || T2 <= 0) return RTFS_EINVAL;
    const int BT = B * T2;
    hipLaunchKernelGGL(attn_qkv_kernel<NT>, dim3((BT + 1) / 2), dim3(256), 0, (hipStream_t)stream, G, Wt, bias, slope, gq, bq, gk, bk, gv, bv, Q, K, V,
                       Ypre_or_null, BT, T2);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

template <int NT>
static int attn_core_impl(const float* Q, const float* K, const float* V, float* O, float* LSE_or_null, int B, int T2, void* stream) {
    if (B <= 0 || T2 <= 0) return RTFS_EINVAL;
    dim3 grid((T2 + 31) / 32, kHeads, B);
    const int fsplit = (long long)grid.x * kHeads * B <= 128 ? 2 : 1;  // below half a workgroup per CU: two workgroups per query tile (see attn_core_kernel)
    if (T2 > 1024) {  // past 16.4 s of audio the [32][T2] score tile no longer fits the LDS: key-blocked two-sweep kernel
        hipLaunchKernelGGL((attn_core_long_kernel<NT>), grid, dim3(256), 0, (hipStream_t)stream, Q, K, V, O, LSE_or_null, T2);
        RTFS_LAUNCH_CHECK();
        return RTFS_OK;
    }
    if (T2 <= 128)
        hipLaunchKernelGGL((attn_core_kernel<4, NT>), dim3(grid.x * fsplit, grid.y, grid.z), dim3(256), 0, (hipStream_t)stream, Q, K, V, O, LSE_or_null, T2, fsplit);
    else if (T2 <= 256)
        hipLaunchKernelGGL((attn_core_kernel<8, NT>), dim3(grid.x * fsplit, grid.y, grid.z), dim3(256), 0, (hipStream_t)stream, Q, K, V, O, LSE_or_null, T2, fsplit);
    else if (T2 <= 512)
        hipLaunchKernelGGL((attn_core_kernel<16, NT>), dim3(grid.x * fsplit, grid.y, grid.z), dim3(256), 0, (hipStream_t)stream, Q, K, V, O, LSE_or_null, T2, fsplit);
    else
        hipLaunchKernelGGL((attn_core_kernel<32, NT>), dim3(grid.x * fsplit, grid.y, grid.z), dim3(256), 0, (hipStream_t)stream, Q, K, V, O, LSE_or_null, T2, fsplit);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

template <int NT>
static int attn_out_impl(const float* O, const float* W, const float* bias, float slope, const float* gamma_fc, const float* beta_fc, float* G,
                         float* Ypre_or_null, int B, int T2, void* stream, const float* Gres = nullptr) {
    if (B <= 0 || T2 <= 0) return RTFS_EINVAL;
    hipLaunchKernelGGL(attn_out_kernel<NT>, dim3(B * T2), dim3(256), 0, (hipStream_t)stream, O, W, bias, slope, gamma_fc, beta_fc, Gres ? Gres : G, G, Ypre_or_null);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

extern "C" {

int rtfs_attn_qkv_fwd(const float* G, const float* Wt, const float* bias, const float* slope, const float* gq, const float* bq, const float* gk,
                      const float* bk, const float* gv, const float* bv, float* Q, float* K, float* V, float* Ypre_or_null, int B, int T2,
                      void* stream) {
    return attn_qkv_impl<0>(G, Wt, bias, slope, gq, bq, gk, bk, gv, bv, Q, K, V, Ypre_or_null, B, T2, stream);
}
int rtfs_attn_qkv_fwd_bf16(const float* G, const void* Wpk, const float* bias, const float* slope, const float* gq, const float* bq, const float* gk,
                           const float* bk, const float* gv, const float* bv, float* Q, float* K, float* V, float* Ypre_or_null, int B, int T2, int terms,
                           void* stream) {
    const float* W = (const float*)Wpk;
    RTFS_TERMS_DISPATCH(terms, attn_qkv_impl<1>(G, W, bias, slope, gq, bq, gk, bk, gv, bv, Q, K, V, Ypre_or_null, B, T2, stream),
                        attn_qkv_impl<3>(G, W, bias, slope, gq, bq, gk, bk, gv, bv, Q, K, V, Ypre_or_null, B, T2, stream),
                        attn_qkv_impl<6>(G, W, bias, slope, gq, bq, gk, bk, gv, bv, Q, K, V, Ypre_or_null, B, T2, stream));
}

int rtfs_attn_core_fwd(const float* Q, const float* K, const float* V, float* O, float* LSE_or_null, int B, int T2, void* stream) {
    return attn_core_impl<0>(Q, K, V, O, LSE_or_null, B, T2, stream);
}
// QK^T and PV on v_mfma_f32_32x32x16_bf16 (terms 1) or as three-term split-bf16 products (terms 3); softmax in fp32
int rtfs_attn_core_fwd_bf16(const float* Q, const float* K, const float* V, float* O, float* LSE_or_null, int B, int T2, int terms, void* stream) {
    RTFS_TERMS_DISPATCH(terms, attn_core_impl<1>(Q, K, V, O, LSE_or_null, B, T2, stream), attn_core_impl<3>(Q, K, V, O, LSE_or_null, B, T2, stream), attn_core_impl<6>(Q, K, V, O, LSE_or_null, B, T2, stream));
}

int rtfs_attn_out_fwd(const float* O, const float* W, const float* bias, float slope, const float* gamma_fc, const float* beta_fc, float* G,
                      float* Ypre_or_null, int B, int T2, void* stream) {
    return attn_out_impl<0>(O, W, bias, slope, gamma_fc, beta_fc, G, Ypre_or_null, B, T2, stream);
}
int rtfs_attn_out_fwd_bf16(const float* O, const void* Wpk, const float* bias, float slope, const float* gamma_fc, const float* beta_fc, float* G,
                           float* Ypre_or_null, int B, int T2, int terms, void* stream) {
    const float* W = (const float*)Wpk;
    RTFS_TERMS_DISPATCH(terms, attn_out_impl<1>(O, W, bias, slope, gamma_fc, beta_fc, G, Ypre_or_null, B, T2, stream),
                        attn_out_impl<3>(O, W, bias, slope, gamma_fc, beta_fc, G, Ypre_or_null, B, T2, stream),
                        attn_out_impl<6>(O, W, bias, slope, gamma_fc, beta_fc, G, Ypre_or_null, B, T2, stream));
}
// Gout = Gin + LN(PReLU(out-projection)): the out-of-place form (training step: Gin is the attention's input, kept for the adjoint).  Gin == Gout is the in-place call.
int rtfs_attn_out_fwd_to(const float* O, const float* W, const float* bias, float slope, const float* gamma_fc, const float* beta_fc, const float* Gin, float* Gout,
                         float* Ypre_or_null, int B, int T2, void* stream) {
    return attn_out_impl<0>(O, W, bias, slope, gamma_fc, beta_fc, Gout, Ypre_or_null, B, T2, stream, Gin);
}
int rtfs_attn_out_fwd_to_bf16(const float* O, const void* Wpk, const float* bias, float slope, const float* gamma_fc, const float* beta_fc, const float* Gin, float* Gout,
                              float* Ypre_or_null, int B, int T2, int terms, void* stream) {
    const float* W = (const float*)Wpk;
    RTFS_TERMS_DISPATCH(terms, attn_out_impl<1>(O, W, bias, slope, gamma_fc, beta_fc, Gout, Ypre_or_null, B, T2, stream, Gin),
                        attn_out_impl<3>(O, W, bias, slope, gamma_fc, beta_fc, Gout, Ypre_or_null, B, T2, stream, Gin),
                        attn_out_impl<6>(O, W, bias, slope, gamma_fc, beta_fc, Gout, Ypre_or_null, B, T2, stream, Gin));
}

}  // extern "C"
