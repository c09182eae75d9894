/* rtfs_hip.h -- C-ABI of librtfs_hip.so: the MI355X (gfx950) kernels behind RTFS-Net's separation path
 * (forward, and the adjoint chain of the training step).
 *
 * The reference (spkgyk/RTFS-Net) is pure Python: its "plugin interface" for this path is the nn.Module API
 * `src.models.AVNet` (src/models/tdavnet.py:14-97); there is no FFI in it.  This header is the boundary BELOW
 * that API: what a maintainer binds (ctypes / pybind / cffi) to replace the ATen/cuDNN/
 * `sru` calls each reference function makes.  Every entry point
 *   - takes raw DEVICE pointers and sizes only (no torch types), caller-allocated outputs and workspaces,
 *   - enqueues on the given `hipStream_t` (passed as void*; NULL = default stream) and never synchronises,
 *   - keeps no global state and is re-entrant per stream (exception: the parameter-gradient reducers of the training step share
 *     one zeroed 8 MB scratch per device and the deferred-finish state of rtfs_spread_defer, csrc/spread.hip -- order their launches on
 *     one stream per device),
 *   - returns 0 on success, RTFS_EINVAL (-1) for unsupported shapes, RTFS_ELAUNCH (-2) if the launch failed.
 *
 * Tensor layout below the boundary is CHANNELS-LAST fp32:
 *     full resolution  [B][T][F=129][C]     C = 256 (encoder width) or 64 (RTFS hidden width)
 *     compressed       [B][T2][F2=64][64]   T2 = (T-2)/2 + 1
 * with T = 1 + L/128 STFT frames.  The kernels are specialised to the RTFS-Net family
 * (config/{lrs2,lrs3,voxceleb2}_RTFSNet_{4,6,12}_layer.yaml): win 256, hop 128, C 256, hid 64, dw kernel 4,
 * SRU hidden 32 x 2 directions, window 8, 4 attention heads, CAF "kernel_size" 4, lip embedding 512;
 * B, L (hence T, T2), Tv and the number of blocks are free.
 *
 * gLN statistics (`stats`): double[B][16], entries 0 and 1 = (sum, sum of squares) over one utterance (one 128-byte line per
 * utterance so that the accumulating atomics of different utterances never queue on the same line), ACCUMULATED by producers
 * with fp64 atomics -- the caller zeroes the slot before the producing launch.
 */
#ifndef RTFS_HIP_H
#define RTFS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define RTFS_OK 0
#define RTFS_EINVAL (-1)
#define RTFS_ELAUNCH (-2)

/* ---- a1: STFTEncoder.forward, src/models/TDAVNet/encoder.py:161-175 ------------------------------------- */
/* torch.stft(n_fft 256, hop 128, hann, center, reflect, onesided) + stack(re,im): wav [B][L] -> spec [B][T][129][2] */
int rtfs_stft_fwd(const float* wav, float* spec, int B, int L, void* stream);
/* Conv2d(2->256,3x3,'same',no bias): Wp [18][256] (tap = ci*9+dt*3+df); also accumulates the bottleneck gLN stats of a_emb */
int rtfs_enc_conv_fwd(const float* spec, const float* Wp, float* a_emb, double* stats, int B, int T, void* stream);

/* ---- a2: audio_bottleneck ConvNormAct(pre gLN, pre ReLU, 1x1), tdavnet.py:59,89; conv_layers.py:65-129 ---- */
int rtfs_bottleneck_fwd(const float* a_emb, const double* stats, const float* gamma, const float* beta, const float* Wt /*[256][256] out,in*/,
                        const float* bias, float* a0, int B, int TF, void* stream);

/* ---- a5.1-a5.2: gateway + projection, separators/tdanet.py:34-49,108-109 ------------------------------------ */
/* y (pre-gLN) = Wp . prelu(s*gw+gb) + bias, [B][TF][64]; stats_out accumulates (sum, sumsq) of y */
int rtfs_proj_fwd(const float* s, const float* gw, const float* gb, float gslope, const float* Wt /*[64][256]*/, const float* bias, float* y,
                  double* stats_out, int B, int TF, void* stream);

/* ---- a5.3, a5.6: depth-wise 4x4 convolutions, tdanet.py:61-76,112-114; layers/fusion.py:25-52 ---------------- */
/* nconv in {1,2,4} convolutions of one input (stride 2: nconv = 1); mode 0 raw, 1 gLN(in), 2 PReLU(gLN(in)); stride 1 ('same') or 2 (pad 1) */
int rtfs_dwconv_fwd(const float* in, const double* stats_in, const float* gamma, const float* beta, float slope, int mode, int stride, int nconv,
                    const float* const* w /*[16][64]*/, const float* const* bias, float* const* out, double* const* stats_out, int B, int Tin,
                    int Fin, void* stream);

/* the same stride-1 convolutions applied to InjectionMultiSum's mix gLN(loc) * sigmoid(gLN(gate)^) + gLN(glob)^ (layers/fusion.py:59-67; what
 * rtfs_tfar_mix_fwd writes) without materialising it: loc [B][T][F][64]; gate, glob [B][Tg][Fg][64] (^ = nearest up-sampling); nconv in {1,2} */
int rtfs_dwconv_mix_fwd(const float* loc, const double* loc_stats, const float* loc_g, const float* loc_b, const float* gate,
                        const double* gate_stats, const float* gate_g, const float* gate_b, const float* glob, const double* glob_stats,
                        const float* glob_g, const float* glob_b, int nconv, const float* const* w /*[16][64]*/, const float* const* bias,
                        float* const* out, double* const* stats_out, int B, int T, int F, int Tg, int Fg, void* stream);

/* ---- a5.4: global pooling, tdanet.py:117-118 ---------------------------------------------------------------- */
int rtfs_pool_fwd(const float* d0, const double* d0_stats, const float* d0_g, const float* d0_b, const float* d1, const double* d1_stats,
                  const float* d1_g, const float* d1_b, float* G, int B, int T, int T2, void* stream);

/* ---- a6-a7: DualPathRNN.forward, layers/rnn_layers.py:136-162; sru.SRU (external, oracle/sru_ref.py) --------- */
/* dim 4: sequences along F (one per (b,t2)); dim 3: along T (one per (b,f2)).  S sequences of L = npos-7 windows. */
/* variant: 0 = the library's choice (fp32 at >= 512 tiles of 63 pair rows: weight-stationary 2-parallel fast-FIR kernel - three half-rate 4-tap correlations, 0.775x the MFMAs, sums re-ordered; else LDS-staged
 * kernels on tiles cut from the flattened (sequence, window) row index), 1 = tiles padded per sequence, 2 = LDS-staged flattened tiles (A/B: 1 and 2
 * give the same bits), 3 = the direct weight-stationary kernel (W0 resident in registers, >= 1024 flattened 64-row tiles; its LayerNorm uses v_rsq_f32 and
 * agrees with 1 / 2 to 1 ulp of rstd).  The bf16 entry takes the same range (its 0 and 3 are the direct weight-stationary kernel). */
int rtfs_dp_unfold_gemm_fwd(const float* G, const float* gamma, const float* beta, const float* Wt /*[256][512]*/, float* U0, int B, int T2, int dim,
                            int variant, void* stream);
int rtfs_sru_scan_fwd(const float* U, const float* X, const float* wc, const float* bias, float scale_x, float* H, int S, int L, int km,
                      void* stream);
/* SRU layers 1-3 with the input projection U = Hprev . W fused into the recurrence (U never reaches HBM): Wt [192][64], row = m*64 + dir*32 + j;
 * training (both or neither): Cout cell states [S][L][64] and Uout pre-activations [S][L][3][64] for rtfs_sru_scan_bwd.
 * Same result as rtfs_gemm_rows_fwd(64 -> 192) followed by rtfs_sru_scan_fwd(km = 3). */
int rtfs_sru_layer_fwd(const float* Hprev, const float* Wt, const float* weight_c, const float* bias, float scale_x, float* Hout, float* Cout_or_null,
                       float* Uout_or_null, int S, int L, void* stream);
/* the same with the kernel form named (inference only; all forms give the same bits): 0 = the library's choice by S, 1 = one wave per sequence,
 * 2 = one wave per (sequence, direction), 3 = one workgroup per (sequence, direction), its waves splitting the gates (the smallest batches) */
int rtfs_sru_layer_fwd_form(const float* Hprev, const float* Wt, const float* weight_c, const float* bias, float scale_x, float* Hout, float* Cout_or_null,
                            float* Uout_or_null, int S, int L, int form, void* stream);
int rtfs_gemm_rows_fwd(const float* X, const float* Wt, const float* bias_or_null, float* Y, int M, int K, int N, void* stream);
/* row GEMM Y (= or +=) X . Wt^T for the (K, N) pairs of the path: SRU input projections, decoder taps, every input-gradient GEMM of the backward pass (Wt = transposed
 * weight).  Round 6: bias_or_null must be NULL (no row GEMM of the path carries a bias) and `accumulate` exists for (K, N) = (192, 64) and (96, 64) only - the two input
 * gradients that add to a residual gradient; other requests return RTFS_EINVAL (their kernels were instantiated and never launched). */
int rtfs_gemm_rows(const float* X, const float* Wt, const float* bias_or_null, float* Y, int M, int K, int N, int accumulate, void* stream);
int rtfs_dp_convt_fwd(const float* H3, const float* Wt /*[64][512]*/, const float* bias, float* G /*in place*/, int B, int T2, int dim, void* stream);
/* the same with the kernel form named: 0 = the library's choice (fp32 at >= 1024 tiles of 63 pair rows: the fast-FIR kernel of rtfs_dp_unfold_gemm_fwd in its
 * ConvTranspose mode - three half-rate 4-tap correlations over the zero-padded rows, sums re-ordered), 1 = the direct 8-tap kernels (A/B and tests) */
int rtfs_dp_convt_fwd_form(const float* H3, const float* Wt /*[64][512]*/, const float* bias, float* G /*in place*/, int B, int T2, int dim, int variant,
                           void* stream);
/* out of place: Gout = Gin + ConvTranspose1d(H3) + bias (the training step keeps Gin - the stage's input - for the adjoint: rnn_layers.py:146-156 under autograd);
 * Gin == Gout is rtfs_dp_convt_fwd.  The fast-FIR kernel reads Gin directly, the other forms copy it to Gout on the stream first. */
int rtfs_dp_convt_fwd_to(const float* H3, const float* Wt /*[64][512]*/, const float* bias, const float* Gin, float* Gout, int B, int T2, int dim, void* stream);

/* ---- a8: MultiHeadSelfAttention2D.forward, layers/attention.py:149-189 ------------------------------------- */
int rtfs_attn_qkv_fwd(const float* G, const float* Wt /*[96][64]*/, const float* bias, const float* slope, const float* gq, const float* bq,
                      const float* gk, const float* bk, const float* gv, const float* bv, float* Q, float* K, float* V, float* Ypre_or_null /*[B*T2*64][96], training*/,
                      int B, int T2, void* stream);
int rtfs_attn_core_fwd(const float* Q, const float* K, const float* V, float* O, float* LSE_or_null /*[B][4][T2], training*/, int B, int T2, void* stream);
int rtfs_attn_out_fwd(const float* O, const float* W /*[64][64] out,in*/, const float* bias, float slope, const float* gamma_fc, const float* beta_fc,
                      float* G /*in place*/, float* Ypre_or_null /*[B*T2][64 f][64 co], training*/, int B, int T2, void* stream);
/* out of place: Gout = Gin + LN4D(PReLU(out-projection of O)) (attention.py:183-189; the training step keeps Gin, the attention's input); Gin == Gout is the call above */
int rtfs_attn_out_fwd_to(const float* O, const float* W /*[64][64] out,in*/, const float* bias, float slope, const float* gamma_fc, const float* beta_fc,
                         const float* Gin, float* Gout, float* Ypre_or_null, int B, int T2, void* stream);

/* The three readers of gLN(D0) in one pass: rtfs_dwconv_fwd(mode 1, stride 1, w1 -> out1: fusion_layers[0].local_embedding, fusion.py:25-52),
 * rtfs_dwconv_fwd(mode 1, stride 2, w2 + bias2 -> out2: downsample_layers[1], tdanet.py:112-114) and the adaptive_avg_pool2d(gLN(D0)) term of
 * rtfs_pool_fwd (tdanet.py:117-118) -> pooled [B][T2][64][64]; rtfs_pool_add_fwd finishes G = pooled + gLN(D1).  T2 = (T - 2) / 2 + 1. */
int rtfs_dwconv_trio_fwd(const float* d0, const double* d0_stats, const float* d0_g, const float* d0_b, const float* w1, float* out1, double* stats1,
                         const float* w2, const float* bias2, float* out2, double* stats2, float* pooled, int B, int T, int T2, void* stream);
int rtfs_pool_add_fwd(const float* pooled, const float* d1, const double* d1_stats, const float* d1_g, const float* d1_b, float* G, int B, int T2,
                      void* stream);
/* rtfs_pool_add_fwd + rtfs_dwconv_fwd(mode 1, stride 1, one convolution, no bias) in ONE pass over `in` (D1): out = conv(gLN(in)) with its gLN partial
 * sums (fusion_layers[1].local_embedding, layers/fusion.py:25-52) and G = pooled + gLN(in) (tdanet.py:117-118). */
int rtfs_dwconv_gadd_fwd(const float* in, const double* stats_in, const float* gamma, const float* beta, const float* w /*[16][64]*/, float* out,
                         double* stats_out, const float* pooled, float* G, int B, int T, int F, void* stream);

/* ---- a5.6: TFAR, InjectionMultiSum.forward, layers/fusion.py:54-69 ------------------------------------------ */
int rtfs_tfar_mix_fwd(const float* loc, const double* loc_stats, const float* loc_g, const float* loc_b, const float* gate,
                      const double* gate_stats, const float* gate_g, const float* gate_b, const float* glob, const double* glob_stats,
                      const float* glob_g, const float* glob_b, float* out, int B, int T, int F, int Tg, int Fg, void* stream);

/* ---- a5.6-a5.7: concat_layers[0] tail + residual_conv + gateway residual, tdanet.py:127-131 ------------------ */
int rtfs_resid_fwd(const float* cl, const double* cl_stats, const float* cl_g, const float* cl_b, const float* d0, const double* d0_stats,
                   const float* d0_g, const float* d0_b, const float* cg, const double* cg_stats, const float* cg_g, const float* cg_b,
                   const float* cgate, const double* cgate_stats, const float* cgate_g, const float* cgate_b, const float* Wt /*[256][64]*/,
                   const float* bias, const float* s_in, const float* gw, const float* gb, float gslope, const float* a0_or_null, float* out, int B,
                   int T, int T2, void* stream);
/* rtfs_resid_fwd of block i (with a0) fused with rtfs_proj_fwd of block i+1 (the blocks share their weights, tdanet.py:9-59 / `shared`):
 * out as above, plus py = Wp . prelu(out*gw+gb) + pbias ([B][T*129][64], pre-gLN) and its gLN partial sums in pstats - the next
 * block's projection (tdanet.py:108-109) is computed from the output tile while it is still in LDS.
 * variant (also rtfs_resid_caf_fwd): kernel form at large batch (>= 2048 64-pixel tiles), same result up to the grouping of the fp32 partial sums
 * behind pstats.  0 = the library's choice; 1 = two 4-wave workgroups per CU; 2 = one 4-wave workgroup per CU with the whole register file;
 * 3 = one 8-wave workgroup per CU, MFMA waves + memory / element-wise waves (csrc/gemm.hip resid_ws_kernel).  Anything else: RTFS_EINVAL. */
int rtfs_resid_proj_fwd(const float* cl, const double* cl_stats, const float* cl_g, const float* cl_b, const float* d0, const double* d0_stats,
                        const float* d0_g, const float* d0_b, const float* cg, const double* cg_stats, const float* cg_g, const float* cg_b,
                        const float* cgate, const double* cgate_stats, const float* cgate_g, const float* cgate_b, const float* Wt,
                        const float* bias, const float* s_in, const float* gw, const float* gb, float gslope, const float* a0, float* out,
                        const float* Wp, const float* pbias, float* py, double* pstats, int B, int T, int T2, int variant, void* stream);

/* ---- a10: CAF, ATTNFusionCell.forward, layers/fusion.py:252-274 --------------------------------------------- */
int rtfs_caf_video_fwd(const float* v /*[B][512][Tv]*/, const float* att_w, const float* att_b, const float* att_g, const float* att_be,
                       const float* rs_w, const float* rs_b, const float* rs_g, const float* rs_be, float* att_out, float* rsz_out, int B, int Tv,
                       void* stream);
int rtfs_caf_fuse_fwd(const float* x, const float* ks, const float* kb, const float* vs, const float* vb, const float* att, const float* rsz,
                      const float* a0_or_null, float* out, int B, int T, int Tv, void* stream);
/* adjoint of rtfs_caf_video_fwd (training step; the forward is the same kernel in both modes - the video side has no BatchNorm): datt, drsz
 * [B][Tv][256] -> dv [B][512][Tv] (written) and the eight parameter gradients, ADDED into the caller's zeroed buffers (shapes of the parameters) */
int rtfs_caf_video_bwd(const float* v, const float* att_w, const float* att_b, const float* att_g, const float* att_be, const float* rs_w,
                       const float* rs_b, const float* rs_g, const float* rs_be, const float* datt, const float* drsz, float* dv, float* d_att_w,
                       float* d_att_b, float* d_att_g, float* d_att_be, float* d_rs_w, float* d_rs_b, float* d_rs_g, float* d_rs_be, int B, int Tv,
                       void* stream);

/* Block 0 of the refinement loop: rtfs_resid_fwd (a0_or_null = NULL: the block runs on a0 itself, refinement_module.py:55) fused with
 * rtfs_caf_fuse_fwd (ATTNFusionCell's audio side, fusion.py:259-272, applied to the block output in the epilogue registers; add_input != 0
 * adds s_in, which IS a0 for block 0, refinement_module.py:60) and - when Wp_or_null != NULL (needs add_input) - with rtfs_proj_fwd of
 * block 1: the block output never reaches HBM.  att / rsz: [B][Tv][256] from rtfs_caf_video_fwd; ks/kb/vs/vb: [256] as rtfs_caf_fuse_fwd. */
int rtfs_resid_caf_fwd(const float* cl, const double* cl_stats, const float* cl_g, const float* cl_b, const float* d0, const double* d0_stats,
                       const float* d0_g, const float* d0_b, const float* cg, const double* cg_stats, const float* cg_g, const float* cg_b,
                       const float* cgate, const double* cgate_stats, const float* cgate_g, const float* cgate_b, const float* Wt,
                       const float* bias, const float* s_in, const float* gw, const float* gb, float gslope, const float* ks, const float* kb,
                       const float* vs, const float* vb, const float* att, const float* rsz, int Tv, int add_input, float* out,
                       const float* Wp_or_null, const float* pbias, float* py, double* pstats, int B, int T, int T2, int variant, void* stream);

/* ---- a11: MaskGenerator.forward + __apply_masks (RI_split), TDAVNet/mask_generator.py:67-99 ------------------ */
int rtfs_mask_fwd(const float* x, float slope, const float* Wt /*[256][256]*/, const float* bias, const float* a_emb, float* masked,
                  float* m_or_null /*post-ReLU mask, training*/, int B, int TF, void* stream);

/* ---- a12: STFTDecoder.forward, TDAVNet/decoder.py:110-132 ---------------------------------------------------- */
/* taps [B][T][129][32] = rtfs_gemm_rows_fwd(masked, Wdec' [32][256]) ; frames: workspace [B][T][256]; out [B][L] */
int rtfs_istft_fwd(const float* taps, float* frames, float* out, int B, int L, void* stream);


/* ======================================================================================================================
 * Backward (training step, BASELINE configs 3-5).  The reference has no backward code -- it is torch autograd over the
 * forward modules (train.py:148 via Lightning) plus sru's CUDA backward kernel; every entry point below is the adjoint of
 * the forward entry point it names, checked against autograd of the oracle (tests/test_hip_backward.py).
 * Parameter gradients are ACCUMULATED (+=) into caller-zeroed buffers (the RTFS block's weights are shared by all blocks).
 * ====================================================================================================================== */
int rtfs_colsum_add(const float* X, float* out, long long M, int N, void* stream);
int rtfs_axpy(const float* x, float a, float* y, long long n, void* stream);
/* out = xs[0] + ... + xs[n-1] (n <= 8, count % 4 == 0; xs: host array of device pointers): the running d(a0) sum of the step, formed once */
int rtfs_sum_n(const float* const* xs, int n, float* out, long long count, void* stream);
/* Deferred mode of the parameter-gradient reducers (every *_bwd / rtfs_wgrad-style entry point that adds per-workgroup partial sums into dgamma / dbeta / dW /
 * dslope ... through the spread scratch): between rtfs_spread_defer(1) and rtfs_spread_defer(0) their small finish launches (219 per training step) are
 * recorded and applied by one launch per ~20 producers, with fp32 atomic adds (two producers may name the same destination: the RTFS blocks share their
 * weights).  Destinations are complete after rtfs_spread_flush / rtfs_spread_defer(0), stream-ordered.  Outside a deferred section every entry point finishes
 * its own sums before it returns to the stream, as in rounds 1-3.  Reference: parameter-gradient accumulation of autograd over separators/tdanet.py:106-133. */
int rtfs_spread_defer(int on, void* stream);
int rtfs_spread_flush(void* stream);
/* scratch lane (0 / 1) used by the following reducer launches and rtfs_spread_defer / _flush: lane 1 for launches on a second stream that may run
 * concurrently with lane 0's (the training step's weight-gradient side stream); deferred sections are per lane */
int rtfs_spread_lane(int lane);
/* GroupNorm(1,C) adjoint; act: 0 none, 1 PReLU after the norm (C=64), 2 ReLU after the norm (C=256); red: double[B][16] (entries 0, 1 used) zeroed by caller */
int rtfs_gln_bwd_reduce(const float* dY, const float* X, const double* stats, const float* gamma, const float* beta, int act, float slope, double* red,
                        float* dgamma, float* dbeta, float* dslope, int B, int rows, int C, void* stream);
int rtfs_gln_bwd_apply(const float* dY, const float* X, const double* stats, const float* gamma, const float* beta, int act, float slope,
                       const double* red, float* dX, int accumulate, int B, int rows, int C, void* stream);
/* adjoints of rtfs_dwconv_fwd */
int rtfs_dwconv_bwd_input(const float* dOut, const float* w, float* dIn, int accumulate, int stride, int B, int Tin, int Fin, void* stream);
int rtfs_dwconv_bwd_weight(const float* dOut, const float* in, const double* stats_in, const float* gamma, const float* beta, float slope, int mode,
                           int stride, float* dW, float* dbias_or_null, int B, int Tin, int Fin, void* stream);
/* adjoints of rtfs_pool_fwd / rtfs_tfar_mix_fwd; rtfs_expand_fwd materialises the TFAR tail for d(residual_conv.weight) */
int rtfs_pool_bwd(const float* dG, float* dN0, int B, int T, int T2, void* stream);
/* the tail of an RTFS block's backward in one pass over d(gLN(D0)): dN0 += rtfs_dwconv_bwd_input(dD1, w, stride 2) + rtfs_pool_bwd(dG), then the
 * reduce pass of rtfs_gln_bwd_reduce(dN0, D0, act 0) on the finished values (red: double[B][16] zeroed by the caller; dgamma / dbeta [64]
 * accumulate).  Adjoint of /root/reference/src/models/separators/tdanet.py:112-118 (downsample_layers[1] and adaptive_avg_pool2d both read
 * gLN(D0)).  F = 129 -> 64; T2 = (T - 2) / 2 + 1. */
int rtfs_d0_tail_bwd(const float* dD1, const float* w, const float* dG, float* dN0, const float* D0, const double* d0_stats, const float* gamma,
                     const float* beta, double* red, float* dgamma, float* dbeta, int B, int T, int T2, void* stream);
int rtfs_mix_bwd(const float* dOut, const float* loc, const double* loc_stats, const float* loc_g, const float* loc_b, const float* gate,
                 const double* gate_stats, const float* gate_g, const float* gate_b, float* dNloc, float* dNgate, float* dNglob, int B, int T, int F, int Tg,
                 int Fg, void* stream);
/* rtfs_mix_bwd + the gLN adjoint of the local branch (rtfs_gln_bwd_reduce / _apply on dNloc) in two passes over dOut and loc, dNloc never stored:
 * dLoc = gradient w.r.t. the local conv's output (pre-norm).  The reduce passes of the gate / global branches' gLN adjoints ride in the first
 * pass as well: red = double[3][B][16] (loc, gate, glob) zeroed by the caller - the caller finishes those two branches with
 * rtfs_gln_bwd_apply(dNgate, ..., red + B*16, ...) / (dNglob, ..., red + 2*B*16, ...); dgb = host array of six device pointers to [64]
 * accumulators (dgamma, dbeta of loc, gate, glob).
 * Replaces autograd over InjectionMultiSum.forward's `local_feat * sigmoid(gate) + global` (/root/reference/src/models/layers/fusion.py:59-67)
 * together with the GroupNorms of its three embeddings. */
int rtfs_mix_gln_bwd(const float* dOut, const float* loc, const double* loc_stats, const float* loc_g, const float* loc_b, const float* gate,
                     const double* gate_stats, const float* gate_g, const float* gate_b, const float* glob, const double* glob_stats, const float* glob_g,
                     const float* glob_b, float* dLoc, float* dNgate, float* dNglob, double* red, float* const* dgb, int B, int T, int F, int Tg, int Fg,
                     void* stream);
int rtfs_expand_fwd(const float* cl, const double* cl_stats, const float* cl_g, const float* cl_b, const float* d0, const double* d0_stats,
                    const float* d0_g, const float* d0_b, const float* cg, const double* cg_stats, const float* cg_g, const float* cg_b, const float* cgate,
                    const double* cgate_stats, const float* cgate_g, const float* cgate_b, float* E, int B, int T, int T2, void* stream);
/* gateway adjoint (dw1x1 + PReLU, tdanet.py:34-49) with its parameter reductions, applied to dG = dx + dy0 . Wp formed on the fly (WpT [256][64]); ds (= or +=),
 * acc_mode 0 none, 1 acc = ds, 2 acc += ds (a running d(a0)).  (Until round 6 there was also the un-fused rtfs_gateway_bwd on a materialised dG: no caller, deleted.) */
int rtfs_proj_gateway_bwd(const float* dy0, const float* WpT, const float* dx, const float* s, const float* gw, const float* gb, float slope, float* ds,
                          int accumulate, float* acc, int acc_mode, float* dgw, float* dgb, float* dslope, long long rows, void* stream);
/* round 6: the same (plain form) + the residual conv's input gradient of the block whose output gradient ds is - the next block of the backward pass, tdanet.py:127-131 -
 * from the finished ds rows while they are in LDS: next_dE [rows][64] = ds . next_WrT^T (next_WrT [64][256], the weight rtfs_gemm_rows(.., 256, 64) takes; same bits) */
int rtfs_proj_gateway_bwd_next(const float* dy0, const float* WpT, const float* dx, const float* s, const float* gw, const float* gb, float slope, float* ds,
                               float* dgw, float* dgb, float* dslope, const float* next_WrT, float* next_dE, long long rows, void* stream);
/* weight gradient of any 1x1 conv / linear map; rows may be segmented and nshift > 1 computes the taps of a Toeplitz map
 * (unfold / ConvTranspose1d) in one launch: dW[n][z*KIN+k] += sum dY[seq,l][n] * X[seq, l+x_off+z][k].  pro: X as stored (0), or re-derived on load - the gateway
 * (1: p0 / p1 = gw / gb, slope; NOUT = 64, KIN = 256 - the projection's map), PReLU (2) or ReLU(gLN) (3: p0 / p1 = gamma / beta, stats, rows_per_b) with
 * NOUT a multiple of 128; other prologue / shape pairs return RTFS_EINVAL */
int rtfs_wgrad(const float* dY, int ldy, const float* X, int ldx, float* dW, int ldw, float* dbias_or_null, long long M, int seg_len, int x_seg, int x_off, int nshift,
               int NOUT, int KIN, int pro, const float* p0, const float* p1, float slope, const double* stats, int rows_per_b, void* stream);
/* adjoints of rtfs_dp_unfold_gemm_fwd (input side) and rtfs_dp_convt_fwd */
int rtfs_fold_gemm_bwd(const float* dU0, const float* Wt /*[64][2048]*/, float* dxn, int B, int T2, int dim, void* stream);
int rtfs_convt_bwd_input(const float* dG, const float* Wt /*[64][512]*/, float* dH3, int B, int T2, int dim, void* stream);
/* the same with the kernel form named: 0 = the library's choice (fp32 at >= 1024 tiles of 63 pair rows: the fast-FIR kernel), 1 = the direct 8-tap kernel */
int rtfs_convt_bwd_input_form(const float* dG, const float* Wt /*[64][512]*/, float* dH3, int B, int T2, int dim, int variant, void* stream);
/* SRU recurrence: training forward (stores the cell state) and adjoint */
int rtfs_sru_scan_train_fwd(const float* U, const float* X, const float* wc, const float* bias, float scale_x, float* H, float* C, int S, int L,
                            int km, void* stream);
int rtfs_sru_scan_bwd(const float* U, const float* X, const float* C, const float* wc, const float* bias, float scale_x, const float* dH, float* dU,
                      float* dX, float* dwc, float* dbias, int S, int L, int km, void* stream);
/* rtfs_sru_scan_bwd with the incoming gradient in two parts: dH + dH2 (dH2 may be null) */
int rtfs_sru_scan_bwd2(const float* U, const float* X, const float* C, const float* wc, const float* bias, float scale_x, const float* dH,
                       const float* dH2_or_null, float* dU, float* dX, float* dwc, float* dbias, int S, int L, int km, void* stream);
/* the adjoint of rtfs_sru_layer_fwd (layers 1-3, training mode) in one launch - rtfs_sru_scan_bwd(km = 3) + rtfs_wgrad + rtfs_gemm_rows with dU held in
 * LDS (reference: autograd over sru.SRU's layer, rnn_layers.py:100-105): W [192][64] as the forward takes it; incoming gradient dH (+ dH2 when not
 * null); the input gradient leaves as two fully written parts dX0 + dX1 [S][L][64] (one per scan direction; the next layer down takes them as dH,
 * dH2); dW [192][64], dwc, dbias [2][64] accumulated into; work: rtfs_sru_layer_bwd_work_floats(S) floats of device scratch.  S * L < 2.79 M positions. */
int rtfs_sru_layer_bwd_work_floats(int S);
int rtfs_sru_layer_bwd(const float* U, const float* X, const float* C, const float* W, const float* wc, const float* bias, float scale_x, const float* dH,
                       const float* dH2_or_null, float* dX0, float* dX1, float* work, float* dW, float* dwc, float* dbias, int S, int L, void* stream);
int rtfs_ln4d_c_bwd(const float* dxn, const float* G, const float* gamma, float* dG, float* dgamma, float* dbeta, long long rows, void* stream);
int rtfs_seq_gather(const float* G, const float* gamma, const float* beta, int ln, float* out, int B, int T2, int dim, void* stream);
/* attention adjoints */
int rtfs_attn_out_norm_bwd(const float* dOut, const float* Ypre, float slope, const float* gamma_fc, float* dYpre, float* dgamma_fc, float* dbeta_fc,
                           float* dslope, int ntok, void* stream);
int rtfs_attn_qkv_norm_bwd(const float* dQ, const float* dK, const float* dV, const float* Ypre, const float* slope, const float* gq, const float* gk,
                           const float* gv, float* dYpre, float* dgq, float* dbq, float* dgk, float* dbk, float* dgv, float* dbv, float* dslope, int B,
                           int T2, void* stream);
int rtfs_attn_core_bwd(const float* Q, const float* K, const float* V, const float* O, const float* dO, const float* LSE, float* Dws, float* dQ,
                       float* dK, float* dV, int B, int T2, void* stream);
int rtfs_transpose_tok(const float* in, float* out, int ntok, void* stream);
/* S3 mask, CAF (training-mode BatchNorm), decoder / encoder ends */
/* rtfs_mask_bwd_elem: dz = gradient w.r.t. the mask pre-activation (ReLU gate from m), da_emb = gradient that reaches the encoder features through
 * the complex product - WRITTEN (the first contribution; round 3 accumulated into a zeroed buffer) */
int rtfs_mask_bwd_elem(const float* dmasked, const float* a_emb, const float* m, float* dz, float* da_emb, long long rows, void* stream);
/* rtfs_gemm_rows(dtaps, dec_wT, K = 32, N = 256) with rtfs_mask_bwd_elem in its epilogue: d(masked) never reaches HBM.  dtaps [rows][32] = the iSTFT /
 * decoder adjoint's tap gradients (18 used), dec_wT [256][32].  Adjoint of /root/reference/src/models/TDAVNet/mask_generator.py:70-82 behind the decoder's
 * ConvTranspose2d (decoder.py). */
int rtfs_decoder_mask_bwd(const float* dtaps, const float* dec_wT, const float* a_emb, const float* m, float* dz, float* da_emb, long long rows,
                          void* stream);
int rtfs_prelu_bwd(const float* dy, const float* x, float slope, float* dx, int accumulate, float* dslope, long long n, void* stream);
/* Round 6: the two 256 -> 256 input-gradient GEMMs of the step whose consumer is an activation's adjoint, with that adjoint in the GEMM's epilogue
 * (weight-stationary kernel at large maps; the launches they replace, in order, at small ones - same results).  Wt: [256][256] as rtfs_gemm_rows takes it
 * (Y = X . Wt^T), utterance layout [B][rows][256].
 *   rtfs_gemm_prelu_bwd            dx = prelu'(x) * (dz . Wt^T), dslope[0] += sum (dz . Wt^T) x [x <= 0]: the adjoint of Conv2d(PReLU(x)) w.r.t. x
 *                                  (mask_generator.py:47-48; = rtfs_gemm_rows + rtfs_prelu_bwd, dx may not alias dz or x)
 *   rtfs_gemm_gln_relu_bwd_reduce  dR = dy . Wt^T and the REDUCE pass of relu(gLN(x))'s adjoint on it (red: double[B][16] zeroed by the caller, dgamma /
 *                                  dbeta [256] accumulated; tdavnet.py:59,89's pre-norm + pre-act; = rtfs_gemm_rows + rtfs_gln_bwd_reduce(act 2));
 *                                  rtfs_gln_bwd_apply(dR, x, ..., act 2, red, ...) finishes the adjoint as before. */
int rtfs_gemm_prelu_bwd(const float* dz, const float* Wt, const float* x, float slope, float* dx, float* dslope, int B, int rows, void* stream);
int rtfs_gemm_gln_relu_bwd_reduce(const float* dy, const float* Wt, const float* x, const double* stats, const float* gamma, const float* beta, float* dR,
                                  double* red, float* dgamma, float* dbeta, int B, int rows, void* stream);
int rtfs_chan_stats(const float* x, double* sum, double* sumsq, long long rows, void* stream);
int rtfs_caf_bwd_reduce(const float* dOut, const float* x, const float* ks, const float* kb, const float* vs, const float* vb, const float* att,
                        const float* rsz, float* datt, float* drsz, float* R, int B, int T, int Tv, void* stream);
int rtfs_caf_bwd_apply(const float* dOut, const float* x, const float* ks, const float* kb, const float* att, const float* rsz, const float* coef,
                       float* dx, int accumulate, int B, int T, int Tv, void* stream);
int rtfs_istft_bwd(const float* dout, float* dspec, float* dtaps, int B, int L, void* stream);
int rtfs_spec_patches(const float* spec, float* patches, int B, int T, void* stream);

/* ---- a9 / f3: VP block = TDANetBlock with is2d = False (separators/tdanet.py:106-133) + GlobalAttention (layers/attention.py:28-73,
 * 192-220), eval mode, one launch: x, out [B][512][Tv] (NCT, as the reference hands the lip embedding); params: rtfs_vp_param_count()
 * floats packed as VpOff in csrc/vp.hip (BatchNorm1d folded); pe: rows of the PositionalEncoding buffer [>= 16][64].  3 <= Tv <= 100. */
int rtfs_vp_param_count(void);
int rtfs_vp_block_fwd(const float* x, const float* params, const float* pe, float* out, int B, int Tv, void* stream);

/* ---- f1: loss head, PairwiseNegSDR.forward src/losses/matrix.py:13-53 (the PIT search over the tiny [B][n][n] matrix stays on the host,
 * src/losses/pit_wrapper.py:82-107) ------------------------------------------------------------------------------------------- */
/* est, tgt [B][n_src][T]; sums [B][n_src][n_src][6] doubles, zeroed by the caller: (S_e, S_t, S_ee, S_tt, S_et, S_(e-t)^2) of pair (est i, target j) */
int rtfs_neg_sdr_sums(const float* est, const float* tgt, double* sums, int B, int n_src, int T, void* stream);
/* kind 0 snr / 1 sisdr / 2 sdsdr -> pw [B][n][n] = -10 log10(sdr + EPS) (take_log) and coef [B][n][n][4] = (ce, ct, mean e, mean t) */
int rtfs_neg_sdr_finish(const double* sums, int kind, int zero_mean, int take_log, float* pw, float* coef, int B, int n_src, int T, void* stream);
/* dest [B][n][T] = sum_j G[b][i][j] (ce (e_i - mean e_i) + ct (t_j - mean t_j)): the adjoint of pw w.r.t. the estimates */
int rtfs_neg_sdr_grad(const float* est, const float* tgt, const float* coef, const float* G, float* dest, int B, int n_src, int T, void* stream);

/* ---- f2: frozen lip encoder, FRCNNVideoModel with the ResNet-18 trunk, eval mode (src/models/videomodels/frcnn_videomodel.py:16-72,
 * resnet.py:27-130); channels-last fp32 activations [frame][y][x][channel], BatchNorm folded into weight rows (scale) and bias (shift) ---- */
/* frontend3D[0..2]: Conv3d(1->64, 5x7x7, stride (1,2,2), pad (2,3,3)) + BatchNorm3d + PReLU(64) (frcnn_videomodel.py:43-54).
 * P: zero-padded frames [B][T+4][H+6][W+6]; Ws [64][256], k = (dt*7+dy)*7+dx (k >= 245 zero); tapoff [256] ints =
 * (dt*(H+6)+dy)*(W+6)+dx (0 for k >= 245); bias, slope [64]; out [B*T][Hc][Wc][64], Hc = (H-1)/2+1, Wc = (W-1)/2+1. */
int rtfs_lip_stem_fwd(const float* P, const float* Ws, const int* tapoff, const float* bias, const float* slope, float* out, int B, int T, int H,
                      int W, void* stream);
/* frontend3D[3]: MaxPool3d((1,3,3), (1,2,2), (0,1,1)) (frcnn_videomodel.py:55): in [N][H][W][64] -> out [N][(H-1)/2+1][(W-1)/2+1][64] */
int rtfs_lip_maxpool_fwd(const float* in, float* out, int N, int H, int W, void* stream);
/* conv3x3 / 1x1 downsample + BatchNorm2d (+ residual) (+ PReLU) of BasicBlock (resnet.py:5-13, 49-66): in [N][H][W][Cin];
 * Wk [Cout][ks*ks*Cin], k = (dy*ks+dx)*Cin+ci; pad = ks/2; stride 1|2; bias, slope [Cout] or NULL; res [N][Ho][Wo][Cout] or NULL (added
 * before the activation); out [N][Ho][Wo][Cout], Ho = (H+2*pad-ks)/stride+1.  Cin, Cout multiples of 64; ks 1 or 3. */
int rtfs_conv_nhwc_fwd(const float* in, const float* Wk, const float* bias, const float* slope, const float* res, float* out, int N, int H, int W,
                       int Cin, int Cout, int ks, int stride, void* stream);
/* AdaptiveAvgPool2d(1) + view(B, T, C).transpose(1, 2) (resnet.py:124-126, frcnn_videomodel.py:66): in [B*T][HW][C] -> out [B][C][T] */
int rtfs_lip_avgpool_fwd(const float* in, float* out, int B, int T, int HW, int C, void* stream);
/* ---- f4: mouth-ROI preprocessing, get_preprocessing_pipelines() of src/datas/transform.py:151-167 (Normalize(0,255) -> Center/RandomCrop ->
 * HorizontalFlip -> Normalize(0.421, 0.165)) fused with the stem's zero padding.  roi [B][T][H][W] uint8; crop [B][3] ints (dy, dx, flip)
 * or NULL = centre crop (transform.py:96-101); lut [256] = the value map evaluated on the host in float64; P [B][T+4][ch+6][cw+6]. */
int rtfs_lip_roi_fwd(const unsigned char* roi, const int* crop, const float* lut, float* P, int B, int T, int H, int W, int ch, int cw, void* stream);

/* =====================================================================================================================
 * bf16 MFMA variants of the inference path (BASELINE config 5: "bf16 with MFMA attention").
 * Same computation as the fp32 entry point of the same name - activations stay fp32 in HBM, accumulation, norm statistics,
 * the SRU recurrence, softmax and (i)STFT stay fp32 - but every dense contraction runs on v_mfma_f32_32x32x16_bf16 /
 * v_mfma_f32_16x16x32_bf16 (16x the fp32 MFMA rate):
 *     terms = 1   operands rounded to bfloat16                                       (waveform ~4e-3 rel. of fp32, RTFS-Net-12)
 *     terms = 3   split-bf16, a.b ~ a_lo b_hi + a_hi b_lo + a_hi b_hi (three MFMAs)   (waveform ~8e-6 rel. of fp32)
 *     terms = 6   a = a_hi + a_mid + a_lo (three bf16 = the whole fp32 mantissa), six MFMAs (waveform ~7e-7: fp32-equivalent).  Weight arguments are
 *                 then the PLAIN fp32 matrices of the fp32 entry point (operands are split in registers), not host-packed ones
 * Weight arguments typed `const void* Wpk` are HOST-PACKED: the fp32 matrix W[n][k] of the fp32 entry point, every group of 4
 * consecutive k replaced in place by 8 bfloat16 {hi(k0..k3), lo(k0..k3)} (lo = bf16(w - hi); same byte size and indexing;
 * rtfs_net_amd/models/hip_path.py:pack_bf16).  rtfs_sru_layer_fwd_bf16 takes the plain fp32 weight.
 * Replaces the same reference lines as the fp32 siblings (attention.py:171-173 for the attention core, rnn_layers.py:146-150, ...).
 */
int rtfs_bottleneck_fwd_bf16(const float* a_emb, const double* stats, const float* gamma, const float* beta, const void* Wpk, const float* bias,
                             float* a0, int B, int TF, int terms, void* stream);
int rtfs_proj_fwd_bf16(const float* s, const float* gw, const float* gb, float gslope, const void* Wpk, const float* bias, float* y, double* stats_out,
                       int B, int TF, int terms, void* stream);
int rtfs_dp_unfold_gemm_fwd_bf16(const float* G, const float* gamma, const float* beta, const void* Wpk, float* U0, int B, int T2, int dim, int variant,
                                 int terms, void* stream);
int rtfs_sru_layer_fwd_bf16(const float* Hprev, const float* Wt, const float* weight_c, const float* bias, float scale_x, float* Hout,
                            float* Cout_or_null, float* Uout_or_null, int S, int L, int terms, void* stream);
int rtfs_dp_convt_fwd_bf16(const float* H3, const void* Wpk, const float* bias, float* G, int B, int T2, int dim, int terms, void* stream);
int rtfs_dp_convt_fwd_to_bf16(const float* H3, const void* Wpk, const float* bias, const float* Gin, float* Gout, int B, int T2, int dim, int terms, void* stream);
int rtfs_attn_qkv_fwd_bf16(const float* G, const void* Wpk, const float* bias, const float* slope, const float* gq, const float* bq, const float* gk,
                           const float* bk, const float* gv, const float* bv, float* Q, float* K, float* V, float* Ypre_or_null, int B, int T2, int terms,
                           void* stream);
int rtfs_attn_core_fwd_bf16(const float* Q, const float* K, const float* V, float* O, float* LSE_or_null, int B, int T2, int terms, void* stream);
int rtfs_attn_out_fwd_bf16(const float* O, const void* Wpk, const float* bias, float slope, const float* gamma_fc, const float* beta_fc, float* G,
                           float* Ypre_or_null, int B, int T2, int terms, void* stream);
int rtfs_attn_out_fwd_to_bf16(const float* O, const void* Wpk, const float* bias, float slope, const float* gamma_fc, const float* beta_fc, const float* Gin,
                              float* Gout, float* Ypre_or_null, int B, int T2, int terms, void* stream);
int rtfs_resid_fwd_bf16(const float* cl, const double* cl_stats, const float* cl_g, const float* cl_b, const float* d0, const double* d0_stats,
                        const float* d0_g, const float* d0_b, const float* cg, const double* cg_stats, const float* cg_g, const float* cg_b,
                        const float* cgate, const double* cgate_stats, const float* cgate_g, const float* cgate_b, const void* Wpk, const float* bias,
                        const float* s_in, const float* gw, const float* gb, float gslope, const float* a0_or_null, float* out, int B, int T, int T2,
                        int terms, void* stream);
int rtfs_resid_proj_fwd_bf16(const float* cl, const double* cl_stats, const float* cl_g, const float* cl_b, const float* d0, const double* d0_stats,
                             const float* d0_g, const float* d0_b, const float* cg, const double* cg_stats, const float* cg_g, const float* cg_b,
                             const float* cgate, const double* cgate_stats, const float* cgate_g, const float* cgate_b, const void* Wpk,
                             const float* bias, const float* s_in, const float* gw, const float* gb, float gslope, const float* a0, float* out,
                             const void* Wp_pk, const float* pbias, float* py, double* pstats, int B, int T, int T2, int variant, int terms,
                             void* stream);
int rtfs_resid_caf_fwd_bf16(const float* cl, const double* cl_stats, const float* cl_g, const float* cl_b, const float* d0, const double* d0_stats,
                            const float* d0_g, const float* d0_b, const float* cg, const double* cg_stats, const float* cg_g, const float* cg_b,
                            const float* cgate, const double* cgate_stats, const float* cgate_g, const float* cgate_b, const void* Wpk,
                            const float* bias, const float* s_in, const float* gw, const float* gb, float gslope, const float* ks, const float* kb,
                            const float* vs, const float* vb, const float* att, const float* rsz, int Tv, int add_input, float* out,
                            const void* Wp_pk_or_null, const float* pbias, float* py, double* pstats, int B, int T, int T2, int variant, int terms,
                            void* stream);
int rtfs_mask_fwd_bf16(const float* x, float slope, const void* Wpk, const float* bias, const float* a_emb, float* masked, float* m_or_null, int B,
                       int TF, int terms, void* stream);
int rtfs_gemm_rows_fwd_bf16(const float* X, const void* Wpk, const float* bias_or_null, float* Y, int M, int K, int N, int terms, void* stream);
/* bf16 / split-bf16 siblings of the MFMA entry points of the TRAINING step (same arguments + `terms`; fp32 accumulation; weight operands
 * host-packed, activation / gradient operands packed inside the kernels).  Gradients agree with the fp32 step to ~1e-5 with terms = 3. */
int rtfs_gemm_rows_bf16(const float* X, const void* Wpk, const float* bias_or_null, float* Y, int M, int K, int N, int accumulate, int terms, void* stream);
int rtfs_wgrad_bf16(const float* dY, int ldy, const float* X, int ldx, float* dW, int ldw, float* dbias, long long M, int seg_len, int x_seg, int x_off,
                    int nshift, int NOUT, int KIN, int pro, const float* p0, const float* p1, float slope, const double* stats, int rows_per_b, int terms,
                    void* stream);
int rtfs_decoder_mask_bwd_bf16(const float* dtaps, const void* dec_wT_pk, const float* a_emb, const float* m, float* dz, float* da_emb, long long rows,
                               int terms, void* stream);
int rtfs_proj_gateway_bwd_bf16(const float* dy0, const void* WpT_pk, const float* dx, const float* s, const float* gw, const float* gb, float slope,
                               float* ds, int accumulate, float* acc, int acc_mode, float* dgw, float* dgb, float* dslope, long long rows, int terms,
                               void* stream);
int rtfs_fold_gemm_bwd_bf16(const float* dU0, const void* Wpk, float* dxn, int B, int T2, int dim, int terms, void* stream);
int rtfs_convt_bwd_input_bf16(const float* dG, const void* Wpk, float* dH3, int B, int T2, int dim, int terms, void* stream);

/* =====================================================================================================================
 * VP (video) block, TRAINING step: the convolution / BatchNorm1d chain of the 1-D TDANetBlock (separators/tdanet.py:106-133,
 * is2d = False) with batch statistics, and its adjoints.  Tensors [B][64][T] (512 channels for block input / output), fp32.  A BatchNorm is never a
 * kernel of its own: producers write the pre-norm tensor and accumulate per-channel (sum, sum of squares) into `stats` [2][64] (float64, like the
 * gLN slots of the audio branch: order-independent sums, variance differenced in float64); consumers take
 * (stats, gamma, beta, inv_n = 1 / positions) and normalise on read - the host all-reduces the slot in between under SyncBatchNorm (train.py:145).
 * NULL stats = no normalisation.  Adjoint: a kernel leaves the gradient w.r.t. a BatchNorm OUTPUT, rtfs_vp_bn_bwd_reduce forms
 * sums [2][64] (float64) = (sum dyhat, sum dyhat * xhat) = (dbeta, dgamma), rtfs_vp_dwconv_bwd / rtfs_vp_gate_proj_bwd apply the BatchNorm adjoint on read
 * (batch_stats = 0: running statistics, no coupling terms).  The GlobalAttention stage of the block: rtfs_vp_attn_fwd / _bwd below.
 */
int rtfs_vp_gate_proj_fwd(const float* x, const float* gw, const float* gb, float gslope, const float* Wp /*[64][512]*/, const float* bp, float* r, float* y,
                          double* stats, int B, int T, void* stream);
/* depth-wise k = 3 (+bias) of BN(src) [PReLU'd if in_act = 1]; stride 1 ('same') or 2 (pad 1); optional second convolution (w1 -> out1, stats1) of the same input */
int rtfs_vp_dwconv_fwd(const float* src, const double* in_stats, const float* in_gamma, const float* in_beta, float in_inv_n, int in_act, float in_slope,
                       const float* w0 /*[64][3]*/, const float* b0_or_null, float* out0, double* stats0, const float* w1_or_null, float* out1, double* stats1,
                       int B, int Tin, int Tout, int stride, void* stream);
/* g = sum_i adaptive_avg_pool1d(BN_i(raw_i), Tg)  (tdanet.py:117-118); inv_n_i = 1 / positions behind the statistics of tensor i */
int rtfs_vp_pool_fwd(const float* const* raw, const double* const* stats, const float* const* gamma, const float* const* beta, int T0, int T1, int T2, int T3,
                     float inv_n0, float inv_n1, float inv_n2, float inv_n3, float* g, int B, int Tg, void* stream);
/* InjectionMultiSum mix (layers/fusion.py:59-67): out = BN(loc) * sigmoid(BN(gate))^ + BN(emb)^ (+ BN(res)); ^ = nearest up-sampling To -> Tn */
int rtfs_vp_mix_fwd(const float* loc, const double* loc_stats, const float* loc_g, const float* loc_b, float inv_n_loc, const float* gate,
                    const double* gate_stats, const float* gate_g, const float* gate_b, const float* emb, const double* emb_stats, const float* emb_g,
                    const float* emb_b, float inv_n_glob, const float* res_or_null, const double* res_stats, const float* res_g, const float* res_b, float* out,
                    int B, int Tn, int To, void* stream);
int rtfs_vp_resid_fwd(const float* e, const float* Wr /*[512][64]*/, const float* br, const float* r, float* out, int B, int T, void* stream);
int rtfs_vp_resid_bwd(const float* dout, const float* e, const float* Wr, float* de, float* dWr, float* dbr, int B, int T, void* stream);
int rtfs_vp_mix_bwd(const float* dout, const float* loc, const double* loc_stats, const float* loc_g, const float* loc_b, float inv_n_loc, const float* gate,
                    const double* gate_stats, const float* gate_g, const float* gate_b, float inv_n_glob, float* dloc, float* dgate, float* demb,
                    float* dres_acc_or_null, int B, int Tn, int To, void* stream);
int rtfs_vp_bn_bwd_reduce(const float* dyhat, const float* raw, const double* stats, const float* gamma, const float* beta, float inv_n, double* sums, int B,
                          int T, void* stream);
/* dW: [3][64] (tap-major) accumulated; dsrc = gradient w.r.t. the BatchNorm output of `src` (through the PReLU if in_act = 1), stored or accumulated */
int rtfs_vp_dwconv_bwd(const float* dyhat, const float* raw, const double* out_stats, const float* out_gamma, const float* out_beta, float out_inv_n,
                       const double* sums, float inv_n_all, int batch_stats, const float* src, const double* in_stats, const float* in_gamma,
                       const float* in_beta, float in_inv_n, int in_act, float in_slope, const float* w, float* dW, float* dbias_or_null,
                       float* dsrc_or_null, int accumulate, float* dslope_or_null, int B, int Tin, int Tout, int stride, void* stream);
int rtfs_vp_pool_bwd(const float* dg, float* const* d /*accumulated*/, int T0, int T1, int T2, int T3, int B, int Tg, void* stream);
int rtfs_vp_gate_proj_bwd(const float* dyhat, const float* y, const double* y_stats, const float* y_gamma, const float* y_beta, float y_inv_n,
                          const double* sums, float inv_n_all, int batch_stats, const float* dout, const float* x, const float* r, const float* gw,
                          const float* gb, float gslope, const float* Wp, float* dWp, float* dbp, float* dgw, float* dgb, float* dgslope, float* dx, int B,
                          int T, void* stream);

/* GlobalAttention of the VP block in the training step (layers/attention.py:28-73,192-220 MultiHeadSelfAttention incl. nn.MultiheadAttention,
 * layers/conv_layers.py:218-259 FeedForwardNetwork), csrc/vp_attn.hip: g, out, dout, dg [B][64][Tg], 2 <= Tg <= 16; params / dparams
 * [rtfs_vp_attn_param_count()] in the order norm1 (w, b), in_proj (w [192][64], b), out_proj (w [64][64], b), norm2 (w, b), FFN encoder conv
 * [128][64], its gLN (w, b), refiner dw conv [128][3] + bias, decoder conv [64][128], its gLN (w, b); pe: rows of the positional encoding
 * [>= Tg][64]; masks_or_null: per utterance rtfs_vp_attn_mask_size(Tg) multiplicative keep-masks of ONE step - attention-probability dropout
 * [8][Tg][Tg], dropout of the attention output [Tg][64], the three DropPath factors (after the MHSA, the refiner, the decoder) - or NULL for no
 * dropout.  rtfs_vp_attn_bwd recomputes the forward, writes dg and ADDS the parameter gradients of all B utterances into dparams. */
int rtfs_vp_attn_param_count(void);
int rtfs_vp_attn_mask_size(int Tg);
int rtfs_vp_attn_fwd(const float* g, const float* params, const float* pe, const float* masks_or_null, float* out, int B, int Tg, void* stream);
/* the same block in eval mode for MORE than 16 pooled tokens (2 <= Tg <= 1024; utterances longer than 5.1 s - the reference has no length limit): no masks, nothing
 * kept for an adjoint; work: [B][rtfs_vp_attn_long_work_floats(Tg)] floats of device scratch */
int rtfs_vp_attn_long_work_floats(int Tg);
int rtfs_vp_attn_long_fwd(const float* g, const float* params, const float* pe, float* out, float* work, int B, int Tg, void* stream);
int rtfs_vp_attn_bwd(const float* g, const float* params, const float* pe, const float* masks_or_null, const float* dout, float* dg, float* dparams, int B,
                     int Tg, void* stream);

/* ---- optimizer step of the training step (BASELINE config 3 / 4): torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW (config yaml:117-120, train.py:135-146
 * hand both to PyTorch / Lightning) as TWO launches over a chunk map of all parameter tensors, csrc/optim.hip; host side rtfs_net_amd/optim.py (FusedAdamW).
 * Tables (DEVICE memory): params / grads / exp_avg / exp_avg_sq = [n_tensors] int64 device pointers to contiguous fp32 tensors, sizes = [n_tensors] int64 element
 * counts, chunks = [n_chunks][2] int32 (tensor index, first element; 1024 elements per chunk).  rtfs_grad_sqnorm zeroes *sqnorm and adds the sum of squares of
 * every gradient element; rtfs_adamw_clip_step scales the gradients in place by min(1, max_norm / (sqrt(*sqnorm) + 1e-6)) (max_norm <= 0: no clipping, sqnorm
 * not read) and applies AdamW with decoupled weight decay in torch.optim.AdamW's operation order; bias_correction1 = 1 - beta1^step, bias_correction2_sqrt =
 * sqrt(1 - beta2^step) of this step; the hyper-parameters are doubles because torch forms 1 - beta, lr / bias_correction1 ... in double before they become fp32. */
int rtfs_grad_sqnorm(const long long* params, const long long* grads, const long long* exp_avg, const long long* exp_avg_sq, const long long* sizes,
                     const int* chunks, int n_chunks, double* sqnorm, void* stream);
int rtfs_adamw_clip_step(const long long* params, const long long* grads, const long long* exp_avg, const long long* exp_avg_sq, const long long* sizes,
                         const int* chunks, int n_chunks, const double* sqnorm, double max_norm, double lr, double beta1, double beta2, double eps,
                         double weight_decay, double bias_correction1, double bias_correction2_sqrt, void* stream);

/* ---- BatchNorm2d (batch statistics) of the CAF cell's key / value embeddings in the training step (fusion.py:249-253), csrc/optim.hip: the per-channel
 * arithmetic between rtfs_chan_stats and the cell (forward: folded scale / shift, running-statistics update) and between rtfs_caf_bwd_reduce and
 * rtfs_caf_bwd_apply (adjoint: coefficients c1, c2, c3 + the three parameter gradients per tag), one launch each.  sums: [2][C] fp64; n: one double ON THE
 * DEVICE (positions behind sums_glob); *_glob: sums over all ranks under SyncBatchNorm, the local pointer otherwise.  R: [4][C] = (sum dy, sum dy x) of key, of value. */
int rtfs_caf_bn_prepare(const double* sums_glob, const double* sums_loc, const double* n, const float* k_dw, const float* k_g, const float* k_be,
                        float* k_run_mean, float* k_run_var, long long* k_batches, float* k_inv, float* k_mean_u, float* k_s, float* k_b, const float* v_dw,
                        const float* v_g, const float* v_be, float* v_run_mean, float* v_run_var, long long* v_batches, float* v_inv, float* v_mean_u, float* v_s,
                        float* v_b, float momentum, float eps, float* mean_x, float* var_x, float* lsum, float* lcov, void* stream);
int rtfs_caf_bn_adjoint(const float* R_loc, const float* R_glob, const double* n, const float* mean_x, const float* lsum, const float* lcov, const float* k_dw,
                        const float* k_g, const float* k_inv, float* k_d_dw, float* k_d_g, float* k_d_be, const float* v_dw, const float* v_g,
                        const float* v_inv, float* v_d_dw, float* v_d_g, float* v_d_be, float* coef, void* stream);

/* ---- the whole adjoint of up to four stride-1 depth-wise 4x4 convolutions that share one input, in one pass (csrc/bwd_dw.hip, round 6) -----------------
 * autograd of ConvNormAct(groups = channels) -> gLN (conv_layers.py:65-129; tdanet.py:61-76, layers/fusion.py:25-52).  nconv in {1, 2, 4}; dy[k]: gradient w.r.t.
 * convolution k's output - or, when x != NULL, w.r.t. its gLN-NORMALISED output, the gLN adjoint  dX = rstd (gamma dN - S1/N - xhat S2/N)  then being applied on load
 * from x[k] (pre-norm output), x_stats[k] (its forward statistics), red[k] (S1, S2: rtfs_gln_bwd_reduce / rtfs_mix_gln_bwd / rtfs_d0_tail_bwd) and gamma[k];
 * in: the common input, transformed as the forward saw it (mode 0 raw, 1 gLN, 2 PReLU(gLN), 3 the TFAR mix gLN(in) * sigmoid(gLN(gate))^ + gLN(glob)^ re-formed per
 * pixel as rtfs_dwconv_mix_fwd forms it: in_mix = host array {gate, gate_stats, gate_gamma, gate_beta, glob, glob_stats, glob_gamma, glob_beta} at in_Tg x in_Fg, NULL
 * otherwise); dIn (=, or += when accumulate) the gradient w.r.t. that TRANSFORMED input; dW[k] [16][64] += tap gradients, dbias[k] [64] += bias gradients (dbias NULL: none).  Pointer arrays are host arrays of device pointers.
 * Replaces rtfs_gln_bwd_apply + rtfs_dwconv_bwd_weight + rtfs_dwconv_bwd_input per convolution: dX never reaches HBM. */
int rtfs_dw_adjoint(int nconv, const float* const* dy, const float* const* x, const double* const* x_stats, const double* const* red,
                    const float* const* gamma, const float* const* w, const float* in, const double* in_stats, const float* in_gamma, const float* in_beta,
                    float in_slope, int mode, const void* const* in_mix, int in_Tg, int in_Fg, float* dIn, int accumulate, float* const* dW,
                    float* const* dbias, int B, int T, int F, void* stream);

/* the same for ONE convolution that is the local branch of an InjectionMultiSum (layers/fusion.py:54-69: out = gLN(loc) * sigmoid(gLN(gate))^ + gLN(glob)^): dOut is
 * the gradient w.r.t. the mix's output; the local branch's mix + gLN adjoint  dX = rstd (gamma dOut s^ - S1/N - xhat S2/N)  is applied on load from loc (pre-norm
 * output of the convolution), its statistics, loc_red = (S1, S2) and gate_sig = s = sigmoid(gLN(gate)) [B][Tg][Fg][64], both written by rtfs_mix_gln_bwd_sig. */
int rtfs_dw_adjoint_mix(const float* dOut, const float* loc, const double* loc_stats, const double* loc_red, const float* loc_gamma, const float* gate_sig,
                        int Tg, int Fg, const float* w, const float* in, const double* in_stats, const float* in_gamma, const float* in_beta, float in_slope,
                        int mode, const void* const* in_mix, int in_Tg, int in_Fg, float* dIn, int accumulate, float* dW, int B, int T, int F, void* stream);
/* rtfs_mix_gln_bwd with one more output: sig (or NULL) receives sigmoid(gLN(gate)), which the reduce pass forms anyway; dLoc may then be NULL - the apply pass of the
 * local branch is left to rtfs_dw_adjoint_mix (dLoc never reaches HBM). */
int rtfs_mix_gln_bwd_sig(const float* dOut, const float* loc, const double* loc_stats, const float* loc_g, const float* loc_b, const float* gate,
                         const double* gate_stats, const float* gate_g, const float* gate_b, const float* glob, const double* glob_stats, const float* glob_g,
                         const float* glob_b, float* dLoc, float* dNgate, float* dNglob, float* sig, double* red, float* const* dgb, int B, int T, int F, int Tg,
                         int Fg, void* stream);

/* ---- module-boundary views (csrc/views.hip): the reference's stage modules called one at a time, forward hooks ------------------------------
 * The reference runs `self.encoder(x)`, `self.audio_bottleneck(...)`, `self.refinement_module(a, v)`, `self.mask_generator(...)`,
 * `self.decoder(...)` as ordinary modules (src/models/tdavnet.py:86-97; base_av_model.py:61-118 calls them one by one) and lets hooks see every
 * sub-module's output.  The fused forward never materialises what the next kernel can form on load; these entry points materialise those views.
 *   rtfs_gln_stats     stats[b] += (sum, sum of squares) of x[b] (per_utt floats per utterance, multiple of 4); the caller zeroes the slot
 *   rtfs_norm_act_fwd  y = act(GroupNorm(1,C)(x)) (normalizations.py:8-17 + activations.py): act 0 none, 1 PReLU(slope), 2 ReLU, 3 sigmoid; x [B][rows][C]
 *   rtfs_gateway_fwd   y = prelu(x * w + b, slope), the gateway's output (tdanet.py:34-41,108); x [rows][256]
 *   rtfs_cl_to_nchw / rtfs_nchw_to_cl   [B][P][C] <-> [B][C][P] (P = T*F pixels) */
int rtfs_gln_stats(const float* x, double* stats, int B, long long per_utt, void* stream);
int rtfs_norm_act_fwd(const float* x, const double* stats, const float* gamma, const float* beta, int act, float slope, float* y, int B, long long rows,
                      int C, void* stream);
int rtfs_gateway_fwd(const float* x, const float* w, const float* bias, float slope, float* y, long long rows, void* stream);
int rtfs_cl_to_nchw(const float* src, float* dst, int B, int P, int C, void* stream);
int rtfs_nchw_to_cl(const float* src, float* dst, int B, int P, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RTFS_HIP_H */
