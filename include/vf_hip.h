/* vf_hip.h -- C ABI of libvf_hip.so, the MI355X (gfx950) kernels of the ViewFusion hot path.
 *
 * The reference (bronemos/view-fusion) has no FFI: its hot path is `model/unet.py` +
 * `model/view_fusion.py` calling stock torch ops.  Each entry point below replaces the torch
 * op(s) cited next to it.  Conventions:
 *   - raw DEVICE pointers to contiguous fp32 NCHW tensors, explicit sizes, explicit stream
 *     (`hipStream_t` passed as void*); no torch types;
 *   - every call only ENQUEUES work: no allocation, no synchronisation, no retained
 *     pointers, re-entrant, HIP-graph capturable; workspaces are passed in by the caller
 *     (the one exception: the five vf_xgmi_* MEMORY calls at the end of this file, which allocate / export / map the
 *     IPC-shared gradient arena once per process, outside any stream);
 *   - return value is a hipError_t as int (0 = success; hipErrorInvalidValue for an
 *     unsupported shape -- there is no fallback path).
 */
#ifndef VF_HIP_H
#define VF_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- GroupNorm (+Swish) : nn.GroupNorm(32,C,eps) -> Swish, unet.py:211-212,254,180-182 ----
 * HW = H*W must be a power of two >= 4 (square power-of-two maps), else hipErrorInvalidValue. */
int vf_gn_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean /*[S*G]*/,
              float* rstd /*[S*G]*/, int S, int C, int HW, int groups, float eps, int silu, void* stream);
/* dgamma_part/dbeta_part: [S][C] per-view partials; reduce over S with vf_colsum.
 * addend (like x, or NULL) is added to dx: gradient of a second consumer of x (residual branch).
 * dx_rowsum ([S][C] or NULL): sum of dx over the map per (view, channel), excluding addend -- the gradient of
 * the conv bias / embedding bias added in front of this GroupNorm (unet.py:160-177); only filled where
 * vf_gn_bwd_emits_rowsum() says so. */
int vf_gn_bwd_emits_rowsum(int C, int HW, int groups);
int vf_gn_bwd(const float* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
              const float* dy, const float* addend, float* dx, float* dgamma_part, float* dbeta_part,
              float* dx_rowsum, int S, int C, int HW, int groups, int silu, void* stream);
/* same on the never-materialised channel concatenation [x (C1 channels) | x2 (C - C1)] of two NCHW tensors
 * (decoder skip connections, unet.py:134); the second-consumer gradients and dx are split the same way.
 * x2 == NULL: plain input; then addend2 (or NULL) is a second full-size tensor added to dx (gradient of a third
 * consumer of x).  Backward with x2: single-pass shapes only (vf_gn_bwd_emits_rowsum). */
int vf_gn_cat_fwd(const float* x, const float* x2, int C1, const float* gamma, const float* beta, float* y, float* mean,
                  float* rstd, int S, int C, int HW, int groups, float eps, int silu, void* stream);
int vf_gn_cat_bwd(const float* x, const float* x2, int C1, const float* gamma, const float* beta, const float* mean,
                  const float* rstd, const float* dy, const float* addend, const float* addend2, float* dx, float* dx2,
                  float* dgamma_part, float* dbeta_part, float* dx_rowsum, int S, int C, int HW, int groups, int silu,
                  void* stream);
int vf_rowsum(const float* x, float* out /*[rows]*/, int rows, int len, void* stream);
/* conv epilogue gradients in one launch: db[C] (|NULL) and dvb[S][C] (|NULL) from dy[S][C][HW] */
int vf_bias_grad(const float* dy, float* db, float* dvb, int S, int C, int HW, void* stream);
int vf_colsum(const float* part /*[batch][S][C]*/, float* out /*[batch][C]*/, int batch, int S, int C,
              void* stream);
/* many column sums in one launch: desc = device int64 [n][6] rows {part, out, S, C, batch, first 64-column block};
 * total_blocks = sum over the rows of ceil(C/64) * batch.  Same arithmetic as vf_colsum per row. */
int vf_colsum_multi(const void* desc, int n, long total_blocks, void* stream);

/* ---- convolution : nn.Conv2d 3x3 / 1x1, nn.Upsample+conv, stride-2 conv,
 *      unet.py:42,189,198,214,238,255,256 (+ FeatureWiseAffine add :160-177, residual add :245,277) ---- */
int vf_conv_pack_sizes(int Cout, int Cin, int KS, long* fwd_floats, long* bwd_floats);
int vf_conv_pack_weights(const float* w_oihw, float* w_fwd, float* w_bwd /*or NULL*/, int Cout, int Cin, int KS,
                         void* stream);
/* every layer of a network in one launch; desc = device int64 [nlayers][9] rows
 * {w, w_fwd, w_bwd, Cout, Cin, KS, fwd_floats, bwd_floats, first_block}, block = 256 elements */
int vf_conv_pack_weights_multi(const void* desc, int nlayers, long total_blocks, void* stream);
/* mode 0 stride-1 | 1 stride-2 (x is 2Hx2W) | 2 nearest-x2 upsampled x (x is H/2xW/2) |
 * 4 sub-pixel dgrad of mode 1: x = dY (HxW), y = dX (2Hx2W), every
 * output parity gets its own taps (no zeros multiplied; w_packed = the dgrad pack; no epilogue operands).
 * H,W = OUTPUT size (mode 4: the dY size), square power of two in [8,128]. */
int vf_conv_fwd(const float* x, const float* w_packed, const float* bias /*[Cout]|NULL*/,
                const float* view_bias /*[S][Cout]|NULL*/, const float* residual /*like y|NULL*/, float* y,
                float* ws /*|NULL*/, long ws_floats, int S, int Cin, int Cout, int H, int W, int KS, int mode,
                void* stream);
/* inference fusion: a_out = [Swish](GroupNorm(groups, eps)(conv output)), the GroupNorm evaluated inside the conv's
 * split-K reduce launch where the conv runs split-K (small S: the sampler), by a GroupNorm launch behind it otherwise.
 * y: a valid output buffer; it holds the conv output afterwards if store_y != 0 (or the unfused route ran).
 * gn_stats: scratch, 2*S*groups floats. */
int vf_conv_fwd_gn(const float* x, const float* w_packed, const float* bias, const float* view_bias, const float* residual,
                   float* y, int store_y, const float* gn_gamma, const float* gn_beta, float* a_out, float* gn_stats,
                   int groups, float eps, int silu, float* ws, long ws_floats, int S, int Cin, int Cout, int H, int W,
                   int KS, int mode, void* stream);
/* The sampler's convolutions at few stacked views (model/view_fusion.py:179-214 driving model/unet.py:42,189,198,214,
 * 238,255,256 with S = 1 ... a dozen): ONE launch per layer, K split over the waves of a workgroup and summed in LDS in a
 * fixed order; reads the UNPACKED OIHW parameter.  x2 != NULL (1x1 only): input channels [C1, Cin) come from x2.
 * H = W = the output map, stride 1: mode 0, or mode 2 for 3x3 (x stored at half size, nearest-x2 upsampled on read: the
 * Upsample conv of unet.py:185-190; round 5) (vf_conv_small_supported says which layers are taken:
 * 1x1 with Cin % 4 == 0, 3x3 with Cin % 32 == 0, power-of-two maps of at least 4x4). */
int vf_conv_small_supported(int Cin, int Cout, int H, int W, int KS, int mode);
int vf_conv_small(const float* x, const float* x2 /*|NULL*/, int C1, const float* w_oihw, const float* bias /*|NULL*/,
                  const float* view_bias /*|NULL*/, const float* residual /*|NULL*/, float* y, int S, int Cin, int Cout,
                  int H, int W, int KS, int mode, void* stream);
/* The same 3x3 launch with the residual block's 1x1 convolution folded in as extra K (unet.py:238,245:
 * block2(h) + res_conv(x)):  y = conv3x3(x, w) + bias + view_bias + conv1x1([rx | rx2], rw_oihw) + rbias.
 * rC = channels of [rx | rx2] (a multiple of 4), the first rC1 of them from rx; rx2 == NULL: all from rx. */
int vf_conv_small_res(const float* x, const float* w_oihw, const float* bias /*|NULL*/, const float* view_bias /*|NULL*/,
                      float* y, int S, int Cin, int Cout, int H, int W, const float* rx, const float* rx2 /*|NULL*/,
                      int rC1, int rC, const float* rw_oihw, const float* rbias /*|NULL*/, void* stream);
/* The general one-launch form (round 5): GroupNorm(+Swish) around the conv without a GroupNorm launch (unet.py:207-218,
 * 254: GroupNorm -> Swish -> conv).  in_stats != NULL: x is the RAW GroupNorm input and in_stats the [S][Cin][2] 64-bit
 * integer sums (x, x^2 in 2^-24 units) left by the launch that produced x; the normalisation is applied while x is
 * staged (not with x2).  out_stats != NULL ([S][Cout][2] uint64, ZERO before the launch): the launch adds the sums of
 * y and y^2 per (view, channel) with integer atomics -- order-independent, bit-reproducible.  rx != NULL: as
 * vf_conv_small_res.  w_packed != NULL (3x3): weights from vf_conv_small_pack.  mode 0, or 2 for a plain 3x3 conv. */
int vf_conv_small_gn(const float* x, const float* x2 /*|NULL*/, int C1, const float* w_oihw, const float* bias /*|NULL*/,
                     const float* view_bias /*|NULL*/, const float* residual /*|NULL*/, float* y, int S, int Cin,
                     int Cout, int H, int W, int KS, const unsigned long long* in_stats /*|NULL*/,
                     const float* in_gamma, const float* in_beta, int in_groups, float eps, int silu,
                     unsigned long long* out_stats /*|NULL*/, const float* rx /*|NULL*/, const float* rx2 /*|NULL*/,
                     int rC1, int rC, const float* rw_oihw, const float* rbias /*|NULL*/, const float* w_packed /*|NULL*/,
                     int mode, void* stream);
/* 3x3 weights in the load order of the one-launch kernel (every weight load of a wave = 1 KB of consecutive memory instead
 * of 16 pieces of 16 OIHW rows); static weights (the sampler) are packed once.  Cin % 32 == 0. */
long vf_conv_small_pack_floats(int Cout, int Cin);
int vf_conv_small_pack(const float* w_oihw, float* w_packed, int Cout, int Cin, void* stream);
/* split-K workspace the call above wants at this shape (0 when the natural grid fills the chip) */
long vf_conv_fwd_ws_floats(int S, int Cin, int Cout, int H, int W, int KS);
/* 1x1 conv on the channel concatenation [x1 (C1 channels, multiple of 64) | x2] (residual conv of the decoder
 * blocks, unet.py:134,238): forward, dgrad into two tensors, wgrad */
int vf_conv1x1_cat_fwd(const float* x1, const float* x2, int C1, const float* w_packed, const float* bias, float* y,
                       float* ws, long ws_floats, int S, int Cin, int Cout, int H, int W, void* stream);
int vf_conv1x1_cat_dgrad(const float* dy, const float* w_packed_bwd, float* dx1, float* dx2, int C1, int S, int Cin,
                         int Cout, int H, int W, void* stream);
int vf_conv1x1_cat_wgrad(const float* x1, const float* x2, int C1, const float* dy, float* dw_oihw, float* ws,
                         long ws_floats, int S, int Cin, int Cout, int H, int W, void* stream);
/* EXPERIMENT (hosts select it with VF_BF16X3=1; default off): the same 1x1 convolutions with fp32-accurate products on
 * the bf16 matrix path -- every operand split into three bf16 pieces, six MFMAs per K=16 block, fp32 accumulation
 * (csrc/conv1x1_bf16x3.hip).  pack: w_oihw [Cout][Cin] -> split + packed forward operand (and the transposed dgrad
 * operand if w3_bwd != NULL); sizes in 32-bit words from vf_conv1x1_bf16x3_pack_dwords(M, K) with (M, K) = (Cout, Cin) /
 * (Cin, Cout).  conv: x may be the concatenation [x | x2] (x2 != NULL: C1in channels in x), y may be split into
 * [y | y2] (y2 != NULL: C1out channels in y); HW a power of two >= 64. */
long vf_conv1x1_bf16x3_pack_dwords(int M, int K);
int vf_conv1x1_bf16x3_pack(const float* w_oihw, void* w3_fwd, void* w3_bwd, int Cout, int Cin, void* stream);
int vf_conv1x1_bf16x3(const float* x, const float* x2, int C1in, const void* w3, const float* bias,
                      const float* view_bias, const float* residual, float* y, float* y2, int C1out, int S, int Cin,
                      int Cout, int HW, void* stream);
long vf_conv_wgrad_ws_floats(int S, int Cin, int Cout, int H, int W, int KS);
int vf_conv_wgrad(const float* x, const float* dy, float* dw_oihw, float* ws, long ws_floats, int S, int Cin,
                  int Cout, int H, int W, int KS, int mode, void* stream);
/* vf_conv_wgrad / vf_conv1x1_cat_wgrad without their follow-up launch (round 6): the main kernel only; desc9 (HOST memory,
 * 9 x int64) receives the row vf_wino44_reduce_multi needs for this layer, *nblocks its workgroup count; ws must stay
 * untouched until that launch (hosts that can postpone the weight gradients to the end of the backward pass). */
int vf_conv_wgrad_main(const float* x, const float* dy, float* dw_oihw, float* ws, long ws_floats, int S, int Cin,
                       int Cout, int H, int W, int KS, int mode, long long* desc9, int* nblocks, void* stream);
int vf_conv1x1_cat_wgrad_main(const float* x1, const float* x2, int C1, const float* dy, float* dw_oihw, float* ws,
                              long ws_floats, int S, int Cin, int Cout, int H, int W, long long* desc9, int* nblocks,
                              void* stream);
int vf_sumpool2(const float* x, float* y, long n_out, int Wo, void* stream);

/* fused Winograd F(2x2,3x3) path for stride-1 3x3 convs on 8x8 / 16x16 / 32x32 / 64x64 maps (forward and
 * dgrad); modes 0 and 2 as above.  Pack layout differs from the direct kernel's. */
int vf_wino_supported(int H, int W, int mode);
int vf_wino_pack_sizes(int Cout, int Cin, long* fwd_floats, long* bwd_floats);
int vf_wino_pack_weights(const float* w_oihw, float* u_fwd, float* u_bwd /*or NULL*/, int Cout, int Cin,
                         void* stream);
/* desc = device int64 [nlayers][8] rows {w, u_fwd, u_bwd, Cout, Cin, fwd_floats, bwd_floats, first_block} */
int vf_wino_pack_weights_multi(const void* desc, int nlayers, long total_blocks, void* stream);
/* ws: room for the K-split partial tiles of a grid that does not divide the 256 CUs (vf_wino_conv_ws_floats;
 * NULL / too small = plain grid) */
long vf_wino_conv_ws_floats(int S, int Cin, int Cout, int H, int W);
/* expected CU fill in percent under the tail plan (+ the workgroup-tile count): policy input for hosts */
int vf_wino_conv_fill_pct(int S, int Cin, int Cout, int H, int W, int* tiles_out);
int vf_wino_conv_fwd(const float* x, const float* u_packed, const float* bias, const float* view_bias,
                     const float* residual, float* y, float* ws, long ws_floats, int S, int Cin, int Cout, int H,
                     int W, int mode, void* stream);
/* The same conv with the GroupNorm(+Swish) BEHIND it (unet.py:207-218: the next Block's norm) evaluated by the conv's
 * fix-up launch -- possible where every tile of the launch is a K-split tail tile (the sampler at a few stacked views,
 * 16x16 ... 64x64 maps): vf_wino_conv_gn_fusable() says where.  a_out = [Swish](GroupNorm(groups, eps)(y)); y itself is
 * written only if store_y.  One graph node less than conv + fix-up + GroupNorm. */
int vf_wino_conv_gn_fusable(int S, int Cin, int Cout, int H, int W, int mode, int groups);
int vf_wino_conv_fwd_gn(const float* x, const float* u_packed, const float* bias /*|NULL*/, const float* view_bias /*|NULL*/,
                        const float* residual /*|NULL*/, float* y, int store_y, const float* gn_gamma, const float* gn_beta,
                        float* a_out, int groups, float eps, int silu, float* ws, long ws_floats, int S, int Cin, int Cout,
                        int H, int W, int mode, void* stream);
/* fused Winograd F(4x4,3x3) path (36 products per 4x4 outputs: 2.25 multiplies per output instead of the nested
 * kernel's 3) for the stride-1 3x3 convs on the 32x32 / 64x64 maps, forward and dgrad -- replaces nn.Conv2d 3x3
 * at reference model/unet.py:42,189,214 on those maps; same calling convention as the vf_wino_* entries above, its own
 * pack layout (U = G4 w G4^T, 36 slices in wave order). */
int vf_wino44_supported(int H, int W, int mode);
int vf_wino44_pack_sizes(int Cout, int Cin, long* fwd_floats, long* bwd_floats);
int vf_wino44_pack_weights(const float* w_oihw, float* u_fwd, float* u_bwd /*or NULL*/, int Cout, int Cin,
                           void* stream);
int vf_wino44_pack_weights_multi(const void* desc, int nlayers, long total_blocks, void* stream);
long vf_wino44_conv_ws_floats(int S, int Cin, int Cout, int H, int W);
int vf_wino44_conv_fill_pct(int S, int Cin, int Cout, int H, int W, int* tiles_out);
int vf_wino44_conv_fwd(const float* x, const float* u_packed, const float* bias, const float* view_bias,
                       const float* residual, float* y, float* ws, long ws_floats, int S, int Cin, int Cout, int H,
                       int W, int mode, void* stream);
/* weight gradient of a stride-1 3x3 conv (modes 0 and 2, output H = W in {8,16,32,64}) through the same
 * transform: dU = sum over tiles of (A dY A^T) (B^T d B)^T per Winograd slice, split over tile ranges into `ws`
 * slabs that are summed in a fixed order, then dW = G^T dU G */
int vf_wino_wgrad_supported(int H, int W, int mode);
long vf_wino_wgrad_ws_floats(int S, int Cin, int Cout, int H, int W);
/* db (or NULL): also the bias gradient sum_{s,p} dY -- the kernel reads every dY tile anyway; db2 (or NULL) is a
 * second [Cout] destination for the same sums (a layer sharing this dY, the residual 1x1 conv, gets its own tensor) */
int vf_wino_wgrad(const float* x, const float* dy, float* dw_oihw, float* db, float* db2, float* ws, long ws_floats,
                  int S, int Cin, int Cout, int H, int W, int mode, void* stream);
/* the same without the follow-up slab-sum launch (round 5): only the main kernel runs; desc9 (HOST memory, 9 x int64)
 * receives this layer's row for vf_wino44_reduce_multi, *nblocks its workgroup count; ws must stay untouched until then.
 * vf_wino44_reduce_multi sums the slabs of many layers in one launch: table = DEVICE copy of the rows, each row's `first`
 * field (the int32 at byte 64) set to the sum of the preceding rows' workgroup counts, nblocks = the total. */
int vf_wino_wgrad_main(const float* x, const float* dy, float* dw_oihw, float* db, float* db2, float* ws, long ws_floats,
                       int S, int Cin, int Cout, int H, int W, int mode, long long* desc9, int* nblocks, void* stream);
int vf_wino44_reduce_multi(const void* table, int nrows, int nblocks, void* stream);

/* ---- grouped time-embedding affine: all FeatureWiseAffine Linear(K->C_g) layers of the UNet on the same
 *      (S,K) embedding in one launch, unet.py:160-177.  desc = device int64 [ngroups][5] rows
 *      {W_g, b_g, C_g, out_off_g (floats), c_off_g}; out/de = flat buffers of the (S,C_g) matrices; CT = sum C_g ---- */
int vf_time_affine_fwd(const void* desc, int ngroups, const float* emb, float* out, int S, int K, int CT,
                       void* stream);
long vf_time_affine_ws_floats(int S, int K);
/* gdst (or NULL): device array of ngroups rows {float* dW_g, float* db_g}: one destination per layer (the slots of
 * a data-parallel gradient buffer) instead of the flat dw / db */
int vf_time_affine_bwd(const void* desc, int ngroups, const float* emb, const float* de, float* dw /*[CT][K]*/,
                       float* db /*[CT]*/, const void* gdst, float* demb /*[S][K] or NULL*/, float* ws, int S, int K,
                       int CT, void* stream);

/* ---- batched GEMM + softmax : torch.einsum / torch.softmax / nn.Linear,
 *      unet.py:267-274 (attention), :29-31,165 (linears) ---- */
int vf_bgemm(const float* A, const float* B, float* C, const float* bias /*[N]|NULL*/, int batch, int M, int N,
             int K, long sAb, long sAm, long sAk, long sBb, long sBk, long sBn, long sCb, long sCm, long sCn,
             float alpha, float beta, void* stream);
/* fused attention forward (unet.py:258-277 core): qkv [S][3C][L] -> out [S][C][L]; optionally
 * P [S][L][L] (softmax probabilities, saved for backward).  L in {64,256}, C % 32 == 0. */
int vf_attention_fwd(const float* qkv, float* out, float* P /*|NULL*/, int S, int C, int L, void* stream);
/* attention backward (autograd of unet.py:267-274), L = 256, C % 32 == 0, first of three launches:
 * dS = P o (dP - rowsum(P o dP)) with dP = dO^T V computed in the kernel, and dQ = K dS^T / sqrt(C) -> q third of dqkv;
 * qkv, dqkv [S][3C][L], dO [S][C][L], P, dS [S][L][L] (dS must not alias P).  dV and dK stay vf_bgemm calls. */
int vf_attention_dscore(const float* qkv, const float* dO, const float* P, float* dS, float* dqkv, int S, int C, int L,
                        void* stream);
/* ... second launch (C % 64 == 0): dV = dO P and dK = q dS / sqrt(C) -> the v and k thirds of dqkv */
int vf_attention_dvdk(const float* qkv, const float* dO, const float* P, const float* dS, float* dqkv, int S, int C, int L,
                      void* stream);
int vf_softmax_fwd(const float* x, float* y, int rows, int cols, void* stream);
int vf_softmax_bwd(const float* y, const float* dy, float* dx, int rows, int cols, void* stream);

/* ---- small elementwise : PositionalEncoding unet.py:142-157, Swish :180-182, torch.cat :134 ---- */
int vf_sincos_embed(const float* level /*[S]*/, const float* angle /*[S]*/, float* out /*[S][dim]*/, int S, int dim,
                    void* stream);
int vf_swish_fwd(const float* x, float* y, long n, void* stream);
int vf_swish_bwd(const float* x, const float* dy, float* dx, long n, void* stream);
int vf_concat_channels(float* a, float* b, float* out, int S, long na, long nb, int split, void* stream);
/* nn.Dropout(p) inside Block, unet.py:207-216 (training mode): y = x * (u >= p) / (1-p), u = caller's uniform draws;
 * applied to dy it is the backward */
int vf_dropout(const float* x, const float* u, float* y, long n, float p, void* stream);

/* ---- ViewFusion : view_fusion.py:162-164 (q_sample), :244-263/:95-115 (stack), :265-298/:116-150
 *      (compose, mean ablation, MSE), :70-84,152-177 (posterior + p_sample), :314-317 (extract) ---- */
int vf_gather_level(const float* gammas, const long long* t, const float* u /*[B]|NULL*/, float* level, int B,
                    void* stream);
int vf_stack_views(const float* y_cond /*[B][Nmax][Cc][HW]*/, const float* y_t /*[B][3][HW]*/,
                   const float* noise /*|NULL*/, const float* level, const float* angle, const int* off /*[B+1]*/,
                   float* x /*[S][Cc+3][HW]*/, float* level_s, float* angle_s, int B, int Nmax,
                   int Cc /* conditioning channels: 3, or 6 for the `relative` configs */, int HW, int S,
                   int copy_cond, void* stream);
int vf_compose_fwd(const float* unet_out, const int* off, const float* target /*|NULL*/, float* noise_hat,
                   float* weights /*[B][maxV][3][HW]|NULL*/, float* loss_part /*[B*64]*/, float* loss, int B,
                   int Cout, int HW, int maxV, int weighting, void* stream);
int vf_compose_mse_bwd(const float* unet_out, const int* off, const float* target, const float* noise_hat,
                       const float* gloss, float* dout, int B, int Cout, int HW, int weighting, void* stream);
int vf_p_sample_tail(const float* unet_out, const int* off, const float* y_t, const float* z /*|NULL*/,
                     const long long* t, const float* sqrt_recip_gammas, const float* sqrt_recipm1_gammas,
                     const float* posterior_log_variance, const float* posterior_mean_coef1,
                     const float* posterior_mean_coef2, float* y_next /*|NULL*/, float* mean_out /*|NULL*/,
                     float* weights /*|NULL*/, int B, int Cout, int HW, int maxV, int weighting, int clip,
                     void* stream);

/* eval metric next to the path: utils/metrics.py:6-8; out[b] = PSNR of image b (n floats each) */
int vf_psnr(const float* generated, const float* target, float* out /*[B]*/, int B, int n, void* stream);

/* ---- optimizer step next to the path: torch.optim.Adam, experiment.py:118-120,293 ----
 * desc = device int64 [ntensors][6] rows {p, g, exp_avg, exp_avg_sq, numel, first_block}, block = 1024 elems */
int vf_adam_multi(const void* desc, int ntensors, long total_blocks, float lr, float beta1, float beta2, float eps,
                  float bias_correction1, float bias_correction2, void* stream);
/* the same update with {lr, 1-beta1^t, 1-beta2^t} read from device memory (float[3]) at run time: the form a HIP-graph
 * capture of the training step uses */
int vf_adam_multi_dev(const void* desc, int ntensors, long total_blocks, const float* scalars, float beta1, float beta2,
                      float eps, void* stream);
/* scalars[0..2] = {lr, bc1, bc2}, enqueued on `stream` (values carried as launch arguments) */
int vf_adam_set_scalars(float* scalars, float lr, float bc1, float bc2, void* stream);

/* ---- gradient exchange next to the path: replaces DistributedDataParallel's bucketed NCCL
 *      all-reduce + optimizer.step(), experiment.py:104-107, 118-120, 292-293 -- a one-shot all-reduce over IPC-mapped
 *      peer gradient arenas fused with the Adam update (csrc/xgmi.hip; host side reducer.XgmiArena, VF_REDUCER=xgmi).
 *      The five memory calls below are the only entry points of the library that allocate / map memory; they run once
 *      per process, outside any stream. ---- */
int vf_xgmi_alloc(void** ptr, long bytes);                 /* zero-filled, IPC-exportable device memory */
int vf_xgmi_free(void* ptr);
int vf_xgmi_export(void* ptr, void* handle64 /*64 bytes out*/);
int vf_xgmi_open(const void* handle64, void** ptr);        /* a peer's allocation mapped into this process */
int vf_xgmi_close(void* ptr);
/* flag[slot][rank] = epoch in every rank's flag block (unsigned [nslots][world]); peer_flags: HOST array of `world`
 * device pointers.  Ordered behind everything `stream` has executed before, released at system scope. */
int vf_xgmi_signal(const void* const* peer_flags, int world, int rank, int slot, unsigned epoch, void* stream);
/* one wave waits until flag[slot][p] >= epoch for every p and slot in [slot_lo, slot_hi) of this rank's block; gives
 * up after timeout_us and stores 1 + slot_lo to *status (device unsigned) instead of hanging the queue */
int vf_xgmi_wait(const void* my_flags, int world, int slot_lo, int slot_hi, unsigned epoch, void* status,
                 long timeout_us, void* stream);
/* g = (arena_0 + ... + arena_{W-1}) / W in rank order -> gavg, then vf_adam_multi_dev's update of this rank's own
 * parameters; desc rows as vf_adam_multi with g = the slot in the LOCAL arena (p == 0: average only);
 * peer_bases: HOST array of `world` arena base pointers (own rank: my_base) */
int vf_xgmi_reduce_adam(const void* desc, int ntensors, long total_blocks, const void* const* peer_bases,
                        const float* my_base, float* gavg_base, int world, const float* scalars, float beta1,
                        float beta2, float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VF_HIP_H */
