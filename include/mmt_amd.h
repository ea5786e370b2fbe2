/*
 * mmt_amd.h -- C ABI of libmmt_amd.so: the MI355X (gfx950) hot path of
 * gnsJhenJie/Multimodal_Transformer (Transformer/model + Transformer/nn_component).
 *
 * The reference is pure Python/PyTorch and has no FFI of its own; its plugin
 * interface is the string-keyed module registry (model/model_registrar.py:22-37)
 * whose entries' forward() methods are what each entry point below replaces.
 * All pointers are DEVICE pointers to contiguous fp32 (unless said otherwise),
 * `stream` is a hipStream_t passed as void*, sizes are element counts.
 * Every function returns 0 on success or a negative MMT_E* code; the message is
 * available from mmt_last_error().  Nothing here allocates device memory: the
 * caller brings outputs and a workspace (size from the *_workspace_floats calls).
 *
 * Weight layout: every nn.Linear weight W[N][K] used by a kernel is first
 * re-laid-out once by mmt_pack_linear into "MFMA streaming order"
 *   packed[slab = n/16][chunk = k/16][lane = 16*q + j][e]  = W[16*slab + j][16*chunk + 4*q + e]
 * (N, K zero-padded to multiples of 16) so that one wave64 `global_load_dwordx4`
 * reads 1 KiB of contiguous weights that feed four v_mfma_f32_16x16x4_f32.
 * Biases / LayerNorm vectors stay plain fp32 arrays padded to a multiple of 16.
 */
#ifndef MMT_AMD_H
#define MMT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMT_MAX_LAYERS 8
#define MMT_MAX_MLP_HIDDEN 3
#define MMT_MAX_CONV 4

#define MMT_OK 0
#define MMT_EINVAL (-1)   /* bad argument / unsupported shape */
#define MMT_ELAUNCH (-2)  /* HIP launch or runtime error     */

/* one packed nn.Linear: y = x W^T + b  (b may be NULL = no bias) */
typedef struct {
  const float *w;   /* packed, mmt_packed_linear_floats(N,K) floats */
  const float *b;   /* N padded to 16, or NULL                      */
  int32_t n, k;     /* logical out / in features                    */
} mmt_linear;

typedef struct {
  const float *g, *b;   /* LayerNorm weight / bias (eps 1e-5) */
} mmt_norm;

/* nn.TransformerEncoderLayer (post-norm, ReLU) -- Transformer_model.py:28-30 */
typedef struct {
  mmt_linear qkv, out, ff1, ff2;
  mmt_norm ln1, ln2;
} mmt_enc_layer;

/* Encoder: Transformer_model.py:11-59 */
typedef struct {
  int32_t d_model, nhead, d_ff, nlayers, n_in;
  mmt_linear input_fc;
  const float *pe;            /* pos_encoder.pe buffer, row stride d_model */
  mmt_enc_layer layer[MMT_MAX_LAYERS];
} mmt_encoder;

/* nn.TransformerDecoderLayer (post-norm, ReLU) -- Transformer_model.py:77-79 */
typedef struct {
  mmt_linear qkv, out;        /* self attention                                   */
  mmt_linear cq, ckv, cout;   /* cross attention: q (rows 0..d), kv (rows d..3d)   */
  mmt_linear ff1, ff2;
  mmt_norm ln1, ln2, ln3;
  /* optional second copies in mmt_pack_lin4_first / _second order (NULL = absent): batches small enough for 4-agent
   * clusters run their linears on the 4x4x1 matrix instruction, which reads this order */
  const float *qkv4, *out4, *cq4, *cout4, *ff14, *ff24;
} mmt_dec_layer;

/* Mlp (MLP.py:6-30), regression form: n_hidden x (Linear, LayerNorm, ReLU) + Linear */
typedef struct {
  int32_t n_hidden;
  mmt_linear hidden[MMT_MAX_MLP_HIDDEN];
  mmt_norm hidden_ln[MMT_MAX_MLP_HIDDEN];
  mmt_linear out;
} mmt_mlp;

/* Trajectory_Decoder + decoder/MLP head (+ optional fusions) -- Transformer_model.py:155-199,
 * mgcvae.py:375-449 */
typedef struct {
  int32_t d_model, nhead, d_ff, nlayers, n_in, n_out;
  mmt_linear input_fc, output_fc;
  const float *pe;
  mmt_dec_layer layer[MMT_MAX_LAYERS];
  mmt_mlp head;               /* VEHICLE/decoder/MLP                                    */
  int32_t has_map_fusion;     /* VEHICLE/fusion/hist_map on cat[h, map_feat]            */
  mmt_mlp map_fusion;
  int32_t has_mem_fusion;     /* VEHICLE/fusion/lane_hist (or mix_fc) on cat[memory, f] */
  mmt_mlp mem_fusion;
} mmt_decoder;

/* CNNMapEncoder: map_encoder.py:6-36 (conv weights plain [Cout][Cin][k][k]) */
typedef struct {
  int32_t nconv, in_ch, in_h, in_w;
  int32_t out_ch[MMT_MAX_CONV], ksize[MMT_MAX_CONV], stride[MMT_MAX_CONV];
  const float *conv_w[MMT_MAX_CONV], *conv_b[MMT_MAX_CONV];
  const float *fc_w, *fc_b;   /* plain row-major [n_out][flat] */
  int32_t n_out;
  int32_t input_affine;       /* 1: the kernel applies map*2-1 (mgcvae.py:290) while reading the raster */
  /* optional: conv weights in MFMA lane order (mmt_pack_conv); with them the reference architecture
   * (3x100x100 -> 10,k5,s2 -> 20,k5,s2 -> 10,k5,s1 -> 1,k3,s1) runs as implicit GEMM on the matrix cores */
  const float *conv_wp[MMT_MAX_CONV];
} mmt_cnn;

/* Lane_Encoder: Lane_encoder.py:7-33 (Conv1d + leaky_relu + Linear) */
typedef struct {
  int32_t in_ch, out_ch, ksize, stride, n_samples, n_out;
  const float *conv_w, *conv_b, *fc_w, *fc_b;   /* plain layouts */
} mmt_lane;

/* vit_pytorch==0.33.2 ViT (third-party; call sites mgcvae.py:145-155,294-296): pre-norm blocks, cls pooling.
 * Sub-matrices are packed separately so that one workgroup can run attention head-group by head-group:
 *   qkv.w : [group][q,k,v] each packed (heads_per_group*dim_head x dim), no bias
 *   out.w : [group] K-slices, each packed (dim x heads_per_group*dim_head); out.b = bias (dim)
 *   patch_fc.w : [panel] K-slices of the (dim x channels*patch^2) weight, each packed (dim x panel_k) */
typedef struct {
  mmt_norm ln1, ln2;
  mmt_linear qkv, out, ff1, ff2;
} mmt_vit_block;

typedef struct {
  int32_t image, patch, channels, dim, depth, heads, dim_head, mlp_dim, n_out, heads_per_group, panel_k;
  int32_t input_affine;                      /* 1: apply map*2-1 while gathering patches (mgcvae.py:296) */
  mmt_linear patch_fc;
  const float *cls_token, *pos_embedding;    /* (dim), (n_patches+1, dim) */
  mmt_vit_block block[MMT_MAX_LAYERS];
  mmt_norm head_ln;
  mmt_linear head_fc;
} mmt_vit;

/* ---- training-step building blocks (train.py:218-256, mgcvae.py:559-627) ------------------------ */
/* C[m][n] (+)= epi( sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] ) ;  epi: +bias[n], +addend[m*ld_add + n], relu,
 * zero where mask[m*ld_mask + n] <= 0.  One kernel serves X W^T (forward), dY W (input grads), dY^T X (weight grads). */
typedef struct {
  const float *A;
  int64_t sam, sak;
  const float *B;
  int64_t sbk, sbn;
  float *C;
  int64_t ldc;
  int32_t M, N, K;
  const float *bias, *addend;
  int64_t ld_add;
  const float *mask;
  int64_t ld_mask;
  int32_t relu, accumulate;
  /* optional LayerNorm backward on the A operand (nn.LayerNorm over its K columns, A unit-stride in k, M <= 256):
   * A'[m][k] = rstd[m] * (A[m][k] g[k] - mean_k(A g) - xhat[m][k] mean_k(A g xhat)) is what gets multiplied, and is
   * also written to ln_dx (row stride ld_ln, like ln_xhat) when ln_dx != NULL.  ln_xhat == NULL: plain A. */
  const float *ln_xhat, *ln_rstd, *ln_g;
  float *ln_dx;
  int64_t ld_ln;
  /* optional split of the K range over `splitk` workgroup slices (weight gradients: K = ph*B rows, both operands with
   * unit stride along their rows): slice z writes its partial product to C + z*part_stride (and asum + z*part_stride);
   * the caller sums the slices.  asum != NULL: asum[m] = sum_k A[m][k] (the bias gradient) from the same pass.
   * splitk <= 1 and asum == NULL: plain GEMM.  Not combined with bias/addend/relu/mask/accumulate. */
  int32_t splitk, pad_;
  int64_t part_stride;
  float *asum;
  /* dropout (training; nn.Dropout / F.dropout, Transformer_model.py:28-30,77-79,126-129): `drop` holds one keep byte per
   * output (0 = dropped, row stride ld_drop bytes, written by mmt_dropout_mask); kept values are multiplied by drop_scale =
   * 1/(1-p).  With v = acc + bias:  drop_after_add ? keep*(v + addend) : keep*[relu](v) + addend.   drop == NULL: no dropout.
   * mask_scale != 0: outputs that pass `mask` are also multiplied by it (backward of relu followed by dropout). */
  const uint8_t *drop;
  int64_t ld_drop;
  float drop_scale;
  int32_t drop_after_add;
  float mask_scale;
  int32_t pad2_;
  /* with ln_xhat: the multiplied operand is keep*A' (the gradient entering the dropped sub-layer), also written to ln_dxm
   * (row stride ld_ln) for the deferred weight gradients; ln_dx still receives A' (the residual path).  NULL: none. */
  const uint8_t *ln_drop;
  int64_t ld_ln_drop;
  float *ln_dxm;
  float ln_drop_scale;
  int32_t pad3_;
} mmt_gemm_args;

/* short-sequence multi-head attention with explicit (batch stride, row stride) per operand, in floats */
typedef struct {
  const float *q, *k, *v;
  float *o, *probs;                 /* probs stash, zero for masked keys: row (b,h,i) at (b*H+h)*p_hs + i*p_rs, Lk floats */
  const uint8_t *pad;               /* (B,Lk) key padding or NULL */
  int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;
  int64_t p_hs, p_rs;               /* probs strides; contiguous (B,H,Lq,Lk) is p_hs = Lq*Lk, p_rs = Lk */
  int32_t B, H, Lq, Lk, dh, causal, causal_offset;
  float scale;
  /* dropout on the attention probabilities (nn.MultiheadAttention(dropout=p) in training): keep bytes addressed like
   * `probs`; o = (keep*probs) v, the stash keeps the undropped probabilities.  NULL: none. */
  const uint8_t *pdrop;
  float pdrop_scale;
  int32_t pad_;
} mmt_attn_args;

typedef struct {
  float *param;
  const float *grad;
  float *exp_avg, *exp_avg_sq;
  int64_t numel;
  float lr_scale;                   /* per-group multiplier on state[1] (map_encoder group: 2.0, train.py:218-219) */
  int32_t pad_;
} mmt_adamw_tensor;

/* one job of mmt_ln_grad_group: the arguments of mmt_ln_param_grads */
typedef struct {
  const float *dy, *xhat;
  float *gw, *gb;
  int64_t ld, part_stride;
  int32_t M, N, splitk, pad_;
} mmt_ln_grad_job;

const char *mmt_last_error(void);
int mmt_version(void);

/* ---- weight re-layout ------------------------------------------------------------------- */
size_t mmt_packed_linear_floats(int n, int k);
/* W row-major [n][ldw]; columns [col0, col0+k) are packed (lets callers split in_proj / cat inputs) */
int mmt_pack_linear(const float *w, int n, int k, int ldw, int col0, float *dst, void *stream);
/* Copies of the decoder's linears for the 4-agent-cluster kernels (v_mfma_f32_4x4x1), laid out in the order those
 * kernels' load instructions run: [slice][part][k-quad step][wave][lane][4].  "first" = a linear whose output rows are
 * split over `nslices` workgroups (in_proj: nparts 3, part_rows d, slice_rows d/nhead; cross q: nparts 1; linear1:
 * slice_rows d_ff/nhead), w = first row of the (nparts*part_rows or nslices*slice_rows) x k matrix, row stride ldw.
 * "second" = a linear whose input columns are split (out_proj, linear2): n x (nslices*slice_k).
 * *_floats returns 0 when the geometry is not one those kernels handle (the decoder then uses the 16-row kernels).
 * No reference counterpart: nn.Linear keeps [n][k]; mmt_dec_layer.*4 point at such copies. */
size_t mmt_lin4_first_floats(int nparts, int slice_rows, int k, int nslices);
int mmt_pack_lin4_first(const float *w, int ldw, int nparts, int part_rows, int slice_rows, int k, int nslices, float *dst,
                        void *stream);
size_t mmt_lin4_second_floats(int n, int slice_k, int nslices);
int mmt_pack_lin4_second(const float *w, int ldw, int n, int slice_k, int nslices, float *dst, void *stream);
/* All re-layouts of a module in ONE launch (a training step refreshes ~200 packed tensors after every optimizer update):
 * a device-resident table of jobs, kind 0 = copy `n` floats, 1 = mmt_pack_linear(src, n, k, ldw, col0), 2 =
 * mmt_pack_lin4_first(src, ldw, nparts, part_rows, slice_rows, k, nslices), 3 = mmt_pack_lin4_second(src, ldw, n,
 * slice_rows (= slice_k), nslices).  `total` = floats written to dst.  No reference counterpart. */
typedef struct {
  const float *src;
  float *dst;
  int64_t total;
  int32_t kind, n, k, ldw, col0, nparts, part_rows, slice_rows, nslices, tr;   /* tr (kinds 2, 3): the matrix is the transpose of src */
} mmt_pack_job;
int mmt_pack_jobs(const mmt_pack_job *table_dev, int njobs, void *stream);

/* ---- Encoder.forward  (Transformer_model.py:45-59; masks of model_utils.py:141-151 built in) --
 * x_st (B,T,n_in) may hold NaN padding; memory (B,T,d) out; pad_mask (B,T) uint8 out (may be NULL). */
int mmt_encoder_forward(const mmt_encoder *enc, const float *x_st, int B, int T, float *memory,
                        uint8_t *pad_mask, void *stream);

/* The same with a workspace (mmt_encoder_workspace_floats floats, ZEROED ONCE by the caller before its first use and again
 * after a call that mmt_decoder_status reported as timed out; never shared by calls that overlap in time).  With it, batches
 * small enough that two workgroups per agent are all resident (2 * ceil(B/8)*8 <= #CUs) split every layer over two
 * workgroups that exchange partial sums through the workspace (Transformer_model.py:45-59 is unchanged mathematically; the K
 * range of the out-projection and linear2 is summed in two halves).  Other batches / geometries: as mmt_encoder_forward.
 * workspace == NULL: mmt_encoder_forward.  No counterpart in the reference. */
size_t mmt_encoder_workspace_floats(const mmt_encoder *enc, int B, int T);
int mmt_encoder_forward_ws(const mmt_encoder *enc, const float *x_st, int B, int T, float *memory,
                           uint8_t *pad_mask, float *workspace, void *stream);

/* ---- autoregressive decode  (mgcvae.py:375-449 at_dec, basic/map branch, KV-cached) -------------
 * memory (B,T,d); pad_mask (B,T) uint8 or NULL; start_pos (B,n_in) standardised; map_feat (B,m) or NULL;
 * mem_feat (B,mf) or NULL (lane feature repeated over T, mgcvae.py:404-407);
 * pred_st (B,ph,n_in) standardised positions out; h_out (B,ph,n_out) or NULL;
 * tgt: NULL = feed predictions back (at_dec); else (B,ph,n_in) teacher-forced tokens
 *      (Trajectory_Decoder.forward, Transformer_model.py:168-199; h_out is then the result).
 * workspace: mmt_decoder_workspace_floats(...) floats. */
size_t mmt_decoder_workspace_floats(const mmt_decoder *dec, int B, int T, int ph);
/* Two implementations sit behind mmt_decoder_forward (tuning knob, no counterpart in the reference):
 *   PERSISTENT: one launch, every workgroup owns <= 16 agents and streams all decoder weights each step;
 *   CLUSTER   : RC <= 16 agents x nhead workgroups share one sub-layer (weights split by head / FFN slice, partial
 *               sums exchanged through memory).  With 4-agent clusters that are all resident at once
 *               (ceil(B/4)*nhead <= #CUs) the 216 sub-layers of a decode run in ONE launch whose clusters hand their
 *               partials over in-kernel (write-through stores, an arrival counter per cluster, bounded waits: a
 *               cluster that cannot complete a wait within 0.2 s makes the call return NaN predictions);
 *               otherwise 3 short launches per layer per step (the kernel boundary is the exchange).
 *   CLUSTER_LAUNCHES: AUTO's choice, but never the one-launch form of CLUSTER (also: environment MMT_NO_FUSED_DECODER=1).
 * AUTO (default) picks CLUSTER whenever the geometry allows it (d <= 256, head dim 16/32, d_ff/nhead <= 64).
 * The one-launch form assumes it has the GPU's CUs to itself: decodes issued concurrently on several streams still
 * complete (workgroups are dispatched in order), a GPU shared with other long-running kernels may time out. */
#define MMT_DECODER_AUTO 0
#define MMT_DECODER_PERSISTENT 1
#define MMT_DECODER_CLUSTER 2
#define MMT_DECODER_CLUSTER_LAUNCHES 3
#define MMT_DECODER_CLUSTER_FUSED 4      /* CLUSTER, one-launch forms: only the fused kernel (every workgroup runs every phase)    */
#define MMT_DECODER_CLUSTER_PIPELINE 5   /* CLUSTER, one-launch forms: only the weight-stationary pipeline (a workgroup = a stage) */
int mmt_decoder_select(int impl);
/* Which kernel family the calling thread's last mmt_decoder_forward actually ran (benchmarks label their lines with it). */
#define MMT_PATH_NONE 0
#define MMT_PATH_PERSISTENT 1          /* mmt_decoder_kernel                                                   */
#define MMT_PATH_CLUSTER_LAUNCHES16 2  /* mmt_dec_phase_kernel per phase (8- / 16-agent clusters, 16x16x4 MFMA) */
#define MMT_PATH_CLUSTER_LAUNCHES4 3   /* mmt_dec_phase4_kernel per phase (4-agent clusters, 4x4x1 MFMA)        */
#define MMT_PATH_CLUSTER_FUSED 4       /* mmt_dec_fused4_kernel: every phase in one launch                      */
#define MMT_PATH_CLUSTER_PIPELINE 5    /* mmt_dec_pipe4_kernel: weight-stationary pipeline, one launch          */
int mmt_decoder_last_path(void);
/* Synchronises `stream` and reports whether a one-launch decode issued on the current device gave up on a wait since the
 * last call: MMT_OK, or MMT_ELAUNCH (that decode's outputs are NaN; the fault is cleared and the one-launch form stays
 * disabled on this device, so re-issuing the call runs launch-per-phase).  Callers check it where they synchronise anyway
 * (the reference copies every prediction to the host, trajectron.py:216-221).  No counterpart in the reference. */
int mmt_decoder_status(void *stream);
/* test hook: bit 0 - the next one-launch decodes lose one workgroup (their waits must time out); bit 1 - the next one-launch
 * backward sweeps (mmt_sweep_pipe_run) do; 0 re-arms.  Not for production. */
int mmt_debug_decoder_stall(int on);
int mmt_decoder_forward(const mmt_decoder *dec, const float *memory, const uint8_t *pad_mask,
                        const float *start_pos, const float *map_feat, int map_dim,
                        const float *mem_feat, int mem_feat_dim, const float *tgt, int B, int T, int ph,
                        float *pred_st, float *h_out, float *workspace, void *stream);

/* The same decode for callers that cannot allocate device memory or bring a workspace (code under HIP-graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  NO counterpart in the reference.  The workspace is a file-scope device
 * array of MMT_DECODER_STATIC_SCRATCH_FLOATS floats (44 MiB per GPU the library runs on, held until the process ends); sizes
 * whose mmt_decoder_workspace_floats() exceeds it are rejected with MMT_EINVAL.  Calls no hipMalloc / hipFree.  Every call on
 * a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race - serialise them.  Results are
 * bit-identical to mmt_decoder_forward.
 * Under HIP-graph capture: the FIRST call on a device (and the first call of each decoder form / shape class on it) makes
 * runtime queries that are not stream operations (hipGetSymbolAddress, hipFuncSetAttribute, the occupancy query) and must
 * therefore happen OUTSIDE a capture - call it once eagerly with the shapes that will be captured, then capture; later calls
 * only enqueue kernels and a memset on `stream` (tests/test_zz_no_alloc.py captures and replays one). */
#define MMT_DECODER_STATIC_SCRATCH_FLOATS (11u << 20)
int mmt_decoder_forward_static(const mmt_decoder *dec, const float *memory, const uint8_t *pad_mask,
                               const float *start_pos, const float *map_feat, int map_dim,
                               const float *mem_feat, int mem_feat_dim, const float *tgt, int B, int T, int ph,
                               float *pred_st, float *h_out, void *stream);

/* conv weight [Co][Ci][ks][ks] -> packed[tap][ci/4][co/16][lane = 16*(ci%4) + co%16], zero padded (new, no counterpart) */
size_t mmt_packed_conv_floats(int Co, int Ci, int ks);
int mmt_pack_conv(const float *w, int Co, int Ci, int ks, float *dst, void *stream);

/* ---- CNNMapEncoder.forward on map*2-1 (map_encoder.py:31-36, mgcvae.py:288-290) ------------------ */
size_t mmt_cnn_workspace_floats(const mmt_cnn *cnn, int B);
int mmt_cnn_forward(const mmt_cnn *cnn, const float *maps, int B, float *out, float *workspace, void *stream);

/* ---- the autoregressive decode of a TRAINING step (mgcvae.py:375-449 with the modules in training mode): the reference's
 * nn.Dropout / attention-probability dropouts of Trajectory_Decoder (Transformer_model.py:77-79,126-129) applied from keep
 * bytes the caller drew (mmt_dropout_mask), so that the positions fed back are the ones the training step's stash pass and
 * backward see.  Sites are laid out like the training step's stash: `pe`, `drop1/2/3[l]` (ph, B, d) step-major, `ff[l]`
 * (ph, B, d_ff), `attn1[l]` (B, H, ph, ph), `attn2[l]` (B, H, ph, T); a NULL site is not dropped; *_scale = 1/(1-p).
 * Runs on the 4-agent-cluster kernels only (mmt_decoder_dropout_supported: head dimension 32, lin4 weight copies present, batch
 * small enough for 4-agent clusters -- B <= 256 on 256 CUs): the one-launch pipeline's DROP instantiation where mmt_decoder_forward
 * would take the pipeline, else the launch-per-phase kernels' (bit-identical to each other; the keep bytes of a visit are requested
 * before its poll); otherwise MMT_EINVAL, and the caller decodes step by step on the training kernels (training.py).  No
 * teacher-forced form: with given tokens the stash pass is the decode. */
typedef struct {
  const uint8_t *pe;
  const uint8_t *attn1[MMT_MAX_LAYERS], *drop1[MMT_MAX_LAYERS], *attn2[MMT_MAX_LAYERS], *drop2[MMT_MAX_LAYERS],
      *ff[MMT_MAX_LAYERS], *drop3[MMT_MAX_LAYERS];
  float pe_scale;
  float attn1_scale[MMT_MAX_LAYERS], drop1_scale[MMT_MAX_LAYERS], attn2_scale[MMT_MAX_LAYERS], drop2_scale[MMT_MAX_LAYERS],
      ff_scale[MMT_MAX_LAYERS], drop3_scale[MMT_MAX_LAYERS];
} mmt_decoder_dropout;
int mmt_decoder_dropout_supported(const mmt_decoder *dec, int B, int T, int ph);
int mmt_decoder_forward_dropout(const mmt_decoder *dec, const float *memory, const uint8_t *pad_mask, const float *start_pos,
                                const float *map_feat, int map_dim, const float *mem_feat, int mem_feat_dim, int B, int T,
                                int ph, float *pred_st, float *workspace, const mmt_decoder_dropout *drop, void *stream);

/* ---- the training step's decode that also keeps what the backward reads (round 3) ---------------------------------------------
 * The autoregressive decode of mmt_decoder_forward[_dropout] on the weight-stationary pipeline kernel, whose stages ALSO write
 * every decoder activation the backward pass reads - each value is in a stage's registers or LDS at some point of the decode -
 * so that the training step needs no second, all-steps forward pass over the decoder (mgcvae.py:375-449 run once, as the
 * reference's autograd tape does).  `stash` is ONE buffer of mmt_decoder_stash_layout.total_floats floats: nlayers blocks of
 * layer_floats (offsets x .. p2 inside a block) and a tail block at `tail` (offsets t_*).  Rows are step-major (row s * B + b):
 *   x      (R, d)    input of the layer (layer 0: embedded token + positional encoding, after its dropout)
 *   qkv    (R, 3d)   self-attention in_proj output   p1 (B, H, ph, ph) its softmax (keys 0..s of query s; before dropout)
 *   a1     (R, d)    self-attention output           xhat1 / rstd1 / x1: norm1's normalised rows, 1/std, output
 *   cq     (R, d)    cross-attention query           p2 (B, H, ph, T)   a2 (R, d)   xhat2 / rstd2 / x2: norm2
 *   f1     (R, d_ff) relu(linear1) after its dropout xhat3 / rstd3: norm3 (its output is the next block's x, or t_x)
 *   t_x (R, d) output of the last layer, t_ho (R, ho_ld) output_fc [map configurations: cat[output_fc, encoded map], ho_ld =
 *   n_out + map_dim, and t_u (R, n_out) = fusion/hist_map of it], t_hu / t_hxhat (R, 32) / t_hrstd (R) decoder/MLP's hidden
 *   layer (ReLU output, normalised rows, 1/std)                                      with R = ph * B.
 * Values are those the decode itself used (hand-off rows carry their 2 tag bits).  Supported where the pipeline runs
 * (mmt_decoder_stash_supported: d = 256, 8 heads, d_ff = 512, n_out = 32, decoder/MLP with one 32-wide hidden layer,
 * fusion/hist_map absent or a single Linear of <= 64 inputs, B <= 256, T, ph <= 16, pipeline resident and not disabled on the device); otherwise MMT_EINVAL and the caller
 * runs the all-steps pass (training.TrainEngine._decoder_steps).  `drop` may be NULL.  No counterpart in the reference. */
typedef struct {
  size_t layer_floats, tail, total_floats;
  size_t x, qkv, a1, xhat1, x1, cq, a2, xhat2, x2, xhat3, f1, rstd1, rstd2, rstd3, p1, p2;
  size_t t_x, t_ho, t_hu, t_hxhat, t_hrstd;
  size_t t_u, ho_ld;      /* map configurations: fusion/hist_map's output (R, n_out); row width of t_ho (n_out, or n_out + map_dim) */
} mmt_decoder_stash_layout;
int mmt_decoder_stash_supported(const mmt_decoder *dec, int map_dim, int B, int T, int ph, int with_dropout);
int mmt_decoder_stash_layout_of(const mmt_decoder *dec, int map_dim, int B, int T, int ph, mmt_decoder_stash_layout *out);
int mmt_decoder_forward_stash(const mmt_decoder *dec, const float *memory, const uint8_t *pad_mask, const float *start_pos,
                              const float *map_feat, int map_dim, const float *mem_feat, int mem_feat_dim, int B, int T, int ph,
                              float *pred_st, float *workspace, const mmt_decoder_dropout *drop, float *stash, size_t stash_floats,
                              void *stream);

/* ---- reverse-time sweep of the training step as ONE weight-stationary pipeline launch (round 3) -----------------------------
 * Replaces the per-(step, layer) kernel chain of the decoder backward (the autograd of train.py:248 `loss.backward()` through
 * mgcvae.py:432-446's position feedback; training.TrainEngine._sweep here): same inputs (the forward stash), same outputs (every
 * gradient buffer the deferred weight-gradient GEMMs read), with or without the training-mode keep masks, autoregressive decode,
 * decoder/MLP head with one hidden layer (Basic / Lane configurations).  No counterpart in the reference (which calls autograd).
 * Stage kinds index the [3] arrays: 0 = feed forward (LayerNorm3), 1 = cross attention (LayerNorm2), 2 = self attention
 * (LayerNorm1).  w1 / w2 / w2k / w2v / w_ofc are 4x4x1-order packs (mmt_pack_jobs kinds 2 / 3 with tr = 1: the sweep multiplies
 * by W where the forward multiplies by W^T):
 *   w1[0] = first(linear2.weight^T, slices of d_ff/H rows)   w2[0] = second(linear1.weight^T, K slices of d_ff/H)
 *   w1[1] = first(cross out_proj^T, slices of dh rows)       w2[1] = second(cross in_proj_weight[:d]^T, K slices of dh)
 *   w1[2] = first(self out_proj^T, slices of dh rows)        w2[2] / w2k / w2v = second(self in_proj_weight[q | k | v rows]^T)
 *   w_ofc = second(output_fc.weight^T, one K slice of n_out)
 * Stash layouts are the training step's: step-major (ph, B, .) rows s*B + b; p1 (B,H,ph,ph), p2 (B,H,ph,T); mkv / gmkv (B*T, 2d).
 * gqkv's k | v columns and gmkv must be ZERO on entry (they accumulate); G is (ph + 1, B, 2) with dL/dpos[t] in row t = 1..ph on
 * entry and the total gradient at pos[t] on return.  A sweep whose hand-offs time out (0.2 s) writes NaN to *poison, raises the
 * fault that mmt_decoder_status reports, and leaves the gradient buffers undefined. */
typedef struct {
  const float *w1[3], *w2[3], *w2k, *w2v;
  const float *gamma[3];                  /* norm3 / norm2 / norm1 .weight */
  const float *xhat[3], *rstd[3];         /* stash of those LayerNorms: (ph,B,d), (ph,B) */
  float *gdy[3], *gdx[3];                 /* out: gradient at each LayerNorm's output / input, (ph,B,d) */
  const float *f1;                        /* relu(linear1) hidden, (ph,B,d_ff) */
  float *gf1;                             /* out: its gradient */
  const float *cq, *mkv, *p2;             /* cross attention: queries (ph,B,d), memory K|V (B*T,2d), probabilities */
  float *gcq, *gmkv;
  const float *qkv, *p1;                  /* self attention: q|k|v (ph,B,3d), probabilities */
  float *gqkv;
  /* training-mode dropout (NULL: that site is not dropped), keep bytes laid out like the tensors they mask: k_ln[kind] = dropout3 /
   * dropout2 / dropout1 (ph,B,d); k_ff = the feed-forward hidden units (ph,B,d_ff; the f1 stash is already dropped);
   * k_att[0] = cross-attention probabilities (B,H,ph,T), k_att[1] = self-attention probabilities (B,H,ph,ph); s_* = 1/(1-p) */
  const uint8_t *k_ln[3], *k_ff, *k_att[2];
  float s_ln[3], s_ff, s_att[2];
} mmt_sweep_layer;
typedef struct {
  int32_t B, T, ph, d, H, nlayers, d_ff, n_out, n_pos, pad_;
  mmt_sweep_layer layer[MMT_MAX_LAYERS];
  float *G;                               /* (ph+1, B, 2) */
  float *gx0;                             /* out: gradient at the embedded tokens, (ph,B,d) */
  const float *w_in;                      /* decoder input_fc.weight (d, 2), natural layout */
  const float *w_hout, *h_gamma, *w_h0;   /* decoder/MLP: out Linear weight (2, n_out), LayerNorm weight, hidden Linear weight (n_out, n_out) */
  const float *w_ofc;
  const float *h_hu, *h_xhat, *h_rstd;    /* head tape: relu(LN(.)) (ph,B,n_out), xhat, rstd (ph,B) */
  float *h_ghum, *h_ghz, *h_gho;          /* out: masked dy of the head's LayerNorm, its dx, gradient at output_fc's output */
  float *poison;                          /* 1 float: NaN when the sweep gave up (the step's loss / optimizer guard), may be NULL */
  const uint8_t *k_pe;                    /* PositionalEncoding.dropout keep bytes (ph,B,d) or NULL */
  float s_pe, pad2_;
  /* Map configurations (mgcvae.py:438-443): decoder/MLP's input is fusion/hist_map(cat[output_fc(x), encoded_map]), here a single
   * Linear (n_out + map_dim -> n_out; fusion_hist_map_layer = 1).  w_fuse = its weight (n_out, n_out + map_dim), natural layout, or
   * NULL (no fusion).  With it: h_gu receives the gradient at the fusion's output (ph,B,n_out) and h_gho the gradient at the
   * whole concatenation, rows gho_ld = n_out + map_dim wide (its first n_out columns are the gradient at output_fc's output,
   * the rest the per-step gradient at the encoded map); without it gho_ld = n_out and h_gu is not written. */
  const float *w_fuse;
  float *h_gu;
  int gho_ld, map_dim;
} mmt_sweep_pipe;
int mmt_sweep_pipe_supported(const mmt_sweep_pipe *args);
size_t mmt_sweep_pipe_workspace_floats(const mmt_sweep_pipe *args);
int mmt_sweep_pipe_run(const mmt_sweep_pipe *args, float *workspace, void *stream);

/* ---- ViT.forward (vit_pytorch 0.33.2 architecture; parity unpinned): maps (B,C,S,S) -> (B,n_out) ------------ */
int mmt_vit_forward(const mmt_vit *vit, const float *maps, int B, float *out, void *stream);

/* ---- Lane_Encoder.forward on the raw (B*L, 2, S) view (Lane_encoder.py:29-33, mgcvae.py:301-306) -- */
int mmt_lane_forward(const mmt_lane *lane, const float *lanes, int n_rows, float *out, void *stream);

/* ---- un-standardise + ADE/FDE + L2 loss (mgcvae.py:614-620,708-715,772; evaluation.py:10-19) -------
 * pred_st (B,ph,2); node_pos (B,2); pred_m (B,ph,2) out = pred_st*80 + node_pos (may be NULL);
 * gt (B,ph,2) in the units of `metric_in_metres` (1: metres vs pred_m, 0: standardised vs pred_st)
 * or NULL; ade/fde (B) out or NULL; loss_sum (1) accumulates sum_b mean_t ||.||2 (caller zeroes). */
int mmt_finalize(const float *pred_st, const float *node_pos, const float *gt, int metric_in_metres,
                 int B, int ph, float *pred_m, float *ade, float *fde, float *loss_sum, void *stream);

/* ---- training ops: all whole-batch, asynchronous on `stream`, graph-capturable ------------------------------------ */
int mmt_gemm(const mmt_gemm_args *g, void *stream);
/* out[n] (+)= sum_m a[m*lda+n] * (b ? b[m*ldb+n] : 1), rows with mask[m*ldm+n] <= 0 skipped: bias / LayerNorm grads */
/* dO = A' @ B as mmt_gemm (optional fused LayerNorm backward, no epilogue, C ignored) handed straight to the backward of an
 * attention with ONE query per batch element (the decoder's per-step sweep: out_proj's input gradient feeding the attention
 * backward): dq (=), dk / dv (+=) as mmt_attention_bwd with d_o = dO, which never goes to memory.  M = a->B <= 256, N = H*32. */
int mmt_gemm_attention_bwd(const mmt_gemm_args *g, const mmt_attn_args *a, float *dq, long dq_bs, float *dk, float *dv,
                           long dk_bs, long dk_rs, void *stream);
/* Convolutions of the map CNN as GEMMs for the training path (CNNMapEncoder, nn_component/map_encoder.py:6-36; autograd of
 * F.leaky_relu(conv(x), 0.2) and of flatten + fc).  Activations are rows (b, y, x) x C columns; patch columns are ordered
 * (c, ky, kx) like a Conv2d weight.  im2col: cols[(b,oy,ox)][(c,ky,kx)] = leaky(alpha*in + beta, slope), `in` NCHW (nhwc = 0) or
 * rows-by-channels with row stride ld_in (nhwc = 1).  col2im: the transposed gather, din[(b,y,x)][c] (C columns, contiguous)
 * = sum of the patch entries that read the pixel, times the activation derivative at z (same layout as din; NULL: none).
 * ld_cols / ld_dcols / ld_din: row strides (>= the logical widths; the training path pads them to multiples of 4). */
int mmt_im2col(const float *in, int B, int C, int H, int W, int kh, int kw, int stride, int nhwc, long ld_in, float alpha,
               float beta, float slope, float *cols, long ld_cols, void *stream);
int mmt_col2im(const float *dcols, long ld_dcols, int B, int C, int H, int W, int kh, int kw, int stride, const float *z,
               float slope, float *din, long ld_din, void *stream);
/* nn.LayerNorm parameter gradients: gw[n] = sum_m dy[m][n]*xhat[m][n], gb[n] = sum_m dy[m][n] (row stride ld for both);
 * rows dealt over `splitk` slices, slice z writing to gw/gb + z*part_stride (the caller sums the slices) */
int mmt_ln_param_grads(const float *dy, const float *xhat, int M, int N, long ld, float *gw, float *gb, int splitk,
                       long part_stride, void *stream);
int mmt_colsum(const float *a, const float *b, const float *mask, int M, int N, long lda, long ldb, long ldm,
               float *out, int accumulate, void *stream);
/* Group launches (no reference counterpart: torch's autograd issues one kernel per parameter gradient, train.py:244-246): the
 * deferred weight gradients of a step are ~70 independent split-K GEMMs (each an mmt_gemm call with splitk >= 1 / asum) and ~30
 * LayerNorm parameter reductions over the same ph*B rows, each far too small to fill 256 CUs; ONE launch runs all tiles of all
 * jobs, tile for tile the arithmetic of the single-job launch (bit-identical results).  *_plan validates the jobs as the
 * single-job entry point would and writes the launch table [njobs jobs][nblocks {job, bx, by, bz}] into host memory
 * (table_host == NULL: sizes only); the caller copies it to 16-byte aligned device memory once and launches it any number of
 * times (graph-capturable; the jobs' pointers are baked into the table). */
int mmt_gemm_group_plan(const mmt_gemm_args *jobs, int njobs, void *table_host, size_t capacity, size_t *bytes, int *nblocks);
int mmt_gemm_group(const void *table_dev, int njobs, int nblocks, void *stream);
int mmt_ln_grad_group_plan(const mmt_ln_grad_job *jobs, int njobs, void *table_host, size_t capacity, size_t *bytes, int *nblocks);
int mmt_ln_grad_group(const void *table_dev, int njobs, int nblocks, void *stream);
/* y = [relu](LayerNorm(x + res)) with eps 1e-5; stashes xhat (M,N), rstd (M)   (nn.LayerNorm; MLP.py:13-16) */
int mmt_layernorm_fwd(const float *x, const float *res, int M, int N, const float *g, const float *b, int relu,
                      float *y, float *xhat, float *rstd, void *stream);
/* drop != NULL (keep bytes (M,N), not with accumulate): dx_dropped = keep * drop_scale * dx, the gradient entering a
 * sub-layer whose output went through dropout before the residual add */
int mmt_layernorm_bwd(const float *dy, const float *xhat, const float *rstd, const float *g, const float *y_relu,
                      int M, int N, float *dx, int accumulate, float *dy_masked, const uint8_t *drop, float drop_scale,
                      float *dx_dropped, void *stream);
/* ---- dropout for the training step (no counterpart call in the reference: torch draws its masks from the global generator;
 * here every mask of a step is generated up front from a counter-based generator so that the forward stash pass and the
 * backward see the same bits).  keep[i] = Philox4x32-10(key = rng_state[0], counter = (i/4, stream_id, rng_state[1]))[i%4]
 * >= p*2^32 for i < n (n a multiple of 4, keep 4-byte aligned); rng_state = {seed, step} lives on the device so that a
 * captured graph draws fresh masks on every replay: mmt_rng_advance adds 1 to the step. */
int mmt_dropout_mask(uint8_t *keep, long n, float p, const uint64_t *rng_state, uint32_t stream_id, void *stream);
int mmt_rng_advance(uint64_t *rng_state, void *stream);
/* y = keep ? x*scale : 0 (y may alias x): F.dropout(encoded_map, 0.5) (mgcvae.py:291-293) and the ViT's dropouts */
int mmt_dropout_apply(const float *x, const uint8_t *keep, float scale, float *y, long n, void *stream);
int mmt_attention_fwd(const mmt_attn_args *a, void *stream);
/* dq (=), dk += , dv += ; dk/dv share strides (dk_bs, dk_rs) */
int mmt_attention_bwd(const mmt_attn_args *a, const float *d_o, long do_bs, long do_rs, float *dq, long dq_bs,
                      long dq_rs, float *dk, float *dv, long dk_bs, long dk_rs, void *stream);
/* pos [ph][B][2], y [B][ph][2]: dpos = d/dpos mean_b mean_t ||pos-y||_2; loss_sum (1) += that mean (may be NULL) */
int mmt_l2_loss_bwd(const float *pos, const float *y, int B, int ph, float *dpos, float *loss_sum, void *stream);
/* clip_grad_value_(clip) + torch.optim.AdamW over a device-resident table; state_dev = {step, lr} floats on the device.
 * guard != NULL: a device float (the step's loss, summed over the data-parallel ranks with the gradients); when it is not
 * finite the whole update is skipped -- a decode that gave up (NaN positions) must not reach the weights */
int mmt_adamw_multi(const mmt_adamw_tensor *table_dev, int ntensors, const float *state_dev, float beta1, float beta2,
                    float eps, float weight_decay, float clip, float grad_scale, const float *guard, void *stream);

/* x0 = nan_to_num(x_st, nan=0) (Transformer_model.py:47); pad[row] = isnan(x_st[row][1]) (model_utils.py:148-151) */
int mmt_history_prepare(const float *x_st, int rows, int n_in, float *x0, uint8_t *pad, void *stream);
/* state[0] += 1 (step count), state[1] *= gamma: ExponentialLR stepped once per batch (train.py:226-227,253-256) */
/* guard as in mmt_adamw_multi: a skipped step leaves state alone and adds 1 to skipped[0] (may be NULL) */
int mmt_lr_step(float *state_dev, float gamma, const float *guard, float *skipped, void *stream);

/* ---- stand-alone Mlp / teacher-forced pieces -----------------------------------------------------------------
 * softmax over the middle axis of a contiguous (outer, n, inner) tensor: nn.Softmax() of Mlp(mode='classification')
 * (MLP.py:18-21; dim=None resolves to dim 0 of the 3-D input it receives: outer=1, n=B, inner=ph*out) */
int mmt_softmax_axis(const float *x, long outer, int n, long inner, float *y, void *stream);
/* torch.cumsum over the middle axis: mlp(h).cumsum(dim=-2) of non_at_dec (mgcvae.py:352,370) */
int mmt_cumsum_axis(const float *x, long outer, int n, long inner, float *y, void *stream);
/* L2_norm (model_utils.py:154-160): out[r] = ||a[r] - b[rb]||_2 over the last axis of length n; b is broadcast over a middle axis
 * of length rep (a: (outer, rep, inner, n), b: (outer, inner, n)); rep = 1 for equal shapes */
int mmt_l2_norm(const float *a, const float *b, long rows, int n, long rep, long inner, float *out, void *stream);
/* y += x (gradient of that cumsum walked backwards in time) */
int mmt_add_inplace(float *y, const float *x, long n, void *stream);

/* ---- ViT map encoder, training pieces (third-party vit_pytorch==0.33.2, call sites mgcvae.py:17,145-155,294-296) ----
 * patch rearrange 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' of alpha*img+beta: img (B,C,S,S) -> out (B*(S/P)^2 rows, ld floats apart) */
int mmt_patchify(const float *img, int B, int C, int S, int P, float alpha, float beta, float *out, long ld, void *stream);
/* nn.GELU (exact, erf form) and its derivative: dz = dy * gelu'(z) */
int mmt_gelu_fwd(const float *z, float *y, long n, void *stream);
int mmt_gelu_bwd(const float *z, const float *dy, float *dz, long n, void *stream);
/* dst[b][r][0..N) (+)= src[b][r][0..N) for views with batch / row strides in floats (a source stride of 0 broadcasts):
 * cls / position token assembly, cls pooling and their gradients */
int mmt_copy3(const float *src, long sbs, long srs, float *dst, long dbs, long drs, int B, int R, int N, int accumulate,
              void *stream);

/* ---- GMM2D (Transformer/model/components/gmm2d.py:7-217; dead code in the live path, named by the north star) ----
 * R = product of the leading dims, N = components (<= 64); mus / log_sigmas / sigmas (R,N,2), corrs & co (R,N).
 * prepare = __init__ (:39-66): normalised log_pis, sigmas, clamp(1-rho^2), Cholesky L (R,N,2,2), covariance (:211-217);
 * log_prob (:136-163) -> logp (R) and/or reg_los (:107-133) -> lreg (R,N);
 * rsample (:79-105): out (S,R,2) = mu_c + L_c z_c, z (S,R,N,2) standard normal from the caller, component c from
 *   comp (S,R) int32 or, when comp is NULL, by inverse CDF on the uniforms u (S,R); comp_out (S,R) optional;
 * mode (:176-204): 0.01 mesh arg max between the component means, one (x,y) per distribution;
 * from_cov (:68-77): covariance matrices (RN,2,2) -> log_sigmas (RN,2), corrs (RN). */
int mmt_gmm2d_prepare(const float *log_pis, const float *log_sigmas, const float *corrs, long R, int N,
                      float *norm_log_pis, float *sigmas, float *one_minus_rho2, float *L, float *cov, void *stream);
int mmt_gmm2d_log_prob(const float *norm_log_pis, const float *mus, const float *log_sigmas, const float *sigmas,
                       const float *corrs, const float *one_minus_rho2, const float *value, long R, int N,
                       float *logp, float *lreg, void *stream);
int mmt_gmm2d_rsample(const float *norm_log_pis, const float *mus, const float *L, const float *z, const int32_t *comp,
                      const float *u, long S, long R, int N, float *out, int32_t *comp_out, void *stream);
int mmt_gmm2d_mode(const float *norm_log_pis, const float *mus, const float *log_sigmas, const float *sigmas,
                   const float *corrs, const float *one_minus_rho2, long R, int N, float *out, void *stream);
int mmt_gmm2d_from_cov(const float *cov, long RN, float *log_sigmas, float *corrs, void *stream);

/* ---- map patch extraction: GeometricMap.get_cropped_maps_from_scene_map_batch (environment/map.py:93-173)
 * + warp_affine_crop / bilinear grid_sample (model/dataset/homography_warper.py:255-497) as one gather-rotate kernel.
 * maps: device array of nmaps device pointers to uint8 rasters [layers][X][Y]; map_dims (nmaps,2) = X,Y; agent_map (B)
 * = raster index per agent; centers (B,2) = the agents' map points (to_map_points, truncated to integers, unpadded);
 * angles_deg (B) heading rotation in degrees or NULL; patch_size[4] = [-x,-y,+x,+y] (HOST pointer);
 * out (B, layers, ps[1]+ps[3], ps[0]+ps[2]) fp32 in [0,1]. */
int mmt_map_crop(const uint8_t *const *maps, const int32_t *map_dims, int layers, const int32_t *agent_map,
                 const int32_t *centers, const float *angles_deg, const int32_t *patch_size, int B, float *out, void *stream);

/* ---- batch construction: get_node_timestep_data / get_timesteps_data (model/dataset/preprocessing.py:79-270)
 * for B (node, timestep) pairs in one launch over a packed fp64 scene table: table rows of node n start at row_off[n] and
 * cover scene timesteps first[n]..last[n] (environment/node.py:77-139), ncol state columns with the position first.
 * Outputs as the reference's batch element (preprocessing.py:100-123): first_history_index (B) int64, x (B,max_ht+1,ncol),
 * y (B,max_ft,npred), x_st / y_st = (. - current position) / scale with NaN padding kept (environment.py:49-61 divides by
 * 80 regardless of the configured std).  pred_col (npred <= 2 column indices, position prediction) is a HOST array. */
int mmt_batch_gather(const double *table, const int64_t *row_off, const int32_t *first, const int32_t *last,
                     const int32_t *node_idx, const int32_t *ts, int B, int max_ht, int max_ft, int ncol,
                     const int32_t *pred_col, int npred, double scale, int64_t *first_history_index, float *x, float *y,
                     float *x_st, float *y_st, void *stream);
/* Lane part of a batch element (preprocessing.py:170-202, collate :43-45, trajectron.py:62-65): lanes = the nodes' lane tables
 * (fp64, metres), one (n_lanes, n_points, 2) block per row of `table` (Node.lanes_point[t - first_timestep]);
 * lane_input (B,n_lanes,n_points,2) = points / 80; lane_t_mask (B,n_lanes) uint8 = the lane's first two points differ;
 * lane_label (B,max_ft,n_lanes) = one-hot of the lane nearest to each standardised future position (np.argmin semantics). */
int mmt_lane_gather(const double *table, const double *lanes, const int64_t *row_off, const int32_t *first,
                    const int32_t *last, const int32_t *node_idx, const int32_t *ts, int B, int max_ft, int ncol,
                    int n_lanes, int n_points, double scale, float *lane_input, float *lane_label, uint8_t *lane_mask,
                    void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MMT_AMD_H */
