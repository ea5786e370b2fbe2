// K5: ViT map encoder (vit_pytorch 0.33.2 architecture) for gfx950, one workgroup per agent.
//
// tokens = n_patches + 1 (17) live in a 32-row LDS tile for the whole network: patchify + Linear
// (K = 1875 streamed in 256-wide panels gathered from the raster with map*2-1 applied on the fly),
// cls token + learned positions, `depth` pre-norm blocks
//     x += to_out(softmax(q k^T / sqrt(dh)) v),   x += W2 gelu(W1 LN(x))
// with attention run head-group by head-group (2 heads = 128 q/k/v columns at a time, the group's
// out-projection K-slice accumulated straight into the residual), then LN + Linear on the cls row.
// Same weight-streaming MFMA engine as the encoder/decoder (mmt_device.cuh).
// Call sites in the reference: mgcvae.py:145-155 (ctor), :294-296 (forward on map*2-1).
#include "mmt_device.cuh"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

int mmt_set_error(int code, const char *fmt, ...);   // transformer.hip

namespace {

struct VitLaunch {
  int B, ntok, npatch_side, ldx, lds_scr, patch_dim;
};

// CLS16 (exactly 16 patches, the shipped 100 / 25 geometry): the patches own rows 0..15 = ONE MFMA row tile, the cls token sits in
// row 16 and goes through every linear on the VALU (tile_linear's XR form) instead of dragging a second row tile that is 15/16
// padding through the matrix cores.  Otherwise token t is row t of a two-tile block.
template <int MB, int MODE, bool CLS16>
__device__ __forceinline__ void vit_linear(const float *X, int ldx, const mmt_linear &L, float *Y, int ldy) {
  if constexpr (CLS16) linear_to_lds_xrow<1, MODE>(X, ldx, L, Y, ldy);
  else linear_to_lds<MB, MODE>(X, ldx, L, Y, ldy);
}

template <int DH, bool CLS16>
__global__ __launch_bounds__(MMT_NT) void mmt_vit_kernel(mmt_vit V, VitLaunch P, const float *__restrict__ maps,
                                                         float *__restrict__ out) {
  extern __shared__ __align__(16) float smem[];
  const int dim = V.dim, ldx = P.ldx, lds = P.lds_scr, ntok = P.ntok;
  constexpr int NR = CLS16 ? 17 : 32;   // rows of every LDS tile
  float *xs = smem;              // [NR][ldx] residual stream
  float *ln = xs + NR * ldx;     // [NR][ldx] LayerNorm output (pre-norm blocks keep x untouched)
  float *scr = ln + NR * ldx;    // [NR][lds] patch panel / q,k,v of one head group / FF hidden
  const int tid = threadIdx.x, nt = blockDim.x;
  const int b = blockIdx.x;
  const int S = V.image, ps = V.patch, C = V.channels;
  const float *img = maps + (size_t)b * C * S * S;

  for (int i = tid; i < NR * (2 * ldx + lds); i += nt) smem[i] = 0.f;
  __syncthreads();

  // ---- patch embedding, K streamed in panels; token row = 1 + patch index, row 0 (cls) stays zero here
  const int PK = V.panel_k;
  const int npanel = (P.patch_dim + PK - 1) / PK;
  const size_t panel_stride = (size_t)round16(dim) * PK;     // packed floats of a full panel
  for (int pn = 0; pn < npanel; ++pn) {
    const int k0 = pn * PK, kw = min(PK, P.patch_dim - k0);
    for (int i = tid; i < (ntok - 1) * PK; i += nt) {
      const int p = i / PK, kk = i - p * PK;
      float v = 0.f;
      if (kk < kw) {
        const int f = k0 + kk;                      // feature index inside '(p1 p2 c)'
        const int c = f % C, pp = f / C, p2 = pp % ps, p1 = pp / ps;
        const int hi = p / P.npatch_side, wi = p - hi * P.npatch_side;
        v = img[((size_t)c * S + hi * ps + p1) * S + wi * ps + p2];
        if (V.input_affine) v = v * 2.0f - 1.0f;
      }
      scr[(CLS16 ? p : 1 + p) * lds + kk] = v;
    }
    __syncthreads();
    mmt_linear L = V.patch_fc;
    L.w = V.patch_fc.w + pn * panel_stride;
    L.k = kw;
    if (pn == 0) {
      linear_to_lds<CLS16 ? 1 : 2, EPI_STORE>(scr, lds, L, xs, ldx);
    } else {
      L.b = nullptr;
      linear_to_lds<CLS16 ? 1 : 2, EPI_ADD>(scr, lds, L, xs, ldx);
    }
    __syncthreads();
  }
  // cls token and learned positions ('x += pos_embedding[:, :n+1]')
  for (int i = tid; i < ntok * dim; i += nt) {
    const int t = i / dim, c = i - t * dim;                 // token t: 0 = cls
    const int r = CLS16 ? (t == 0 ? 16 : t - 1) : t;
    const float base = t == 0 ? V.cls_token[c] : xs[r * ldx + c];
    xs[r * ldx + c] = base + V.pos_embedding[(size_t)t * dim + c];
  }
  __syncthreads();

  const int hg = V.heads_per_group, gw = hg * DH;          // group width (columns of q, k, v each)
  const int ngroups = V.heads / hg;
  const float scale = 1.0f / sqrtf((float)DH);
  const size_t qkv_stride = (size_t)round16(gw) * round16(dim);
  const size_t out_stride = (size_t)round16(dim) * round16(gw);
  for (int Lr = 0; Lr < V.depth; ++Lr) {
    const mmt_vit_block &W = V.block[Lr];
    layer_norm_rows_to(xs, ldx, ln, ldx, ntok, dim, W.ln1);
    __syncthreads();
    for (int g = 0; g < ngroups; ++g) {
      if ((gw & 15) == 0) {                // q | k | v of this head group: three back-to-back blocks of whole slabs = one linear
        mmt_linear L = W.qkv;
        L.w = W.qkv.w + (size_t)(g * 3) * qkv_stride;
        L.n = 3 * gw;
        vit_linear<2, EPI_STORE, CLS16>(ln, ldx, L, scr, lds);
      } else {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          mmt_linear L = W.qkv;
          L.w = W.qkv.w + (size_t)(g * 3 + t) * qkv_stride;
          L.n = gw;
          vit_linear<2, EPI_STORE, CLS16>(ln, ldx, L, scr + t * gw, lds);
        }
      }
      __syncthreads();
      for (int q16 = tid >> 4; q16 < ntok * hg; q16 += nt >> 4) {
        const int r = q16 / hg, hh = q16 - r * hg;
        float *q = scr + r * lds + hh * DH;
        const float *kb = scr + gw + hh * DH;
        attend_group<DH>(q, kb, kb + gw, lds, ntok, 0u, scale, q);
      }
      __syncthreads();
      mmt_linear L = W.out;
      L.w = W.out.w + (size_t)g * out_stride;
      L.k = gw;
      if (g != 0) L.b = nullptr;
      vit_linear<2, EPI_ADD, CLS16>(scr, lds, L, xs, ldx);
      __syncthreads();
    }
    layer_norm_rows_to(xs, ldx, ln, ldx, ntok, dim, W.ln2);
    __syncthreads();
    vit_linear<2, EPI_GELU, CLS16>(ln, ldx, W.ff1, scr, lds);
    __syncthreads();
    vit_linear<2, EPI_ADD, CLS16>(scr, lds, W.ff2, xs, ldx);
    __syncthreads();
  }
  // cls pooling: LN + Linear on the cls row only (the 15 rows after it -- zeros, or with CLS16 whatever follows in LDS -- are
  // other rows of the MFMA tile: they never mix into row 0 and their outputs are not read)
  constexpr int CR = CLS16 ? 16 : 0;
  layer_norm_rows_to(xs + CR * ldx, ldx, ln + CR * ldx, ldx, 1, dim, V.head_ln);
  __syncthreads();
  linear_to_lds<1, EPI_STORE>(ln + CR * ldx, ldx, V.head_fc, scr, lds);
  __syncthreads();
  for (int c = tid; c < V.n_out; c += nt) out[(size_t)b * V.n_out + c] = scr[c];
}

}  // namespace

extern "C" int mmt_vit_forward(const mmt_vit *vit, const float *maps, int B, float *out, void *stream) {
  if (!vit || !maps || !out) return mmt_set_error(MMT_EINVAL, "mmt_vit_forward: null pointer");
  if (B <= 0) return mmt_set_error(MMT_EINVAL, "mmt_vit_forward: empty batch (B=%d)", B);
  const mmt_vit &V = *vit;
  if (V.patch <= 0 || V.image % V.patch) return mmt_set_error(MMT_EINVAL, "mmt_vit_forward: image %d not divisible by patch %d", V.image, V.patch);
  const int side = V.image / V.patch, ntok = side * side + 1;
  if (ntok > 32) return mmt_set_error(MMT_EINVAL, "mmt_vit_forward: %d tokens do not fit the 32-row tile", ntok);
  if (V.depth <= 0 || V.depth > MMT_MAX_LAYERS) return mmt_set_error(MMT_EINVAL, "mmt_vit_forward: depth %d outside 1..%d", V.depth, MMT_MAX_LAYERS);
  if (V.dim % 16 || V.mlp_dim % 16) return mmt_set_error(MMT_EINVAL, "mmt_vit_forward: dim and mlp_dim must be multiples of 16");
  if (V.dim_head != 16 && V.dim_head != 32 && V.dim_head != 64) return mmt_set_error(MMT_EINVAL, "mmt_vit_forward: dim_head must be 16, 32 or 64");
  if (V.heads_per_group <= 0 || V.heads % V.heads_per_group) return mmt_set_error(MMT_EINVAL, "mmt_vit_forward: heads %% heads_per_group != 0");
  if (V.panel_k <= 0 || V.panel_k % 64) return mmt_set_error(MMT_EINVAL, "mmt_vit_forward: panel_k must be a positive multiple of 64");
  VitLaunch P;
  P.B = B;
  P.ntok = ntok;
  P.npatch_side = side;
  P.patch_dim = V.channels * V.patch * V.patch;
  P.ldx = V.dim + MMT_PAD;
  const int gw = V.heads_per_group * V.dim_head;
  P.lds_scr = std::max(std::max(3 * gw, V.mlp_dim), std::max(V.panel_k, (V.n_out + 15) & ~15)) + MMT_PAD;
  const bool cls16 = ntok == 17;
  // (CLS16: 17 rows; the cls pooling's MFMA tile starts at row 16 and runs 15 rows into what follows: keep that inside the allocation)
  const size_t lds = cls16 ? std::max((size_t)17 * (2 * P.ldx + P.lds_scr), (size_t)(17 + 32) * P.ldx + 16 * P.lds_scr) * sizeof(float)
                           : (size_t)32 * (2 * P.ldx + P.lds_scr) * sizeof(float);
  if (lds > 160 * 1024) return mmt_set_error(MMT_EINVAL, "mmt_vit_forward: tile needs %zu B of LDS (> 160 KiB)", lds);
#define VIT_LAUNCH(DHV)                                                                                          \
  do {                                                                                                           \
    if (cls16) {                                                                                                 \
      (void)hipFuncSetAttribute((const void *)mmt_vit_kernel<DHV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL((mmt_vit_kernel<DHV, true>), dim3(B), dim3(MMT_NT), lds, (hipStream_t)stream, V, P, maps, out); \
    } else {                                                                                                     \
      (void)hipFuncSetAttribute((const void *)mmt_vit_kernel<DHV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL((mmt_vit_kernel<DHV, false>), dim3(B), dim3(MMT_NT), lds, (hipStream_t)stream, V, P, maps, out); \
    }                                                                                                            \
  } while (0)
  if (V.dim_head == 16) VIT_LAUNCH(16); else if (V.dim_head == 32) VIT_LAUNCH(32); else VIT_LAUNCH(64);
#undef VIT_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mmt_set_error(MMT_ELAUNCH, "mmt_vit_forward: %s", hipGetErrorString(e));
  return MMT_OK;
}
