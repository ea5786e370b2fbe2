// Encoder and KV-cached autoregressive decoder for gfx950 (MI355X).
//
// K1  mmt_encoder_kernel<MB>  : nan_to_num -> Linear -> +PE -> L x [QKV, causal+pad attention,
//                               out-proj + residual + LN, FFN + residual + LN]; the 16*MB-row tile of
//                               floor(16*MB/T) agents never leaves LDS.          (Transformer_model.py:45-59)
// K2+K3 mmt_decoder_kernel    : prologue = optional memory fusion + cross-attention K/V of all layers
//                               (computed once, the reference redoes it every step), then ph steps of
//                               the 6-layer decoder on ONE new token per agent with a KV cache, the
//                               output_fc, optional map fusion and the MLP head, positions fed back
//                               inside the kernel.                                (mgcvae.py:375-449)
#include "mmt_device.cuh"

#ifdef MMT_PROFILE
__device__ unsigned long long g_prof[512];
#define STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && s == (ph / 2)) g_prof[L * 16 + (k)] = clock64(); } while (0)
#define ESTAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_prof[256 + L * 8 + (k)] = wall_clock64(); } while (0)
#else
#define ESTAMP(k) do { } while (0)
#define STAMP(k) do { } while (0)
#endif

namespace {

// input_fc epilogues: + bias + positional encoding (Transformer_model.py:48-49 / :100-102)
struct EncInEpi {
  const float *__restrict__ bias, *__restrict__ pe;
  float *__restrict__ xs;
  int d, T, ldx;
  template <int M>
  struct Pre {
    float b;
  };
  template <int M>
  __device__ __forceinline__ void prep(Pre<M> &p, int, int c) const { p.b = c < d ? bias[c] : 0.f; }
  template <int M>
  __device__ __forceinline__ void store(int r, int c, float v, const Pre<M> &p, int, int) const {
    if (c < d) xs[r * ldx + c] = v + p.b + pe[(r % T) * d + c];
  }
};
struct DecInEpi {
  const float *__restrict__ bias, *__restrict__ pe;   // pe already offset to the current step's row
  float *__restrict__ xs;
  int d, ldx;
  template <int M>
  struct Pre {
    float b;
  };
  template <int M>
  __device__ __forceinline__ void prep(Pre<M> &p, int, int c) const { p.b = c < d ? bias[c] + pe[c] : 0.f; }
  template <int M>
  __device__ __forceinline__ void store(int r, int c, float v, const Pre<M> &p, int, int) const {
    if (c < d) xs[r * ldx + c] = v + p.b;
  }
};

struct EncLaunch {
  int B, T, ldx, ldb, apw;   // apw = agents per workgroup
};

template <int MB, int DH>
__global__ __launch_bounds__(MMT_NT) void mmt_encoder_kernel(mmt_encoder E, EncLaunch P, const float *__restrict__ x_st,
                                                          float *__restrict__ memory, uint8_t *__restrict__ pad_out) {
  extern __shared__ __align__(16) float smem[];
  constexpr int MT = 16 * MB;
  const int d = E.d_model, H = E.nhead, T = P.T, ldx = P.ldx, ldb = P.ldb;
  constexpr int dh = DH;
  float *xs = smem;
  float *buf = xs + MT * ldx;
  uint32_t *padbits = reinterpret_cast<uint32_t *>(buf + MT * ldb);
  const int tid = threadIdx.x, nt = blockDim.x;
  const int a0 = blockIdx.x * P.apw;
  const int na = min(P.apw, P.B - a0);
  const int nrows = na * T;
  const size_t grow0 = (size_t)a0 * T;

  for (int i = tid; i < MT * (ldx + ldb); i += nt) smem[i] = 0.f;
  __syncthreads();
  // stage the input, nan_to_num(nan=0, +-inf -> +-FLT_MAX)           Transformer_model.py:47
  for (int i = tid; i < nrows * E.n_in; i += nt) {
    const int r = i / E.n_in, c = i - r * E.n_in;
    float v = x_st[(grow0 + r) * E.n_in + c];
    if (isnan(v)) v = 0.f;
    else if (isinf(v)) v = v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    buf[r * ldb + c] = v;
  }
  // key padding = NaN in feature column 1                              model_utils.py:148-151
  if (tid < na) {
    uint32_t bits = 0;
    for (int t = 0; t < T; ++t) {
      const bool p = isnan(x_st[(grow0 + tid * T + t) * E.n_in + 1]);
      bits |= (p ? 1u : 0u) << t;
      if (pad_out) pad_out[grow0 + tid * T + t] = p ? 1 : 0;
    }
    padbits[tid] = bits;
  }
  __syncthreads();
  WRegs pre;
  {
    EncInEpi epi{E.input_fc.b, E.pe, xs, d, T, ldx};
    tile_linear<MB, false>(buf, ldb, E.input_fc.w, round16(E.n_in) >> 4, round16(d) >> 4, epi, pre,
                           next_of(E.layer[0].qkv));
  }
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)dh);
  for (int L = 0; L < E.nlayers; ++L) {
    const mmt_enc_layer &W = E.layer[L];
    ESTAMP(0);
    linear_to_lds<MB, EPI_STORE, true>(xs, ldx, W.qkv, buf, ldb, pre, next_of(W.out));
    __syncthreads();
    ESTAMP(1);
    for (int g = tid >> 4; g < nrows * H; g += nt >> 4) {
      const int r = g / H, h = g - r * H, a = r / T, t = r - a * T;
      float *q = buf + r * ldb + h * dh;
      const float *kb = buf + (a * T) * ldb + d + h * dh;
      attend_group<DH>(q, kb, kb + d, ldb, t + 1, padbits[a], scale, q);   // causal: keys 0..t
    }
    __syncthreads();
    ESTAMP(2);
    linear_to_lds<MB, EPI_ADD, true>(buf, ldb, W.out, xs, ldx, pre, next_of(W.ff1));
    __syncthreads();
    ESTAMP(3);
    layer_norm_rows<false>(xs, ldx, nrows, d, W.ln1);
    __syncthreads();
    ESTAMP(4);
    linear_to_lds<MB, EPI_RELU, true>(xs, ldx, W.ff1, buf, ldb, pre, next_of(W.ff2));
    __syncthreads();
    ESTAMP(5);
    linear_to_lds<MB, EPI_ADD, true>(buf, ldb, W.ff2, xs, ldx, pre,
                                     L + 1 < E.nlayers ? next_of(E.layer[L + 1].qkv) : no_next());
    __syncthreads();
    ESTAMP(6);
    layer_norm_rows<false>(xs, ldx, nrows, d, W.ln2);
    __syncthreads();
    ESTAMP(7);
  }
  for (int i = tid; i < nrows * d; i += nt) {
    const int r = i / d, c = i - r * d;
    memory[(grow0 + r) * d + c] = xs[r * ldx + c];
  }
}

// ---------------------------------------------------------------------------------------------
struct DecLaunch {
  int B, T, ph, ldx, ldb, half, apw;   // half = column offset of scratch region B inside buf
  int map_dim, mem_feat_dim;
};

template <int DH>
__global__ __launch_bounds__(MMT_NT) void mmt_decoder_kernel(mmt_decoder D, DecLaunch P, const float *__restrict__ memory,
                                                          const uint8_t *__restrict__ pad_mask,
                                                          const float *__restrict__ start_pos,
                                                          const float *__restrict__ map_feat,
                                                          const float *__restrict__ mem_feat,
                                                          const float *__restrict__ tgt, float *__restrict__ pred_st,
                                                          float *__restrict__ h_out, float *__restrict__ memkv,
                                                          float *__restrict__ selfkv) {
  extern __shared__ __align__(16) float smem[];
  const int d = D.d_model, H = D.nhead, T = P.T, ph = P.ph, ldx = P.ldx, ldb = P.ldb, B = P.B;
  constexpr int dh = DH;
  const int nin = D.n_in, nout = D.n_out;
  float *xs = smem;                      // [16][ldx]   residual stream of the current token
  float *buf = xs + 16 * ldx;            // [16][ldb]   qkv / q / ffn hidden / small MLP scratch
  float *pos = buf + 16 * ldb;           // [16][16]    fed-back position (standardised)
  float *delta = pos + 16 * 16;          // [16][16]    head output
  uint32_t *padbits = reinterpret_cast<uint32_t *>(delta + 16 * 16);   // [16]
  float *regA = buf, *regB = buf + P.half;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int a0 = blockIdx.x * P.apw;
  const int na = min(P.apw, B - a0);     // agents (= rows) of this workgroup
  const float scale = 1.0f / sqrtf((float)dh);

  for (int i = tid; i < 16 * (ldx + ldb) + 512; i += nt) smem[i] = 0.f;
  __syncthreads();
  if (tid < na) {
    uint32_t bits = 0;
    if (pad_mask)
      for (int t = 0; t < T; ++t) bits |= (pad_mask[(size_t)(a0 + tid) * T + t] ? 1u : 0u) << t;
    padbits[tid] = bits;
  }
  if (!tgt)
    for (int i = tid; i < na * nin; i += nt) {
      const int r = i / nin, c = i - r * nin;
      pos[r * 16 + c] = start_pos[(size_t)(a0 + r) * nin + c];
    }

  // ---- prologue: (optional) memory fusion, then cross-attention K/V of every layer, 16 memory rows at a time
  const int mrows = na * T;
  for (int m0 = 0; m0 < mrows; m0 += 16) {
    const int nr = min(16, mrows - m0);
    __syncthreads();
    if (D.has_mem_fusion) {
      const int mf = P.mem_feat_dim;
      for (int i = tid; i < 16 * (d + mf); i += nt) {
        const int r = i / (d + mf), c = i - r * (d + mf);
        float v = 0.f;
        if (r < nr) {
          const int grow = m0 + r, a = grow / T;
          v = c < d ? memory[((size_t)a0 * T + grow) * d + c] : mem_feat[(size_t)(a0 + a) * mf + (c - d)];
        }
        regA[r * ldb + c] = v;
      }
      __syncthreads();
      mlp_stack_tile(D.mem_fusion, regA, regB, ldb, xs, ldx, nr);
    } else {
      for (int i = tid; i < 16 * d; i += nt) {
        const int r = i / d, c = i - r * d;
        xs[r * ldx + c] = r < nr ? memory[((size_t)a0 * T + m0 + r) * d + c] : 0.f;
      }
      __syncthreads();
    }
    for (int L = 0; L < D.nlayers; ++L) {
      linear_to_lds<1, EPI_STORE>(xs, ldx, D.layer[L].ckv, buf, ldb);
      __syncthreads();
      float *dst = memkv + ((size_t)L * B * T + (size_t)a0 * T + m0) * 2 * d;
      for (int i = tid; i < nr * 2 * d; i += nt) {
        const int r = i / (2 * d), c = i - r * 2 * d;
        dst[(size_t)r * 2 * d + c] = buf[r * ldb + c];
      }
      __syncthreads();
    }
  }
  __syncthreads();

  // ---- ph autoregressive steps, one new token per agent per step
  WRegs pre;
  for (int s = 0; s < ph; ++s) {
    for (int i = tid; i < 16 * 16; i += nt) {
      const int r = i >> 4, c = i & 15;
      float v = 0.f;
      if (r < na && c < nin) v = tgt ? tgt[((size_t)(a0 + r) * ph + s) * nin + c] : pos[r * 16 + c];
      buf[r * ldb + c] = v;
    }
    __syncthreads();
    {
      DecInEpi epi{D.input_fc.b, D.pe + (size_t)s * d, xs, d, ldx};
      tile_linear<1, false>(buf, ldb, D.input_fc.w, round16(nin) >> 4, round16(d) >> 4, epi, pre,
                            next_of(D.layer[0].qkv));
    }
    __syncthreads();
    for (int L = 0; L < D.nlayers; ++L) {
      const mmt_dec_layer &W = D.layer[L];
      STAMP(0);
      // self attention on the KV cache
      linear_to_lds<1, EPI_STORE, true>(xs, ldx, W.qkv, buf, ldb, pre, next_of(W.out));
      __syncthreads();
      STAMP(1);
      float *cache = selfkv + ((size_t)L * B + a0) * ph * 2 * d;
      for (int i = tid; i < na * 2 * d; i += nt) {
        const int r = i / (2 * d), c = i - r * 2 * d;
        cache[((size_t)r * ph + s) * 2 * d + c] = buf[r * ldb + d + c];
      }
      __syncthreads();
      STAMP(2);
      for (int g = tid >> 4; g < na * H; g += nt >> 4) {
        const int r = g / H, h = g - r * H;
        float *q = buf + r * ldb + h * dh;
        const float *kb = cache + (size_t)r * ph * 2 * d + h * dh;
        attend_group<DH>(q, kb, kb + d, 2 * d, s + 1, 0u, scale, q);
      }
      __syncthreads();
      STAMP(3);
      linear_to_lds<1, EPI_ADD, true>(buf, ldb, W.out, xs, ldx, pre, next_of(W.cq));
      __syncthreads();
      STAMP(4);
      layer_norm_rows<false>(xs, ldx, na, d, W.ln1);
      __syncthreads();
      STAMP(5);
      // cross attention on the precomputed memory K/V; only key padding masks it (memory_mask is dropped)
      linear_to_lds<1, EPI_STORE, true>(xs, ldx, W.cq, buf, ldb, pre, next_of(W.cout));
      __syncthreads();
      STAMP(6);
      const float *mkv = memkv + ((size_t)L * B + a0) * T * 2 * d;
      for (int g = tid >> 4; g < na * H; g += nt >> 4) {
        const int r = g / H, h = g - r * H;
        float *q = buf + r * ldb + h * dh;
        const float *kb = mkv + (size_t)r * T * 2 * d + h * dh;
        attend_group<DH>(q, kb, kb + d, 2 * d, T, padbits[r], scale, q);
      }
      __syncthreads();
      STAMP(7);
      linear_to_lds<1, EPI_ADD, true>(buf, ldb, W.cout, xs, ldx, pre, next_of(W.ff1));
      __syncthreads();
      STAMP(8);
      layer_norm_rows<false>(xs, ldx, na, d, W.ln2);
      __syncthreads();
      STAMP(9);
      linear_to_lds<1, EPI_RELU, true>(xs, ldx, W.ff1, buf, ldb, pre, next_of(W.ff2));
      __syncthreads();
      STAMP(10);
      linear_to_lds<1, EPI_ADD, true>(buf, ldb, W.ff2, xs, ldx, pre,
                                      L + 1 < D.nlayers ? next_of(D.layer[L + 1].qkv) : next_of(D.output_fc));
      __syncthreads();
      STAMP(11);
      layer_norm_rows<false>(xs, ldx, na, d, W.ln3);
      __syncthreads();
      STAMP(12);
    }
    linear_to_lds<1, EPI_STORE, true>(xs, ldx, D.output_fc, regA, ldb, pre, no_next());
    __syncthreads();
    if (h_out)
      for (int i = tid; i < na * nout; i += nt) {
        const int r = i / nout, c = i - r * nout;
        h_out[((size_t)(a0 + r) * ph + s) * nout + c] = regA[r * ldb + c];
      }
    if (!tgt) {
      float *zin = regA, *zalt = regB;
      if (D.has_map_fusion) {
        const int md = P.map_dim;
        for (int i = tid; i < na * md; i += nt) {
          const int r = i / md, c = i - r * md;
          regA[r * ldb + nout + c] = map_feat[(size_t)(a0 + r) * md + c];
        }
        __syncthreads();
        // fusion output lands at the start of region B, which then feeds the head
        float *fz = regB;
        mlp_stack_tile(D.map_fusion, regA, regB + P.half / 2, ldb, fz, ldb, na);
        zin = regB;
        zalt = regA;
      }
      mlp_stack_tile(D.head, zin, zalt, ldb, delta, 16, na);
      for (int i = tid; i < na * nin; i += nt) {
        const int r = i / nin, c = i - r * nin;
        const float np = pos[r * 16 + c] + delta[r * 16 + c];
        pos[r * 16 + c] = np;
        pred_st[((size_t)(a0 + r) * ph + s) * nin + c] = np;
      }
    }
    __syncthreads();
  }
}

__global__ void mmt_pack_linear_kernel(const float *__restrict__ w, int n, int k, int ldw, int col0,
                                       float *__restrict__ dst, int K16, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = i & 3, lane = (i >> 2) & 63;
    const size_t tile = i >> 8;
    const int chunk = tile % K16, slab = tile / K16;
    const int row = slab * 16 + (lane & 15), col = chunk * 16 + (lane >> 4) * 4 + e;
    dst[i] = (row < n && col < k) ? w[(size_t)row * ldw + col0 + col] : 0.f;
  }
}

// With loss_sum the host launches ONE block: every thread adds its agents in index order, the waves are reduced by a fixed
// shuffle tree and summed in wave order - the same bits on every run (a float atomicAdd per wave is not).
__global__ __launch_bounds__(1024) void mmt_finalize_kernel(const float *__restrict__ pred_st, const float *__restrict__ node_pos,
                                                            const float *__restrict__ gt, int metres, int B, int ph,
                                                            float *__restrict__ pred_m, float *__restrict__ ade,
                                                            float *__restrict__ fde, float *__restrict__ loss_sum) {
  __shared__ float wsum[16];
  float mine = 0.f;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
    const float nx = node_pos ? node_pos[b * 2] : 0.f, ny = node_pos ? node_pos[b * 2 + 1] : 0.f;
    float acc = 0.f, last = 0.f;
    for (int t = 0; t < ph; ++t) {
      const size_t o = ((size_t)b * ph + t) * 2;
      const float sx = pred_st[o], sy = pred_st[o + 1];
      const float mx = sx * 80.0f + nx, my = sy * 80.0f + ny;      // mgcvae.py:772
      if (pred_m) {
        pred_m[o] = mx;
        pred_m[o + 1] = my;
      }
      if (gt) {
        const float dx = (metres ? mx : sx) - gt[o], dy = (metres ? my : sy) - gt[o + 1];
        last = sqrtf(dx * dx + dy * dy);
        acc += last;
      }
    }
    const float mean_err = acc / (float)ph;
    if (ade) ade[b] = mean_err;
    if (fde) fde[b] = last;
    mine += mean_err;
  }
  if (loss_sum) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
      float total = 0.f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) total += wsum[w];
      *loss_sum += total;
    }
  }
}

}  // namespace

// ================================================================================ host side
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>

static thread_local char g_err[512] = "";
int mmt_set_error(int code, const char *fmt, ...) {   // shared with context.hip (not part of the C ABI)
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define fail mmt_set_error
static int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MMT_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return MMT_OK;
}

extern "C" const char *mmt_last_error(void) { return g_err; }
extern "C" int mmt_version(void) { return 1; }
#ifdef MMT_PROFILE
extern "C" int mmt_debug_read_prof(unsigned long long *host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * n);
}
#endif

extern "C" size_t mmt_packed_linear_floats(int n, int k) {
  return (size_t)((n + 15) & ~15) * (size_t)((k + 15) & ~15);
}

extern "C" int mmt_pack_linear(const float *w, int n, int k, int ldw, int col0, float *dst, void *stream) {
  if (!w || !dst || n <= 0 || k <= 0 || ldw < col0 + k || col0 < 0) return fail(MMT_EINVAL, "mmt_pack_linear: bad arguments");
  const size_t total = mmt_packed_linear_floats(n, k);
  const int K16 = ((k + 15) & ~15) >> 4;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(mmt_pack_linear_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, n, k, ldw, col0, dst,
                     K16, total);
  return check_launch("mmt_pack_linear");
}

static int max_mlp_width(const mmt_mlp &m) {
  int w = std::max(m.out.k, m.out.n);
  for (int i = 0; i < m.n_hidden; ++i) w = std::max(w, std::max(m.hidden[i].k, m.hidden[i].n));
  return (w + 15) & ~15;
}

static int check_tf(int d, int H, int f, int L, const char *who) {
  if (d <= 0 || H <= 0 || d % H || L <= 0 || L > MMT_MAX_LAYERS || f <= 0)
    return fail(MMT_EINVAL, "%s: bad transformer config d=%d H=%d f=%d L=%d", who, d, H, f, L);
  const int dh = d / H;
  if (d % 16 || (dh != 16 && dh != 32 && dh != 64)) return fail(MMT_EINVAL, "%s: need d%%16==0 and head dim 16, 32 or 64 (d=%d dh=%d)", who, d, dh);
  return MMT_OK;
}

// encoder_split.hip
int mmt_encoder_forward_split(const mmt_encoder *enc, const float *x_st, int B, int T, float *memory, uint8_t *pad_mask,
                              float *workspace, void *stream);

static int encoder_forward_impl(const mmt_encoder *enc, const float *x_st, int B, int T, float *memory, uint8_t *pad_mask,
                                float *workspace, void *stream);
extern "C" int mmt_encoder_forward(const mmt_encoder *enc, const float *x_st, int B, int T, float *memory,
                                   uint8_t *pad_mask, void *stream) {
  return encoder_forward_impl(enc, x_st, B, T, memory, pad_mask, nullptr, stream);
}
extern "C" int mmt_encoder_forward_ws(const mmt_encoder *enc, const float *x_st, int B, int T, float *memory,
                                      uint8_t *pad_mask, float *workspace, void *stream) {
  return encoder_forward_impl(enc, x_st, B, T, memory, pad_mask, workspace, stream);
}
static int encoder_forward_impl(const mmt_encoder *enc, const float *x_st, int B, int T, float *memory, uint8_t *pad_mask,
                                float *workspace, void *stream) {
  if (!enc || !x_st || !memory) return fail(MMT_EINVAL, "mmt_encoder_forward: null pointer");
  if (B <= 0) return fail(MMT_EINVAL, "mmt_encoder_forward: empty batch (B=%d)", B);
  if (T <= 0 || T > 32) return fail(MMT_EINVAL, "mmt_encoder_forward: history length T=%d outside 1..32", T);
  if (int rc = check_tf(enc->d_model, enc->nhead, enc->d_ff, enc->nlayers, "mmt_encoder_forward")) return rc;
  if (enc->n_in <= 0 || enc->n_in > 64) return fail(MMT_EINVAL, "mmt_encoder_forward: n_in=%d", enc->n_in);
  if (workspace) {      // small batches: two workgroups per agent tile (encoder_split.hip); 0 = not this time
    const int rc = mmt_encoder_forward_split(enc, x_st, B, T, memory, pad_mask, workspace, stream);
    if (rc != 0) return rc < 0 ? rc : MMT_OK;
  }
  const int d = enc->d_model;
  const int wide = std::max(std::max(3 * d, (enc->d_ff + 15) & ~15), (enc->n_in + 15) & ~15);
  EncLaunch P;
  P.B = B;
  P.T = T;
  P.ldx = d + MMT_PAD;
  P.ldb = wide + MMT_PAD;
  // 32-row tiles hold floor(32/T) agents (3 at T=9: 84 % of the MFMA rows are real); use them once the
  // batch is large enough to fill the chip anyway, otherwise one 16-row tile per agent group.
  const int apw16 = 16 / T, apw32 = 32 / T;
  const bool big = (T > 16) || (B >= 768 && apw32 * 16 > apw16 * 32);
  const int MB = big ? 2 : 1;
  P.apw = MB == 2 ? apw32 : apw16;
  const int MT = 16 * MB;
  const size_t lds = ((size_t)MT * (P.ldx + P.ldb) + 64) * sizeof(float);
  if (lds > 160 * 1024) return fail(MMT_EINVAL, "mmt_encoder_forward: tile needs %zu B of LDS (> 160 KiB)", lds);
  const int grid = (B + P.apw - 1) / P.apw;
  const int dh = d / enc->nhead;
#define ENC_LAUNCH(MBV, DHV)                                                                                      \
  do {                                                                                                            \
    (void)hipFuncSetAttribute((const void *)mmt_encoder_kernel<MBV, DHV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds);                                                                          \
    hipLaunchKernelGGL((mmt_encoder_kernel<MBV, DHV>), dim3(grid), dim3(MMT_NT), lds, (hipStream_t)stream, *enc, P,   \
                       x_st, memory, pad_mask);                                                                   \
  } while (0)
  if (MB == 1) {
    if (dh == 16) ENC_LAUNCH(1, 16); else if (dh == 32) ENC_LAUNCH(1, 32); else ENC_LAUNCH(1, 64);
  } else {
    if (dh == 16) ENC_LAUNCH(2, 16); else if (dh == 32) ENC_LAUNCH(2, 32); else ENC_LAUNCH(2, 64);
  }
#undef ENC_LAUNCH
  return check_launch("mmt_encoder_forward");
}

// decoder_cluster.hip
bool mmt_cluster_decoder_supports(const mmt_decoder *dec, int map_dim, bool autoregressive, int T, int ph);
size_t mmt_cluster_decoder_extra_floats(const mmt_decoder *dec, int B, int T);
bool mmt_cluster_decoder_dropout_ok(const mmt_decoder *dec, int B);
int mmt_cluster_decoder_forward(const mmt_decoder *dec, const float *memory, const uint8_t *pad_mask, const float *start_pos,
                                const float *map_feat, int map_dim, const float *mem_feat, int mem_feat_dim,
                                const float *tgt, int B, int T, int ph, float *pred_st, float *h_out, float *workspace,
                                int allow_fused, void *stream, const mmt_decoder_dropout *drop, float *stash);
bool mmt_cluster_decoder_stash_ok(const mmt_decoder *dec, int map_dim, int B, int T, int ph, bool drop);
void mmt_cluster_decoder_stash_layout(const mmt_decoder *dec, int map_dim, int B, int T, int ph, mmt_decoder_stash_layout *out);

static thread_local int g_last_path = MMT_PATH_NONE;
void mmt_note_decoder_path(int path) { g_last_path = path; }
extern "C" int mmt_decoder_last_path(void) { return g_last_path; }
static std::atomic<int> g_decoder_impl_atomic{MMT_DECODER_AUTO};     // process-wide tuning knob; each call reads it once
extern "C" int mmt_decoder_select(int impl) {
  if (impl != MMT_DECODER_AUTO && impl != MMT_DECODER_PERSISTENT && impl != MMT_DECODER_CLUSTER && impl != MMT_DECODER_CLUSTER_LAUNCHES &&
      impl != MMT_DECODER_CLUSTER_FUSED && impl != MMT_DECODER_CLUSTER_PIPELINE)
    return fail(MMT_EINVAL, "mmt_decoder_select: unknown implementation %d", impl);
  g_decoder_impl_atomic.store(impl, std::memory_order_relaxed);
  return MMT_OK;
}

extern "C" size_t mmt_decoder_workspace_floats(const mmt_decoder *dec, int B, int T, int ph) {
  if (!dec || B <= 0 || T <= 0 || ph <= 0) return 0;
  return (size_t)dec->nlayers * B * (size_t)(T + ph) * 2 * dec->d_model + mmt_cluster_decoder_extra_floats(dec, B, T);
}

static int decoder_forward_impl(const mmt_decoder *dec, const float *memory, const uint8_t *pad_mask,
                                const float *start_pos, const float *map_feat, int map_dim, const float *mem_feat,
                                int mem_feat_dim, const float *tgt, int B, int T, int ph, float *pred_st,
                                float *h_out, float *workspace, void *stream, const mmt_decoder_dropout *drop,
                                float *stash = nullptr) {
  if (!dec || !memory || !workspace) return fail(MMT_EINVAL, "mmt_decoder_forward: null pointer");
  const int g_decoder_impl = g_decoder_impl_atomic.load(std::memory_order_relaxed);
  if (B <= 0) return fail(MMT_EINVAL, "mmt_decoder_forward: empty batch (B=%d)", B);
  if (T <= 0 || T > 32) return fail(MMT_EINVAL, "mmt_decoder_forward: memory length T=%d outside 1..32", T);
  if (ph <= 0 || ph > 32) return fail(MMT_EINVAL, "mmt_decoder_forward: horizon ph=%d outside 1..32", ph);
  if (int rc = check_tf(dec->d_model, dec->nhead, dec->d_ff, dec->nlayers, "mmt_decoder_forward")) return rc;
  if (dec->n_in <= 0 || dec->n_in > 16) return fail(MMT_EINVAL, "mmt_decoder_forward: n_in=%d outside 1..16", dec->n_in);
  if (!tgt && (!start_pos || !pred_st)) return fail(MMT_EINVAL, "mmt_decoder_forward: autoregressive mode needs start_pos and pred_st");
  if (tgt && !h_out) return fail(MMT_EINVAL, "mmt_decoder_forward: teacher-forced mode needs h_out");
  if (!tgt && dec->head.out.n != dec->n_in) return fail(MMT_EINVAL, "mmt_decoder_forward: head width %d != token width %d", dec->head.out.n, dec->n_in);
  if (dec->has_map_fusion && !tgt && (!map_feat || map_dim <= 0)) return fail(MMT_EINVAL, "mmt_decoder_forward: map fusion configured but no map_feat");
  if (dec->has_mem_fusion && (!mem_feat || mem_feat_dim <= 0)) return fail(MMT_EINVAL, "mmt_decoder_forward: memory fusion configured but no mem_feat");
  const int d = dec->d_model;
  const bool cluster_ok = mmt_cluster_decoder_supports(dec, map_dim, tgt == nullptr, T, ph);
  const bool want_cluster = g_decoder_impl == MMT_DECODER_CLUSTER || g_decoder_impl == MMT_DECODER_CLUSTER_FUSED ||
                            g_decoder_impl == MMT_DECODER_CLUSTER_PIPELINE;
  if (want_cluster && !cluster_ok)
    return fail(MMT_EINVAL, "mmt_decoder_forward: the cluster decoder does not support this geometry (d=%d H=%d f=%d)", d,
                dec->nhead, dec->d_ff);
  int small = max_mlp_width(dec->head);
  small = std::max(small, (dec->n_out + 15) & ~15);
  if (dec->has_map_fusion) {
    small = std::max(small, max_mlp_width(dec->map_fusion));
    if (((dec->n_out + map_dim + 15) & ~15) > small) small = (dec->n_out + map_dim + 15) & ~15;
    if (dec->map_fusion.n_hidden ? dec->map_fusion.hidden[0].k != dec->n_out + map_dim : dec->map_fusion.out.k != dec->n_out + map_dim)
      return fail(MMT_EINVAL, "mmt_decoder_forward: map fusion expects %d inputs, got n_out+map_dim=%d",
                  dec->map_fusion.n_hidden ? dec->map_fusion.hidden[0].k : dec->map_fusion.out.k, dec->n_out + map_dim);
  }
  if (dec->has_mem_fusion) {
    small = std::max(small, max_mlp_width(dec->mem_fusion));
    const int kin = dec->mem_fusion.n_hidden ? dec->mem_fusion.hidden[0].k : dec->mem_fusion.out.k;
    if (kin != d + mem_feat_dim) return fail(MMT_EINVAL, "mmt_decoder_forward: memory fusion expects %d inputs, got d+mem_feat_dim=%d", kin, d + mem_feat_dim);
    if (dec->mem_fusion.out.n != d) return fail(MMT_EINVAL, "mmt_decoder_forward: memory fusion must output d_model");
  }
  // AUTO: the cluster kernels win up to ~2.5 rounds of 16-agent clusters x nhead workgroups over the chip (measured on MI355X,
  // 8 heads: B = 1024: 4.7 vs 5.1 ms, B = 2048: 9.1 vs 6.1 ms); beyond that the persistent kernel's weight reuse is faster.
  bool use_cluster = cluster_ok && g_decoder_impl != MMT_DECODER_PERSISTENT;
  if (use_cluster && !want_cluster) {
    int dev = 0, ncu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    use_cluster = 2 * ((B + 15) / 16) * dec->nhead <= 5 * ncu;
  }
  if (drop && !cluster_ok) return fail(MMT_EINVAL, "mmt_decoder_forward_dropout: this geometry has no cluster decoder");
  if (stash && (!cluster_ok || g_decoder_impl == MMT_DECODER_PERSISTENT || g_decoder_impl == MMT_DECODER_CLUSTER_LAUNCHES ||
                g_decoder_impl == MMT_DECODER_CLUSTER_FUSED))
    return fail(MMT_EINVAL, "mmt_decoder_forward_stash: only the pipeline decode keeps the stash");
  if (use_cluster || drop || stash)
    return mmt_cluster_decoder_forward(dec, memory, pad_mask, start_pos, map_feat, map_dim, mem_feat, mem_feat_dim, tgt, B, T,
                                       ph, pred_st, h_out, workspace,
                                       g_decoder_impl == MMT_DECODER_CLUSTER_LAUNCHES ? 0 : g_decoder_impl == MMT_DECODER_CLUSTER_FUSED ? 1
                                       : g_decoder_impl == MMT_DECODER_CLUSTER_PIPELINE ? 2 : 3, stream, drop, stash);
  mmt_note_decoder_path(MMT_PATH_PERSISTENT);
  // buf = [ region A | region B ], region B additionally split in two for the map-fusion ping-pong
  const int half = std::max((3 * d + 1) / 2, 2 * (small + MMT_PAD));
  const int halfr = (half + 15) & ~15;
  const int wide = std::max(std::max(3 * d, (dec->d_ff + 15) & ~15), 2 * halfr);
  DecLaunch P;
  P.B = B;
  P.T = T;
  P.ph = ph;
  P.ldx = d + MMT_PAD;
  P.ldb = wide + MMT_PAD;
  P.half = halfr;
  P.map_dim = map_dim;
  P.mem_feat_dim = mem_feat_dim;
  int dev = 0, ncu = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  P.apw = std::min(16, std::max(1, (B + ncu - 1) / ncu));
  const size_t lds = ((size_t)16 * (P.ldx + P.ldb) + 512 + 64) * sizeof(float);
  if (lds > 160 * 1024) return fail(MMT_EINVAL, "mmt_decoder_forward: tile needs %zu B of LDS (> 160 KiB)", lds);
  const int grid = (B + P.apw - 1) / P.apw;
  float *memkv = workspace;
  float *selfkv = workspace + (size_t)dec->nlayers * B * T * 2 * d;
  const int dh = d / dec->nhead;
#define DEC_LAUNCH(DHV)                                                                                             \
  do {                                                                                                              \
    (void)hipFuncSetAttribute((const void *)mmt_decoder_kernel<DHV>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                              (int)lds);                                                                            \
    hipLaunchKernelGGL((mmt_decoder_kernel<DHV>), dim3(grid), dim3(MMT_NT), lds, (hipStream_t)stream, *dec, P, memory,  \
                       pad_mask, start_pos, map_feat, mem_feat, tgt, pred_st, h_out, memkv, selfkv);                \
  } while (0)
  if (dh == 16) DEC_LAUNCH(16); else if (dh == 32) DEC_LAUNCH(32); else DEC_LAUNCH(64);
#undef DEC_LAUNCH
  return check_launch("mmt_decoder_forward");
}

extern "C" int mmt_decoder_forward(const mmt_decoder *dec, const float *memory, const uint8_t *pad_mask,
                                   const float *start_pos, const float *map_feat, int map_dim, const float *mem_feat,
                                   int mem_feat_dim, const float *tgt, int B, int T, int ph, float *pred_st,
                                   float *h_out, float *workspace, void *stream) {
  return decoder_forward_impl(dec, memory, pad_mask, start_pos, map_feat, map_dim, mem_feat, mem_feat_dim, tgt, B, T, ph, pred_st,
                              h_out, workspace, stream, nullptr);
}

// The autoregressive decode of a TRAINING step: the same kernels reading the step's dropout keep masks (include/mmt_amd.h).
extern "C" int mmt_decoder_dropout_supported(const mmt_decoder *dec, int B, int T, int ph) {
  if (!dec || B <= 0) return 0;
  if (!mmt_cluster_decoder_supports(dec, dec->has_map_fusion ? 1 : 0, true, T, ph)) return 0;
  return mmt_cluster_decoder_dropout_ok(dec, B) ? 1 : 0;
}
extern "C" int mmt_decoder_forward_dropout(const mmt_decoder *dec, const float *memory, const uint8_t *pad_mask,
                                           const float *start_pos, const float *map_feat, int map_dim, const float *mem_feat,
                                           int mem_feat_dim, int B, int T, int ph, float *pred_st, float *workspace,
                                           const mmt_decoder_dropout *drop, void *stream) {
  if (!drop) return fail(MMT_EINVAL, "mmt_decoder_forward_dropout: null mask descriptor");
  return decoder_forward_impl(dec, memory, pad_mask, start_pos, map_feat, map_dim, mem_feat, mem_feat_dim, nullptr, B, T, ph, pred_st,
                              nullptr, workspace, stream, drop);
}

// The training step's decode that also keeps what the backward reads (include/mmt_amd.h).
static bool stash_static_ok(const mmt_decoder *dec, int map_dim, int B, int T, int ph) {
  return dec && B > 0 && T > 0 && ph > 0 && map_dim >= 0 && (map_dim > 0) == (dec->has_map_fusion != 0) &&
         mmt_cluster_decoder_supports(dec, map_dim, true, T, ph);
}
extern "C" int mmt_decoder_stash_supported(const mmt_decoder *dec, int map_dim, int B, int T, int ph, int with_dropout) {
  if (!stash_static_ok(dec, map_dim, B, T, ph)) return 0;
  const int impl = g_decoder_impl_atomic.load(std::memory_order_relaxed);
  if (impl != MMT_DECODER_AUTO && impl != MMT_DECODER_CLUSTER && impl != MMT_DECODER_CLUSTER_PIPELINE) return 0;
  return mmt_cluster_decoder_stash_ok(dec, map_dim, B, T, ph, with_dropout != 0) ? 1 : 0;
}
extern "C" int mmt_decoder_stash_layout_of(const mmt_decoder *dec, int map_dim, int B, int T, int ph, mmt_decoder_stash_layout *out) {
  if (!out || !stash_static_ok(dec, map_dim, B, T, ph))
    return fail(MMT_EINVAL, "mmt_decoder_stash_layout_of: bad arguments or unsupported geometry");
  mmt_cluster_decoder_stash_layout(dec, map_dim, B, T, ph, out);
  return MMT_OK;
}
extern "C" int mmt_decoder_forward_stash(const mmt_decoder *dec, const float *memory, const uint8_t *pad_mask, const float *start_pos,
                                         const float *map_feat, int map_dim, const float *mem_feat, int mem_feat_dim, int B, int T,
                                         int ph, float *pred_st, float *workspace, const mmt_decoder_dropout *drop, float *stash,
                                         size_t stash_floats, void *stream) {
  if (!stash || ((uintptr_t)stash & 15)) return fail(MMT_EINVAL, "mmt_decoder_forward_stash: stash must be a 16-byte aligned device buffer");
  if (!mmt_decoder_stash_supported(dec, map_dim, B, T, ph, drop != nullptr))
    return fail(MMT_EINVAL, "mmt_decoder_forward_stash: not supported for this decoder / batch / device state (mmt_decoder_stash_supported)");
  mmt_decoder_stash_layout lay;
  mmt_cluster_decoder_stash_layout(dec, map_dim, B, T, ph, &lay);
  if (stash_floats < lay.total_floats)
    return fail(MMT_EINVAL, "mmt_decoder_forward_stash: stash of %zu floats, need %zu", stash_floats, lay.total_floats);
  return decoder_forward_impl(dec, memory, pad_mask, start_pos, map_feat, map_dim, mem_feat, mem_feat_dim, nullptr, B, T, ph, pred_st,
                              nullptr, workspace, stream, drop, stash);
}

// ---- the same decode for callers that cannot allocate (include/mmt_amd.h: mmt_decoder_forward_static) ----------------------
// The workspace is a file-scope device array.  Its NAME is only a host-side stand-in for the symbol: the device address is
// looked up once per device with hipGetSymbolAddress (each GPU holds its own copy at its own address) and only that pointer is
// ever handed to kernels.
static __device__ __align__(256) float scratch[MMT_DECODER_STATIC_SCRATCH_FLOATS];
static float *g_scratch_ptr[64];
static std::mutex g_scratch_mu;

static int static_scratch(float **out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return fail(MMT_ELAUNCH, "mmt_decoder_forward_static: hipGetDevice: %s", hipGetErrorString(e));
  if (dev < 0 || dev >= 64) return fail(MMT_EINVAL, "mmt_decoder_forward_static: device index %d outside 0..63", dev);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    float *p = nullptr;
    e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));     // with device `dev` current: that device's copy
    if (e != hipSuccess || !p)
      return fail(MMT_ELAUNCH, "mmt_decoder_forward_static: hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return MMT_OK;
}

extern "C" int mmt_decoder_forward_static(const mmt_decoder *dec, const float *memory, const uint8_t *pad_mask,
                                          const float *start_pos, const float *map_feat, int map_dim, const float *mem_feat,
                                          int mem_feat_dim, const float *tgt, int B, int T, int ph, float *pred_st, float *h_out,
                                          void *stream) {
  if (!dec) return fail(MMT_EINVAL, "mmt_decoder_forward_static: null pointer");
  if (B <= 0 || T <= 0 || ph <= 0) return fail(MMT_EINVAL, "mmt_decoder_forward_static: bad sizes B=%d T=%d ph=%d", B, T, ph);
  const size_t need = mmt_decoder_workspace_floats(dec, B, T, ph);
  if (need == 0 || need > (size_t)MMT_DECODER_STATIC_SCRATCH_FLOATS)
    return fail(MMT_EINVAL, "mmt_decoder_forward_static: B=%d T=%d ph=%d needs %zu floats of scratch, the static buffer holds %zu "
                            "(use mmt_decoder_forward with a caller workspace)", B, T, ph, need, (size_t)MMT_DECODER_STATIC_SCRATCH_FLOATS);
  float *ws = nullptr;
  if (int rc = static_scratch(&ws)) return rc;
  // nothing in the scratch has to start at zero: every decode zeroes its own hand-off counters (cross-K/V kernel) and writes
  // each partial / cache row before it reads it
  return mmt_decoder_forward(dec, memory, pad_mask, start_pos, map_feat, map_dim, mem_feat, mem_feat_dim, tgt, B, T, ph, pred_st,
                             h_out, ws, stream);
}

extern "C" int mmt_finalize(const float *pred_st, const float *node_pos, const float *gt, int metric_in_metres, int B,
                            int ph, float *pred_m, float *ade, float *fde, float *loss_sum, void *stream) {
  if (!pred_st || B <= 0 || ph <= 0) return fail(MMT_EINVAL, "mmt_finalize: bad arguments");
  if ((pred_m || metric_in_metres) && !node_pos) return fail(MMT_EINVAL, "mmt_finalize: node_pos required for metres");
  if ((ade || fde || loss_sum) && !gt) return fail(MMT_EINVAL, "mmt_finalize: metrics need gt");
  hipLaunchKernelGGL(mmt_finalize_kernel, dim3(loss_sum ? 1 : (B + 255) / 256), dim3(loss_sum ? 1024 : 256), 0, (hipStream_t)stream,
                     pred_st, node_pos, gt, metric_in_metres, B, ph, pred_m, ade, fde, loss_sum);
  return check_launch("mmt_finalize");
}
