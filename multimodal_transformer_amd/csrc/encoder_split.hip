// History encoder (Transformer_model.py:11-59) with every agent tile shared by TWO workgroups (gfx950 / MI355X).
//
// Why: mmt_encoder_kernel gives one 16-row tile (one agent's T <= 16 tokens) to one workgroup, which streams all 12.7 MB of
// encoder weights; its time is the MFMA issue time of those weights on ONE CU (a 16x16x4 MFMA consumes 64 weights per 32
// cycles per SIMD whether 9 or 16 rows are real), and at the benchmark batch (128 agents) half of the 256 CUs idle.  Here the
// two workgroups of a tile split every layer Megatron-style:
//   QKV            : workgroup h computes q, k, v of heads [h*H/2, (h+1)*H/2)         (column split, no exchange)
//   attention      : its own heads                                                      (local)
//   out-projection : K-slice = its heads' columns of the attention output  -> partial [T][d]   --- exchange 1 ---
//   x = LN(x + bias + p0 + p1)   (both workgroups, fixed order: identical bits on both sides)
//   linear1 + ReLU : hidden columns [h*f/2, (h+1)*f/2)                                  (column split)
//   linear2        : K-slice = its hidden columns                          -> partial [T][d]   --- exchange 2 ---
//   x = LN(x + bias + p0 + p1)
// Each workgroup streams half of the weights; two 9 KB exchanges per layer go through memory.
//
// Exchange protocol (cdna_hip_programming.md Guideline 16, form R1): the partial is stored write-through (sc1, 16 bytes per
// lane), every wave drains its stores, workgroup barrier, ONE lane stores the workgroup's progress word (sc1); the partner
// polls that ONE word (relaxed sc1 loads, bounded), workgroup barrier, then loads the partial with sc1 loads (past the L1).
// Progress words only ever grow: both workgroups of a tile complete the same number of exchanges per call, so at kernel entry
// a workgroup's OWN word (which only it writes) is the base both count from - nothing is reset per call, nothing is frozen
// into a graph.  The caller zeroes the workspace ONCE (and again after a call that reported a time-out).
// Both workgroups of every tile must be resident at once: the host only takes this path when 2 * tiles <= #CUs.
#include "mmt_device.cuh"

#include <algorithm>
#include <atomic>
#include <cstdlib>

int mmt_set_error(int code, const char *fmt, ...);   // transformer.hip

// Sticky fault word of this file's kernel (a workgroup gave up waiting for its partner): read and cleared through
// mmt_split_encoder_take_fault() by mmt_decoder_status().
__device__ unsigned g_encoder_fault;

namespace {

constexpr int ENT = 512;                             // threads per workgroup
constexpr unsigned long long ENC_WAIT_TICKS = 20000000ull;     // 0.2 s of the 100 MHz wall clock

// which slabs (16 output columns) and chunks (16 k) of a packed linear a workgroup multiplies: virtual slab v ->
// real slab (v / grp) * stride + base + v % grp; chunks [c_off, c_off + K16) of the K16full the rows are laid out with
struct LinG {
  const float *w;
  int K16, K16full, c_off, N16, grp, stride, base, v0;      // v0: first virtual slab of this call (a linear may be run in two calls)
  __device__ __forceinline__ int slab(int v) const {
    v = max(min(v, N16 - 1), 0) + v0;                 // a dead slab re-reads the last one (in bounds, discarded)
    return (v / grp) * stride + base + v % grp;
  }
};
__device__ __forceinline__ LinG no_lin() { return LinG{nullptr, 0, 0, 0, 0, 1, 0, 0, 0}; }

template <int SL = WSL>
__device__ __forceinline__ void wload_g(WRegs &r, const f32x4 *W4, const LinG &G, int s0, int c0) {
#pragma unroll
  for (int s = 0; s < SL; ++s) {
    const f32x4 *p = W4 + ((size_t)G.slab(s0 + s) * G.K16full + G.c_off + c0) * 64;
#pragma unroll
    for (int u = 0; u < WU; ++u) r.w[s][u] = p[(size_t)u * 64];
  }
}

// tile_linear (mmt_device.cuh) for one 16-row tile with a slab / chunk selection; K16 must be a multiple of WU.
// X columns are REAL k positions (the chunk offset applies to X too); epilogue columns are REAL output columns.
// SL slabs per wave per set (the linear that follows is prefetched with its own NSL).
template <bool PRE, int SL, int NSL, class Epi>
__device__ __forceinline__ void tile_linear_g(const float *__restrict__ X, int ldx, const LinG G, Epi epi, WRegs &pre, const LinG next) {
  const int lane = lane_id(), nw = n_waves(), first = wave_id() * SL;
  const float *xrow = X + (lane & 15) * ldx + (lane >> 4) * 4 + G.c_off * 16;
  const f32x4 *W4 = reinterpret_cast<const f32x4 *>(G.w) + lane;
  const f32x4 *N4 = reinterpret_cast<const f32x4 *>(next.w) + lane;
  const bool do_next = next.w != nullptr;
  f32x4 acc[SL];
#pragma unroll
  for (int s = 0; s < SL; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
  typename Epi::template Pre<1> pb[SL];
  const int nsets = first < G.N16 ? (G.N16 - first + nw * SL - 1) / (nw * SL) : 0;
  const int gps = G.K16 / WU, total = nsets * gps;
  WRegs cur;
  if (PRE) cur = pre;
  else if (total > 0) wload_g<SL>(cur, W4, G, first, 0);
  int set = 0, kg = 0;
  for (int g = 0; g < total; ++g) {
    int kg1 = kg + 1, set1 = set;
    if (kg1 == gps) {
      kg1 = 0;
      ++set1;
    }
    WRegs nxt = cur;
    if (g + 1 < total) wload_g<SL>(nxt, W4, G, first + set1 * nw * SL, kg1 * WU);
    else if (do_next) wload_g<NSL>(nxt, N4, next, wave_id() * NSL, 0);
    const int s0 = first + set * nw * SL;
    if (kg == gps - 1) {
#pragma unroll
      for (int s = 0; s < SL; ++s) epi.prep(pb[s], (lane >> 4) * 4, G.slab(s0 + s) * 16 + (lane & 15));
    }
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const f32x4 xv = *reinterpret_cast<const f32x4 *>(xrow + (kg * WU + u) * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int s = 0; s < SL; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[e], cur.w[s][u][e], acc[s], 0, 0, 0);
    }
    if (kg == gps - 1) {
#pragma unroll
      for (int s = 0; s < SL; ++s) {
        if (s0 + s < G.N16) {
          const int col = G.slab(s0 + s) * 16 + (lane & 15);
#pragma unroll
          for (int r = 0; r < 4; ++r) epi.store((lane >> 4) * 4 + r, col, acc[s][r], pb[s], 0, r);
        }
        acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    cur = nxt;
    kg = kg1;
    set = set1;
  }
  if (total == 0 && do_next) wload_g<NSL>(cur, N4, next, wave_id() * NSL, 0);
  pre = cur;
}

struct InEpi {      // input_fc: + bias + positional encoding (Transformer_model.py:48-49)
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

struct SplitArgs {
  int B, T, ldx, ldb, ncu_pad;
  float *slots;            // [tile][half][2][16][d]  partials on their way to the partner
  unsigned *flags;         // [tile][half] progress words, one 128-byte line each
};

struct Xchg {
  gu32 *mine, *theirs;
  float *out[2];           // my slot, by parity of the exchange
  const float *in[2];      // the partner's slot
  unsigned base, e;
  int *abort;              // LDS
};

// my partial (LDS tile pb, rows < nrows) -> my slot, write-through; then the progress word
__device__ __forceinline__ void publish(Xchg &X, const float *pb, int ldp, int nrows, int d) {
  float *dst = X.out[X.e & 1];
  const int nq = d >> 2;
  for (int i = threadIdx.x; i < nrows * nq; i += ENT) {
    const int r = i / nq, q = i - r * nq;
    bs4x<AUX_SC1>(mk_rsrc(dst + (size_t)r * d), (unsigned)q * 16u, *reinterpret_cast<const f32x4 *>(pb + r * ldp + 4 * q));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave, before the barrier
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(X.mine, X.base + X.e + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until the partner has published exchange e; false = gave up
__device__ __forceinline__ bool await(Xchg &X) {
  if (threadIdx.x == 0) {
    const unsigned target = X.base + X.e + 1u;
    if ((int)(__hip_atomic_load(X.theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      const unsigned long long t0 = wall_clock64();
      for (;;) {
        __builtin_amdgcn_s_sleep(1);
        if ((int)(__hip_atomic_load(X.theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) break;
        if (wall_clock64() - t0 > ENC_WAIT_TICKS) {
          *X.abort = 1;
          break;
        }
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");        // compiler only: no load of handed-off bytes moves above the poll
  return *X.abort == 0;
}

// xs = LayerNorm(xs + bias + p0 + p1) in ONE pass: 32 lanes per row (16 rows = the 512 threads), two float4 column groups per
// lane (d <= 256); p(mine) from the LDS tile, p(partner) from its slot (sc1 loads).  The two partials are added in the order of
// the halves and the row statistics are reduced in a fixed order, so both workgroups of a tile hold the same bits.
__device__ __forceinline__ void combine_norm(Xchg &X, int half, float *xs, int ldx, const float *pb, int ldp, const float *bias,
                                             const mmt_norm &N, int nrows, int d) {
  const float *src = X.in[X.e & 1];
  const int nq = d >> 2, row = threadIdx.x >> 5, l32 = threadIdx.x & 31;
  const bool live = row < nrows;
  f32x4 v[2];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = l32 + 32 * j;
    const bool ok = live && q < nq;
    const f32x4 theirs = bl4x<AUX_SC1>(mk_rsrc(src + (size_t)row * d), ok ? (unsigned)q * 16u : OOB);
    v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (ok) {
      const f32x4 mine = *reinterpret_cast<const f32x4 *>(pb + row * ldp + 4 * q);
      const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + 4 * q);
      const f32x4 x = *reinterpret_cast<const f32x4 *>(xs + row * ldx + 4 * q);
      v[j] = (x + b) + ((half == 0 ? mine : theirs) + (half == 0 ? theirs : mine));
    }
    s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 32);
  const float mu = s / (float)d;
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j)
    if (l32 + 32 * j < nq) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dl = v[j][e] - mu;
        var += dl * dl;
      }
    }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) var += __shfl_xor(var, o, 32);
  const float rstd = 1.0f / sqrtf(var / (float)d + MMT_LN_EPS);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = l32 + 32 * j;
    if (live && q < nq) {
      const f32x4 g = *reinterpret_cast<const f32x4 *>(N.g + 4 * q), be = *reinterpret_cast<const f32x4 *>(N.b + 4 * q);
      f32x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = (v[j][e] - mu) * rstd * g[e] + be[e];
      *reinterpret_cast<f32x4 *>(xs + row * ldx + 4 * q) = y;
    }
  }
  __syncthreads();
  ++X.e;
}

template <int DH>
__global__ __launch_bounds__(ENT) void mmt_encoder_split_kernel(mmt_encoder E, SplitArgs P, const float *__restrict__ x_st,
                                                                float *__restrict__ memory, uint8_t *__restrict__ pad_out) {
  extern __shared__ __align__(16) float smem[];
  const int d = E.d_model, H = E.nhead, T = P.T, ldx = P.ldx, ldb = P.ldb, f = E.d_ff;
  float *xs = smem;                    // [16][ldx]  residual stream
  float *buf = xs + 16 * ldx;          // [16][ldb]  q|k|v (real columns), attention output in place of q, FFN hidden
  float *pb = buf + 16 * ldb;          // [16][ldx]  this workgroup's partial of the out-projection / linear2
  uint32_t *padbits = reinterpret_cast<uint32_t *>(pb + 16 * ldx);
  int *abort_flag = reinterpret_cast<int *>(padbits + 1);
  const int tid = threadIdx.x;
  // blocks b and b + 8 share an XCD (speed only): tile = (b / 16) * 8 + b % 8, half = (b / 8) & 1
  const int b = blockIdx.x, tile = (b >> 4) * 8 + (b & 7), half = (b >> 3) & 1;
  if (tile >= P.B) return;             // (grid padded to a multiple of 16; both halves of such a tile leave)
  const int nrows = T;
  const size_t grow0 = (size_t)tile * T;

  for (int i = tid; i < 16 * (2 * ldx + ldb) + 8; i += ENT) smem[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < nrows * E.n_in; i += ENT) {       // nan_to_num(nan=0, +-inf -> +-FLT_MAX)   Transformer_model.py:47
    const int r = i / E.n_in, c = i - r * E.n_in;
    float v = x_st[(grow0 + r) * E.n_in + c];
    if (isnan(v)) v = 0.f;
    else if (isinf(v)) v = v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    buf[r * ldb + c] = v;
  }
  if (tid == 0) {                                          // key padding = NaN in feature column 1   model_utils.py:148-151
    uint32_t bits = 0;
    for (int t = 0; t < T; ++t) {
      const bool p = isnan(x_st[(grow0 + t) * E.n_in + 1]);
      bits |= (p ? 1u : 0u) << t;
      if (pad_out && half == 0) pad_out[grow0 + t] = p ? 1 : 0;
    }
    padbits[0] = bits;
  }
  Xchg X;
  {
    unsigned *fl = P.flags + (size_t)tile * 64;            // two 128-byte lines per tile
    X.mine = (gu32 *)(fl + 32 * half);
    X.theirs = (gu32 *)(fl + 32 * (1 - half));
    X.base = __hip_atomic_load(X.mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // only this workgroup ever writes it
    float *sl = P.slots + (size_t)tile * 2 * 2 * 16 * d;
    for (int k = 0; k < 2; ++k) {
      X.out[k] = sl + ((size_t)half * 2 + k) * 16 * d;
      X.in[k] = sl + ((size_t)(1 - half) * 2 + k) * 16 * d;
    }
    X.e = 0;
    X.abort = abort_flag;
  }
  __syncthreads();
  const int D16 = d >> 4, F16 = f >> 4, hh = H / 2, hs16 = hh * DH / 16;      // hs16: slabs of one of q / k / v for my heads
  const LinG gIn{E.input_fc.w, round16(E.n_in) >> 4, round16(E.n_in) >> 4, 0, D16, D16, 0, 0, 0};
  // q|k|v of my heads = 3 * hs16 slabs; run as a body of whole 2-slabs-per-wave sets and a tail at 1 slab per wave (24 slabs on
  // 8 waves: 16 + 8 instead of two sets of 16 of which the second is half empty)
  const int nw = ENT / 64, qkv_n = 3 * hs16, qkv_body = (qkv_n / (2 * nw)) * (2 * nw), qkv_tail = qkv_n - qkv_body;
  auto gQKV = [&](int L) { return LinG{E.layer[L].qkv.w, D16, D16, 0, qkv_body, hs16, D16, half * hs16, 0}; };
  auto gQKVt = [&](int L) { return LinG{E.layer[L].qkv.w, D16, D16, 0, qkv_tail, hs16, D16, half * hs16, qkv_body}; };
  auto gOut = [&](int L) { return LinG{E.layer[L].out.w, hs16, D16, half * hs16, D16, D16, 0, 0, 0}; };
  auto gFf1 = [&](int L) { return LinG{E.layer[L].ff1.w, D16, D16, 0, F16 / 2, F16 / 2, 0, half * (F16 / 2), 0}; };
  auto gFf2 = [&](int L) { return LinG{E.layer[L].ff2.w, F16 / 2, F16, half * (F16 / 2), D16, D16, 0, 0, 0}; };
  WRegs pre;
  {
    // input_fc has K16 = 1 < WU: plain loop (tile_linear's small-K path), both workgroups compute all of it
    InEpi epi{E.input_fc.b, E.pe, xs, d, T, ldx};
    WRegs scratch;
    tile_linear<1, false>(buf, ldb, E.input_fc.w, gIn.K16, D16, epi, scratch, no_next());
    wload_g<2>(pre, reinterpret_cast<const f32x4 *>(E.layer[0].qkv.w) + lane_id(), gQKV(0), wave_id() * 2, 0);
  }
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)DH);
  bool ok = true;
  for (int L = 0; L < E.nlayers && ok; ++L) {
    const mmt_enc_layer &W = E.layer[L];
    {
      LdsEpi<EPI_STORE> epi{W.qkv.b, buf, 3 * d, ldb};
      tile_linear_g<true, 2, 1>(xs, ldx, gQKV(L), epi, pre, gQKVt(L));
      tile_linear_g<true, 1, 2>(xs, ldx, gQKVt(L), epi, pre, gOut(L));
    }
    __syncthreads();
    for (int g = tid >> 4; g < nrows * hh; g += ENT >> 4) {
      const int r = g / hh, h = half * hh + (g - r * hh);
      float *q = buf + r * ldb + h * DH;
      const float *kb = buf + d + h * DH;
      attend_group<DH>(q, kb, kb + d, ldb, r + 1, padbits[0], scale, q);   // causal: keys 0..r
    }
    __syncthreads();
    {
      LdsEpi<EPI_STORE> epi{nullptr, pb, d, ldx};
      tile_linear_g<true, 2, 2>(buf, ldb, gOut(L), epi, pre, gFf1(L));
    }
    __syncthreads();
    publish(X, pb, ldx, nrows, d);
    ok = await(X);
    if (!ok) break;
    combine_norm(X, half, xs, ldx, pb, ldx, W.out.b, W.ln1, nrows, d);
    {
      LdsEpi<EPI_RELU> epi{W.ff1.b, buf, f, ldb};
      tile_linear_g<true, 2, 2>(xs, ldx, gFf1(L), epi, pre, gFf2(L));
    }
    __syncthreads();
    {
      LdsEpi<EPI_STORE> epi{nullptr, pb, d, ldx};
      tile_linear_g<true, 2, 2>(buf, ldb, gFf2(L), epi, pre, L + 1 < E.nlayers ? gQKV(L + 1) : no_lin());
    }
    __syncthreads();
    publish(X, pb, ldx, nrows, d);
    ok = await(X);
    if (!ok) break;
    combine_norm(X, half, xs, ldx, pb, ldx, W.ff2.b, W.ln2, nrows, d);
  }
  if (!ok) {                              // the partner never arrived: fail loudly (NaN rows + the sticky fault word)
    if (tid == 0) atomicOr(&g_encoder_fault, 1u);
    for (int i = tid; i < nrows * d; i += ENT) memory[grow0 * d + i] = __builtin_nanf("");
    return;
  }
  // both workgroups hold the same rows: each writes half of the columns
  const int dq = d >> 3;                  // float4s per row per half
  for (int i = tid; i < nrows * dq; i += ENT) {
    const int r = i / dq, q = half * dq + (i - r * dq);
    reinterpret_cast<f32x4 *>(memory + (grow0 + r) * d)[q] = *reinterpret_cast<const f32x4 *>(xs + r * ldx + 4 * q);
  }
}

std::atomic<bool> g_split_disabled[64];

}  // namespace

// The split form needs: one tile per agent (T <= 16), an even head count, d and d_ff/2 multiples of 16 * WU, and every
// workgroup resident (2 * B <= #CUs).
static bool split_ok(const mmt_encoder *enc, int B, int T, int ncu) {
  const int d = enc->d_model, H = enc->nhead, f = enc->d_ff;
  if (T > 16 || (H & 1) || d % H) return false;
  const int dh = d / H;
  if (dh != 16 && dh != 32 && dh != 64) return false;
  if ((d / 16) % WU || ((H / 2) * dh / 16) % WU || (f % 32) || ((f / 32) % WU) || d % 8) return false;
  const int grid = ((B + 7) / 8) * 16;
  return grid <= ncu;
}

// Workspace layout: [progress words: ENC_MAX_TILES tiles x 2 lines of 128 bytes][slots].  The words sit at offsets that do not
// depend on the batch size, so calls with different B may share one (zeroed-once) workspace: slot data never lands on a word.
constexpr size_t ENC_MAX_TILES = 512;
extern "C" size_t mmt_encoder_workspace_floats(const mmt_encoder *enc, int B, int T) {
  if (!enc || B <= 0) return 0;
  const size_t tiles = ((size_t)(B + 7) / 8) * 8;
  return ENC_MAX_TILES * 64 + tiles * 2 * 2 * 16 * (size_t)enc->d_model + 64;
}

// 1 when a split-encoder launch on the current device timed out since the last call (its memory rows are NaN); clears the word
// and keeps the split form off on this device.
unsigned mmt_split_encoder_take_fault() {
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_encoder_fault), sizeof(v)) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  if (v) {
    const unsigned zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_encoder_fault), &zero, sizeof(zero));
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64) g_split_disabled[dev].store(true);
  }
  return v;
}
void mmt_split_encoder_disable() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64) g_split_disabled[dev].store(true);
}
void mmt_split_encoder_enable() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64) g_split_disabled[dev].store(false);
}

// returns 1 when it launched, 0 when this geometry / batch takes the one-workgroup kernel, < 0 on error
int mmt_encoder_forward_split(const mmt_encoder *enc, const float *x_st, int B, int T, float *memory, uint8_t *pad_mask,
                              float *workspace, void *stream) {
  int dev = 0, ncu = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  static const bool env_off = getenv("MMT_NO_SPLIT_ENCODER") != nullptr;      // read once
  if (dev < 0 || dev >= 64 || env_off || g_split_disabled[dev].load() || !split_ok(enc, B, T, ncu)) return 0;
  const int d = enc->d_model;
  const size_t tiles = ((size_t)(B + 7) / 8) * 8;
  SplitArgs P;
  P.B = B;
  P.T = T;
  P.ldx = d + MMT_PAD;
  P.ldb = std::max(std::max(3 * d, (enc->d_ff + 15) & ~15), (enc->n_in + 15) & ~15) + MMT_PAD;
  P.ncu_pad = 0;
  P.flags = reinterpret_cast<unsigned *>((reinterpret_cast<uintptr_t>(workspace) + 127) & ~(uintptr_t)127);
  P.slots = reinterpret_cast<float *>(P.flags + ENC_MAX_TILES * 64);
  if (tiles > ENC_MAX_TILES) return 0;
  const size_t lds = ((size_t)16 * (2 * P.ldx + P.ldb) + 64) * sizeof(float);
  if (lds > 160 * 1024) return 0;
  const int grid = (int)tiles * 2, dh = d / enc->nhead;
#define SPLIT_LAUNCH(DHV)                                                                                               \
  do {                                                                                                                  \
    (void)hipFuncSetAttribute((const void *)mmt_encoder_split_kernel<DHV>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                              (int)lds);                                                                                \
    int per_cu = 0;                                                                                                     \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mmt_encoder_split_kernel<DHV>, ENT, lds) != hipSuccess || \
        per_cu < 1 || grid > per_cu * ncu) {                                                                            \
      (void)hipGetLastError();                                                                                          \
      return 0;                                                                                                         \
    }                                                                                                                   \
    hipLaunchKernelGGL((mmt_encoder_split_kernel<DHV>), dim3(grid), dim3(ENT), lds, (hipStream_t)stream, *enc, P, x_st, \
                       memory, pad_mask);                                                                               \
  } while (0)
  if (dh == 16) SPLIT_LAUNCH(16); else if (dh == 32) SPLIT_LAUNCH(32); else SPLIT_LAUNCH(64);
#undef SPLIT_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mmt_set_error(MMT_ELAUNCH, "mmt_encoder_forward_ws: %s", hipGetErrorString(e));
  return 1;
}
