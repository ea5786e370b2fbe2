// Cluster decoder for gfx950 (MI355X): the KV-cached autoregressive decode (mgcvae.py:375-449,
// Transformer_model.py:62-109,155-199, MLP.py:6-30) as a chain of short, wide kernels.
//
// Why: one new token per agent per step leaves only B rows of work per sub-layer.  The persistent kernel
// (transformer.hip) gives every workgroup the whole weight stream (18.9 MB per step through one CU, which
// takes in ~50 GB/s); here RC <= 16 agents form a "cluster" whose sub-layer is split Megatron-style over
// H workgroups:
//   phase A (self attention)  : workgroup h computes q_h,k_h,v_h, appends to the KV cache, attends, and
//                               multiplies by the K-slice h of the out-projection     -> partial [RC][d]
//   phase B (cross attention) : workgroup h: q_h, attention on the precomputed memory K/V, K-slice of out-proj
//   phase C (feed forward)    : workgroup g: d_ff/H hidden columns (ReLU), K-slice of linear2
// Every phase starts by summing the previous phase's H partials (+ bias + residual) and applying the
// LayerNorm, redundantly in each workgroup of the cluster (fixed order: bit-reproducible and independent
// of the batch size and of RC).  A kernel boundary is the all-reduce: 3 launches per layer per step, no
// in-kernel spinning.  Everything a workgroup needs from memory (its weight slice <= 128 KB, biases, the
// partials, its heads' K/V cache rows) is requested at kernel entry in one burst; K/V rows are read as
// whole 128-byte lines and staged in LDS.
// The head (output_fc, optional map fusion, decoder MLP, position feedback) and the embedding of the next
// token ride at the front of the next step's first phase A.
#define MMT_OPAQUE_TID
#include "mmt_device.cuh"
#include "mmt_gfx950.cuh"

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <mutex>
#include <type_traits>
#include <cstdint>
#include <cstdlib>
#include <cstring>

int mmt_set_error(int code, const char *fmt, ...);   // transformer.hip
void mmt_note_decoder_path(int path);                // transformer.hip: what mmt_decoder_last_path() reports

// Sticky per-device fault word: raised by the final head launch of a decode whose in-kernel hand-offs timed out (the outputs of
// that call are NaN).  Read and cleared by mmt_decoder_status().
__device__ unsigned g_decoder_fault;
__device__ unsigned g_decoder_fault_info[4];   // first give-up of the pipeline kernel: {1, workgroup, phase, cluster}

#ifdef MMT_PROFILE
// diagnostic build: wall-clock (100 MHz) stamps per launch / workgroup / stage, read by tools/cluster_profile.py
#define CLP_LAUNCHES 256
#define CLP_WGS 64
__device__ unsigned long long g_clprof[CLP_LAUNCHES][CLP_WGS][8];
#define CSTAMP(k) do { if (thread_idx() == 0 && blockIdx.x < CLP_WGS && a.prof_idx < CLP_LAUNCHES) g_clprof[a.prof_idx][blockIdx.x][k] = wall_clock64(); } while (0)
extern "C" int mmt_debug_read_clprof(unsigned long long *host, size_t n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_clprof), sizeof(unsigned long long) * n);
}
// pipeline kernel: stamps per workgroup and visit (step * 16 + local cluster index), read by tools/pipe_profile.py
#define PPR_WGS 256
#define PPR_VISITS 192
__device__ unsigned long long g_pprof[PPR_WGS][PPR_VISITS][16];
#define PSTAMP(visit, k) do { const int v_ = ((visit) >> 12) * 16 + ((visit) & 4095); if (thread_idx() == 0 && blockIdx.x < PPR_WGS && ((visit) & 4095) < 16 && v_ < PPR_VISITS) g_pprof[blockIdx.x][v_][k] = wall_clock64(); } while (0)
#define PSTAMPC(visit, k) do { const int v_ = ((visit) >> 12) * 16 + ((visit) & 4095); if (thread_idx() == 0 && blockIdx.x < PPR_WGS && ((visit) & 4095) < 16 && v_ < PPR_VISITS) g_pprof[blockIdx.x][v_][k] = clock64(); } while (0)
extern "C" int mmt_debug_read_pprof(unsigned long long *host, size_t n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pprof), sizeof(unsigned long long) * n);
}
#else
#define CSTAMP(k) do { } while (0)
#define PSTAMP(visit, k) do { } while (0)
#define PSTAMPC(visit, k) do { } while (0)
#endif

namespace {

constexpr int CNT = 512;          // threads per workgroup
constexpr int CNW = 8;            // waves per workgroup
constexpr int KS_SMAX = 6;        // most 16-column slabs a K-split linear produces (3*DH/16 at DH = 32)
constexpr int LDO = 72;           // row stride of the small LDS tiles (<= 64 columns + 8)
constexpr int LDTOK = 24;
constexpr int CHAIN_MAX = 4;
constexpr int KMAX = 16;          // most keys an attention sees (history / horizon <= 16; else: persistent kernel)

enum { XM_EMBED = 0, XM_REDUCE = 1, XM_HEAD_EMBED = 2 };
enum { PH_A = 0, PH_B = 1, PH_C = 2, PH_H = 3 };

// The first 160 bytes are what every launch needs before it can request anything from memory; the kernel pulls them
// in with one batch of scalar loads (hot_args).  The host resolves the per-phase pointers.
struct ClArgs {
  int B, T, ph, d, H, RC, s, L, n_in, n_out, FS16, map_dim, has_map_fusion, prof_idx;
  int n_chain;                      // the small linears behind output_fc: fusion/hist_map then decoder/MLP
  int tagged;                       // 4-agent-cluster kernels: rows handed from phase to phase carry the phase's tag bit (below)
  const float *part_in, *xres_in;
  float *part_out, *xres_out;
  const float *red_bias;
  mmt_norm red_ln;
  const float *w1, *b1, *w2;        // first linear (+ bias) and second linear of this phase, in the order the kernel variant reads
  float *kv;                        // K/V cache of this phase: selfkv (A) or memkv (B)
  const uint8_t *pad_mask;
  mmt_linear input_fc, output_fc;
  const float *pe;
  mmt_linear chain[CHAIN_MAX];
  mmt_norm chain_ln[CHAIN_MAX];     // LayerNorm + ReLU after chain[i] when chain_ln[i].g != NULL
  const float *start_pos, *tgt, *map_feat;
  float *pred_st, *h_out;
  const unsigned *tmo;              // time-out word of the fused kernel (NULL: launch-per-phase path); read by the final head launch
  // training-mode dropout (mmt_decoder_forward_dropout; DROP instantiations of the launch-per-phase kernels only): keep bytes of
  // THIS launch, already offset to its decoder step (NULL: that site is not dropped), and the 1/(1-p) of each
  const uint8_t *dk_red, *dk_pe, *dk_ff, *dk_att;      // sub-layer output summed by this launch's reduce | embedding + pe | FFN hidden | attention probabilities
  float ds_red, ds_pe, ds_ff, ds_att;
};
static_assert(offsetof(ClArgs, input_fc) == 160, "hot block of ClArgs");

// One batch of scalar loads for the hot block: left alone the compiler fetches each field where it is first needed,
// which makes four or five dependent scalar-cache misses (~0.3 us each) before the first vector load can go out.
__device__ __forceinline__ void hot_args(const ClArgs &a) {
  asm volatile("" ::"s"(a.B), "s"(a.T), "s"(a.ph), "s"(a.d), "s"(a.H), "s"(a.RC), "s"(a.s), "s"(a.L), "s"(a.FS16), "s"(a.part_in),
               "s"(a.xres_in), "s"(a.part_out), "s"(a.xres_out), "s"(a.red_bias), "s"(a.red_ln.g), "s"(a.red_ln.b), "s"(a.w1),
               "s"(a.b1), "s"(a.w2), "s"(a.kv), "s"(a.pad_mask));
}

// ------------------------------------------------------------------------------- K-split linear
// Y[16][16*S] = X[16][16*K16] W^T with the K range dealt over the waves (CPW chunks of 16 each): every wave
// requests all of its weights at once (ks_load), multiplies (ks_mfma) and leaves S partial 16x16 tiles in
// LDS; ks_reduce sums them in wave order.
template <int SMAX, int CPW>
struct KsW {
  f32x4 w[SMAX][CPW];
};

template <int SMAX, int CPW, class SlabFn>
__device__ __forceinline__ void ks_load(KsW<SMAX, CPW> &r, const float *Wp, int K16, int S, SlabFn slab_of) {
  const int w = wave_id();
  const rsrc_t rs = mk_rsrc(Wp);
  const unsigned lane_off = lane_id() * 16u;
#pragma unroll
  for (int s = 0; s < SMAX; ++s)
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int c = w * CPW + i;
      r.w[s][i] = bl4(rs, (s < S && c < K16) ? (unsigned)(slab_of(s) * K16 + c) * 1024u + lane_off : OOB);
    }
}

template <int SMAX, int CPW>
__device__ __forceinline__ void ks_mfma(const KsW<SMAX, CPW> &r, const float *X, int ldx, int K16, int S, float *red) {
  const int lane = lane_id(), w = wave_id();
  if (w * CPW >= K16) return;                       // this wave holds no chunk (not read by ks_reduce)
  const float *xrow = X + (lane & 15) * ldx + (lane >> 4) * 4;
  f32x4 acc[SMAX];
#pragma unroll
  for (int s = 0; s < SMAX; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int c = w * CPW + i;
    if (c < K16) {
      const f32x4 xv = *reinterpret_cast<const f32x4 *>(xrow + c * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int s = 0; s < SMAX; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[e], r.w[s][i][e], acc[s], 0, 0, 0);
    }
  }
#pragma unroll
  for (int s = 0; s < SMAX; ++s)
    if (s < S) *reinterpret_cast<f32x4 *>(red + ((w * KS_SMAX + s) * 64 + lane) * 4) = acc[s];
}

// thread -> output element of reduction pass `it`: idx = tid + CNT*it -> (row, col); col < 16*S when valid
__device__ __forceinline__ bool ks_elem(int it, int S, int &row, int &col) {
  const int idx = thread_idx() + CNT * it;
  const int s = idx >> 8, ln = (idx & 255) >> 2, r = idx & 3;
  row = 4 * (ln >> 4) + r;
  col = s * 16 + (ln & 15);
  return idx < S * 256;
}
// epi(row, col, sum, it); NIT passes cover S <= 2*NIT slabs
template <int CPW, int NIT, class Epi>
__device__ __forceinline__ void ks_reduce(const float *red, int K16, int S, Epi epi) {
  const int nwu = min(CNW, (K16 + CPW - 1) / CPW);
  // every LDS read of every pass is issued before the first add (a loop over the nwu tiles pays one LDS round trip per
  // tile and pass); the tiles of waves >= nwu were never written and are read but not added.  Same order of additions.
  float v[NIT][CNW];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = thread_idx() + CNT * it, s = min(idx >> 8, KS_SMAX - 1), rem = idx & 255;
#pragma unroll
    for (int w = 0; w < CNW; ++w) v[it][w] = red[(w * KS_SMAX + s) * 256 + rem];
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = thread_idx() + CNT * it;
    if (idx < S * 256) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < CNW; ++w) sum = w < nwu ? sum + v[it][w] : sum;
      int row, col;
      ks_elem(it, S, row, col);
      epi(row, col, sum, it);
    }
  }
}

// ------------------------------------------------------------------------------- N-split linear
// part[rows][d] = X[16][16*K2] Wslice^T, Wslice = chunks [chunk0, chunk0+K2) of a packed (d x Kfull) matrix; wave w
// owns slabs w and w+8 (d <= 256).  Results go straight from the accumulators to global memory.
template <int NS_KMAX>
struct NsW {
  f32x4 w[2][NS_KMAX];
};
template <int NS_KMAX>
__device__ __forceinline__ void ns_load(NsW<NS_KMAX> &r, const float *Wp, int Kfull16, int chunk0, int K2, int D16) {
  const int w = wave_id();
  const rsrc_t rs = mk_rsrc(Wp);
  const unsigned lane_off = lane_id() * 16u;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int c = 0; c < NS_KMAX; ++c) {
      const int slab = w + CNW * j;
      r.w[j][c] = bl4(rs, (slab < D16 && c < K2) ? (unsigned)(slab * Kfull16 + chunk0 + c) * 1024u + lane_off : OOB);
    }
}
template <int NS_KMAX>
__device__ __forceinline__ void ns_mfma_store(const NsW<NS_KMAX> &r, const float *X, int ldx, int K2, int D16, float *part, int d,
                                              int nrows) {
  const int lane = lane_id(), w = wave_id();
  const float *xrow = X + (lane & 15) * ldx + (lane >> 4) * 4;
  f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int c = 0; c < NS_KMAX; ++c) {
    if (c < K2) {
      const f32x4 xv = *reinterpret_cast<const f32x4 *>(xrow + c * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[e], r.w[j][c][e], acc[j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int slab = w + CNW * j;
    if (slab < D16) {
      const int col = slab * 16 + (lane & 15), row0 = (lane >> 4) * 4;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        if (row0 + rr < nrows) part[(size_t)(row0 + rr) * d + col] = acc[j][rr];
    }
  }
}

// ------------------------------------------------------------------------------- x = LN(res + bias + sum_g part[g])
// 32 lanes per row, two float4 column groups per lane (d <= 256).  Summation order is fixed.
// four keep bytes (one aligned 32-bit load) -> the factors 0 | scale of four consecutive columns
__device__ __forceinline__ f32x4 keep4w(unsigned kb, float scale) {
  return (f32x4){(kb & 0xffu) ? scale : 0.f, (kb & 0xff00u) ? scale : 0.f, (kb & 0xff0000u) ? scale : 0.f, (kb & 0xff000000u) ? scale : 0.f};
}
__device__ __forceinline__ f32x4 keep4(const uint8_t *p, size_t off, float scale) {
  const unsigned kb = *reinterpret_cast<const unsigned *>(p + off);
  return (f32x4){(kb & 0xffu) ? scale : 0.f, (kb & 0xff00u) ? scale : 0.f, (kb & 0xff0000u) ? scale : 0.f, (kb & 0xff000000u) ? scale : 0.f};
}

struct RedRegs {
  f32x4 res[2], bias[2], gam[2], bet[2], p[2][8];
};
__device__ __forceinline__ void reduce_issue(RedRegs &R, const ClArgs &a, int cluster) {
  const int tid = thread_idx(), row = tid >> 5, l32 = tid & 31, nq = a.d >> 2;
  const rsrc_t res = mk_rsrc(a.xres_in + (size_t)cluster * 16 * a.d), pp = mk_rsrc(a.part_in + (size_t)cluster * a.H * 16 * a.d);
  const rsrc_t rb = mk_rsrc(a.red_bias), rg = mk_rsrc(a.red_ln.g), rbt = mk_rsrc(a.red_ln.b);
  const unsigned gstride = 16u * a.d * 4u;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = l32 + 32 * j;
    const bool ok = q < nq && row < a.RC;
    const unsigned ro = ok ? (unsigned)(row * a.d + 4 * q) * 4u : OOB, co = ok ? (unsigned)q * 16u : OOB;
    R.res[j] = bl4(res, ro);
#pragma unroll
    for (int g = 0; g < 8; ++g) R.p[j][g] = bl4(pp, (ok && g < a.H) ? g * gstride + ro : OOB);
    R.bias[j] = bl4(rb, co);
    R.gam[j] = bl4(rg, co);
    R.bet[j] = bl4(rbt, co);
  }
}
// DROP: res + keep * (bias + sum_g part[g]): the summed sub-layer output went through dropout before the residual add
template <bool DROP = false>
__device__ __forceinline__ void reduce_finish(const RedRegs &R, const ClArgs &a, int cluster, float *xs, int ldx, bool write_res) {
  const int tid = thread_idx(), row = tid >> 5, l32 = tid & 31, nq = a.d >> 2;
  const bool live = row < a.RC;
  f32x4 v[2];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    f32x4 acc = R.res[j] + R.bias[j];
    if constexpr (DROP) {
      const int ag = cluster * a.RC + row, q = l32 + 32 * j;
      if (a.dk_red && live && ag < a.B && q < nq) {
        f32x4 t = R.bias[j];
#pragma unroll
        for (int g = 0; g < 8; ++g) t += R.p[j][g];
        acc = R.res[j] + keep4(a.dk_red, (size_t)ag * a.d + 4 * q, a.ds_red) * t;
      } else {
#pragma unroll
        for (int g = 0; g < 8; ++g) acc += R.p[j][g];
      }
    } else
#pragma unroll
    for (int g = 0; g < 8; ++g) acc += R.p[j][g];      // unused slots hold zeros
    v[j] = acc;
    s += (acc[0] + acc[1]) + (acc[2] + acc[3]);
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 32);
  const float mu = s / (float)a.d;
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (l32 + 32 * j < nq) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dlt = v[j][e] - mu;
        var += dlt * dlt;
      }
    }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) var += __shfl_xor(var, o, 32);
  const float rstd = 1.0f / sqrtf(var / (float)a.d + MMT_LN_EPS);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = l32 + 32 * j;
    if (q < nq) {
      f32x4 y = (f32x4){0.f, 0.f, 0.f, 0.f};          // rows beyond the cluster stay zero (finite MFMA operands)
      if (live) {
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (v[j][e] - mu) * rstd * R.gam[j][e] + R.bet[j][e];
      }
      *reinterpret_cast<f32x4 *>(xs + row * ldx + 4 * q) = y;
      if (write_res && live) reinterpret_cast<f32x4 *>(a.xres_out + ((size_t)cluster * 16 + row) * a.d)[q] = y;
    }
  }
}

// ---- tag bits (4-agent-cluster kernels) --------------------------------------------------------------------------------
// Every row that one phase hands to the next (the H partial sums, the residual rows, the embedded token) has the two lowest
// mantissa bits of each float REPLACED by a tag: (p >> 3) & 3 for phase p = step * 3L + 3 * layer + sub-layer (s & 3 for the
// embedded token of step s).  The pipeline kernel rotates its hand-off rows over 8 slots (2 buffers x 4 row blocks of the
// 16-row cluster slots, of which 4-agent clusters use only 4 rows), so consecutive contents of a slot carry consecutive tags,
// and its consumers poll the DATA itself (no counter, no drain): a word is valid when it carries the expected tag and is not
// the 0xFFFFFFFF the buffers are filled with before every decode.  A stage can start polling for a cluster at most 3L phases
// before the rows are written (it has visited the cluster one step earlier): in that window a slot sees <= 3L/8 + 1 <= 4
// generations (L <= 8), which 2 bits tell apart.  The tag is a deterministic change of <= 3 ulp of an intermediate sum; it
// is applied by ALL 4-agent-cluster forms (launch-per-phase, fused, pipeline), which therefore stay bit-identical.
constexpr unsigned TAG_MASK = 3u;
__device__ __forceinline__ float tag1(float v, unsigned tag) {
  return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, v) & ~TAG_MASK) | tag);
}
__device__ __forceinline__ f32x4 tag4(f32x4 v, unsigned tag) {
  return (f32x4){tag1(v[0], tag), tag1(v[1], tag), tag1(v[2], tag), tag1(v[3], tag)};
}
__device__ __forceinline__ bool tag_ok(f32x4 v, unsigned tag) {
  unsigned bad = 0;                      // branch-free: all four words of the 16-byte load are looked at
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned w = __builtin_bit_cast(unsigned, v[e]);
    bad |= ((w ^ tag) & TAG_MASK) | (w == 0xFFFFFFFFu ? 1u : 0u);
  }
  return bad == 0;
}
__device__ __forceinline__ unsigned tag_of_phase(unsigned p) { return (p >> 3) & TAG_MASK; }
__device__ __forceinline__ unsigned phase_tag(const ClArgs &a) { return tag_of_phase((unsigned)a.prof_idx); }   // prof_idx = phase p

// x = tok W_in^T + b + pe[s]  (Transformer_model.py:100-102); tok is a 16x16 LDS tile (n_in <= 16)
// DROP: x = dropout(...) with this step's keep bytes (PositionalEncoding.dropout, Transformer_model.py:126-129)
template <bool SC1 = false, bool DROP = false>
__device__ __forceinline__ void embed_token(const ClArgs &a, int cluster, const float *tok, float *xs, int ldx, bool write_res) {
  const int lane = lane_id(), w = wave_id(), D16 = a.d >> 4;
  const f32x4 *W4 = reinterpret_cast<const f32x4 *>(a.input_fc.w) + lane;
  const f32x4 xv = *reinterpret_cast<const f32x4 *>(tok + (lane & 15) * LDTOK + (lane >> 4) * 4);
  for (int slab = w; slab < D16; slab += CNW) {
    const f32x4 wv = W4[(size_t)slab * 64];
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[e], wv[e], acc, 0, 0, 0);
    const int col = slab * 16 + (lane & 15), row0 = (lane >> 4) * 4;
    const float add = a.input_fc.b[col] + a.pe[(size_t)a.s * a.d + col];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      float y = acc[rr] + add;
      if constexpr (DROP) {
        const int ag = cluster * a.RC + row0 + rr;
        if (a.dk_pe && row0 + rr < a.RC && ag < a.B) y = a.dk_pe[(size_t)ag * a.d + col] ? y * a.ds_pe : 0.f;
      }
      if (a.tagged) y = tag1(y, (unsigned)a.s & TAG_MASK);            // the embedded token of step s
      xs[(row0 + rr) * ldx + col] = y;
      if (write_res && row0 + rr < a.RC) {
        float *dst = a.xres_out + ((size_t)cluster * 16 + row0 + rr) * a.d + col;
        if (a.tagged) y = tag1(y, phase_tag(a));                      // as the residual row of this phase
        if constexpr (SC1) __hip_atomic_store((gf32 *)dst, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
        else *dst = y;
      }
    }
  }
}

// the same with input_fc resident (the pipeline's head workgroups embed one token per visit): wave w owns slabs w and w + 8
// (d <= 256); pe[] = this step's positional-encoding values of the lane's two columns.  Only rows 0..3 are written.
struct EmbedW {
  f32x4 wv[2];
  float b[2];
};
__device__ __forceinline__ void embed_load(EmbedW &E, const ClArgs &a) {
  const int lane = lane_id(), w = wave_id(), D16 = a.d >> 4;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int slab = w + CNW * j;
    E.wv[j] = bl4(mk_rsrc(a.input_fc.w), slab < D16 ? (unsigned)(slab * 64 + lane) * 16u : OOB);
    E.b[j] = bl1(mk_rsrc(a.input_fc.b), slab < D16 ? (unsigned)(slab * 16 + (lane & 15)) * 4u : OOB);
  }
}
__device__ __forceinline__ void embed_pe(float (&pe)[2], const ClArgs &a, int s) {
  const int lane = lane_id(), w = wave_id(), D16 = a.d >> 4;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int slab = w + CNW * j;
    pe[j] = bl1(mk_rsrc(a.pe + (size_t)s * a.d), slab < D16 ? (unsigned)(slab * 16 + (lane & 15)) * 4u : OOB);
  }
}
// kpe (DROP): keep bytes of the lane's 2 x 4 outputs, one word per slab (byte rr = row rr), loaded by embed_keep
template <bool DROP = false>
__device__ __forceinline__ void embed_token_res(const ClArgs &a, const EmbedW &E, const float (&pe)[2], const float *tok, float *xs, int ldx,
                                                const unsigned (&kpe)[2] = {0u, 0u}) {
  const int lane = lane_id(), w = wave_id(), D16 = a.d >> 4;
  const f32x4 xv = *reinterpret_cast<const f32x4 *>(tok + (lane & 15) * LDTOK + (lane >> 4) * 4);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int slab = w + CNW * j;
    if (slab < D16) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[e], E.wv[j][e], acc, 0, 0, 0);
      const float add = E.b[j] + pe[j];
      if (lane < 16) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          float y = acc[rr] + add;
          if constexpr (DROP) {
            if (a.dk_pe) y = ((kpe[j] >> (8 * rr)) & 0xffu) ? y * a.ds_pe : 0.f;     // (rows beyond the batch are never read)
          }
          if (a.tagged) y = tag1(y, (unsigned)a.s & TAG_MASK);
          xs[rr * ldx + slab * 16 + lane] = y;
        }
      }
    }
  }
}

// the keep bytes embed_token_res<true> applies: lane < 16 owns column slab*16 + lane of rows 0..3 (agents a0..a0+3)
__device__ __forceinline__ void embed_keep(unsigned (&kpe)[2], const ClArgs &a, int a0, int na) {
  const int lane = lane_id(), w = wave_id(), D16 = a.d >> 4;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int slab = w + CNW * j;
    kpe[j] = 0u;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const unsigned v = __builtin_amdgcn_raw_buffer_load_b8(mk_rsrc(a.dk_pe), (a.dk_pe && slab < D16 && lane < 16 && rr < na)
                                                                 ? (unsigned)((a0 + rr) * a.d + slab * 16 + lane) : OOB, 0, 0);
      kpe[j] |= (v & 0xffu) << (8 * rr);
    }
  }
}

__device__ __forceinline__ void small_load(KsW<4, 1> &r, const mmt_linear &lin) {
  ks_load<4, 1>(r, lin.w, round16(lin.k) >> 4, round16(lin.n) >> 4, [](int s) { return s; });
}

// K/V cache rows of a cluster, staged through registers: unit u -> (agent, key, k|v, 16-byte piece) of the padded
// [RC][KMAX][2][DH/4] index space (shifts only; the DH/4 lanes of a row read one contiguous 4*DH-byte line).
template <int DH>
__device__ __forceinline__ void kv_unit(int u, int &ag, int &j, int &kv, int &c4) {
  static_assert(KMAX == 16, "kv_unit");
  c4 = u % (DH / 4);
  const int row = u / (DH / 4);
  kv = row & 1;
  j = (row >> 1) & 15;
  ag = row >> 5;
}

// what a phase requests from memory for its own sub-layer
template <int DH, int PHASE>
struct PhaseGeom {      // register tiles sized per phase: every load instruction of the burst is a useful one
  static constexpr int S1MAX = PHASE == PH_A ? 3 * DH / 16 : PHASE == PH_B ? DH / 16 : 4;
  static constexpr int K2MAX = PHASE == PH_C ? 4 : DH / 16;
};
template <int DH, int CPW, int PHASE>
struct MainRegs {
  KsW<PhaseGeom<DH, PHASE>::S1MAX, CPW> w1;
  NsW<PhaseGeom<DH, PHASE>::K2MAX> w2;
  float b1[3];               // bias of the first linear for this thread's reduction elements
  f32x4 kv[DH / 4];          // K/V cache rows (whole 128-byte lines), staged into LDS later
  int pad;                   // key-padding flag of (agent tid>>4, key tid&15), phase B
};

template <int DH, int CPW, int PHASE>
__device__ __forceinline__ void main_issue(MainRegs<DH, CPW, PHASE> &M, const ClArgs &a, int a0, int na, int g) {
  const int d = a.d, D16 = d >> 4, K16 = D16;
  constexpr int S_A = 3 * DH / 16, S_B = DH / 16;
  if constexpr (PHASE == PH_A) {
    ks_load<PhaseGeom<DH, PHASE>::S1MAX, CPW>(M.w1, a.w1, K16, S_A, [&](int s) { return (s / S_B) * D16 + g * S_B + (s % S_B); });
  } else if constexpr (PHASE == PH_B) {
    ks_load<PhaseGeom<DH, PHASE>::S1MAX, CPW>(M.w1, a.w1, K16, S_B, [&](int s) { return g * S_B + s; });
  } else {
    ks_load<PhaseGeom<DH, PHASE>::S1MAX, CPW>(M.w1, a.w1, K16, a.FS16, [&](int s) { return g * a.FS16 + s; });
  }
  const int S1 = PHASE == PH_A ? S_A : PHASE == PH_B ? S_B : a.FS16;
  const rsrc_t rb1 = mk_rsrc(a.b1);
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    int row, col;
    const bool ok = ks_elem(it, S1, row, col);
    int bi;
    if constexpr (PHASE == PH_A) bi = (col / DH) * d + g * DH + (col % DH);
    else if constexpr (PHASE == PH_B) bi = g * DH + col;
    else bi = g * a.FS16 * 16 + col;
    M.b1[it] = bl1(rb1, ok ? (unsigned)bi * 4u : OOB);
  }
  if constexpr (PHASE == PH_A || PHASE == PH_B) {
    // unit u -> (agent, key, k|v, 16-byte piece): the DH/4 lanes of a row read one contiguous 4*DH-byte line
    const int nglob = PHASE == PH_A ? a.s : a.T;
    const float *base = a.kv + (((size_t)a.L * a.B + a0) * (PHASE == PH_A ? a.ph : a.T)) * 2 * d;
    const unsigned astride = (unsigned)(PHASE == PH_A ? a.ph : a.T) * 2 * d;
    const rsrc_t rkv = mk_rsrc(base);
#pragma unroll
    for (int it = 0; it < DH / 4; ++it) {
      int ag, j, kv, c4;
      kv_unit<DH>(thread_idx() + CNT * it, ag, j, kv, c4);
      M.kv[it] = bl4(rkv, (ag < na && j < nglob) ? (ag * astride + (unsigned)j * 2 * d + kv * d + g * DH + 4 * c4) * 4u : OOB);
    }
  }
  M.pad = 0;
  if constexpr (PHASE == PH_B) {
    const int ar = thread_idx() >> 4, l16 = thread_idx() & 15;
    M.pad = __builtin_amdgcn_raw_buffer_load_b8(mk_rsrc(a.pad_mask), (a.pad_mask && thread_idx() < 256 && ar < na && l16 < a.T)
                                                    ? (unsigned)((a0 + ar) * a.T + l16) : OOB, 0, 0);
  }
  if constexpr (PHASE == PH_C) ns_load(M.w2, a.w2, a.H * a.FS16, g * a.FS16, a.FS16, D16);
  else ns_load(M.w2, a.w2, K16, g * S_B, S_B, D16);
}


// head of step hs = s - 1 on the LayerNorm'ed output of the last layer (xs): output_fc -> [fusion/hist_map] -> decoder/MLP ->
// position feedback; leaves the next token in tok (autoregressive) and pred_st / h_out in global memory (workgroup g == 0).
// out_only: stop behind output_fc (ha holds its result; the pipeline's head workgroups run the chain on one wave: head_chain_wave)
template <int CPW>
__device__ __forceinline__ void head_step(const ClArgs &a, int g, int a0, int na, const float *xs, int ldx, float *red, float *ha,
                                          float *hb, float *tok, const KsW<4, CPW> &wo, KsW<4, 1> sm_cur, float *pos = nullptr,
                                          bool out_only = false) {
  // head of step hs = s - 1: output_fc -> [fusion/hist_map] -> decoder/MLP -> position feedback (mgcvae.py:435-446)
  const int hs = a.s - 1, tid = thread_idx(), K16 = a.d >> 4;
  __syncthreads();
  const int So = round16(a.n_out) >> 4;
  ks_mfma<4, CPW>(wo, xs, ldx, K16, So, red);
  if (!a.tgt && a.has_map_fusion)
    for (int i = tid; i < na * a.map_dim; i += CNT) {
      const int r = i / a.map_dim, c = i - r * a.map_dim;
      ha[r * LDO + a.n_out + c] = a.map_feat[(size_t)(a0 + r) * a.map_dim + c];
    }
  __syncthreads();
  ks_reduce<CPW, 2>(red, K16, So, [&](int r, int c, float v, int) {
    if (c < a.n_out) {
      v += a.output_fc.b[c];
      ha[r * LDO + c] = v;
      if (g == 0 && a.h_out && r < na) a.h_out[((size_t)(a0 + r) * a.ph + hs) * a.n_out + c] = v;
    }
  });
  __syncthreads();
  if (!a.tgt && !out_only) {
    float *cur = ha, *alt = hb;
#pragma unroll
    for (int i = 0; i < CHAIN_MAX; ++i) {
      if (i < a.n_chain) {
        KsW<4, 1> sm_nxt = sm_cur;
        if (i + 1 < CHAIN_MAX && i + 1 < a.n_chain) small_load(sm_nxt, a.chain[i + 1 < CHAIN_MAX ? i + 1 : i]);
        const int k16 = round16(a.chain[i].k) >> 4, s16 = round16(a.chain[i].n) >> 4, n = a.chain[i].n;
        const float *bias = a.chain[i].b;
        ks_mfma<4, 1>(sm_cur, cur, LDO, k16, s16, red);
        __syncthreads();
        ks_reduce<1, 2>(red, k16, s16, [&](int r, int c, float v, int) { alt[r * LDO + c] = c < n ? v + (bias ? bias[c] : 0.f) : 0.f; });
        __syncthreads();
        if (a.chain_ln[i].g) {
          layer_norm_rows<true>(alt, LDO, a.RC, n, a.chain_ln[i]);      // rows beyond the cluster: bias values, never read
          __syncthreads();
        }
        float *t = cur;
        cur = alt;
        alt = t;
        sm_cur = sm_nxt;
      }
    }
    // cur holds delta (n_in columns)
    for (int i = tid; i < na * a.n_in; i += CNT) {
      const int r = i / a.n_in, c = i - r * a.n_in;
      // previous position: what workgroup 0 of the cluster stored in an earlier LAUNCH, or (fused kernel, where a plain
      // load of another workgroup's store could be stale) this workgroup's own copy in LDS
      const float prev = hs == 0 ? a.start_pos[(size_t)(a0 + r) * a.n_in + c]
                         : pos   ? pos[r * 16 + c]
                                 : a.pred_st[((size_t)(a0 + r) * a.ph + hs - 1) * a.n_in + c];
      const float np = prev + cur[r * LDO + c];
      tok[r * LDTOK + c] = np;
      if (pos) pos[r * 16 + c] = np;
      if (g == 0) a.pred_st[((size_t)(a0 + r) * a.ph + hs) * a.n_in + c] = np;
    }
  }
}

// ---- the head's chain of small linears (decoder/MLP: Linear + LayerNorm + ReLU, Linear) for the 4 rows of a cluster on ONE wave.
// head_step runs it on the matrix cores with two or three workgroup barriers per linear (K-split partial tiles through LDS,
// LayerNorm on a wave per row): ~4 us of barriers and LDS round trips on every step of the decode's critical path.  Here lane
// (row r = lane / 16, c = lane % 16) owns output columns c and c + 16 and nothing leaves the wave.  Same sums in the same order:
// v_mfma_f32_16x16x4_f32 adds its four products as a forward chain of fused multiply-adds (tools/probe/mfma16_order.hip), a
// 16-wide K chunk is four such steps (k = 4q + e, e outer), ks_reduce adds the chunks in order to 0, and layer_norm_rows'
// 64-lane xor butterfly pairs (c, c ^ 16) first and then c ^ 8, 4, 2, 1 - which row rotations by 8, 4, 2, 1 reproduce once the
// values have the matching period.  Taken when every linear of the chain has <= CWK inputs and <= 32 outputs (chain_fits_wave).
// CWK = 32: the Basic / lane configurations; CWK = 64 (an instantiation of its own, so that the code of the first stays as it
// was measured): the map configurations, whose fusion/hist_map takes output_fc's 32 values + the map feature's 32.
template <int CWK>
constexpr int CW_LD = CWK + 4;      // row stride of the staged weights W[col][k] (+ 4: the 16 lanes of a row read different banks)
template <int CWK>
__device__ __forceinline__ bool chain_fits_wave(const ClArgs &a) {
  if (a.tgt || a.n_chain < 1 || a.n_in > 16) return false;
  // (no a.chain[a.n_chain - 1]: an index known only at run time sends the whole argument block to scratch.  The last linear's
  // width is n_in - mmt_decoder_forward refuses anything else - and a narrower one would only leave zeros in the missing columns.)
  bool ok = true;
#pragma unroll
  for (int i = 0; i < CHAIN_MAX; ++i)
    if (i < a.n_chain) ok = ok && a.chain[i].k <= CWK && a.chain[i].n <= 32 && (!a.chain_ln[i].g || a.chain[i].n == 32);
  return ok;
}
// packed [n/16][k/16][lane = 16q + j][e] = W[16s + j][16c + 4q + e]  ->  cw[i][col][k], zero beyond the padded matrix
template <int CWK>
__device__ __forceinline__ void chain_stage(const ClArgs &a, float *cw) {
#pragma unroll
  for (int i = 0; i < CHAIN_MAX; ++i)
    if (i < a.n_chain) {
      const int kp = round16(a.chain[i].k), np = round16(a.chain[i].n), K16 = kp >> 4;
      for (int idx = thread_idx(); idx < 32 * CWK; idx += CNT) {
        const int col = idx / CWK, k = idx % CWK;
        float v = 0.f;
        if (col < np && k < kp) v = a.chain[i].w[((size_t)((col >> 4) * K16 + (k >> 4)) * 64 + 16 * ((k & 15) >> 2) + (col & 15)) * 4 + (k & 3)];
        cw[(i * 32 + col) * CW_LD<CWK> + k] = v;
      }
    }
}
template <int N>
__device__ __forceinline__ float row_ror_add(float t) {      // t + (t of lane (l + N) % 16 of the same row)
  return t + dpp_mov<0x120 + N>(t);
}
// the 64-lane butterfly of layer_norm_rows over a row of 32 values held two per lane (x0 = column c, x1 = column c + 16)
__device__ __forceinline__ float row32_total(float x0, float x1) {
  float t = (x0 + 0.f) + (x1 + 0.f);        // step 32 meets the idle lanes' zeros, step 16 the other half of the row
  t = row_ror_add<8>(t);
  t = row_ror_add<4>(t);
  t = row_ror_add<2>(t);
  t = row_ror_add<1>(t);
  return t;
}
// STASH: st_ho = output_fc's rows (n_out wide), st_hu / st_hxhat / st_hrstd = the hidden layer's ReLU output, normalised row and
// 1/std (32 wide; one hidden layer: mmt_decoder_stash_supported), all at the stash row of the cluster's first agent
// (map configurations: st_ho rows are cat[output_fc, encoded map], ho_ld wide, and st_u = fusion/hist_map's output - the chain's
// first linear, which has no LayerNorm)
template <int CWK, bool STASH = false>
__device__ __forceinline__ void head_chain_wave(const ClArgs &a, int a0, int na, float *ha, float *hb, float *tok, float *pos,
                                                const float *cw, float *st_ho = nullptr, float *st_hu = nullptr,
                                                float *st_hxhat = nullptr, float *st_hrstd = nullptr, float *st_u = nullptr) {
#pragma clang fp contract(off)
  const int hs = a.s - 1, lane = lane_id(), r = lane >> 4, c = lane & 15;
  float *cur = ha, *alt = hb;
  float out[2] = {0.f, 0.f};
  if constexpr (STASH) {
    if (r < na) {
      const int ho_ld = a.n_out + (a.has_map_fusion ? a.map_dim : 0);
#pragma unroll
      for (int t = 0; t < CWK / 16; ++t)
        if (c + 16 * t < ho_ld) st_ho[(size_t)r * ho_ld + c + 16 * t] = ha[r * LDO + c + 16 * t];
    }
  }
#pragma unroll
  for (int i = 0; i < CHAIN_MAX; ++i) {
    if (i < a.n_chain) {
      const int k16 = round16(a.chain[i].k) >> 4, n = a.chain[i].n;
      const float *bias = a.chain[i].b;
      if (CWK == 32 || k16 <= 2) {      // (always, or wave-uniform) both chunks' reads issued up front
        f32x4 hv[2][4];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int q = 0; q < 4; ++q) hv[ch][q] = *reinterpret_cast<const f32x4 *>(cur + r * LDO + 16 * ch + 4 * q);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int col = c + 16 * t;
          const float *wrow = cw + (i * 32 + col) * CW_LD<CWK>;
          float sum = 0.f;
#pragma unroll
          for (int ch = 0; ch < 2; ++ch) {
            f32x4 wv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) wv[q] = *reinterpret_cast<const f32x4 *>(wrow + 16 * ch + 4 * q);
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int q = 0; q < 4; ++q) acc = fmaf(hv[ch][q][e], wv[q][e], acc);
            sum = ch < k16 ? sum + acc : sum;
          }
          out[t] = col < n ? sum + (bias ? bias[col] : 0.f) : 0.f;
        }
      } else {      // chunk by chunk in a loop (32 operand registers at a time: four chunks unrolled need 254 VGPRs)
        float sum[2] = {0.f, 0.f};
#pragma clang loop unroll(disable)
        for (int ch = 0; ch < k16; ++ch) {
          f32x4 hv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) hv[q] = *reinterpret_cast<const f32x4 *>(cur + r * LDO + 16 * ch + 4 * q);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const float *wrow = cw + (i * 32 + c + 16 * t) * CW_LD<CWK> + 16 * ch;
            f32x4 wv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) wv[q] = *reinterpret_cast<const f32x4 *>(wrow + 4 * q);
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int q = 0; q < 4; ++q) acc = fmaf(hv[q][e], wv[q][e], acc);
            sum[t] = sum[t] + acc;
          }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) out[t] = (c + 16 * t) < n ? sum[t] + (bias ? bias[c + 16 * t] : 0.f) : 0.f;
      }
      if (a.chain_ln[i].g) {      // n == 32: LayerNorm + ReLU of the row (layer_norm_rows<true>)
        const float mu = row32_total(out[0], out[1]) * (1.0f / 32.0f);
        const float d0 = out[0] - mu, d1 = out[1] - mu;
        const float var = row32_total(d0 * d0, d1 * d1) * (1.0f / 32.0f);
        const float rstd = 1.0f / sqrtf(var + MMT_LN_EPS);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float xh = (out[t] - mu) * rstd;
          const float y = fmaf(xh, a.chain_ln[i].g[c + 16 * t], a.chain_ln[i].b[c + 16 * t]);
          out[t] = fmaxf(y, 0.f);
          if constexpr (STASH) {      // (the chain's one LayerNorm: decoder/MLP's hidden layer)
            if (r < na) {
              st_hxhat[(size_t)r * 32 + c + 16 * t] = xh;
              st_hu[(size_t)r * 32 + c + 16 * t] = out[t];
            }
          }
        }
        if constexpr (STASH) {
          if (r < na && c == 0) st_hrstd[r] = rstd;
        }
      } else if constexpr (STASH) {
        if (i == 0 && st_u && r < na) {     // fusion/hist_map's output
#pragma unroll
          for (int t = 0; t < 2; ++t)
            if (c + 16 * t < n) st_u[(size_t)r * n + c + 16 * t] = out[t];
        }
      }
      alt[r * LDO + c] = out[0];
      alt[r * LDO + c + 16] = out[1];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // written and read by this wave only
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float *t_ = cur;
      cur = alt;
      alt = t_;
    }
  }
  // out[0] of lane (r, c < n_in) = delta of row r, column c: position feedback (head_step's last loop)
  if (r < na && c < a.n_in) {
    const float prev = hs == 0 ? a.start_pos[(size_t)(a0 + r) * a.n_in + c]
                       : pos   ? pos[r * 16 + c]
                               : a.pred_st[((size_t)(a0 + r) * a.ph + hs - 1) * a.n_in + c];
    const float np = prev + out[0];
    tok[r * LDTOK + c] = np;
    if (pos) pos[r * 16 + c] = np;
    a.pred_st[((size_t)(a0 + r) * a.ph + hs) * a.n_in + c] = np;
  }
}

// blockIdx -> (cluster, slice g).  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one), and the G
// workgroups of a cluster exchange their partials through memory at every kernel boundary: keep them on ONE XCD, so that the
// partials a launch reads were written through the same L2 (measured at B = 128: decode 1.77 -> 1.41 ms).  Groups of 8 clusters
// take 8*G consecutive blocks (XCD x of the group holds cluster x); the clusters left over use the plain order.  Placement is
// for speed only: any mapping that is a bijection gives the same results.
__device__ __forceinline__ void cluster_of_block(int G, int &cluster, int &g) {
  const int ncl = gridDim.x / G, b = blockIdx.x;
  const int grouped = (ncl >> 3) << 3;                 // clusters that come in groups of 8
  if (b < grouped * G) {
    const int group = b / (8 * G), r = b - group * 8 * G;
    cluster = group * 8 + (r & 7);
    g = r >> 3;
  } else {
    cluster = b / G;
    g = b - cluster * G;
  }
}

template <int DH, int CPW, int PHASE, int XMODE, bool DROP = false>
__global__ __launch_bounds__(CNT) void mmt_dec_phase_kernel(ClArgs a) {
  extern __shared__ __align__(16) float smem[];
  const int d = a.d, ldx = d + MMT_PAD, D16 = d >> 4, K16 = D16;
  constexpr int LD1 = (3 * DH > 64 ? 3 * DH : 64) + 8;
  constexpr int LDK = DH + 4;                    // K/V staging row stride: conflict-free b128 reads per 16-lane group
  float *xs = smem;                              // [16][ldx]   LayerNorm'ed input of this sub-layer
  float *red = xs + 16 * ldx;                    // [8][KS_SMAX][256] K-split partial tiles
  float *t1 = red + CNW * KS_SMAX * 256;         // [16][LD1]   q|k|v of the head, or the FFN hidden slice
  float *ob = t1 + 16 * LD1;                     // [16][LDO]   attention output of the head
  float *ha = ob + 16 * LDO;                     // [16][LDO]   head-MLP ping
  float *hb = ha + 16 * LDO;                     // [16][LDO]   head-MLP pong
  float *tok = hb + 16 * LDO;                    // [16][LDTOK] token fed to input_fc
  float *ps = tok + 16 * LDTOK;                  // [16][16]    attention probabilities
  float *Vs = ps + 256;                          // [16][KMAX][LDK]   (phases A, B only)
  float *Ks = red;                               // [16][KMAX][LDK]   reuses the partial-tile region once it is summed
  static_assert(16 * KMAX * LDK <= CNW * KS_SMAX * 256, "K staging must fit the reduction scratch");
  const int tid = thread_idx();
  const int G = PHASE == PH_H ? 1 : a.H;
  int cluster, g;
  cluster_of_block(G, cluster, g);
  const int a0 = cluster * a.RC, na = min(a.RC, a.B - a0);
  constexpr int S_A = 3 * DH / 16, S_B = DH / 16;
  constexpr bool HEAD = PHASE == PH_H || (PHASE == PH_A && XMODE == XM_HEAD_EMBED);
  constexpr bool REDUCES = !(PHASE == PH_A && XMODE == XM_EMBED);

  CSTAMP(0);
  hot_args(a);
  // ---- 1. one burst of requests: what the previous launch produced first, then this workgroup's own operands
  RedRegs R;                                        // deliberately not initialised: only waves with rows < RC touch it
  const bool red_wave = REDUCES && wave_id() * 2 < a.RC;   // wave-uniform (RC is even): idle waves issue nothing
  if (red_wave) reduce_issue(R, a, cluster);
  MainRegs<DH, CPW, PHASE> M = {};
  KsW<4, CPW> wo = {};
  KsW<4, 1> sm_cur = {};
  if constexpr (HEAD) {
    ks_load<4, CPW>(wo, a.output_fc.w, K16, round16(a.n_out) >> 4, [](int s) { return s; });
    if (!a.tgt) small_load(sm_cur, a.chain[0]);
  } else {
    main_issue<DH, CPW, PHASE>(M, a, a0, na, g);     // (behind the head when there is one: its registers are busy)
  }
  CSTAMP(1);
  // t1, ob and ps are written in full before they are read; the head's tiles and the token tile are not
  if constexpr (HEAD || (PHASE == PH_A && XMODE != XM_REDUCE))
    for (int i = tid; i < 2 * 16 * LDO + 16 * LDTOK; i += CNT) ha[i] = 0.f;
  __syncthreads();

  // ---- 2. input of this sub-layer
  if (red_wave) reduce_finish<DROP>(R, a, cluster, xs, ldx, g == 0 && !HEAD);
  else if (REDUCES) {                               // rows beyond the cluster: zeros (finite MFMA operands)
    const int row = tid >> 5, l32 = tid & 31;
    for (int q = l32; q < (d >> 2); q += 32) *reinterpret_cast<f32x4 *>(xs + row * ldx + 4 * q) = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if constexpr (HEAD) {
    head_step<CPW>(a, g, a0, na, xs, ldx, red, ha, hb, tok, wo, sm_cur);
    if constexpr (PHASE == PH_H) {
      // the fused kernel gave up on a wait (a workgroup of a cluster was not resident): fail loudly, never return stale numbers
      if (a.tmo && __hip_atomic_load((const gu32 *)a.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __syncthreads();
        if (tid == 0 && cluster == 0) atomicOr(&g_decoder_fault, 1u);      // sticky: mmt_decoder_status() reports and clears it
        if (a.pred_st)
          for (int i = tid; i < na * a.ph * a.n_in; i += CNT) a.pred_st[(size_t)a0 * a.ph * a.n_in + i] = __builtin_nanf("");
        if (a.h_out)
          for (int i = tid; i < na * a.ph * a.n_out; i += CNT) a.h_out[(size_t)a0 * a.ph * a.n_out + i] = __builtin_nanf("");
      }
      return;
    }
    main_issue<DH, CPW, PHASE>(M, a, a0, na, g);
    __syncthreads();
  }
  if constexpr (PHASE == PH_H) return;
  if constexpr (PHASE == PH_A && XMODE != XM_REDUCE) {
    if (a.tgt || XMODE == XM_EMBED)
      for (int i = tid; i < na * a.n_in; i += CNT) {
        const int r = i / a.n_in, c = i - r * a.n_in;
        tok[r * LDTOK + c] = a.tgt ? a.tgt[((size_t)(a0 + r) * a.ph + a.s) * a.n_in + c]
                                   : a.start_pos[(size_t)(a0 + r) * a.n_in + c];
      }
    __syncthreads();
    embed_token(a, cluster, tok, xs, ldx, g == 0);
  }
  __syncthreads();

  CSTAMP(2);
  // ---- 3. first linear (K-split over the waves), bias (+ReLU) in the reduction
  {
    const int S1 = PHASE == PH_A ? S_A : PHASE == PH_B ? S_B : a.FS16;
    ks_mfma<PhaseGeom<DH, PHASE>::S1MAX, CPW>(M.w1, xs, ldx, K16, S1, red);
    __syncthreads();
    ks_reduce<CPW, 3>(red, K16, S1, [&](int r, int c, float v, int it) {
      v += M.b1[it];
      if (PHASE == PH_C) v = fmaxf(v, 0.f);
      t1[r * LD1 + c] = v;
    });
  }
  CSTAMP(3);

  // ---- 4. attention of head g for the agents of the cluster, K/V staged in LDS
  const float *x2 = t1;
  int ld2 = LD1, K2 = a.FS16;
  if constexpr (PHASE == PH_A || PHASE == PH_B) {
    const int nglob = PHASE == PH_A ? a.s : a.T;
    const int nkeys = PHASE == PH_A ? a.s + 1 : a.T;
    __syncthreads();       // every wave is done reading `red`
#pragma unroll
    for (int it = 0; it < DH / 4; ++it) {
      int ag, j, kv, c4;
      kv_unit<DH>(tid + CNT * it, ag, j, kv, c4);
      if (ag < na && j < nglob) *reinterpret_cast<f32x4 *>((kv ? Vs : Ks) + (ag * KMAX + j) * LDK + 4 * c4) = M.kv[it];
    }
    __syncthreads();       // t1 (q | k | v) complete
    if constexpr (PHASE == PH_A) {   // the new key / value: into the cache (read by later launches) and the staging tiles
      for (int i = tid; i < na * 2 * (DH / 4); i += CNT) {
        const int r = i / (2 * (DH / 4)), rem = i - r * 2 * (DH / 4), kv = rem / (DH / 4), c4 = rem - kv * (DH / 4);
        const f32x4 val = *reinterpret_cast<const f32x4 *>(t1 + r * LD1 + (1 + kv) * DH + 4 * c4);
        *reinterpret_cast<f32x4 *>(a.kv + (((size_t)a.L * a.B + a0 + r) * a.ph + a.s) * 2 * d + kv * d + g * DH + 4 * c4) = val;
        *reinterpret_cast<f32x4 *>((kv ? Vs : Ks) + (r * KMAX + a.s) * LDK + 4 * c4) = val;
      }
      __syncthreads();
    }
    const int ar = tid >> 4, l16 = tid & 15;
    if (tid < 256) {
      float sc = -INFINITY;
      bool masked = ar >= na || l16 >= nkeys;
      if constexpr (PHASE == PH_B) masked = masked || M.pad != 0;
      if (!masked) {
        const float *q = t1 + ar * LD1, *k = Ks + (ar * KMAX + l16) * LDK;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < DH / 4; ++c) {
          const f32x4 qv = *reinterpret_cast<const f32x4 *>(q + 4 * c), kv4 = *reinterpret_cast<const f32x4 *>(k + 4 * c);
          acc = fmaf(qv[0], kv4[0], acc);
          acc = fmaf(qv[1], kv4[1], acc);
          acc = fmaf(qv[2], kv4[2], acc);
          acc = fmaf(qv[3], kv4[3], acc);
        }
        sc = acc * (1.0f / sqrtf((float)DH));
      }
      const float mx = row16_max(sc);
      float p = mx > -INFINITY ? expf(sc - mx) : 0.f;    // a fully masked row yields zeros (reference: undefined / NaN)
      const float sum = row16_sum(p);
      p *= sum > 0.f ? 1.0f / sum : 0.f;
      ps[ar * 16 + l16] = p;
    }
    __syncthreads();
    if (tid < 256) {
      constexpr int DPL = DH / 16;
      float o[DPL];
#pragma unroll
      for (int e = 0; e < DPL; ++e) o[e] = 0.f;
      const int nk = ar < na ? nkeys : 0;
      // all LDS reads in flight at once, then the same fused multiply-adds in the same order (a loop with a run-time trip
      // count pays one LDS round trip per key); rows beyond nk hold stale bits and are never multiplied
      typedef float vrow __attribute__((ext_vector_type(DPL)));
      f32x4 pr[KMAX / 4];
      vrow vv[KMAX];
#pragma unroll
      for (int c = 0; c < KMAX / 4; ++c) pr[c] = *reinterpret_cast<const f32x4 *>(ps + ar * 16 + 4 * c);
#pragma unroll
      for (int j = 0; j < KMAX; ++j) vv[j] = *reinterpret_cast<const vrow *>(Vs + (ar * KMAX + j) * LDK + l16 * DPL);
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {
        const float pj = pr[j >> 2][j & 3];
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[e] = j < nk ? fmaf(pj, vv[j][e], o[e]) : o[e];
      }
#pragma unroll
      for (int e = 0; e < DPL; ++e) ob[ar * LDO + l16 * DPL + e] = o[e];
    }
    x2 = ob;
    ld2 = LDO;
    K2 = S_B;
  }
  __syncthreads();
  CSTAMP(4);

  // ---- 5. K-slice of the output projection: partial [RC][d] of this workgroup
  ns_mfma_store(M.w2, x2, ld2, K2, D16, a.part_out + ((size_t)cluster * a.H + g) * 16 * d, d, a.RC);
  CSTAMP(5);
}

// ------------------------------------------------------------------------------- 4-agent clusters on v_mfma_f32_4x4x1
// With RC = 4 agents a 16x16x4 tile is 3/4 padding and the phase is bound by MFMA issue (32 cycles per 64 weights).
// v_mfma_f32_4x4x1_16B_f32 multiplies 16 independent 4x4 blocks in 8 cycles: block b = lanes 4b..4b+3 computes
// D_b[i][j] += A_b[i] B_b[j] (lane 4b+j holds column j, register i holds row i).  A block is given the 4 agents as A
// (every block reads x[lane & 3][k]) and 4 output columns as B: 4x fewer matrix-core cycles.
//
// The 16 blocks of a wave are dealt as (column group, K part) with ALL K parts of a column group in one wave, so the
// K parts are summed with DPP / permlane-swap adds inside the wave: no LDS round trip, no barrier.  Wave w owns the
// column groups [w*cgw, (w+1)*cgw) (cgw = column groups / 8); block b -> column group b % cgw, K part b / cgw; a block
// runs over q k-quads of its K part (one 16-byte weight load = 4 MFMAs).  The weights come from copies packed in
// exactly the order (slice, part, k-quad step, wave, lane) the load instructions run in (mmt_pack_lin4_*): every wave
// load is one contiguous KiB at a wave-uniform offset.
struct Lin4 {      // geometry of one linear on the 4x4x1 path; all powers of two (mmt_lin4_*_floats checks)
  int cgw, lg_cgw, q;
};
__host__ __device__ __forceinline__ Lin4 lin4(int col_groups, int kquads) {
  Lin4 L;
  L.cgw = col_groups >> 3;
  L.lg_cgw = __builtin_ctz(L.cgw);
  const int lg_nkp = 4 - L.lg_cgw;                   // K parts per column group = 16 / cgw
  L.q = kquads >> lg_nkp;
  if (L.q < 1) L.q = 1;                              // fewer k-quads than parts: the surplus blocks meet zero weights
  return L;
}
// sum over the K parts of a column group: lanes that differ in the block bits above lg_cgw
// (order: across the wave halves, across the row pairs, then inside a row - the static form below halves its registers in
// the first two steps and must add in the same tree to stay bit-identical)
__device__ __forceinline__ float kparts_sum(float v, int cgw) {
  if (cgw <= 8) v = xor32_sum(v);
  if (cgw <= 4) v = xor16_sum(v);
  if (cgw <= 1) v += dpp_mov<0x124>(v);              // row_ror:4
  if (cgw <= 2) v += dpp_mov<0x128>(v);              // row_ror:8  (after ror:4 this covers all four rotations)
  return v;
}

template <int PHASE>
struct Geom4 {
  static constexpr int NPART = PHASE == PH_A ? 3 : 1;     // q | k | v
  static constexpr int QMAX1 = PHASE == PH_C ? 8 : 4;     // most k-quads per block, first linear
  static constexpr int QMAX2 = PHASE == PH_C ? 8 : 4;     // second linear
};
template <int DH, int PHASE>
struct MainRegs4 {
  f32x4 w1[Geom4<PHASE>::NPART][Geom4<PHASE>::QMAX1];
  f32x4 w2[Geom4<PHASE>::QMAX2];
  float b1;               // bias of the first linear, on the lanes that write its result
  f32x4 kv[DH / 16];      // 4 agents: a quarter of the padded K/V index space
  int pad;
};
struct Red4Regs {
  f32x4 res, bias, gam, bet, p[8];
};

// x = LN(res + bias + sum_g part[g]) for 4 rows: wave w < 4 owns row w, a lane one float4 (d <= 256).  Same summation order
// over the parts as the 16-row version.
template <int AUX = AUX_PLAIN>
__device__ __forceinline__ void reduce4_issue(Red4Regs &R, const ClArgs &a, int cluster) {
  const int row = wave_id(), q = lane_id();
  const rsrc_t res = mk_rsrc(a.xres_in + ((size_t)cluster * 16 + row) * a.d), pp = mk_rsrc(a.part_in + ((size_t)cluster * a.H * 16 + row) * a.d);
  const unsigned co = q < (a.d >> 2) ? (unsigned)q * 16u : OOB, gstride = 16u * a.d * 4u;
  R.res = bl4x<AUX>(res, co);
#pragma unroll
  for (int g = 0; g < 8; ++g) R.p[g] = bl4x<AUX>(pp, g < a.H ? co : OOB, g < a.H ? g * gstride : 0u);
  R.bias = bl4(mk_rsrc(a.red_bias), co);
  R.gam = bl4(mk_rsrc(a.red_ln.g), co);
  R.bet = bl4(mk_rsrc(a.red_ln.b), co);
}
// PAIRS (the reduction that follows the cross-attention phase): the parts are added two by two, each pair sum carrying the
// producing phase's tag - exactly what the pipeline kernel's cross-attention workgroups, which own two heads each, hand over
// as ONE pre-summed part (half the rows for its consumers to load); the forms stay bit-identical.
// pre_keep (DROP): the row's four keep bytes, already loaded (the pipeline requests them before it polls: a load behind the poll
// would be a memory round trip on every hop)
// STASH (the training step's decode, mmt_decoder_forward_stash): the row's LayerNorm output, normalised row and 1/std also go to
// st_x / st_xhat / st_rstd (this wave's row of the stash; NULL: not kept)
template <int AUX = AUX_PLAIN, bool DROP = false, bool PAIRS = false, bool STASH = false>
__device__ __forceinline__ void reduce4_finish(const Red4Regs &R, const ClArgs &a, int cluster, float *xs, int ldx, bool write_res,
                                               bool tag_rows = true, const unsigned *pre_keep = nullptr, float *st_x = nullptr,
                                               float *st_xhat = nullptr, float *st_rstd = nullptr) {
  const int row = wave_id(), q = lane_id();
  const bool live = q < (a.d >> 2);
  f32x4 part[PAIRS ? 4 : 8];
  if constexpr (PAIRS) {
    const unsigned tag_in = tag_of_phase((unsigned)a.prof_idx - 1u);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      part[k] = R.p[2 * k] + R.p[2 * k + 1];
      if (a.tagged) part[k] = tag4(part[k], tag_in);
      if (2 * k >= a.H) part[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  } else {
#pragma unroll
    for (int g = 0; g < 8; ++g) part[g] = R.p[g];
  }
  constexpr int NPT = PAIRS ? 4 : 8;
  f32x4 acc = R.res + R.bias;
  bool dropped = false;
  if constexpr (DROP) {            // res + keep * (bias + sum of the parts)
    const int ag = cluster * 4 + row;
    if (a.dk_red && live && ag < a.B) {
      f32x4 t = R.bias;
#pragma unroll
      for (int g = 0; g < NPT; ++g) t += part[g];
      acc = R.res + (pre_keep ? keep4w(*pre_keep, a.ds_red) : keep4(a.dk_red, (size_t)ag * a.d + 4 * q, a.ds_red)) * t;
      dropped = true;
    }
  }
  if (!dropped) {
#pragma unroll
    for (int g = 0; g < NPT; ++g) acc += part[g];       // unused slots hold zeros
  }
  // One row per wave: the two totals are wave-uniform (SGPRs) and 1/d is a multiplication (exact for the powers of two this
  // path is built for): this sits on the critical path of every one of the decode's 228 hops.
  const float inv_d = 1.0f / (float)a.d;
  const float mu = wave64_total((acc[0] + acc[1]) + (acc[2] + acc[3])) * inv_d;     // lanes beyond d/4 hold zeros
  float var = 0.f;
  if (live) {
#pragma unroll
    for (int e = 0; e < 4; ++e) var += (acc[e] - mu) * (acc[e] - mu);
  }
  // (1 / sqrt stays IEEE: v_rsq_f32's 1 ulp, applied 228 times in a row, moved the full-size training gradients from 4e-4 to
  // 9e-4 of the reference's - measured; it would save 1.1 % of the decode)
  const float rstd = 1.0f / sqrtf(wave64_total(var) * inv_d + MMT_LN_EPS);
  if (live) {
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = (acc[e] - mu) * rstd * R.gam[e] + R.bet[e];
    if (a.tagged && tag_rows) y = tag4(y, phase_tag(a));       // this phase's input = the next phase's residual rows
    *reinterpret_cast<f32x4 *>(xs + row * ldx + 4 * q) = y;
    if (write_res) bs4x<AUX>(mk_rsrc(a.xres_out + ((size_t)cluster * 16 + row) * a.d), (unsigned)q * 16u, y);
    if constexpr (STASH) {
      if (st_x) {
        f32x4 xh;
#pragma unroll
        for (int e = 0; e < 4; ++e) xh[e] = (acc[e] - mu) * rstd;
        *reinterpret_cast<f32x4 *>(st_x + 4 * q) = y;
        *reinterpret_cast<f32x4 *>(st_xhat + 4 * q) = xh;
        if (q == 0) *st_rstd = rstd;
      }
    }
  }
}

template <int DH, int PHASE, int AUX = AUX_PLAIN>
__device__ __forceinline__ void main_issue4(MainRegs4<DH, PHASE> &M, const ClArgs &a, int a0, int na, int g) {
  typedef Geom4<PHASE> G4;
  const int d = a.d, FS = a.FS16 * 16, w = wave_id(), lane = lane_id(), b = lane >> 2, j = lane & 3;
  const unsigned lane_off = lane * 16u;
  const Lin4 L1 = lin4((PHASE == PH_C ? FS : DH) / 4, d / 4);
  const rsrc_t r1 = mk_rsrc(a.w1);
#pragma unroll
  for (int p = 0; p < G4::NPART; ++p)
#pragma unroll
    for (int t = 0; t < G4::QMAX1; ++t)
      M.w1[p][t] = bl4s(r1, t < L1.q ? lane_off : OOB, t < L1.q ? (unsigned)((((g * G4::NPART + p) * L1.q + t) * 8 + w) * 1024) : 0u);
  {   // bias: the lanes of the first blocks write the result (block b -> part b in phase A, column group b else)
    int bi;
    if constexpr (PHASE == PH_A) bi = b * d + g * DH + w * 4 + j;
    else if constexpr (PHASE == PH_B) bi = g * DH + w * 4 + j;
    else bi = g * FS + (w * L1.cgw + b) * 4 + j;
    M.b1 = bl1(mk_rsrc(a.b1), b < (PHASE == PH_C ? L1.cgw : G4::NPART) ? (unsigned)bi * 4u : OOB);
  }
  if constexpr (PHASE == PH_A || PHASE == PH_B) {
    const int nglob = PHASE == PH_A ? a.s : a.T;
    const float *base = a.kv + (((size_t)a.L * a.B + a0) * (PHASE == PH_A ? a.ph : a.T)) * 2 * d;
    const unsigned astride = (unsigned)(PHASE == PH_A ? a.ph : a.T) * 2 * d;
    const rsrc_t rkv = mk_rsrc(base);
#pragma unroll
    for (int it = 0; it < DH / 16; ++it) {
      int ag, jj, kv, c4;
      kv_unit<DH>(thread_idx() + CNT * it, ag, jj, kv, c4);
      M.kv[it] = bl4x<AUX>(rkv, (ag < na && jj < nglob) ? (ag * astride + (unsigned)jj * 2 * d + kv * d + g * DH + 4 * c4) * 4u : OOB);
    }
  }
  M.pad = 0;
  if constexpr (PHASE == PH_B) {
    const int ar = thread_idx() >> 4, l16 = thread_idx() & 15;
    M.pad = __builtin_amdgcn_raw_buffer_load_b8(mk_rsrc(a.pad_mask), (a.pad_mask && ar < na && l16 < a.T)
                                                    ? (unsigned)((a0 + ar) * a.T + l16) : OOB, 0, 0);
  }
  {   // second linear: all d output columns, K slice g of out_proj / linear2
    const Lin4 L2 = lin4(d / 4, (PHASE == PH_C ? FS : DH) / 4);
    const rsrc_t r2 = mk_rsrc(a.w2);
#pragma unroll
    for (int t = 0; t < G4::QMAX2; ++t)
      M.w2[t] = bl4s(r2, t < L2.q ? lane_off : OOB, t < L2.q ? (unsigned)(((g * L2.q + t) * 8 + w) * 1024) : 0u);
  }
}

// the stationary part of a phase's operands: its weight slices and the first linear's bias (slice g)
template <int DH, int PHASE>
__device__ __forceinline__ void weights_issue4(MainRegs4<DH, PHASE> &M, const ClArgs &a, int g) {
  typedef Geom4<PHASE> G4;
  const int d = a.d, FS = a.FS16 * 16, w = wave_id(), lane = lane_id(), b = lane >> 2, j = lane & 3;
  const unsigned lane_off = lane * 16u;
  const Lin4 L1 = lin4((PHASE == PH_C ? FS : DH) / 4, d / 4);
  const rsrc_t r1 = mk_rsrc(a.w1);
#pragma unroll
  for (int p = 0; p < G4::NPART; ++p)
#pragma unroll
    for (int t = 0; t < G4::QMAX1; ++t)
      M.w1[p][t] = bl4s(r1, t < L1.q ? lane_off : OOB, t < L1.q ? (unsigned)((((g * G4::NPART + p) * L1.q + t) * 8 + w) * 1024) : 0u);
  {   // bias: the lanes of the first blocks write the result (block b -> part b in phase A, column group b else)
    const int bb = b & 3;         // gemm1_4s: in-row block bb of EVERY DPP row writes (row r of the wave = row r of the result)
    int bi;
    if constexpr (PHASE == PH_A) bi = bb * d + g * DH + w * 4 + j;
    else if constexpr (PHASE == PH_B) bi = g * DH + w * 4 + j;
    else bi = g * FS + (w * L1.cgw + bb) * 4 + j;
    M.b1 = bl1(mk_rsrc(a.b1), bb < (PHASE == PH_C ? L1.cgw : G4::NPART) ? (unsigned)bi * 4u : OOB);
  }
  {   // second linear: all d output columns, K slice g of out_proj / linear2
    const Lin4 L2 = lin4(d / 4, (PHASE == PH_C ? FS : DH) / 4);
    const rsrc_t r2 = mk_rsrc(a.w2);
#pragma unroll
    for (int t = 0; t < G4::QMAX2; ++t)
      M.w2[t] = bl4s(r2, t < L2.q ? lane_off : OOB, t < L2.q ? (unsigned)(((g * L2.q + t) * 8 + w) * 1024) : 0u);
  }
}
// the per-(cluster, step) part: head g's K/V cache rows of the cluster's agents and the key-padding flags
template <int DH, int PHASE, int AUX = AUX_PLAIN>
__device__ __forceinline__ void kv_issue4(MainRegs4<DH, PHASE> &M, const ClArgs &a, int a0, int na, int g, int step = -1) {
  const int d = a.d;
  if constexpr (PHASE == PH_A || PHASE == PH_B) {
    const int nglob = PHASE == PH_A ? (step >= 0 ? step : a.s) : a.T;
    const float *base = a.kv + (((size_t)a.L * a.B + a0) * (PHASE == PH_A ? a.ph : a.T)) * 2 * d;
    const unsigned astride = (unsigned)(PHASE == PH_A ? a.ph : a.T) * 2 * d;
    const rsrc_t rkv = mk_rsrc(base);
#pragma unroll
    for (int it = 0; it < DH / 16; ++it) {
      int ag, jj, kv, c4;
      kv_unit<DH>(thread_idx() + CNT * it, ag, jj, kv, c4);
      M.kv[it] = bl4x<AUX>(rkv, (ag < na && jj < nglob) ? (ag * astride + (unsigned)jj * 2 * d + kv * d + g * DH + 4 * c4) * 4u : OOB);
    }
  }
  M.pad = 0;
  if constexpr (PHASE == PH_B) {
    const int ar = thread_idx() >> 4, l16 = thread_idx() & 15;
    M.pad = __builtin_amdgcn_raw_buffer_load_b8(mk_rsrc(a.pad_mask), (a.pad_mask && ar < na && l16 < a.T)
                                                    ? (unsigned)((a0 + ar) * a.T + l16) : OOB, 0, 0);
  }
}

// t1[0..3][columns of head g] = x W1slice^T + b (+ReLU in phase C)
// DROP (phase C): hidden = dropout(relu(.)) with the keep bytes of (agent a0 + r, hidden column g*FS + col)
template <int DH, int PHASE, bool DROP = false>
__device__ __forceinline__ void gemm1_4(const MainRegs4<DH, PHASE> &M, const ClArgs &a, const float *xs, int ldx, float *t1, int ld1,
                                        int a0 = 0, int na = 0, int g = 0) {
  typedef Geom4<PHASE> G4;
  const int KQ = a.d >> 2, FS = a.FS16 * 16, w = wave_id(), lane = lane_id(), b = lane >> 2, j = lane & 3;
  const Lin4 L1 = lin4((PHASE == PH_C ? FS : DH) / 4, KQ);
  const int kq0 = (b >> L1.lg_cgw) * L1.q;
  const float *xr = xs + j * ldx;
  f32x4 acc[G4::NPART];
#pragma unroll
  for (int p = 0; p < G4::NPART; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < G4::QMAX1; ++t) {
    if (t < L1.q) {
      const f32x4 xv = *reinterpret_cast<const f32x4 *>(xr + 4 * min(kq0 + t, KQ - 1));   // surplus blocks meet zero weights
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int p = 0; p < G4::NPART; ++p) acc[p] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[e], M.w1[p][t][e], acc[p], 0, 0, 0);
    }
  }
#pragma unroll
  for (int p = 0; p < G4::NPART; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[p][r] = kparts_sum(acc[p][r], L1.cgw);
  if constexpr (PHASE == PH_A) {
    if (b < 3) {
      const f32x4 v = b == 0 ? acc[0] : b == 1 ? acc[G4::NPART > 1 ? 1 : 0] : acc[G4::NPART > 2 ? 2 : 0];
#pragma unroll
      for (int r = 0; r < 4; ++r) t1[r * ld1 + b * DH + w * 4 + j] = v[r] + M.b1;
    }
  } else if (b < (PHASE == PH_C ? L1.cgw : 1)) {
    const int col = (w * L1.cgw + b) * 4 + j;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = PHASE == PH_C ? fmaxf(acc[0][r] + M.b1, 0.f) : acc[0][r] + M.b1;
      if constexpr (DROP && PHASE == PH_C) {
        if (a.dk_ff && r < na) v = a.dk_ff[(size_t)(a0 + r) * (FS * a.H) + g * FS + col] ? v * a.ds_ff : 0.f;
      }
      t1[r * ld1 + col] = v;
    }
  }
}

// part[0..3][d] = x2 W2slice^T (K slice g), straight from the accumulators to global memory
template <int DH, int PHASE, int AUX = AUX_PLAIN>
__device__ __forceinline__ void gemm2_4(const MainRegs4<DH, PHASE> &M, const ClArgs &a, const float *x2, int ld2, float *part) {
  typedef Geom4<PHASE> G4;
  const int d = a.d, K2q = (PHASE == PH_C ? a.FS16 * 16 : DH) / 4, w = wave_id(), lane = lane_id(), b = lane >> 2, j = lane & 3;
  const Lin4 L2 = lin4(d / 4, K2q);
  const int kq0 = (b >> L2.lg_cgw) * L2.q;
  const float *xr = x2 + j * ld2;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < G4::QMAX2; ++t) {
    if (t < L2.q) {
      const f32x4 xv = *reinterpret_cast<const f32x4 *>(xr + 4 * min(kq0 + t, K2q - 1));
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[e], M.w2[t][e], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = kparts_sum(acc[r], L2.cgw);
  quad_transpose(acc, j);                            // lane j of a quad: row j, the quad's 4 columns -> one 16-byte store
  if (a.tagged) acc = tag4(acc, phase_tag(a));
  bs4x<AUX>(mk_rsrc(part), b < L2.cgw ? (unsigned)(j * d + (w * L2.cgw + b) * 4) * 4u : OOB, acc);
}

// ---- hand-off between the H workgroups of a cluster INSIDE one launch (fused kernel) -----------------------------------
// Every byte another workgroup reads (partials, residual rows) is stored write-through (sc1) and loaded past the L1 (sc1);
// a phase ends with: every wave drains its stores -> workgroup barrier -> ONE lane adds 1 to the cluster's arrival counter;
// the next phase starts with: ONE lane polls the counter (relaxed, bounded) -> workgroup barrier -> sc1 loads.  The counter
// only grows (phase p waits for H*p arrivals); it is zeroed by the cross-K/V kernel that runs before.  Correctness does not
// depend on which XCD a workgroup runs on; cluster_of_block only makes the traffic XCD-local.
struct FusedSync {
  gu32 *ctr, *tmo;       // arrival counter of this cluster (a 128-byte line of its own), time-out word of the launch
  int *abort;            // LDS
  unsigned phase;        // phases this workgroup has completed
  int H;
};
constexpr unsigned long long FUSED_WAIT_TICKS = 20000000ull;     // 0.2 s of the 100 MHz wall clock: a wait this long = give up

__device__ __forceinline__ bool cluster_wait(const FusedSync &fs) {
  if (thread_idx() == 0) {
    const unsigned target = fs.phase * (unsigned)fs.H;
    if (__hip_atomic_load(fs.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      const unsigned long long t0 = wall_clock64();
      for (;;) {
        __builtin_amdgcn_s_sleep(1);
        if (__hip_atomic_load(fs.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
        if (wall_clock64() - t0 > FUSED_WAIT_TICKS) {      // a sibling never arrived (not resident, or it gave up): every
          __hip_atomic_store(fs.tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // workgroup reaches this exit
          *fs.abort = 1;
          break;
        }
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");        // compiler only: no load of handed-off bytes moves above the poll
  return *fs.abort == 0;
}
__device__ __forceinline__ void cluster_arrive(FusedSync &fs) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave, before the barrier
  __syncthreads();
  if (thread_idx() == 0) __hip_atomic_fetch_add(fs.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ++fs.phase;
}

// One phase (A, B or C) of one layer-step for a cluster of 4 agents; the head-only launch stays on mmt_dec_phase_kernel.
// FUSED: called from the fused kernel's loop - waits for the cluster's previous phase instead of a kernel boundary, stores what
// other workgroups read write-through and loads it past the L1.
template <int DH, int CPW, int PHASE, int XMODE, bool FUSED, bool DROP = false>
__device__ __forceinline__ bool phase4_body(const ClArgs &a, int cluster, int g, float *smem, FusedSync &fs) {
  static_assert(DH == 32 && PHASE != PH_H, "4x4x1 path: 8 column groups per head, one per wave");
  constexpr int AUX = FUSED ? AUX_SC1 : AUX_PLAIN;
  const int d = a.d, ldx = d + MMT_PAD, K16 = d >> 4;
  constexpr int LD1 = 3 * DH + 8;
  constexpr int LDK = DH + 4;
  float *xs = smem;                              // [16][ldx]   rows 0..3: LayerNorm'ed input (16 rows where the head runs)
  float *red = xs + 16 * ldx;                    // K-split partial tiles of the head's small linears
  float *t1 = red + CNW * KS_SMAX * 256;         // [4][LD1]    q|k|v of the head, or the FFN hidden slice
  float *ob = t1 + 16 * LD1;                     // [4][LDO]    attention output of the head
  float *ha = ob + 16 * LDO;
  float *hb = ha + 16 * LDO;
  float *tok = hb + 16 * LDO;
  float *ps = tok + 16 * LDTOK;                  // [4][16]     attention probabilities
  float *Ks = ps + 256;                          // [4][KMAX][LDK]
  float *Vs = Ks + 4 * KMAX * LDK;               // [4][KMAX][LDK]
  float *pos = ps + 64;                          // [4][16]     fused kernel: the cluster's last positions (ps uses 64 floats)
  const int tid = thread_idx();
  const int a0 = cluster * 4, na = min(4, a.B - a0);
  constexpr bool HEAD = XMODE == XM_HEAD_EMBED;
  constexpr bool REDUCES = XMODE != XM_EMBED;
  constexpr bool EMBEDS = PHASE == PH_A && XMODE != XM_REDUCE;

  CSTAMP(0);
  if constexpr (!FUSED) hot_args(a);
  // ---- 1. one burst of requests
  if constexpr (FUSED && REDUCES) {
    if (!cluster_wait(fs)) return false;
  }
  Red4Regs R;
  const bool red_wave = REDUCES && wave_id() < 4;
  if (red_wave) reduce4_issue<AUX>(R, a, cluster);
  MainRegs4<DH, PHASE> M = {};
  KsW<4, CPW> wo = {};
  KsW<4, 1> sm_cur = {};
  if constexpr (HEAD) {
    ks_load<4, CPW>(wo, a.output_fc.w, K16, round16(a.n_out) >> 4, [](int s) { return s; });
    if (!a.tgt) small_load(sm_cur, a.chain[0]);
  } else {
    main_issue4<DH, PHASE, AUX>(M, a, a0, na, g);
  }
  CSTAMP(1);
  if constexpr (EMBEDS) {                          // the head's 16-row tiles and the token tile must be finite
    for (int i = tid; i < 2 * 16 * LDO + 16 * LDTOK; i += CNT) ha[i] = 0.f;
    if (HEAD && tid >= 256)                        // rows 4..15 of the head's input tile
      for (int i = tid - 256; i < 12 * (d >> 2); i += 256)
        *reinterpret_cast<f32x4 *>(xs + (4 + i / (d >> 2)) * ldx + 4 * (i % (d >> 2))) = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
  }

  // ---- 2. input of this sub-layer
  if (red_wave) reduce4_finish<AUX, DROP, PHASE == PH_C>(R, a, cluster, xs, ldx, g == 0 && !HEAD, !HEAD);
  if constexpr (HEAD) {
    head_step<CPW>(a, g, a0, na, xs, ldx, red, ha, hb, tok, wo, sm_cur, FUSED ? pos : nullptr);
    main_issue4<DH, PHASE, AUX>(M, a, a0, na, g);
    __syncthreads();
  }
  if constexpr (EMBEDS) {
    if (a.tgt || XMODE == XM_EMBED)
      for (int i = tid; i < na * a.n_in; i += CNT) {
        const int r = i / a.n_in, c = i - r * a.n_in;
        tok[r * LDTOK + c] = a.tgt ? a.tgt[((size_t)(a0 + r) * a.ph + a.s) * a.n_in + c]
                                   : a.start_pos[(size_t)(a0 + r) * a.n_in + c];
      }
    __syncthreads();
    embed_token<FUSED, DROP>(a, cluster, tok, xs, ldx, g == 0);
  }
  __syncthreads();
  CSTAMP(2);

  // ---- 3. first linear; K/V cache rows from registers to their staging tiles
  gemm1_4<DH, PHASE, DROP>(M, a, xs, ldx, t1, LD1, a0, na, g);
  if constexpr (PHASE != PH_C) {
    const int nglob = PHASE == PH_A ? a.s : a.T;
#pragma unroll
    for (int it = 0; it < DH / 16; ++it) {
      int ag, j, kv, c4;
      kv_unit<DH>(tid + CNT * it, ag, j, kv, c4);
      if (ag < na && j < nglob) *reinterpret_cast<f32x4 *>((kv ? Vs : Ks) + (ag * KMAX + j) * LDK + 4 * c4) = M.kv[it];
    }
  }
  __syncthreads();
  CSTAMP(3);

  // ---- 4. attention of head g: wave 0, one (agent, key) pair per lane; the key / value of this step come from t1
  const float *x2 = t1;
  int ld2 = LD1;
  if constexpr (PHASE != PH_C) {
    const int nglob = PHASE == PH_A ? a.s : a.T;
    const int nkeys = PHASE == PH_A ? a.s + 1 : a.T;
    if (tid < 64) {
      const int ar = tid >> 4, l16 = tid & 15;
      float sc = -INFINITY;
      bool masked = ar >= na || l16 >= nkeys;
      if constexpr (PHASE == PH_B) masked = masked || M.pad != 0;
      if (!masked) {
        const float *q = t1 + ar * LD1;
        const float *k = (PHASE == PH_A && l16 == nglob) ? q + DH : Ks + (ar * KMAX + l16) * LDK;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < DH / 4; ++c) {
          const f32x4 qv = *reinterpret_cast<const f32x4 *>(q + 4 * c), kv4 = *reinterpret_cast<const f32x4 *>(k + 4 * c);
          acc = fmaf(qv[0], kv4[0], acc);
          acc = fmaf(qv[1], kv4[1], acc);
          acc = fmaf(qv[2], kv4[2], acc);
          acc = fmaf(qv[3], kv4[3], acc);
        }
        sc = acc * (1.0f / sqrtf((float)DH));
      }
      const float mx = row16_max(sc);
      float p = mx > -INFINITY ? expf(sc - mx) : 0.f;    // a fully masked row yields zeros (reference: undefined / NaN)
      const float sum = row16_sum(p);
      p *= sum > 0.f ? 1.0f / sum : 0.f;
      if constexpr (DROP) {          // dropout on the attention probabilities (nn.MultiheadAttention(dropout=p), training mode)
        const int Lk = PHASE == PH_A ? a.ph : a.T;
        if (a.dk_att && !masked) p = a.dk_att[((size_t)(a0 + ar) * a.H + g) * a.ph * Lk + l16] ? p * a.ds_att : 0.f;
      }
      ps[tid] = p;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // ps is written and read by this wave only
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      constexpr int DPL = DH / 16;
      float o[DPL];
#pragma unroll
      for (int e = 0; e < DPL; ++e) o[e] = 0.f;
      const int nk = ar < na ? nglob : 0;
      {   // all LDS reads in flight at once, then the same fused multiply-adds in the same order (as attend4)
        typedef float vrow __attribute__((ext_vector_type(DPL)));
        f32x4 pr[KMAX / 4];
        vrow vv[KMAX];
#pragma unroll
        for (int c = 0; c < KMAX / 4; ++c) pr[c] = *reinterpret_cast<const f32x4 *>(ps + ar * 16 + 4 * c);
#pragma unroll
        for (int j = 0; j < KMAX; ++j) vv[j] = *reinterpret_cast<const vrow *>(Vs + (ar * KMAX + j) * LDK + l16 * DPL);
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
          const float pj = pr[j >> 2][j & 3];
#pragma unroll
          for (int e = 0; e < DPL; ++e) o[e] = j < nk ? fmaf(pj, vv[j][e], o[e]) : o[e];
        }
      }
      if (PHASE == PH_A && ar < na) {
        const float pj = ps[ar * 16 + nglob];
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[e] = fmaf(pj, t1[ar * LD1 + 2 * DH + l16 * DPL + e], o[e]);
      }
#pragma unroll
      for (int e = 0; e < DPL; ++e) ob[ar * LDO + l16 * DPL + e] = o[e];
    } else if (PHASE == PH_A && tid >= 256) {        // meanwhile: the new key / value into the cache (read by later launches /
      const int i = tid - 256;                       // later phases of this same workgroup only)
      if (i < na * 2 * (DH / 4)) {
        const int r = i / (2 * (DH / 4)), rem = i - r * 2 * (DH / 4), kv = rem / (DH / 4), c4 = rem - kv * (DH / 4);
        *reinterpret_cast<f32x4 *>(a.kv + (((size_t)a.L * a.B + a0 + r) * a.ph + a.s) * 2 * d + kv * d + g * DH + 4 * c4) =
            *reinterpret_cast<const f32x4 *>(t1 + r * LD1 + (1 + kv) * DH + 4 * c4);
      }
    }
    x2 = ob;
    ld2 = LDO;
    __syncthreads();
  }
  CSTAMP(4);

  // ---- 5. K-slice of the output projection: partial [4][d] of this workgroup
  gemm2_4<DH, PHASE, AUX>(M, a, x2, ld2, a.part_out + ((size_t)cluster * a.H + g) * 16 * d);
  CSTAMP(5);
  if constexpr (FUSED) {
    cluster_arrive(fs);
    CSTAMP(6);
  }
  return true;
}

template <int DH, int CPW, int PHASE, int XMODE, bool DROP = false>
__global__ __launch_bounds__(CNT) void mmt_dec_phase4_kernel(ClArgs a) {
  extern __shared__ __align__(16) float smem[];
  int cluster, g;
  cluster_of_block(a.H, cluster, g);
  FusedSync fs = {};
  phase4_body<DH, CPW, PHASE, XMODE, false, DROP>(a, cluster, g, smem, fs);
}

// ------------------------------------------------------------------------------- fused kernel: every phase of a decode in ONE launch
// When all clusters are resident at once (4-agent clusters: ceil(B/4)*H workgroups <= #CUs, one per CU) the 216 kernel
// boundaries of a decode become 216 cluster-level hand-offs: the H workgroups of a cluster wait only for each other.  The final
// head (PH_H) stays a launch of its own.  Measured at B = 128: 1.22 ms against 1.42 ms for the launch-per-phase form.
// Requesting the next phase's weights early (behind the arrival, or under the attention of the phase before) was measured and
// is SLOWER (1.31 / 1.38 ms): loads return in order, so the partials - the critical path - queue behind 128 KB of weights, and
// the drain that precedes an arrival waits for everything in flight.
struct FusedLayer {
  const float *w1[3], *b1[3], *w2[3];          // phases A, B, C: first linear (+ bias), second linear (4x4x1 order)
  const float *red_bias[3];                    // what a phase's prologue adds and normalises with (phase A: the previous layer's
  mmt_norm red_ln[3];                          //   linear2 bias / norm3 - filled in per layer by the host, layer 0: the last layer's)
};
struct FusedArgs {
  ClArgs a;
  FusedLayer layer[MMT_MAX_LAYERS];
  float *part[2], *xres[2], *selfkv, *memkv;
  unsigned *sync;                              // [0]: time-out word; [32 * (1 + cluster)]: arrival counter
  int nlayers;
  int test_stall;                              // fault injection (MMT_TEST_FUSED_STALL=1, tests only): one workgroup walks away
};

// the per-phase fields of the argument block; everything else stays as the host filled it in
template <int PHASE>
__device__ __forceinline__ void fused_set_phase(ClArgs &x, const FusedArgs &f, int s, int l, unsigned p) {
  const FusedLayer &W = f.layer[l];
  // (scalars too: re-read per phase through an empty volatile asm, or everything derived from them is hoisted out of the loop
  // and spilled)
  auto fresh = [](int v) { asm volatile("" : "+s"(v)); return v; };
  x.d = fresh(f.a.d);
  x.B = fresh(f.a.B);
  x.T = fresh(f.a.T);
  x.H = fresh(f.a.H);
  x.ph = fresh(f.a.ph);
  x.FS16 = fresh(f.a.FS16);
  x.s = s;
  x.L = l;
  x.w1 = W.w1[PHASE];
  x.b1 = W.b1[PHASE];
  x.w2 = W.w2[PHASE];
  x.red_bias = W.red_bias[PHASE];
  x.red_ln = W.red_ln[PHASE];
  x.kv = PHASE == PH_A ? f.selfkv : f.memkv;
  x.part_out = f.part[p & 1];
  x.xres_out = f.xres[p & 1];
  x.part_in = f.part[(p + 1) & 1];
  x.xres_in = f.xres[(p + 1) & 1];
  x.prof_idx = (int)p;
}

template <int DH, int CPW>
__global__ __launch_bounds__(CNT) void mmt_dec_fused4_kernel(FusedArgs f) {
  extern __shared__ __align__(16) float smem[];
  // (XCD-local clusters matter here too although the partials are stored write-through: with slice g on XCD g % 8 instead - an
  // XCD's L2 would then keep its 1/8 of the weights for the whole decode - the decode took 1.32 ms instead of 1.22 ms)
  int cluster, g;
  cluster_of_block(f.a.H, cluster, g);
  const int L = f.nlayers, ph = f.a.ph;
  constexpr int LD1 = 3 * DH + 8;
  float *ps = smem + 16 * (f.a.d + MMT_PAD) + CNW * KS_SMAX * 256 + 16 * LD1 + 3 * 16 * LDO + 16 * LDTOK;
  FusedSync fs;
  fs.ctr = (gu32 *)(f.sync + 32 * (1 + cluster));
  fs.tmo = (gu32 *)f.sync;
  fs.abort = reinterpret_cast<int *>(ps + 128);
  fs.phase = 0;
  fs.H = f.a.H;
  if (thread_idx() == 0) *fs.abort = 0;             // read only behind a workgroup barrier
  ClArgs a = f.a;
  unsigned p = 0;
#pragma clang loop unroll(disable)
  for (int s = 0; s < ph; ++s) {
#pragma clang loop unroll(disable)
    for (int l = 0; l < L; ++l) {
      if (f.test_stall && p == 15 && cluster == 0 && g == 1) return;      // its cluster must time out, the call must turn NaN
      fused_set_phase<PH_A>(a, f, s, l, p);
      bool ok;
      if (l > 0) ok = phase4_body<DH, CPW, PH_A, XM_REDUCE, true>(a, cluster, g, smem, fs);
      else if (s == 0) ok = phase4_body<DH, CPW, PH_A, XM_EMBED, true>(a, cluster, g, smem, fs);
      else ok = phase4_body<DH, CPW, PH_A, XM_HEAD_EMBED, true>(a, cluster, g, smem, fs);
      if (!ok) return;
      fused_set_phase<PH_B>(a, f, s, l, p + 1);
      if (!phase4_body<DH, CPW, PH_B, XM_REDUCE, true>(a, cluster, g, smem, fs)) return;
      fused_set_phase<PH_C>(a, f, s, l, p + 2);
      if (!phase4_body<DH, CPW, PH_C, XM_REDUCE, true>(a, cluster, g, smem, fs)) return;
      p += 3;
    }
  }
}

// ---- the same two linears with the geometry known at compile time (pipeline kernel): every LDS read of the activations is
// issued before the first MFMA and the K-part sums are straight-line code.  Same products, same summation order as gemm1_4 /
// gemm2_4 (bit-identical results).
template <int CG, int KQ>
struct Lin4S {
  static constexpr int cgw = CG >> 3;
  static constexpr int lg_cgw = cgw == 1 ? 0 : cgw == 2 ? 1 : cgw == 4 ? 2 : cgw == 8 ? 3 : 4;
  static constexpr int q = (KQ >> (4 - lg_cgw)) < 1 ? 1 : (KQ >> (4 - lg_cgw));
};
template <int CGW>
__device__ __forceinline__ float kparts_sum_s(float v) {
  if constexpr (CGW <= 8) v = xor32_sum(v);
  if constexpr (CGW <= 4) v = xor16_sum(v);
  if constexpr (CGW <= 1) v += dpp_mov<0x124>(v);
  if constexpr (CGW <= 2) v += dpp_mov<0x128>(v);
  return v;
}
// The same tree for the four rows of one accumulator at a quarter of the instructions: a permlane swap exchanges halves (row
// pairs) between TWO registers, so one swap + one add finishes two values.  Returns, in DPP row r of the wave, the total of
// acc[r] (every in-row block holds it, each with the association kparts_sum gives that block).
template <int CGW>
__device__ __forceinline__ float kparts_sum_rows(const f32x4 &acc) {
  static_assert(CGW <= 4, "both cross-row steps are needed");
  const float u0 = pair32_sum(acc[0], acc[2]);       // lanes 0..31: row 0, lanes 32..63: row 2
  const float u1 = pair32_sum(acc[1], acc[3]);       //              row 1,               row 3
  float v = pair16_sum(u0, u1);                      // DPP row r: accumulator row r
  if constexpr (CGW <= 1) v += dpp_mov<0x124>(v);
  if constexpr (CGW <= 2) v += dpp_mov<0x128>(v);
  return v;
}
// ff_keep >= 0 (phase C, training-mode dropout): the keep byte of the ONE hidden value this lane writes, ff_scale = 1/(1-p)
template <int DH, int PHASE, int D, int FS>
__device__ __forceinline__ void gemm1_4s(const MainRegs4<DH, PHASE> &M, const float *xs, int ldx, float *t1, int ld1, int ff_keep = -1,
                                         float ff_scale = 1.f) {
  typedef Geom4<PHASE> G4;
  typedef Lin4S<(PHASE == PH_C ? FS : DH) / 4, D / 4> L1;
  static_assert(L1::q <= G4::QMAX1, "first linear exceeds the register tile");
  constexpr int KQ = D / 4;
  const int w = wave_id(), lane = lane_id(), b = lane >> 2, j = lane & 3;
  const int kq0 = (b >> L1::lg_cgw) * L1::q;
  const float *xr = xs + j * ldx;
  f32x4 xv[L1::q];
#pragma unroll
  for (int t = 0; t < L1::q; ++t) xv[t] = *reinterpret_cast<const f32x4 *>(xr + 4 * min(kq0 + t, KQ - 1));
  f32x4 acc[G4::NPART];
#pragma unroll
  for (int p = 0; p < G4::NPART; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < L1::q; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int p = 0; p < G4::NPART; ++p) acc[p] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[t][e], M.w1[p][t][e], acc[p], 0, 0, 0);
  // lane (DPP row r, in-row block bb, j) ends with row r of every part; block bb writes part bb (phase A) / column group bb
  float tot[G4::NPART];
#pragma unroll
  for (int p = 0; p < G4::NPART; ++p) tot[p] = kparts_sum_rows<L1::cgw>(acc[p]);
  const int r = b >> 2, bb = b & 3;
  if constexpr (PHASE == PH_A) {
    if (bb < 3) t1[r * ld1 + bb * DH + w * 4 + j] = (bb == 0 ? tot[0] : bb == 1 ? tot[G4::NPART > 1 ? 1 : 0] : tot[G4::NPART > 2 ? 2 : 0]) + M.b1;
  } else if (bb < (PHASE == PH_C ? L1::cgw : 1)) {
    float v = PHASE == PH_C ? fmaxf(tot[0] + M.b1, 0.f) : tot[0] + M.b1;
    if (PHASE == PH_C && ff_keep >= 0) v = ff_keep ? v * ff_scale : 0.f;
    t1[r * ld1 + (w * L1::cgw + bb) * 4 + j] = v;
  }
}
// the accumulators of one second linear, K parts summed (every lane of a column group's blocks holds the total)
template <int DH, int PHASE, int D, int FS>
__device__ __forceinline__ f32x4 gemm2_4s_acc(const MainRegs4<DH, PHASE> &M, const float *x2, int ld2) {
  typedef Geom4<PHASE> G4;
  constexpr int K2q = (PHASE == PH_C ? FS : DH) / 4;
  typedef Lin4S<D / 4, K2q> L2;
  static_assert(L2::q <= G4::QMAX2, "second linear exceeds the register tile");
  const int lane = lane_id(), b = lane >> 2, j = lane & 3;
  const int kq0 = (b >> L2::lg_cgw) * L2::q;
  const float *xr = x2 + j * ld2;
  f32x4 xv[L2::q];
#pragma unroll
  for (int t = 0; t < L2::q; ++t) xv[t] = *reinterpret_cast<const f32x4 *>(xr + 4 * min(kq0 + t, K2q - 1));
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < L2::q; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[t][e], M.w2[t][e], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = kparts_sum_s<L2::cgw>(acc[r]);
  return acc;
}
// two heads of one workgroup (cross attention): ONE part = tag(tag(part of head g0) + tag(part of head g0 + 1)), in the slot of g0
template <int DH, int PHASE, int D, int FS, int AUX>
__device__ __forceinline__ void gemm2_4s_pair(const MainRegs4<DH, PHASE> (&M)[2], const float *x2, int ld2, int hstride, float *part,
                                              unsigned tag, int ldp = D) {
  typedef Lin4S<D / 4, (PHASE == PH_C ? FS : DH) / 4> L2;
  const int w = wave_id(), lane = lane_id(), b = lane >> 2, j = lane & 3;
  const f32x4 a0 = gemm2_4s_acc<DH, PHASE, D, FS>(M[0], x2, ld2), a1 = gemm2_4s_acc<DH, PHASE, D, FS>(M[1], x2 + hstride, ld2);
  f32x4 acc = tag4(tag4(a0, tag) + tag4(a1, tag), tag);
  quad_transpose(acc, j);
  bs4x<AUX>(mk_rsrc(part), b < L2::cgw ? (unsigned)(j * ldp + (w * L2::cgw + b) * 4) * 4u : OOB, acc);
}
template <int DH, int PHASE, int D, int FS, int AUX>
__device__ __forceinline__ void gemm2_4s(const MainRegs4<DH, PHASE> &M, const float *x2, int ld2, float *part, unsigned tag, int ldp = D) {
  typedef Geom4<PHASE> G4;
  constexpr int K2q = (PHASE == PH_C ? FS : DH) / 4;
  typedef Lin4S<D / 4, K2q> L2;
  static_assert(L2::q <= G4::QMAX2, "second linear exceeds the register tile");
  const int w = wave_id(), lane = lane_id(), b = lane >> 2, j = lane & 3;
  const int kq0 = (b >> L2::lg_cgw) * L2::q;
  const float *xr = x2 + j * ld2;
  f32x4 xv[L2::q];
#pragma unroll
  for (int t = 0; t < L2::q; ++t) xv[t] = *reinterpret_cast<const f32x4 *>(xr + 4 * min(kq0 + t, K2q - 1));
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < L2::q; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[t][e], M.w2[t][e], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = kparts_sum_s<L2::cgw>(acc[r]);
  quad_transpose(acc, j);                            // lane j of a quad: row j, the quad's 4 columns -> one 16-byte store
  acc = tag4(acc, tag);
  bs4x<AUX>(mk_rsrc(part), b < L2::cgw ? (unsigned)(j * ldp + (w * L2::cgw + b) * 4) * 4u : OOB, acc);
}

// ------------------------------------------------------------------------------- weight-stationary pipeline (one launch)
// The fused kernel above re-requests its 128 KB weight slice in every one of the 216 phases (0.9 us through one CU's address
// path per phase, 2 GB per decode through the L2s).  Here a workgroup is bound to ONE (layer, sub-layer, slice) for the whole
// decode and keeps that slice in registers (64 VGPRs per lane); the 4-agent clusters flow through the 3L + 1 stages:
//   stage kinds per layer:  A  self attention,  H workgroups (one head each)
//                           B  cross attention, H/2 workgroups (two heads each: half the weights of A / C per head)
//                           C  feed forward,    H workgroups (d_ff/H hidden columns each)
//   + head workgroups: reduce + LayerNorm of the last layer, output_fc, [fusion/hist_map], decoder/MLP, position feedback and
//     the embedding of the next token (stage 0 of every step).
// The whole pipeline is replicated R times (replica r owns the clusters m = r mod R) to fill the chip: 2 x (6 x 20 + 4) = 248
// workgroups on 256 CUs at the shipped size.  Hand-offs, buffers and arithmetic are those of the fused kernel (bit-identical
// outputs): phase p = step * (3L + 1) + stage writes part[p & 1] / xres[p & 1], reads the other pair, and may start on cluster m
// once the cluster's arrival counter shows H * p (every stage adds H in total: A and C workgroups 1 each, B workgroups 2, the head
// workgroup H).  A stage visits (step, cluster) in lexicographic order and depends only on the previous stage of the same
// cluster, so with all workgroups resident nothing can dead-lock; every wait is bounded as in the fused kernel.
struct PipeArgs {
  ClArgs a;
  FusedLayer layer[MMT_MAX_LAYERS];
  float *part[2], *xres[2], *selfkv, *memkv;
  unsigned *sync;                              // [0]: time-out word
  float *x0buf;                                // [cluster][4][d]: the embedded token of the current step, head -> first stage
  const float *head_bias;                      // what the head stage's reduction adds / normalises with: the last layer's
  mmt_norm head_ln;                            //   linear2 bias and norm3
  int nlayers, test_stall, R, NHW, NC, pad_;
  mmt_decoder_dropout drop;                    // DROP instantiation only: the training step's keep-byte sites
  float *stash;                                // STASH instantiation only: the training step's activation stash (stash_off)
};
// Layout of the training stash (include/mmt_amd.h, mmt_decoder_stash_layout): one block per layer + a tail block; rows are
// step-major (row s * B + b), the attention probabilities (B, H, ph, Lk) - the layouts the training kernels read.
__host__ __device__ __forceinline__ mmt_decoder_stash_layout stash_off(int B, int T, int ph, int d, int dff, int H, int n_out, int nlayers,
                                                                       int map_dim = 0) {
  const size_t R = (size_t)ph * B, Rd = R * d;
  mmt_decoder_stash_layout o;
  o.x = 0, o.qkv = Rd, o.a1 = 4 * Rd, o.xhat1 = 5 * Rd, o.x1 = 6 * Rd, o.cq = 7 * Rd, o.a2 = 8 * Rd, o.xhat2 = 9 * Rd, o.x2 = 10 * Rd;
  o.xhat3 = 11 * Rd, o.f1 = 12 * Rd;
  o.rstd1 = o.f1 + R * dff, o.rstd2 = o.rstd1 + R, o.rstd3 = o.rstd2 + R;
  o.p1 = o.rstd3 + R, o.p2 = o.p1 + (size_t)B * H * ph * ph;
  o.layer_floats = (o.p2 + (size_t)B * H * ph * T + 63) & ~(size_t)63;
  o.tail = (size_t)nlayers * o.layer_floats;
  o.ho_ld = (size_t)(n_out + map_dim);
  o.t_x = 0, o.t_ho = Rd, o.t_hu = o.t_ho + R * o.ho_ld, o.t_hxhat = o.t_hu + R * 32, o.t_u = o.t_hxhat + R * 32;
  o.t_hrstd = o.t_u + (map_dim > 0 ? R * n_out : 0);
  o.total_floats = o.tail + ((o.t_hrstd + R + 63) & ~(size_t)63);
  return o;
}
constexpr int PIPE_MAXPOS = 16;                // clusters whose last positions a head workgroup keeps in LDS (others: pred_st)

struct PipeLds {
  float *xs, *red, *ha, *hb, *tok, *t1, *ob, *ps, *Ks, *Vs, *pos, *prm;
  int *abort;
};
constexpr int PIPE_PRM = 64 * (1 + 3 * CHAIN_MAX);      // head workgroups: biases and LayerNorm vectors of the small linears
template <int DH>
__host__ __device__ __forceinline__ size_t pipe_lds_floats(int d) {
  constexpr int LD1 = 3 * DH + 8, LDK = DH + 4;
  return (size_t)16 * (d + MMT_PAD) + CNW * KS_SMAX * 256 + 2 * 16 * LDO + 16 * LDTOK + 2 * 4 * LD1 + 2 * 4 * LDO + 256 +
         2 * (2 * 4 * KMAX * LDK) + PIPE_MAXPOS * 64 + PIPE_PRM;
}
template <int DH>
__device__ __forceinline__ PipeLds pipe_lds(float *smem, int d) {
  constexpr int LD1 = 3 * DH + 8, LDK = DH + 4;
  PipeLds L;
  L.xs = smem;                                   // [16][ldx]  rows 0..3: input of the sub-layer (16 rows where the head runs)
  L.red = L.xs + 16 * (d + MMT_PAD);             // K-split partial tiles of the head's small linears
  L.ha = L.red + CNW * KS_SMAX * 256;
  L.hb = L.ha + 16 * LDO;
  L.tok = L.hb + 16 * LDO;
  L.t1 = L.tok + 16 * LDTOK;                     // [2][4][LD1]  q|k|v of a head, or the FFN hidden slice
  L.ob = L.t1 + 2 * 4 * LD1;                     // [2][4][LDO]  attention output of a head
  L.ps = L.ob + 2 * 4 * LDO;                     // [2][64]      attention probabilities
  L.abort = reinterpret_cast<int *>(L.ps + 128);
  L.Ks = L.ps + 256;                             // [2][4][KMAX][LDK]
  L.Vs = L.Ks + 2 * 4 * KMAX * LDK;
  L.pos = L.Vs + 2 * 4 * KMAX * LDK;             // [PIPE_MAXPOS][4][16]
  L.prm = L.pos + PIPE_MAXPOS * 64;              // [PIPE_PRM]
  return L;
}

#ifndef FW_PROBES
#define FW_PROBES 1
#endif
// (timing experiments: which of the stash-keeping decode's stores cost what - all on in every shipped build)
#ifndef STASH_RED
#define STASH_RED 1
#endif
#ifndef STASH_T1
#define STASH_T1 1
#endif
#ifndef STASH_ATT
#define STASH_ATT 1
#endif
#ifndef FW_PROBE_SLOT0
#define FW_PROBE_SLOT0 1        // probe the part of slice 0 (that workgroup also writes the residual row): 1.049 -> 1.044 ms at B = 128
#endif
#ifndef FW_POLL_FIRST
#define FW_POLL_FIRST 0
#endif
#ifndef FW_COND_WAIT
#define FW_COND_WAIT 0
#endif
#ifndef FW_ROW_PARTS
#define FW_ROW_PARTS 0
#endif
// ---- data-tagged hand-off (see "tag bits" above): the consumer's red waves poll the rows themselves --------------------------
constexpr unsigned long long PIPE_WAIT_TICKS = 20000000ull;     // 0.2 s of the 100 MHz wall clock: a wait this long = give up
struct PipeSync {
  gu32 *tmo;             // time-out word of the launch (global)
  int *abort;            // LDS
};
// give up: raise the launch's time-out word (every other workgroup follows at its next check) and this workgroup's abort flag
__device__ __forceinline__ void pipe_give_up(const PipeSync &ps, int phase, int cluster) {
  if (atomicCAS(&g_decoder_fault_info[0], 0u, 1u) == 0u) {      // the first one to give up says where (mmt_decoder_status)
    g_decoder_fault_info[1] = blockIdx.x;
    g_decoder_fault_info[2] = (unsigned)phase;
    g_decoder_fault_info[3] = (unsigned)cluster;
  }
  __hip_atomic_store(ps.tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  *ps.abort = 1;
}
// A red wave (row `row` of the cluster's 4) fetches what the previous phase left for it - residual row and the H partial rows,
// or (X0) the embedded token - and re-fetches until every word carries the expected tag.  The first attempt takes everything (in
// the steady state the rows are there or land within a round trip); after a miss the wave probes ONE 64-byte piece (four lanes)
// of the last partial row, with s_sleep, until it shows the tag - idle stages must not load the memory system - and then takes
// everything again.  Returns false when it gave up (wave-uniform).
// GS: stride of the part slots that exist (2 behind the cross-attention stage, whose workgroups hand over one pre-summed part
// for their two heads, in the slot of the even head)
// the loads of one full attempt (poll_rows<.., PRE = true>: the caller issues the first attempt AHEAD of the visit's other requests -
// vector memory returns in order)
template <bool X0, int GS = 1>
__device__ __forceinline__ void poll_issue(Red4Regs &R, const ClArgs &a, const float *x0buf, int cluster) {
  const int row = wave_id(), q = lane_id();
  const unsigned co = q < (a.d >> 2) ? (unsigned)q * 16u : OOB, gstride = 16u * a.d * 4u;
  R.res = bl4x<AUX_SC1>(mk_rsrc(X0 ? x0buf + ((size_t)cluster * 4 + row) * a.d : a.xres_in + ((size_t)cluster * 16 + row) * a.d), co);
  if constexpr (!X0) {
    const rsrc_t pp = mk_rsrc(a.part_in + ((size_t)cluster * a.H * 16 + row) * a.d);
#pragma unroll
    for (int g = 0; g < 8 / GS; ++g) R.p[g] = bl4x<AUX_SC1>(pp, g * GS < a.H ? co : OOB, g * GS < a.H ? g * GS * gstride : 0u);
  }
}
template <bool X0, int GS = 1, bool PRE = false>
__device__ __forceinline__ bool poll_rows(Red4Regs &R, const ClArgs &a, const float *x0buf, int cluster, unsigned tag, const PipeSync &ps) {
  const int row = wave_id(), q = lane_id();
  const bool live = q < (a.d >> 2);
#if FW_ROW_PARTS
  // parts of a cluster as [16 rows][H][d]: the H parts of a row are contiguous (as in the backward sweep, decoder_bwd_pipe.cuh)
  const unsigned co = live ? (unsigned)q * 16u : OOB, gstride = (unsigned)a.d * 4u;
  const rsrc_t pp = mk_rsrc(a.part_in + ((size_t)cluster * 16 + row) * a.H * a.d);
#else
  const unsigned co = live ? (unsigned)q * 16u : OOB, gstride = 16u * a.d * 4u;
  const rsrc_t pp = mk_rsrc(a.part_in + ((size_t)cluster * a.H * 16 + row) * a.d);
#endif
  const unsigned probe = q < 4 ? (unsigned)q * 16u : OOB;
  const rsrc_t res = mk_rsrc(X0 ? x0buf + ((size_t)cluster * 4 + row) * a.d : a.xres_in + ((size_t)cluster * 16 + row) * a.d);
  unsigned long long t0 = 0;
  for (int tries = 0;; ++tries) {
    asm volatile("" ::: "memory");       // the loads below are re-issued every pass (buffer loads are not volatile to the compiler)
    if (!(PRE && tries == 0)) R.res = bl4x<AUX_SC1>(res, co);
    bool ok = !live || tag_ok(R.res, tag);
    if constexpr (!X0) {
#pragma unroll
      for (int g = 0; g < 8 / GS; ++g) {
        if (!(PRE && tries == 0)) R.p[g] = bl4x<AUX_SC1>(pp, g * GS < a.H ? co : OOB, g * GS < a.H ? g * GS * gstride : 0u);
        ok = ok && (g * GS >= a.H || !live || tag_ok(R.p[g], tag));
      }
      if constexpr (GS > 1) {
#pragma unroll
        for (int g = 8 / GS; g < 8; ++g) R.p[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    if (__all(ok)) return true;
    if (!t0) t0 = wall_clock64();
    if (tries > 0) {                     // the probe matched but a row is still on its way: look again shortly
      __builtin_amdgcn_s_sleep(1);
      if ((tries & 63) == 0 && wall_clock64() - t0 > PIPE_WAIT_TICKS) {
        if (q == 0) pipe_give_up(ps, a.prof_idx + 1000 * row, cluster);
        return false;
      }
      continue;
    }
#if FW_PROBES <= 1
    for (unsigned n = 0;; ++n) {         // thin polling
      __builtin_amdgcn_s_sleep(2);
      asm volatile("" ::: "memory");
      const f32x4 v = X0 ? bl4x<AUX_SC1>(res, probe) : bl4x<AUX_SC1>(pp, probe, FW_PROBE_SLOT0 ? 0u : (unsigned)(a.H - GS) * gstride);
      if (__all(q >= 4 || tag_ok(v, tag))) break;
      if ((n & 15u) == 15u) {
        const bool late = wall_clock64() - t0 > PIPE_WAIT_TICKS;
        if (late || __hip_atomic_load(ps.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          if (q == 0) {
            if (late) pipe_give_up(ps, a.prof_idx + 1000 * row + (X0 ? 100000 : 0), cluster);
            else *ps.abort = 1;          // another workgroup gave up: follow
          }
          return false;
        }
      }
    }
#else
    {   // FW_PROBES thin probes in flight (round 3, as the backward sweep's bwd_poll_rows): a probe takes a memory round trip, and
        // every workgroup of a stage notices its input at a random phase of its probe period
      auto probe_load = [&]() -> f32x4 {
        return X0 ? bl4x<AUX_SC1>(res, probe) : bl4x<AUX_SC1>(pp, probe, FW_PROBE_SLOT0 ? 0u : (unsigned)(a.H - GS) * gstride);
      };
      f32x4 pv[FW_PROBES];
#pragma unroll
      for (int i = 0; i < FW_PROBES; ++i) {
        asm volatile("" ::: "memory");
        pv[i] = probe_load();
      }
      bool out = false;
      for (unsigned n = 0; !out; ++n) {
#pragma unroll
        for (int i = 0; i < FW_PROBES; ++i) {
          if (out) break;
          const f32x4 v = pv[i];
          asm volatile("" ::: "memory");
          pv[i] = probe_load();
          asm volatile("" ::: "memory");
          if (__all(q >= 4 || tag_ok(v, tag))) out = true;
          else __builtin_amdgcn_s_sleep(3);
        }
        if (!out && (n & 7u) == 7u) {
          const bool late = wall_clock64() - t0 > PIPE_WAIT_TICKS;
          if (late || __hip_atomic_load(ps.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            if (q == 0) {
              if (late) pipe_give_up(ps, a.prof_idx + 1000 * row + (X0 ? 100000 : 0), cluster);
              else *ps.abort = 1;
            }
            return false;
          }
        }
      }
    }
#endif
  }
}

// attention of one head for the 4 agents of a cluster, on ONE wave: one (agent, key) pair per lane; phase A takes the key / value
// of this step from t1 (same arithmetic, same order as phase4_body)
#ifndef PIPE_ATT_ZEROED
#define PIPE_ATT_ZEROED true
#endif
// ZEROED: the staging tiles were zeroed once and only rows that hold keys are ever written, so rows beyond the keys are finite
// zeros and meet probabilities that are zeros: no select per key (16 lane masks = 32 spilled SGPRs in this kernel)
template <int DH, int PHASE, bool ZEROED = false>
__device__ __forceinline__ void attend4(const ClArgs &a, int na, int pad, const float *t1, int ld1, const float *Ks, const float *Vs,
                                        float *ps, float *ob, int visit = 0, int att_keep = -1, float *st_p = nullptr,
                                        size_t st_p_stride = 0, float *st_o = nullptr) {
  // st_p / st_o (training stash): the softmax (before its dropout) of agent ar goes to st_p[ar * st_p_stride + key], the head's
  // attention output to st_o[ar * d + column]
  constexpr int LDK = DH + 4;
  const int nglob = PHASE == PH_A ? a.s : a.T;
  const int nkeys = PHASE == PH_A ? a.s + 1 : a.T;
  const int lane = lane_id(), ar = lane >> 4, l16 = lane & 15;
  float sc = -INFINITY;
  bool masked = ar >= na || l16 >= nkeys;
  if constexpr (PHASE == PH_B) masked = masked || pad != 0;
  if (!masked) {
    const float *q = t1 + ar * ld1;
    const float *k = (PHASE == PH_A && l16 == nglob) ? q + DH : Ks + (ar * KMAX + l16) * LDK;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) {
      const f32x4 qv = *reinterpret_cast<const f32x4 *>(q + 4 * c), kv4 = *reinterpret_cast<const f32x4 *>(k + 4 * c);
      acc = fmaf(qv[0], kv4[0], acc);
      acc = fmaf(qv[1], kv4[1], acc);
      acc = fmaf(qv[2], kv4[2], acc);
      acc = fmaf(qv[3], kv4[3], acc);
    }
    sc = acc * (1.0f / sqrtf((float)DH));
  }
  PSTAMP(visit, 8);
  const float mx = row16_max(sc);
  float p = mx > -INFINITY ? expf(sc - mx) : 0.f;    // a fully masked row yields zeros (reference: undefined / NaN)
  const float sum = row16_sum(p);
  p *= sum > 0.f ? 1.0f / sum : 0.f;
  if (st_p && ar < na && l16 < nkeys) st_p[(size_t)ar * st_p_stride + l16] = p;
  if (att_keep >= 0 && !masked) p = att_keep ? p * a.ds_att : 0.f;      // dropout on the probabilities (as phase4_body<DROP>)
  ps[lane] = p;
  PSTAMP(visit, 9);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // ps is written and read by this wave only
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  constexpr int DPL = DH / 16;
  float o[DPL];
#pragma unroll
  for (int e = 0; e < DPL; ++e) o[e] = 0.f;
  const int nk = ar < na ? nglob : 0;
  // all LDS reads in flight at once (a loop with a run-time trip count pays one LDS round trip per key), then the same
  // fused multiply-adds in the same order; rows beyond nk hold stale bits and are never multiplied
  typedef float vrow __attribute__((ext_vector_type(DPL)));
  f32x4 pr[KMAX / 4];
  vrow vv[KMAX];
#pragma unroll
  for (int c = 0; c < KMAX / 4; ++c) pr[c] = *reinterpret_cast<const f32x4 *>(ps + ar * 16 + 4 * c);
#pragma unroll
  for (int j = 0; j < KMAX; ++j) vv[j] = *reinterpret_cast<const vrow *>(Vs + (ar * KMAX + j) * LDK + l16 * DPL);
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    const float pj = pr[j >> 2][j & 3];
#pragma unroll
    for (int e = 0; e < DPL; ++e) o[e] = (ZEROED || j < nk) ? fmaf(pj, vv[j][e], o[e]) : o[e];
  }
  if (PHASE == PH_A && ar < na) {
    const float pj = ps[ar * 16 + nglob];
#pragma unroll
    for (int e = 0; e < DPL; ++e) o[e] = fmaf(pj, t1[ar * ld1 + 2 * DH + l16 * DPL + e], o[e]);
  }
#pragma unroll
  for (int e = 0; e < DPL; ++e) ob[ar * LDO + l16 * DPL + e] = o[e];
  if (st_o && ar < na) {
#pragma unroll
    for (int e = 0; e < DPL; ++e) st_o[(size_t)ar * a.d + l16 * DPL + e] = o[e];
  }
  PSTAMP(visit, 10);
}

// One visit of a stage workgroup to one cluster: NH heads / slices starting at g0, weights already in M.
// kv_ready: M[].kv / pad already hold this visit's K/V cache rows (requested during the previous visit); nx_*: the next visit,
// whose rows are requested behind this visit's attention (nx_cluster < 0: none, or they are not final yet).
// STASH: stl = this layer's block of the training stash (stash_off), nlayers the stack's depth
template <int DH, int PHASE, int NH, int D, int FS, bool DROP = false, bool STASH = false>
__device__ __forceinline__ bool pipe_phase(const ClArgs &a, const float *x0buf, int cluster, int g0, bool loadx, const PipeLds &S,
                                           const PipeSync &ps, MainRegs4<DH, PHASE> (&M)[NH], int &pad, bool kv_ready, int nx_step,
                                           int nx_cluster, int visit, bool wait_appends = true, bool last_phase = false,
                                           float *stl = nullptr, int nlayers = 0) {
  constexpr int AUX = AUX_SC1;
  constexpr int LD1 = 3 * DH + 8, LDK = DH + 4;
  const int d = a.d, ldx = d + MMT_PAD, tid = thread_idx();
  const int a0 = cluster * 4, na = min(4, a.B - a0);
  mmt_decoder_stash_layout so = {};
  size_t st_row0 = 0;                 // stash row of the cluster's first agent at this step
  if constexpr (STASH) {
    so = stash_off(a.B, a.T, a.ph, d, FS * a.H, a.H, a.n_out, nlayers, a.has_map_fusion ? a.map_dim : 0);
    st_row0 = (size_t)a.s * a.B + a0;
  }
  const unsigned tag_out = phase_tag(a), tag_in = tag_of_phase((unsigned)a.prof_idx - 1u);
  PSTAMP(visit, 0);
  PSTAMPC(visit, 12);
  auto kv_request = [&](int cl, int step) {
    const int b0 = cl * 4, nb = min(4, a.B - b0);
#pragma unroll
    for (int h = 0; h < NH; ++h) kv_issue4<DH, PHASE, AUX_PLAIN>(M[h], a, b0, nb, g0 + h, step);
    if constexpr (PHASE == PH_B) {
      const int ar = lane_id() >> 4, l16 = lane_id() & 15;
      pad = __builtin_amdgcn_raw_buffer_load_b8(mk_rsrc(a.pad_mask), (a.pad_mask && wave_id() < NH && ar < nb && l16 < a.T)
                                                    ? (unsigned)((b0 + ar) * a.T + l16) : OOB, 0, 0);
    }
  };
  Red4Regs R;
  const bool red_wave = wave_id() < 4;
#if FW_POLL_FIRST
  // ---- 0a. the first look at the previous phase's rows goes out AHEAD of everything else the visit requests
  if (red_wave) {
    if (loadx) poll_issue<true>(R, a, x0buf, cluster);
    else poll_issue<false, PHASE == PH_C ? 2 : 1>(R, a, x0buf, cluster);
  }
  __builtin_amdgcn_sched_barrier(0);
#endif
  // ---- 0. this cluster's K/V cache rows do not depend on the previous phase: requested first, staged once the poll is through
  if (!kv_ready) {
#if FW_POLL_FIRST || FW_COND_WAIT
    // (the cache rows of this cluster were appended a whole step ago unless the replica has a single cluster)
    if constexpr (PHASE == PH_A) { if (wait_appends) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#else
    if constexpr (PHASE == PH_A) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's earlier cache appends have left
#endif
    kv_request(cluster, a.s);
  }
  // ---- 1. the previous phase's rows, polled until they carry its tag
  bool got = true;
  // training-mode dropout: this visit's keep bytes do not depend on the previous phase either -- requested before the poll
  // (branch-free: a load under a branch makes the compiler wait for everything in flight where the branch rejoins)
  unsigned k_red = 0u, kb_ff = 0u, kb_att = 0u;
  bool on_ff = false, on_att = false;
  if constexpr (DROP) {
    if (red_wave && !loadx) {
      const int ag = a0 + wave_id();
      k_red = __builtin_amdgcn_raw_buffer_load_b32(mk_rsrc(a.dk_red), (a.dk_red && lane_id() < (d >> 2) && ag < a.B)
                                                       ? (unsigned)(ag * d + 4 * lane_id()) : OOB, 0, 0);
    }
    if constexpr (PHASE == PH_C) {      // gemm1_4s: lane (DPP row r, in-row block bb, j) writes hidden column (w*cgw + bb)*4 + j of row r
      constexpr int cgw = (FS / 4) >> 3;
      const int b = lane_id() >> 2, j = lane_id() & 3, r = b >> 2, bb = b & 3;
      on_ff = a.dk_ff && bb < cgw && r < na;
      kb_ff = __builtin_amdgcn_raw_buffer_load_b8(mk_rsrc(a.dk_ff), on_ff ? (unsigned)((a0 + r) * (FS * a.H) + g0 * FS + (wave_id() * cgw + bb) * 4 + j) : OOB, 0, 0);
    } else {
      const int ar = lane_id() >> 4, l16 = lane_id() & 15, Lk = PHASE == PH_A ? a.ph : a.T;
      on_att = a.dk_att && wave_id() < NH && ar < na && l16 < (PHASE == PH_A ? a.s + 1 : a.T);
      kb_att = __builtin_amdgcn_raw_buffer_load_b8(mk_rsrc(a.dk_att), on_att ? (unsigned)(((a0 + ar) * a.H + g0 + wave_id()) * a.ph * Lk + l16) : OOB, 0, 0);
    }
  }
  if (red_wave) {
    if (!loadx) {
      const unsigned co = lane_id() < (d >> 2) ? (unsigned)lane_id() * 16u : OOB;
      R.bias = bl4(mk_rsrc(a.red_bias), co);
      R.gam = bl4(mk_rsrc(a.red_ln.g), co);
      R.bet = bl4(mk_rsrc(a.red_ln.b), co);
    }
    got = loadx ? poll_rows<true, 1, FW_POLL_FIRST != 0>(R, a, x0buf, cluster, (unsigned)a.s & TAG_MASK, ps)
                : poll_rows<false, PHASE == PH_C ? 2 : 1, FW_POLL_FIRST != 0>(R, a, x0buf, cluster, tag_in, ps);
  }
  (void)got;                                           // a wave that gave up has raised the abort flag
  PSTAMP(visit, 1);
  // ---- 2. input of this sub-layer
  if (red_wave) {
    // STASH: the slice-0 workgroup keeps the sub-layer's input row (x / x1 / x2) and the LayerNorm that made it (self-attention
    // stage: the PREVIOUS layer's norm3)
    // (parked in rows 4..8 of the input tile, which only the head workgroups use, and written out behind the visit's hand-off:
    // stash_flush below - a global store ahead of the hand-off costs 0.2 us per hop)
    float *st_x = nullptr, *st_xhat = nullptr, *st_rstd = nullptr;
    if constexpr (STASH && STASH_RED != 0) {
      if (g0 == 0 && wave_id() < na) {
        st_x = S.xs + (12 + wave_id()) * ldx;          // (reduce4_finish also writes the row to xs itself: this copy is not read)
        st_xhat = S.xs + (4 + wave_id()) * ldx;
        st_rstd = S.xs + 8 * ldx + wave_id();
      }
    }
    if (loadx) {
      if (lane_id() < (d >> 2)) {
        *reinterpret_cast<f32x4 *>(S.xs + wave_id() * ldx + 4 * lane_id()) = R.res;
        if (g0 == 0) bs4x<AUX>(mk_rsrc(a.xres_out + ((size_t)cluster * 16 + wave_id()) * d), (unsigned)lane_id() * 16u, tag4(R.res, tag_out));
      }
    } else {
      reduce4_finish<AUX, DROP, false, STASH>(R, a, cluster, S.xs, ldx, g0 == 0, true, DROP ? &k_red : nullptr, st_x, st_xhat, st_rstd);
    }
  }
  // K/V cache rows (the self-attention cache was appended by this very workgroup in earlier steps): registers -> staging tiles
  if constexpr (PHASE != PH_C) {
    const int nglob = PHASE == PH_A ? a.s : a.T;
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int it = 0; it < DH / 16; ++it) {
        int ag, j, kv, c4;
        kv_unit<DH>(tid + CNT * it, ag, j, kv, c4);
        if (ag < na && j < nglob)
          *reinterpret_cast<f32x4 *>((kv ? S.Vs : S.Ks) + ((h * 4 + ag) * KMAX + j) * LDK + 4 * c4) = M[h].kv[it];
      }
  }
  const int pad_now = pad;
  PSTAMP(visit, 11);
  __syncthreads();
  if (*ps.abort) return false;
  PSTAMP(visit, 2);
  // ---- 3. first linear
#pragma unroll
  for (int h = 0; h < NH; ++h) gemm1_4s<DH, PHASE, D, FS>(M[h], S.xs, ldx, S.t1 + h * 4 * LD1, LD1, (DROP && on_ff) ? (int)(kb_ff & 0xffu) : -1, a.ds_ff);
  PSTAMP(visit, 7);
  __syncthreads();
  PSTAMP(visit, 3);
  if constexpr (STASH && STASH_T1 != 0) {      // the first linear's result (q|k|v of the head, the heads' cross queries, the hidden slice): waves 2, 3
    constexpr int C4 = DH / 4;
    const int i = tid - 128;
    if constexpr (PHASE == PH_A) {
      if (i >= 0 && i < 4 * 3 * C4) {
        const int r = i / (3 * C4), rem = i - r * 3 * C4, part = rem / C4, c4 = rem - part * C4;
        if (r < na)
          *reinterpret_cast<f32x4 *>(stl + so.qkv + (st_row0 + r) * 3 * d + part * d + g0 * DH + 4 * c4) =
              *reinterpret_cast<const f32x4 *>(S.t1 + r * LD1 + part * DH + 4 * c4);
      }
    } else if constexpr (PHASE == PH_B) {
      if (i >= 0 && i < NH * 4 * C4) {
        const int h = i / (4 * C4), rem = i - h * 4 * C4, r = rem / C4, c4 = rem - r * C4;
        if (r < na)
          *reinterpret_cast<f32x4 *>(stl + so.cq + (st_row0 + r) * d + (g0 + h) * DH + 4 * c4) =
              *reinterpret_cast<const f32x4 *>(S.t1 + (h * 4 + r) * LD1 + 4 * c4);
      }
    } else {
      if (i >= 0 && i < 4 * (FS / 4)) {
        const int r = i / (FS / 4), c4 = i - r * (FS / 4);
        if (r < na)
          *reinterpret_cast<f32x4 *>(stl + so.f1 + (st_row0 + r) * (size_t)(FS * a.H) + g0 * FS + 4 * c4) =
              *reinterpret_cast<const f32x4 *>(S.t1 + r * LD1 + 4 * c4);
      }
    }
  }
  // stash: row w of the sub-layer's input (and its LayerNorm's normalised row, 1/std, parked in rows 4..8 of the input tile) goes
  // out on wave `wv` - a wave that idles through the attention (A, B), or the row's own red wave behind the hand-off (C)
  auto stash_rows = [&](int w, int q) {
    if (g0 != 0 || w >= na || q >= (d >> 2)) return;
    const size_t row = st_row0 + w;
    *reinterpret_cast<f32x4 *>(stl + (PHASE == PH_A ? so.x : PHASE == PH_B ? so.x1 : so.x2) + row * d + 4 * q) =
        *reinterpret_cast<const f32x4 *>(S.xs + w * ldx + 4 * q);
    if (!loadx) {
      float *lnl = PHASE == PH_A ? stl - so.layer_floats : stl;
      *reinterpret_cast<f32x4 *>(lnl + (PHASE == PH_A ? so.xhat3 : PHASE == PH_B ? so.xhat1 : so.xhat2) + row * d + 4 * q) =
          *reinterpret_cast<const f32x4 *>(S.xs + (4 + w) * ldx + 4 * q);
      if (q == 0) lnl[(PHASE == PH_A ? so.rstd3 : PHASE == PH_B ? so.rstd1 : so.rstd2) + row] = S.xs[8 * ldx + w];
    }
  };
  // ---- 4. attention: wave h serves head g0 + h
  if constexpr (PHASE != PH_C) {
    if constexpr (STASH && STASH_RED != 0) {
      if (wave_id() >= 4) stash_rows(wave_id() - 4, lane_id());      // (xs is not written again before the next visit's poll)
    }
    if (wave_id() < NH) {
      const int h = wave_id();
      float *st_p = nullptr;      // the softmax before its dropout: parked in row 9 of the input tile (16 keys per agent), see stash_flush
      if constexpr (STASH && STASH_ATT != 0) st_p = S.xs + 9 * ldx + h * 64;
      attend4<DH, PHASE, PIPE_ATT_ZEROED>(a, na, pad_now, S.t1 + h * 4 * LD1, LD1, S.Ks + h * 4 * KMAX * LDK, S.Vs + h * 4 * KMAX * LDK, S.ps + h * 64,
                         S.ob + h * 4 * LDO, visit, (DROP && on_att) ? (int)(kb_att & 0xffu) : -1, st_p, 16, nullptr);
    } else if (PHASE == PH_A && tid >= 256) {        // meanwhile: the new key / value into the cache (read back by this workgroup only)
      const int i = tid - 256;
      if (i < na * 2 * (DH / 4)) {
        const int r = i / (2 * (DH / 4)), rem = i - r * 2 * (DH / 4), kv = rem / (DH / 4), c4 = rem - kv * (DH / 4);
        *reinterpret_cast<f32x4 *>(a.kv + (((size_t)a.L * a.B + a0 + r) * a.ph + a.s) * 2 * d + kv * d + g0 * DH + 4 * c4) =
            *reinterpret_cast<const f32x4 *>(S.t1 + r * LD1 + (1 + kv) * DH + 4 * c4);
      }
    }
    // the next visit's K/V cache rows: requested now, staged at the start of that visit (a memory round trip off its path)
    if (nx_cluster >= 0) {
      if constexpr (PHASE == PH_A) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      kv_request(nx_cluster, nx_step);
    }
    __syncthreads();
  }
  PSTAMP(visit, 4);
  // ---- 5. K-slices of the output projection: one tagged partial [4][d] per head / slice, write-through, nothing to wait for
#if FW_ROW_PARTS
  // row-major parts inside the pipeline; the decode's LAST phase keeps the [H][16][d] layout the final head launch reads
  const int ldp = last_phase ? D : a.H * D;
  auto part_of = [&](int g) -> float * {
    return last_phase ? a.part_out + ((size_t)cluster * a.H + g) * 16 * d : a.part_out + (size_t)cluster * 16 * a.H * d + (size_t)g * d;
  };
#else
  constexpr int ldp = D;
  auto part_of = [&](int g) -> float * { return a.part_out + ((size_t)cluster * a.H + g) * 16 * d; };
#endif
  if constexpr (NH == 2) {
    gemm2_4s_pair<DH, PHASE, D, FS, AUX>(M, S.ob, LDO, 4 * LDO, part_of(g0), tag_out, ldp);
  } else {
#pragma unroll
    for (int h = 0; h < NH; ++h)
      gemm2_4s<DH, PHASE, D, FS, AUX>(M[h], PHASE == PH_C ? S.t1 : S.ob + h * 4 * LDO, PHASE == PH_C ? LD1 : LDO, part_of(g0 + h), tag_out, ldp);
  }
  PSTAMP(visit, 5);
  if constexpr (STASH) {
    // stash_flush: what was parked in LDS goes to the stash now, behind the hand-off.  Every wave writes out what it parked itself
    // (and will overwrite itself at the next visit), so no barrier is needed.
    if constexpr (PHASE == PH_C && STASH_RED != 0) {
      if (wave_id() < 4) stash_rows(wave_id(), lane_id());      // (the row's own red wave: it is the one that rewrites the row)
    }
    // softmax rows and attention outputs: on waves 4.., which have nothing to do until the next visit's first barrier (and reach
    // that barrier only behind these reads, so the next attention cannot overwrite the tiles under them)
    if constexpr (PHASE != PH_C && STASH_ATT != 0) {
      if (wave_id() >= 4 && wave_id() < 4 + NH) {
        const int h = wave_id() - 4, lane = lane_id(), ar = lane >> 4, l16 = lane & 15, Lk = PHASE == PH_A ? a.ph : a.T;
        const int nkeys = PHASE == PH_A ? a.s + 1 : a.T;
        if (ar < na) {
          if (l16 < nkeys)
            stl[(PHASE == PH_A ? so.p1 : so.p2) + ((((size_t)a0 + ar) * a.H + g0 + h) * a.ph + a.s) * Lk + l16] = S.xs[9 * ldx + h * 64 + lane];
          constexpr int DPL = DH / 16;
          typedef float vrow __attribute__((ext_vector_type(DPL)));
          *reinterpret_cast<vrow *>(stl + (PHASE == PH_A ? so.a1 : so.a2) + (st_row0 + ar) * d + (g0 + h) * DH + l16 * DPL) =
              *reinterpret_cast<const vrow *>(S.ob + (h * 4 + ar) * LDO + l16 * DPL);
        }
      }
    }
  }
  PSTAMP(visit, 6);
  PSTAMPC(visit, 13);
  return true;
}

template <int DH, int PHASE, int NH, int D, int FS, bool DROP = false, bool STASH = false>
__device__ __forceinline__ void pipe_stage(const PipeArgs &f, int layer, int g0, int replica, const PipeLds &S, const PipeSync &ps) {
  static_assert(PHASE != PH_A || NH == 1, "the K/V cache append serves one head");
  ClArgs a = f.a;
  const FusedLayer &W = f.layer[layer];
  a.L = layer;
  a.w1 = W.w1[PHASE];
  a.b1 = W.b1[PHASE];
  a.w2 = W.w2[PHASE];
  a.red_bias = W.red_bias[PHASE];
  a.red_ln = W.red_ln[PHASE];
  a.kv = PHASE == PH_A ? f.selfkv : f.memkv;
  MainRegs4<DH, PHASE> M[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    M[h] = {};
    weights_issue4<DH, PHASE>(M[h], a, g0 + h);      // ONCE: the slice stays in registers for the whole decode
  }
  if constexpr (PHASE != PH_C) {      // attend4<ZEROED>: rows of the K/V staging tiles that never hold a key read as zeros
    constexpr int LDK = DH + 4;
    for (int i = thread_idx(); i < 2 * (2 * 4 * KMAX * LDK); i += CNT) S.Ks[i] = 0.f;       // Ks and Vs are adjacent
    __syncthreads();
  }
  const int NP = 3 * f.nlayers, k = 3 * layer + PHASE;
  const bool loadx = PHASE == PH_A && layer == 0;    // the first stage takes the embedded token as it is (no reduction, no LayerNorm)
  // the K/V rows of the next visit are requested a visit ahead unless that visit is to the same cluster (then the row appended
  // in this visit would be missing)
  const bool ahead = false && PHASE != PH_C && replica + f.R < f.NC;     // (measured: no gain - the hand-off round trip covers the loads)
  int pad = 0;
  bool kv_ready = false;
#pragma clang loop unroll(disable)
  for (int s = 0; s < f.a.ph; ++s) {
    const unsigned p = (unsigned)(s * NP + k);
    a.s = s;
    a.prof_idx = (int)p;
    // slot of phase q: buffer q & 1, row block ((p_last - q) >> 1) & 3 of the 16-row cluster slots (period 8; the last phase of the
    // decode lands in rows 0..3, where the final head launch looks for it)
    const unsigned p_last = (unsigned)(NP * f.a.ph - 1);
    const size_t rb_out = (size_t)4 * (((p_last - p) >> 1) & 3u) * f.a.d, rb_in = (size_t)4 * (((p_last - (p - 1u)) >> 1) & 3u) * f.a.d;
#if FW_ROW_PARTS
    const bool last_phase = p == p_last;
    a.part_out = f.part[p & 1] + (last_phase ? rb_out : rb_out * f.a.H);
    a.part_in = f.part[(p + 1) & 1] + rb_in * f.a.H;
#else
    const bool last_phase = false;
    a.part_out = f.part[p & 1] + rb_out;
    a.part_in = f.part[(p + 1) & 1] + rb_in;
#endif
    a.xres_out = f.xres[p & 1] + rb_out;
    a.xres_in = f.xres[(p + 1) & 1] + rb_in;
    if constexpr (DROP) {      // this step's keep-byte sites, as launch_phases' drop_of gives them to the launch-per-phase kernels
      const mmt_decoder_dropout &Dr = f.drop;
      const size_t Bd = (size_t)f.a.B * f.a.d, Bf = (size_t)f.a.B * FS * f.a.H;
      auto site = [](const uint8_t *base, size_t off) -> const uint8_t * { return base ? base + off : nullptr; };
      a.dk_red = a.dk_pe = a.dk_ff = a.dk_att = nullptr;
      if constexpr (PHASE == PH_A) {
        if (layer > 0) a.dk_red = site(Dr.drop3[layer - 1], (size_t)s * Bd), a.ds_red = Dr.drop3_scale[layer - 1];
        a.dk_att = site(Dr.attn1[layer], (size_t)s * f.a.ph), a.ds_att = Dr.attn1_scale[layer];
      } else if constexpr (PHASE == PH_B) {
        a.dk_red = site(Dr.drop1[layer], (size_t)s * Bd), a.ds_red = Dr.drop1_scale[layer];
        a.dk_att = site(Dr.attn2[layer], (size_t)s * f.a.T), a.ds_att = Dr.attn2_scale[layer];
      } else {
        a.dk_red = site(Dr.drop2[layer], (size_t)s * Bd), a.ds_red = Dr.drop2_scale[layer];
        a.dk_ff = site(Dr.ff[layer], (size_t)s * Bf), a.ds_ff = Dr.ff_scale[layer];
      }
    }
    int li = 0;
#pragma clang loop unroll(disable)
    for (int m = replica; m < f.NC; m += f.R, ++li) {
      if (f.test_stall && s == 1 && layer == 1 && PHASE == PH_B && g0 == 0 && m == replica) return;   // its successors must time out
      int nx_cluster = -1, nx_step = s;
      if (ahead) {
        if (m + f.R < f.NC) nx_cluster = m + f.R;
        else if (s + 1 < f.a.ph) nx_cluster = replica, nx_step = s + 1;
      }
      float *stl = nullptr;
      if constexpr (STASH) stl = f.stash + (size_t)layer * stash_off(a.B, a.T, a.ph, a.d, FS * a.H, a.H, a.n_out, f.nlayers, a.has_map_fusion ? a.map_dim : 0).layer_floats;
      if (!pipe_phase<DH, PHASE, NH, D, FS, DROP, STASH>(a, f.x0buf, m, g0, loadx, S, ps, M, pad, kv_ready, nx_step, nx_cluster, s * 4096 + li,
                                                         f.NC <= f.R, last_phase, stl, f.nlayers)) return;
      kv_ready = nx_cluster >= 0;
    }
  }
}

// The head workgroups: head of the previous step (s > 0) and embedding of the next token, for the clusters
// replica + R * (j + NHW * i), i = 0, 1, ...  The token rows go to x0buf (tag s & 1), a buffer of their own.
// STASH: one more pass (s = ph) that only runs the head of the last step - the stash needs its rows too (the final head launch
// behind the kernel recomputes the same positions)
template <int CPW, bool DROP = false, int CWK = 32, bool STASH = false>
__device__ __forceinline__ void pipe_head(const PipeArgs &f, int replica, int j, const PipeLds &S, const PipeSync &ps) {
  ClArgs a = f.a;
  const int d = a.d, ldx = d + MMT_PAD, K16 = d >> 4, tid = thread_idx();
  a.red_bias = f.head_bias;
  a.red_ln = f.head_ln;
  const int NP = 3 * f.nlayers;
  KsW<4, CPW> wo = {};
  KsW<4, 1> sm0 = {};
  ks_load<4, CPW>(wo, a.output_fc.w, K16, round16(a.n_out) >> 4, [](int s) { return s; });     // resident for the whole decode
  if (!a.tgt) small_load(sm0, a.chain[0]);
  EmbedW emb;
  embed_load(emb, a);
  {   // the biases and LayerNorm vectors the head's epilogues read, copied into LDS once (every one of them was a dependent L2
      // round trip per visit on the decode's critical path)
    int off = 0;
    auto stage = [&](const float *&p, int n) {
      if (p) {
        for (int i = tid; i < n; i += CNT) S.prm[off + i] = p[i];
        p = S.prm + off;
      }
      off += 64;
    };
    stage(a.output_fc.b, a.n_out);
#pragma unroll
    for (int i = 0; i < CHAIN_MAX; ++i)
      if (i < a.n_chain) {
        stage(a.chain[i].b, a.chain[i].n);
        stage(a.chain_ln[i].g, a.chain[i].n);
        stage(a.chain_ln[i].b, a.chain[i].n);
      }
  }
  const bool wave_chain = chain_fits_wave<CWK>(a);      // (a.chain[] biases / LayerNorm vectors already point into LDS)
  float *cw = S.Ks;                                // the head workgroups stage no K/V rows: the chain's weights live there
  static_assert(CHAIN_MAX * 32 * CW_LD<CWK> <= 2 * (2 * 4 * KMAX * (32 + 4)), "the staged chain must fit the K/V staging tiles (Ks + Vs)");
  if (wave_chain) chain_stage<CWK>(a, cw);
  // once: the head's tiles and the token tile hold finite values (their padding columns must stay zero: nothing else writes
  // them) and rows 4..15 of the head's input tile are zero (a visit writes rows 0..3 only)
  for (int i = tid; i < 2 * 16 * LDO + 16 * LDTOK; i += CNT) S.ha[i] = 0.f;
  if (tid >= 256)
    for (int i = tid - 256; i < 12 * (d >> 2); i += 256)
      *reinterpret_cast<f32x4 *>(S.xs + (4 + i / (d >> 2)) * ldx + 4 * (i % (d >> 2))) = (f32x4){0.f, 0.f, 0.f, 0.f};
  mmt_decoder_stash_layout so = {};
  if constexpr (STASH) so = stash_off(a.B, a.T, a.ph, d, 64 * a.H, a.H, a.n_out, f.nlayers, a.has_map_fusion ? a.map_dim : 0);
#pragma clang loop unroll(disable)
  for (int s = 0; s < f.a.ph + (STASH ? 1 : 0); ++s) {
    const bool head_only = STASH && s == f.a.ph;
    const unsigned p = (unsigned)(s * NP);           // the phase this step's first stage runs as; the head consumes phase p - 1
    a.s = s;
    a.prof_idx = (int)p;
    float pe[2] = {0.f, 0.f};
    if (!head_only) embed_pe(pe, a, s);
    const unsigned p_last = (unsigned)(NP * f.a.ph - 1);
    const size_t rb_in = (size_t)4 * (((p_last - (p - 1u)) >> 1) & 3u) * f.a.d;
    a.part_in = f.part[(p + 1) & 1] + rb_in * (FW_ROW_PARTS ? f.a.H : 1);
    a.xres_in = f.xres[(p + 1) & 1] + rb_in;
    if constexpr (DROP) {      // dropout3 of the last layer at the step the head consumes; the positional encoder's at this step
      const size_t Bd = (size_t)a.B * a.d;
      a.dk_red = (f.drop.drop3[f.nlayers - 1] && s > 0) ? f.drop.drop3[f.nlayers - 1] + (size_t)(s - 1) * Bd : nullptr;
      a.ds_red = f.drop.drop3_scale[f.nlayers - 1];
      a.dk_pe = (f.drop.pe && !head_only) ? f.drop.pe + (size_t)s * Bd : nullptr;
      a.ds_pe = f.drop.pe_scale;
    }
    int li = 0;
#pragma clang loop unroll(disable)
    for (int m = replica + j * f.R; m < f.NC; m += f.R * f.NHW, ++li) {
      const int a0 = m * 4, na = min(4, a.B - a0);
      PSTAMP(s * 4096 + li, 0);
      Red4Regs R;
      const bool red_wave = s > 0 && wave_id() < 4;
      unsigned k_red = 0u, kpe[2] = {0u, 0u};
      if constexpr (DROP) {
        if (red_wave) {
          const int ag = a0 + wave_id();
          k_red = __builtin_amdgcn_raw_buffer_load_b32(mk_rsrc(a.dk_red), (a.dk_red && lane_id() < (d >> 2) && ag < a.B)
                                                           ? (unsigned)(ag * d + 4 * lane_id()) : OOB, 0, 0);
        }
        embed_keep(kpe, a, a0, na);
      }
      if (red_wave) {
        const unsigned co = lane_id() < (d >> 2) ? (unsigned)lane_id() * 16u : OOB;
        R.bias = bl4(mk_rsrc(a.red_bias), co);
        R.gam = bl4(mk_rsrc(a.red_ln.g), co);
        R.bet = bl4(mk_rsrc(a.red_ln.b), co);
      }
      if (red_wave) (void)poll_rows<false>(R, a, f.x0buf, m, tag_of_phase(p - 1u), ps);
      PSTAMP(s * 4096 + li, 1);
      __syncthreads();
      if (*ps.abort) return;
      const size_t st_row0 = STASH && s > 0 ? (size_t)(s - 1) * a.B + a0 : 0;      // the head works on step s - 1
      if (red_wave) {
        float *st_x = nullptr, *st_xhat = nullptr, *st_rstd = nullptr;
        if constexpr (STASH) {
          if (wave_id() < na) {
            float *last = f.stash + so.tail - so.layer_floats;
            st_x = f.stash + so.tail + so.t_x + (st_row0 + wave_id()) * d;
            st_xhat = last + so.xhat3 + (st_row0 + wave_id()) * d;
            st_rstd = last + so.rstd3 + st_row0 + wave_id();
          }
        }
        reduce4_finish<AUX_SC1, DROP, false, STASH>(R, a, m, S.xs, ldx, false, false, DROP ? &k_red : nullptr, st_x, st_xhat, st_rstd);
      }
      if (s > 0) {
        float *posl = li < PIPE_MAXPOS ? S.pos + li * 64 : nullptr;
        head_step<CPW>(a, 0, a0, na, S.xs, ldx, S.red, S.ha, S.hb, S.tok, wo, sm0, posl, wave_chain);
        if (wave_chain && wave_id() == 0) {
          if constexpr (STASH) {
            float *tl = f.stash + so.tail;
            head_chain_wave<CWK, true>(a, a0, na, S.ha, S.hb, S.tok, posl, cw, tl + so.t_ho + st_row0 * so.ho_ld, tl + so.t_hu + st_row0 * 32,
                                       tl + so.t_hxhat + st_row0 * 32, tl + so.t_hrstd + st_row0,
                                       a.has_map_fusion ? tl + so.t_u + st_row0 * a.n_out : nullptr);
          } else {
            head_chain_wave<CWK>(a, a0, na, S.ha, S.hb, S.tok, posl, cw);
          }
        }
      }
      if (head_only) continue;        // (uniform: the whole workgroup skips the embedding and its two barriers)
      if (a.tgt || s == 0)
        for (int i = tid; i < na * a.n_in; i += CNT) {
          const int r = i / a.n_in, c = i - r * a.n_in;
          S.tok[r * LDTOK + c] = a.tgt ? a.tgt[((size_t)(a0 + r) * a.ph + s) * a.n_in + c] : a.start_pos[(size_t)(a0 + r) * a.n_in + c];
        }
      __syncthreads();
      PSTAMP(s * 4096 + li, 4);
      embed_token_res<DROP>(a, emb, pe, S.tok, S.xs, ldx, kpe);         // rows tagged s & 1
      __syncthreads();
      PSTAMP(s * 4096 + li, 5);
      if (wave_id() < 4 && lane_id() < (d >> 2))       // the embedded token, write-through, 16 bytes per lane
        bs4x<AUX_SC1>(mk_rsrc(f.x0buf + ((size_t)m * 4 + wave_id()) * d), (unsigned)lane_id() * 16u,
                      *reinterpret_cast<const f32x4 *>(S.xs + wave_id() * ldx + 4 * lane_id()));
      // (no barrier here: xs and tok are next written behind the barrier that follows the next visit's poll)
      PSTAMP(s * 4096 + li, 6);
    }
  }
}

template <int DH, int CPW, int D, int FS, bool DROP = false, int CWK = 32, bool STASH = false>
__global__ __launch_bounds__(CNT) void mmt_dec_pipe4_kernel(PipeArgs f) {
  extern __shared__ __align__(16) float smem[];
  const PipeLds S = pipe_lds<DH>(smem, f.a.d);
  const int H = f.a.H, per_layer = 2 * H + H / 2, per_rep = f.nlayers * per_layer + f.NHW;
  const int replica = blockIdx.x / per_rep, rb = blockIdx.x - replica * per_rep;
  PipeSync ps;
  ps.tmo = (gu32 *)f.sync;
  ps.abort = S.abort;
  if (thread_idx() == 0) ps.abort[0] = 0, ps.abort[1] = 0;      // abort flag, wake word of the polling waves
  __syncthreads();
  if (rb < f.nlayers * per_layer) {
    const int layer = rb / per_layer, q = rb - layer * per_layer;
    if (q < H) pipe_stage<DH, PH_A, 1, D, FS, DROP, STASH>(f, layer, q, replica, S, ps);
    else if (q < H + H / 2) pipe_stage<DH, PH_B, 2, D, FS, DROP, STASH>(f, layer, 2 * (q - H), replica, S, ps);
    else pipe_stage<DH, PH_C, 1, D, FS, DROP, STASH>(f, layer, q - H - H / 2, replica, S, ps);
  } else {
    pipe_head<CPW, DROP, CWK, STASH>(f, replica, rb - f.nlayers * per_layer, S, ps);
  }
}

// ------------------------------------------------------------------------------- prologue kernels
struct CkvArgs {
  int rows, d, L, B, T, ldx, ldb;
  const float *src;          // (rows, d): memory or fused memory
  float *memkv;              // [L][rows][2d]
  unsigned *sync;            // fused kernel's time-out word + arrival counters: zeroed here, one launch ahead of their use
  int nsync;
  unsigned *fill[2];         // pipeline kernel: the hand-off buffers, filled with 0xFFFFFFFF (= "nothing here yet") before every decode
  int nfill[2];
  mmt_linear ckv[MMT_MAX_LAYERS];
};
// cross-attention K/V of every layer (computed once per decode; the reference redoes it every step)
__global__ __launch_bounds__(CNT) void mmt_dec_ckv_kernel(CkvArgs a) {
  extern __shared__ __align__(16) float smem[];
  float *xs = smem, *buf = xs + 16 * a.ldx;
  const int tid = thread_idx(), d = a.d, L = blockIdx.y;
  const int r0 = blockIdx.x * 16, nr = min(16, a.rows - r0);
  if (blockIdx.x == 0 && L == 0)
    for (int i = tid; i < a.nsync; i += CNT) a.sync[i] = 0u;
  {
    const int nb = gridDim.x * gridDim.y, bi = blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int n4 = a.nfill[k] >> 2;                         // whole float4s (the buffers are multiples of 4 floats)
      u32x4 *dst = reinterpret_cast<u32x4 *>(a.fill[k]);
      for (int i = bi * CNT + tid; i < n4; i += nb * CNT) dst[i] = (u32x4){0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    }
  }
  for (int i = tid; i < 16 * (a.ldx + a.ldb); i += CNT) smem[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < nr * (d >> 2); i += CNT) {
    const int r = i / (d >> 2), c4 = i - r * (d >> 2);
    *reinterpret_cast<f32x4 *>(xs + r * a.ldx + 4 * c4) = reinterpret_cast<const f32x4 *>(a.src + (size_t)(r0 + r) * d)[c4];
  }
  __syncthreads();
  linear_to_lds<1, EPI_STORE>(xs, a.ldx, a.ckv[L], buf, a.ldb);
  __syncthreads();
  float *dst = a.memkv + ((size_t)L * a.rows + r0) * 2 * d;
  for (int i = tid; i < nr * (d >> 1); i += CNT) {
    const int r = i / (d >> 1), c4 = i - r * (d >> 1);
    reinterpret_cast<f32x4 *>(dst + (size_t)r * 2 * d)[c4] = *reinterpret_cast<const f32x4 *>(buf + r * a.ldb + 4 * c4);
  }
}

struct FuseArgs {
  int rows, d, T, mf, ld;
  const float *memory, *mem_feat;
  float *fused;
  mmt_mlp M;
};
// fusion/lane_hist (or mix_fc) on cat[memory, lane feature repeated over T]   (mgcvae.py:404-409)
__global__ __launch_bounds__(CNT) void mmt_dec_fuse_kernel(FuseArgs a) {
  extern __shared__ __align__(16) float smem[];
  float *ra = smem, *rb = ra + 16 * a.ld, *out = rb + 16 * a.ld;
  const int tid = thread_idx(), d = a.d, mf = a.mf;
  const int r0 = blockIdx.x * 16, nr = min(16, a.rows - r0);
  for (int i = tid; i < 16 * 3 * a.ld; i += CNT) smem[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < nr * (d + mf); i += CNT) {
    const int r = i / (d + mf), c = i - r * (d + mf);
    ra[r * a.ld + c] = c < d ? a.memory[(size_t)(r0 + r) * d + c] : a.mem_feat[(size_t)((r0 + r) / a.T) * mf + (c - d)];
  }
  __syncthreads();
  mlp_stack_tile(a.M, ra, rb, a.ld, out, a.ld, nr);
  for (int i = tid; i < nr * d; i += CNT) {
    const int r = i / d, c = i - r * d;
    a.fused[(size_t)(r0 + r) * d + c] = out[r * a.ld + c];
  }
}

int mlp_width(const mmt_mlp &m) {
  int w = std::max(m.out.k, m.out.n);
  for (int i = 0; i < m.n_hidden; ++i) w = std::max(w, std::max(m.hidden[i].k, m.hidden[i].n));
  return w;
}

// Environment knobs, read ONCE (first use): MMT_CLUSTER_ROWS=4|8|16, MMT_NO_MFMA4, MMT_NO_FUSED_DECODER.
struct DecEnv {
  int cluster_rows;
  bool no_mfma4, no_fused, no_pipe;
  DecEnv() : cluster_rows(0), no_mfma4(getenv("MMT_NO_MFMA4") != nullptr), no_fused(getenv("MMT_NO_FUSED_DECODER") != nullptr),
             no_pipe(getenv("MMT_NO_PIPELINE_DECODER") != nullptr) {
    if (const char *e = getenv("MMT_CLUSTER_ROWS")) {
      const int v = atoi(e);
      if (v == 4 || v == 8 || v == 16) cluster_rows = v;
    }
  }
};
const DecEnv &dec_env() {
  static const DecEnv e;       // thread-safe initialisation (C++11)
  return e;
}

// Per-device host state (the library may be driven from several devices / threads of one process).
constexpr int MAX_DEV = 64;
struct DevState {
  std::atomic<int> ncu{0};
  std::atomic<bool> fused_disabled{false};     // set by mmt_decoder_status() after a timed-out one-launch decode
  std::atomic<unsigned> attr_mask{0};          // kernel families whose LDS attribute has been raised on this device
};
DevState g_dev[MAX_DEV];
std::atomic<int> g_test_stall{0};              // fault injection (tests): mmt_debug_decoder_stall()

int cur_dev() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev < 0 || dev >= MAX_DEV ? 0 : dev;
}
int dev_ncu(int dev) {
  int n = g_dev[dev].ncu.load(std::memory_order_relaxed);
  if (n <= 0) {
    n = 256;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    g_dev[dev].ncu.store(n, std::memory_order_relaxed);
  }
  return n;
}

// Agents per cluster: 16 fills the MFMA tile; small batches use fewer so that more CUs share the work (every
// workgroup takes in its weight slice at ~50 GB/s, and the partials it sums scale with the rows).
bool mfma4_fits(const mmt_decoder *dec);
enum { ONE_LAUNCH_NONE = 0, ONE_LAUNCH_FUSED = 1, ONE_LAUNCH_PIPE = 2 };
// A linear of the head's chain (fusion/hist_map, decoder/MLP) with 33..64 inputs and none wider: the pipeline instantiation whose
// one-wave head chain stages 64 inputs (CWK = 64) - the map configurations.  Otherwise the CWK = 32 instantiation (which runs
// wider chains on the matrix cores, as the other forms do).
static bool pipe_wide_chain(const mmt_decoder *dec, bool teacher_forced) {
  if (teacher_forced) return false;
  int kmax = 0, nmax = 0;
  auto scan = [&](const mmt_mlp &M) {
    for (int i = 0; i < M.n_hidden; ++i) kmax = std::max(kmax, (int)M.hidden[i].k), nmax = std::max(nmax, (int)M.hidden[i].n);
    kmax = std::max(kmax, (int)M.out.k), nmax = std::max(nmax, (int)M.out.n);
  };
  if (dec->has_map_fusion) scan(dec->map_fusion);
  scan(dec->head);
  return kmax > 32 && kmax <= 64 && nmax <= 32;
}
struct PipeGeom {
  int R, NHW, grid;     // replicas of the pipeline, head workgroups per replica, workgroups in all
  float *x0buf;
};
// Replicas / head workgroups that fit on `ncu` CUs (one workgroup per CU); R = 0: the pipeline does not fit.
PipeGeom pipe_geom(int L, int H, int NC, int ncu) {
  PipeGeom g = {0, 0, 0, nullptr};
  if (H < 2 || (H & 1)) return g;
  const int per_stage = L * (2 * H + H / 2);
  g.R = std::min(NC, ncu / (per_stage + 1));
  if (g.R < 1) return g;
  const int per_rep_clusters = (NC + g.R - 1) / g.R;
  g.NHW = std::max(1, std::min(std::min((ncu - g.R * per_stage) / g.R, 4), per_rep_clusters));
  g.grid = g.R * (per_stage + g.NHW);
  return g;
}

// static conditions of the pipeline form (the run-time ones - occupancy, a device that has timed out before - are checked
// per call; where they fail the 4-agent launch-per-phase kernels run)
bool pipe_applies(const mmt_decoder *dec, int NC, int ncu) {
  if (dec_env().no_fused || dec_env().no_pipe || !mfma4_fits(dec)) return false;
  if (dec->d_model != 256 || dec->d_ff != 64 * dec->nhead) return false;
  // tag generations ("tag bits" above): a hand-off slot is rewritten every 8 phases and a stage may start polling a cluster's
  // rows 3L phases before they are written, so a slot shows at most 3L/8 + 1 generations in that window - which the
  // TAG_MASK + 1 tag values must tell apart.  Holds for every L <= MMT_MAX_LAYERS = 8; checked, not assumed.
  static_assert(3 * MMT_MAX_LAYERS / 8 + 1 <= (int)TAG_MASK + 1, "tag bits too few for MMT_MAX_LAYERS");
  if (dec->nlayers < 1 || 3 * dec->nlayers / 8 + 1 > (int)TAG_MASK + 1) return false;
  return pipe_geom(dec->nlayers, dec->nhead, NC, ncu).R >= 1;
}

// The pipeline kernel's grid does not grow with the batch, so beyond the batch whose 4-agent clusters fill the chip one per
// workgroup it runs in throughput mode (about 5 us per cluster visit and stage): B = 160 / 192 / 256 decode in 1.26 / 1.50 / 1.95 ms
// against 2.1 ms on the 16-row launch-per-phase kernels, B = 384 / 512 in 2.87 / 3.80 against 2.4 / 2.5.
constexpr int PIPE_MAX_B = 256;
int cluster_rows(const mmt_decoder *dec, int B) {
  if (dec_env().cluster_rows) return dec_env().cluster_rows;
  const int ncu = dev_ncu(cur_dev()), H = dec->nhead;
  int rc = 16;
  while (rc > 4 && ((B + rc / 2 - 1) / (rc / 2)) * H <= ncu) rc /= 2;
  if (rc > 4 && B <= PIPE_MAX_B && pipe_applies(dec, (B + 3) / 4, ncu)) rc = 4;
  return rc;
}

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
bool lin4_first_ok(int slice_rows, int k) { return slice_rows % 4 == 0 && k % 4 == 0 && pow2(slice_rows / 4) && slice_rows >= 32 && slice_rows <= 512 && pow2(k / 4); }
bool lin4_second_ok(int n, int slice_k) { return n % 4 == 0 && slice_k % 4 == 0 && pow2(n / 4) && n >= 32 && n <= 512 && pow2(slice_k / 4); }

// 4-agent clusters on the 4x4x1 path: packed copies present and every linear within the kernels' register budget.
bool mfma4_fits(const mmt_decoder *dec) {
  if (dec_env().no_mfma4) return false;
  const int d = dec->d_model, H = dec->nhead, DH = d / H, FS = dec->d_ff / H;
  for (int l = 0; l < dec->nlayers; ++l) {
    const mmt_dec_layer &W = dec->layer[l];
    if (!W.qkv4 || !W.out4 || !W.cq4 || !W.cout4 || !W.ff14 || !W.ff24) return false;
  }
  if (DH != 32 || !lin4_first_ok(DH, d) || !lin4_first_ok(FS, d) || !lin4_second_ok(d, DH) || !lin4_second_ok(d, FS)) return false;
  if (lin4(DH / 4, d / 4).q > 4 || lin4(FS / 4, d / 4).q > 8) return false;       // Geom4::QMAX1
  if (lin4(d / 4, DH / 4).q > 4 || lin4(d / 4, FS / 4).q > 8) return false;       // Geom4::QMAX2
  return true;
}

// tr: the matrix is the TRANSPOSE of the array at w (element (row, col) = w[col * ldw + row]): the backward pipeline multiplies
// by W where the forward multiplies by W^T (decoder_bwd_pipe.cuh)
__device__ __forceinline__ float lin4_first_elem(const float *__restrict__ w, int ldw, int nparts, int part_rows, int slice_rows, int k,
                                                 size_t i, bool tr = false) {
  const Lin4 L = lin4(slice_rows / 4, k / 4);
  const int e = i & 3, lane = (i >> 2) & 63, wv = (i >> 8) & 7;
  size_t rest = i >> 11;
  const int t = rest % L.q;
  rest /= L.q;
  const int p = rest % nparts, g = rest / nparts;
  const int b = lane >> 2, j = lane & 3, cgl = b & (L.cgw - 1), kq = (b >> L.lg_cgw) * L.q + t;
  const int row = p * part_rows + g * slice_rows + (wv * L.cgw + cgl) * 4 + j;
  if (kq >= k / 4) return 0.f;
  return tr ? w[(size_t)(4 * kq + e) * ldw + row] : w[(size_t)row * ldw + 4 * kq + e];
}
__global__ void mmt_pack_lin4_first_kernel(const float *__restrict__ w, int ldw, int nparts, int part_rows, int slice_rows, int k,
                                           size_t total, float *__restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + thread_idx();
  if (i < total) dst[i] = lin4_first_elem(w, ldw, nparts, part_rows, slice_rows, k, i);
}
__device__ __forceinline__ float lin4_second_elem(const float *__restrict__ w, int ldw, int n, int slice_k, size_t i, bool tr = false) {
  const Lin4 L = lin4(n / 4, slice_k / 4);
  const int e = i & 3, lane = (i >> 2) & 63, wv = (i >> 8) & 7;
  const size_t rest = i >> 11;
  const int t = rest % L.q, g = rest / L.q;
  const int b = lane >> 2, j = lane & 3, cgl = b & (L.cgw - 1), kq = (b >> L.lg_cgw) * L.q + t;
  const int row = (wv * L.cgw + cgl) * 4 + j;
  if (kq >= slice_k / 4) return 0.f;
  const size_t col = (size_t)g * slice_k + 4 * kq + e;
  return tr ? w[col * ldw + row] : w[(size_t)row * ldw + col];
}
__global__ void mmt_pack_lin4_second_kernel(const float *__restrict__ w, int ldw, int n, int slice_k, size_t total,
                                            float *__restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + thread_idx();
  if (i < total) dst[i] = lin4_second_elem(w, ldw, n, slice_k, i);
}
// every re-layout of a module in one launch: blockIdx.y = job
__global__ __launch_bounds__(256) void mmt_pack_jobs_kernel(const mmt_pack_job *__restrict__ table) {
  const mmt_pack_job J = table[blockIdx.y];
  const size_t total = (size_t)J.total;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + thread_idx(); i < total; i += (size_t)gridDim.x * blockDim.x) {
    float v;
    if (J.kind == 0) {
      v = J.src[i];
    } else if (J.kind == 1) {        // mmt_pack_linear: [slab][chunk][lane][4] MFMA streaming order, zero padded to 16
      const int K16 = (J.k + 15) >> 4, e = i & 3, lane = (i >> 2) & 63;
      const size_t tile = i >> 8;
      const int chunk = tile % K16, slab = tile / K16;
      const int row = slab * 16 + (lane & 15), col = chunk * 16 + (lane >> 4) * 4 + e;
      v = (row < J.n && col < J.k) ? J.src[(size_t)row * J.ldw + J.col0 + col] : 0.f;
    } else if (J.kind == 2) {
      v = lin4_first_elem(J.src, J.ldw, J.nparts, J.part_rows, J.slice_rows, J.k, i, J.tr != 0);
    } else {
      v = lin4_second_elem(J.src, J.ldw, J.n, J.slice_rows, i, J.tr != 0);
    }
    J.dst[i] = v;
  }
}

#include "decoder_bwd_pipe.cuh"

}  // namespace

extern "C" size_t mmt_lin4_first_floats(int nparts, int slice_rows, int k, int nslices) {
  if (nparts < 1 || nslices < 1 || !lin4_first_ok(slice_rows, k)) return 0;
  return (size_t)nslices * nparts * lin4(slice_rows / 4, k / 4).q * 2048;
}
extern "C" int mmt_pack_lin4_first(const float *w, int ldw, int nparts, int part_rows, int slice_rows, int k, int nslices, float *dst,
                                   void *stream) {
  const size_t total = mmt_lin4_first_floats(nparts, slice_rows, k, nslices);
  if (!w || !dst || !total || ldw < k || (nparts > 1 && part_rows < nslices * slice_rows))
    return mmt_set_error(MMT_EINVAL, "mmt_pack_lin4_first: bad arguments or unsupported geometry");
  hipLaunchKernelGGL(mmt_pack_lin4_first_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, ldw, nparts,
                     part_rows, slice_rows, k, total, dst);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MMT_OK : mmt_set_error(MMT_ELAUNCH, "mmt_pack_lin4_first: %s", hipGetErrorString(e));
}
extern "C" size_t mmt_lin4_second_floats(int n, int slice_k, int nslices) {
  if (nslices < 1 || !lin4_second_ok(n, slice_k)) return 0;
  return (size_t)nslices * lin4(n / 4, slice_k / 4).q * 2048;
}
extern "C" int mmt_pack_lin4_second(const float *w, int ldw, int n, int slice_k, int nslices, float *dst, void *stream) {
  const size_t total = mmt_lin4_second_floats(n, slice_k, nslices);
  if (!w || !dst || !total || ldw < nslices * slice_k)
    return mmt_set_error(MMT_EINVAL, "mmt_pack_lin4_second: bad arguments or unsupported geometry");
  hipLaunchKernelGGL(mmt_pack_lin4_second_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, ldw, n, slice_k,
                     total, dst);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MMT_OK : mmt_set_error(MMT_ELAUNCH, "mmt_pack_lin4_second: %s", hipGetErrorString(e));
}

extern "C" int mmt_pack_jobs(const mmt_pack_job *table_dev, int njobs, void *stream) {
  if (!table_dev || njobs <= 0 || njobs > 65535) return mmt_set_error(MMT_EINVAL, "mmt_pack_jobs: bad arguments");
  hipLaunchKernelGGL(mmt_pack_jobs_kernel, dim3(48, njobs), dim3(256), 0, (hipStream_t)stream, table_dev);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MMT_OK : mmt_set_error(MMT_ELAUNCH, "mmt_pack_jobs: %s", hipGetErrorString(e));
}

// Geometry the cluster kernels are written for; anything else runs on the persistent kernel.
bool mmt_cluster_decoder_supports(const mmt_decoder *dec, int map_dim, bool autoregressive, int T, int ph) {
  const int d = dec->d_model, H = dec->nhead;
  if (d % 16 || d > 256 || H < 1 || H > 8 || d % H) return false;
  const int dh = d / H;
  if (dh != 16 && dh != 32) return false;
  if (T > KMAX || ph > KMAX) return false;
  if (dec->d_ff % (16 * H) || dec->d_ff / H > 64) return false;
  if (dec->n_out < 1 || dec->n_out > 64 || dec->n_in < 1 || dec->n_in > 16) return false;
  if (autoregressive) {
    if (dec->head.n_hidden + 1 + (dec->has_map_fusion ? dec->map_fusion.n_hidden + 1 : 0) > CHAIN_MAX) return false;
    if (mlp_width(dec->head) > 64) return false;
    if (dec->has_map_fusion && (mlp_width(dec->map_fusion) > 64 || dec->n_out + map_dim > 64)) return false;
  }
  return true;
}

size_t mmt_cluster_decoder_extra_floats(const mmt_decoder *dec, int B, int T) {
  const int RC = cluster_rows(dec, B);
  const size_t NC = (B + RC - 1) / RC, d = dec->d_model;
  // + the fused kernel's time-out word and per-cluster arrival counters, one 128-byte line each (32 floats of slack to align them)
  // ... and the pipeline kernel's token buffer [NC][4][d]
  return NC * 16 * d * (2 * (size_t)dec->nhead + 2) + (dec->has_mem_fusion ? (size_t)B * T * d : 0) + 64 + (NC + 2) * 32 + NC * 4 * d + 32;
}

#define CL_CHECK(what)                                                                     \
  do {                                                                                     \
    hipError_t e_ = hipGetLastError();                                                     \
    if (e_ != hipSuccess) return mmt_set_error(MMT_ELAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
  } while (0)

template <int DH, int CPW, bool M4>
static int launch_phases(const mmt_decoder *dec, ClArgs a, int B, int ph, float *part[2], float *xres[2], float *memkv, float *selfkv,
                         unsigned *sync, int one_launch, const PipeGeom &pg, hipStream_t st, const mmt_decoder_dropout *drop = nullptr,
                         float *stash = nullptr) {
  const bool fused = one_launch != ONE_LAUNCH_NONE;
  const int d = dec->d_model, H = dec->nhead, NC = (B + a.RC - 1) / a.RC, L = dec->nlayers;
  constexpr int LD1 = (3 * DH > 64 ? 3 * DH : 64) + 8;
  const size_t lds_c = ((size_t)16 * (d + MMT_PAD) + CNW * KS_SMAX * 256 + 16 * LD1 + 3 * 16 * LDO + 16 * LDTOK + 256) * sizeof(float);
  const size_t lds_a = lds_c + (size_t)16 * KMAX * (DH + 4) * sizeof(float);
  // hipFuncSetAttribute is per device: raise the dynamic-LDS limit once per (device, kernel family)
  const unsigned fam_bit = 1u << ((DH == 32 ? 0 : 1) + 2 * (CPW - 1) + 4 * (M4 ? 1 : 0));
  DevState &ds = g_dev[cur_dev()];
  if (!(ds.attr_mask.load(std::memory_order_acquire) & fam_bit)) {
#define SET_LDS(PHASE, XM, BYTES)                                                                                                     \
  do {                                                                                                                                \
    (void)hipFuncSetAttribute((const void *)mmt_dec_phase_kernel<DH, CPW, PHASE, XM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES)); \
    if constexpr (M4 && PHASE != PH_H) {                                                                                              \
      (void)hipFuncSetAttribute((const void *)mmt_dec_phase4_kernel<DH, CPW, PHASE == PH_H ? PH_C : PHASE, XM>,                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES));                                            \
      (void)hipFuncSetAttribute((const void *)mmt_dec_phase4_kernel<DH, CPW, PHASE == PH_H ? PH_C : PHASE, XM, true>,                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES));                                            \
    }                                                                                                                                 \
    if constexpr (M4 && PHASE == PH_H)                                                                                                \
      (void)hipFuncSetAttribute((const void *)mmt_dec_phase_kernel<DH, CPW, PH_H, XM_REDUCE, true>,                                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES));                                            \
  } while (0)
    SET_LDS(PH_A, XM_EMBED, lds_a);
    SET_LDS(PH_A, XM_REDUCE, lds_a);
    SET_LDS(PH_A, XM_HEAD_EMBED, lds_a);
    SET_LDS(PH_B, XM_REDUCE, lds_a);
    SET_LDS(PH_C, XM_REDUCE, lds_c);
    SET_LDS(PH_H, XM_REDUCE, lds_c);
#undef SET_LDS
    if constexpr (M4) {
      (void)hipFuncSetAttribute((const void *)mmt_dec_fused4_kernel<DH, CPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a);
      if constexpr (DH == 32 && CPW == 2)
      {
        (void)hipFuncSetAttribute((const void *)mmt_dec_pipe4_kernel<32, 2, 256, 64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(pipe_lds_floats<32>(256) * sizeof(float)));
        (void)hipFuncSetAttribute((const void *)mmt_dec_pipe4_kernel<32, 2, 256, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(pipe_lds_floats<32>(256) * sizeof(float)));
        (void)hipFuncSetAttribute((const void *)mmt_dec_pipe4_kernel<32, 2, 256, 64, false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(pipe_lds_floats<32>(256) * sizeof(float)));
        (void)hipFuncSetAttribute((const void *)mmt_dec_pipe4_kernel<32, 2, 256, 64, true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(pipe_lds_floats<32>(256) * sizeof(float)));
        (void)hipFuncSetAttribute((const void *)mmt_dec_pipe4_kernel<32, 2, 256, 64, false, 32, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(pipe_lds_floats<32>(256) * sizeof(float)));
        (void)hipFuncSetAttribute((const void *)mmt_dec_pipe4_kernel<32, 2, 256, 64, true, 32, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(pipe_lds_floats<32>(256) * sizeof(float)));
        (void)hipFuncSetAttribute((const void *)mmt_dec_pipe4_kernel<32, 2, 256, 64, false, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(pipe_lds_floats<32>(256) * sizeof(float)));
        (void)hipFuncSetAttribute((const void *)mmt_dec_pipe4_kernel<32, 2, 256, 64, true, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(pipe_lds_floats<32>(256) * sizeof(float)));
      }
    }
    ds.attr_mask.fetch_or(fam_bit, std::memory_order_release);
  }
#define LAUNCH_PHASE(PHASE, XM, BYTES)                                                                     \
  do {                                                                                                     \
    if constexpr (M4) {                                                                                    \
      if (drop) hipLaunchKernelGGL((mmt_dec_phase4_kernel<DH, CPW, PHASE, XM, true>), grid, blk, BYTES, st, a); \
      else hipLaunchKernelGGL((mmt_dec_phase4_kernel<DH, CPW, PHASE, XM>), grid, blk, BYTES, st, a);       \
    } else hipLaunchKernelGGL((mmt_dec_phase_kernel<DH, CPW, PHASE, XM>), grid, blk, BYTES, st, a);        \
  } while (0)
  // training-mode dropout: the keep bytes of one launch (sites laid out like the training step's stash: (ph, B, .) step-major,
  // attention sites (B, H, ph, Lk)); NULL sites stay NULL
  const size_t Bd = (size_t)B * d, Bf = (size_t)B * dec->d_ff;
  auto site = [](const uint8_t *base, size_t off) -> const uint8_t * { return base ? base + off : nullptr; };
  auto drop_of = [&](ClArgs &x, int phase, int l, int s) {
    if (!drop) return;
    x.dk_red = x.dk_pe = x.dk_ff = x.dk_att = nullptr;
    if (phase == PH_A) {
      if (l > 0) x.dk_red = site(drop->drop3[l - 1], (size_t)s * Bd), x.ds_red = drop->drop3_scale[l - 1];
      else if (s > 0) x.dk_red = site(drop->drop3[L - 1], (size_t)(s - 1) * Bd), x.ds_red = drop->drop3_scale[L - 1];
      if (l == 0) x.dk_pe = site(drop->pe, (size_t)s * Bd), x.ds_pe = drop->pe_scale;
      x.dk_att = site(drop->attn1[l], (size_t)s * ph), x.ds_att = drop->attn1_scale[l];
    } else if (phase == PH_B) {
      x.dk_red = site(drop->drop1[l], (size_t)s * Bd), x.ds_red = drop->drop1_scale[l];
      x.dk_att = site(drop->attn2[l], (size_t)s * x.T), x.ds_att = drop->attn2_scale[l];
    } else if (phase == PH_C) {
      x.dk_red = site(drop->drop2[l], (size_t)s * Bd), x.ds_red = drop->drop2_scale[l];
      x.dk_ff = site(drop->ff[l], (size_t)s * Bf), x.ds_ff = drop->ff_scale[l];
    } else {
      x.dk_red = site(drop->drop3[L - 1], (size_t)(ph - 1) * Bd), x.ds_red = drop->drop3_scale[L - 1];
    }
  };
  int p = 0;   // phase counter: phase p writes part[p&1] / xres[p&1] and reads the other pair
  auto bufs = [&](ClArgs &x) {
    x.part_out = part[p & 1];
    x.xres_out = xres[p & 1];
    x.part_in = part[(p + 1) & 1];
    x.xres_in = xres[(p + 1) & 1];
    x.prof_idx = p;
    ++p;
  };
  const mmt_dec_layer &last = dec->layer[L - 1];
  const dim3 grid(NC * H), blk(CNT);
  if constexpr (M4) {
    if (fused) {
      FusedLayer FL[MMT_MAX_LAYERS];
      memset(FL, 0, sizeof(FL));
      for (int l = 0; l < L; ++l) {
        const mmt_dec_layer &W = dec->layer[l];
        const mmt_dec_layer &prev = l == 0 ? last : dec->layer[l - 1];
        FusedLayer &F = FL[l];
        F.w1[PH_A] = W.qkv4, F.b1[PH_A] = W.qkv.b, F.w2[PH_A] = W.out4, F.red_bias[PH_A] = prev.ff2.b, F.red_ln[PH_A] = prev.ln3;
        F.w1[PH_B] = W.cq4, F.b1[PH_B] = W.cq.b, F.w2[PH_B] = W.cout4, F.red_bias[PH_B] = W.out.b, F.red_ln[PH_B] = W.ln1;
        F.w1[PH_C] = W.ff14, F.b1[PH_C] = W.ff1.b, F.w2[PH_C] = W.ff24, F.red_bias[PH_C] = W.cout.b, F.red_ln[PH_C] = W.ln2;
      }
      if (one_launch == ONE_LAUNCH_PIPE) {
        PipeArgs f;
        memset(&f, 0, sizeof(f));
        f.a = a;
        memcpy(f.layer, FL, sizeof(FL));
        for (int i = 0; i < 2; ++i) f.part[i] = part[i], f.xres[i] = xres[i];
        f.selfkv = selfkv;
        f.memkv = memkv;
        f.sync = sync;
        f.x0buf = pg.x0buf;
        f.head_bias = last.ff2.b;
        f.head_ln = last.ln3;
        f.nlayers = L;
        f.test_stall = g_test_stall.load(std::memory_order_relaxed) & 1;
        f.R = pg.R;
        f.NHW = pg.NHW;
        f.NC = NC;
        if (drop) f.drop = *drop;
        f.stash = stash;
        if constexpr (DH == 32 && CPW == 2) {    // the pipeline is built for the shipped geometry (pipe_geom_ok): d = 256, d_ff / H = 64
          const size_t plds = pipe_lds_floats<32>(256) * sizeof(float);
          if (stash && pipe_wide_chain(dec, false)) {      // (mmt_cluster_decoder_stash_ok: fusion/hist_map's 64-input Linear)
            if (drop) hipLaunchKernelGGL((mmt_dec_pipe4_kernel<32, 2, 256, 64, true, 64, true>), dim3(pg.grid), blk, plds, st, f);
            else hipLaunchKernelGGL((mmt_dec_pipe4_kernel<32, 2, 256, 64, false, 64, true>), dim3(pg.grid), blk, plds, st, f);
          } else if (stash) {
            if (drop) hipLaunchKernelGGL((mmt_dec_pipe4_kernel<32, 2, 256, 64, true, 32, true>), dim3(pg.grid), blk, plds, st, f);
            else hipLaunchKernelGGL((mmt_dec_pipe4_kernel<32, 2, 256, 64, false, 32, true>), dim3(pg.grid), blk, plds, st, f);
          } else if (pipe_wide_chain(dec, a.tgt != nullptr)) {      // a head linear with 33..64 inputs (fusion/hist_map): the instantiation that stages them
            if (drop) hipLaunchKernelGGL((mmt_dec_pipe4_kernel<32, 2, 256, 64, true, 64>), dim3(pg.grid), blk, plds, st, f);
            else hipLaunchKernelGGL((mmt_dec_pipe4_kernel<32, 2, 256, 64, false, 64>), dim3(pg.grid), blk, plds, st, f);
          } else if (drop) hipLaunchKernelGGL((mmt_dec_pipe4_kernel<32, 2, 256, 64, true>), dim3(pg.grid), blk, plds, st, f);
          else hipLaunchKernelGGL((mmt_dec_pipe4_kernel<32, 2, 256, 64>), dim3(pg.grid), blk, plds, st, f);
        }
        p = 3 * L * ph;
      } else {
        FusedArgs f;
        memset(&f, 0, sizeof(f));
        f.a = a;
        memcpy(f.layer, FL, sizeof(FL));
        for (int i = 0; i < 2; ++i) f.part[i] = part[i], f.xres[i] = xres[i];
        f.selfkv = selfkv;
        f.memkv = memkv;
        f.sync = sync;
        f.nlayers = L;
        f.test_stall = g_test_stall.load(std::memory_order_relaxed) & 1;
        hipLaunchKernelGGL((mmt_dec_fused4_kernel<DH, CPW>), grid, blk, lds_a, st, f);
        p = 3 * L * ph;
      }
      a.tmo = sync;
    }
  }
  for (int s = 0; s < (fused ? 0 : ph); ++s) {
    a.s = s;
    for (int l = 0; l < L; ++l) {
      a.L = l;
      const mmt_dec_layer &W = dec->layer[l];
      const mmt_dec_layer &prev = l == 0 ? last : dec->layer[l - 1];
      a.w1 = M4 ? W.qkv4 : W.qkv.w;
      a.b1 = W.qkv.b;
      a.w2 = M4 ? W.out4 : W.out.w;
      a.kv = selfkv;
      a.red_bias = prev.ff2.b;
      a.red_ln = prev.ln3;
      bufs(a);
      drop_of(a, PH_A, l, s);
      if (l > 0) LAUNCH_PHASE(PH_A, XM_REDUCE, lds_a);
      else if (s == 0) LAUNCH_PHASE(PH_A, XM_EMBED, lds_a);
      else LAUNCH_PHASE(PH_A, XM_HEAD_EMBED, lds_a);
      a.red_bias = W.out.b;
      a.red_ln = W.ln1;
      a.w1 = M4 ? W.cq4 : W.cq.w;
      a.b1 = W.cq.b;
      a.w2 = M4 ? W.cout4 : W.cout.w;
      a.kv = memkv;
      bufs(a);
      drop_of(a, PH_B, l, s);
      LAUNCH_PHASE(PH_B, XM_REDUCE, lds_a);
      a.red_bias = W.cout.b;
      a.red_ln = W.ln2;
      a.w1 = M4 ? W.ff14 : W.ff1.w;
      a.b1 = W.ff1.b;
      a.w2 = M4 ? W.ff24 : W.ff2.w;
      bufs(a);
      drop_of(a, PH_C, l, s);
      LAUNCH_PHASE(PH_C, XM_REDUCE, lds_c);
    }
  }
  a.s = ph;
  a.red_bias = last.ff2.b;
  a.red_ln = last.ln3;
  bufs(a);
  drop_of(a, PH_H, L - 1, ph);
  if constexpr (M4) {
    if (drop) hipLaunchKernelGGL((mmt_dec_phase_kernel<DH, CPW, PH_H, XM_REDUCE, true>), dim3(NC), blk, lds_c, st, a);
    else hipLaunchKernelGGL((mmt_dec_phase_kernel<DH, CPW, PH_H, XM_REDUCE>), dim3(NC), blk, lds_c, st, a);
  } else
  hipLaunchKernelGGL((mmt_dec_phase_kernel<DH, CPW, PH_H, XM_REDUCE>), dim3(NC), blk, lds_c, st, a);
#undef LAUNCH_PHASE
  CL_CHECK("mmt_decoder_forward (cluster phases)");
  return MMT_OK;
}

// The one-launch kernel spins on its siblings: it is only started when the occupancy query says the whole grid is resident.
static bool one_launch_fits(bool pipe, bool cpw2, int grid, int ncu, int d, bool drop = false, bool wide = false, bool stash = false) {
  constexpr int DH = 32;
  constexpr int LD1 = 3 * DH + 8;
  const size_t lds = pipe ? pipe_lds_floats<DH>(d) * sizeof(float)
                          : ((size_t)16 * (d + MMT_PAD) + CNW * KS_SMAX * 256 + 16 * LD1 + 3 * 16 * LDO + 16 * LDTOK + 256 +
                             (size_t)16 * KMAX * (DH + 4)) * sizeof(float);
  int per_cu = 0;
  hipError_t e;
  if (pipe && stash && wide) e = drop ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mmt_dec_pipe4_kernel<32, 2, 256, 64, true, 64, true>, CNT, lds)
                                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mmt_dec_pipe4_kernel<32, 2, 256, 64, false, 64, true>, CNT, lds);
  else if (pipe && stash) e = drop ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mmt_dec_pipe4_kernel<32, 2, 256, 64, true, 32, true>, CNT, lds)
                                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mmt_dec_pipe4_kernel<32, 2, 256, 64, false, 32, true>, CNT, lds);
  else if (pipe && wide) e = drop ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mmt_dec_pipe4_kernel<32, 2, 256, 64, true, 64>, CNT, lds)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mmt_dec_pipe4_kernel<32, 2, 256, 64, false, 64>, CNT, lds);
  else if (pipe && drop) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mmt_dec_pipe4_kernel<32, 2, 256, 64, true>, CNT, lds);
  else if (pipe) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mmt_dec_pipe4_kernel<32, 2, 256, 64>, CNT, lds);
  else e = cpw2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mmt_dec_fused4_kernel<32, 2>, CNT, lds)
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mmt_dec_fused4_kernel<32, 1>, CNT, lds);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return per_cu >= 1 && grid <= per_cu * ncu;
}

unsigned mmt_split_encoder_take_fault();      // encoder_split.hip

// Reports (and clears) a timed-out one-launch decode / split encoder on the current device; see include/mmt_amd.h.
extern "C" int mmt_decoder_status(void *stream) {
  hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) return mmt_set_error(MMT_ELAUNCH, "mmt_decoder_status: %s", hipGetErrorString(e));
  if (mmt_split_encoder_take_fault())
    return mmt_set_error(MMT_ELAUNCH, "mmt_encoder_forward_ws: a workgroup of the two-workgroups-per-agent encoder gave up waiting for "
                                      "its partner (GPU shared or CU-masked?); that call's memory rows are NaN.  That form is now "
                                      "disabled on this device: re-issue the call");
  unsigned v = 0;
  e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_decoder_fault), sizeof(v));
  if (e != hipSuccess) return mmt_set_error(MMT_ELAUNCH, "mmt_decoder_status: %s", hipGetErrorString(e));
  if (!v) return MMT_OK;
  const unsigned zero = 0;
  unsigned info[4] = {0, 0, 0, 0};
  (void)hipMemcpyFromSymbol(info, HIP_SYMBOL(g_decoder_fault_info), sizeof(info));
  const unsigned zero4[4] = {0, 0, 0, 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_decoder_fault_info), zero4, sizeof(zero4));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_decoder_fault), &zero, sizeof(zero));
  g_dev[cur_dev()].fused_disabled.store(true, std::memory_order_relaxed);
  if (v & 4u)
    return mmt_set_error(MMT_ELAUNCH, "mmt_sweep_pipe_run: the one-launch backward sweep on this device gave up waiting for a stage "
                                      "(GPU shared or CU-masked?); that step's loss is NaN and its gradients are undefined.  The "
                                      "one-launch forms are now disabled on this device: the next step runs the launch-per-kernel "
                                      "sweep [first give-up: workgroup %u phase %u cluster %u]", info[1], info[2], info[3]);
  return mmt_set_error(MMT_ELAUNCH, "mmt_decoder_forward: a one-launch decode on this device gave up waiting for a workgroup of "
                                    "its cluster (GPU shared or CU-masked?); that call's outputs are NaN.  The one-launch form is "
                                    "now disabled on this device: re-issue the call (it runs launch-per-phase) [first give-up: "
                                    "workgroup %u phase %u cluster %u]", info[1], info[2], info[3]);
}
extern "C" int mmt_debug_decoder_stall(int on) {
  g_test_stall.store(on & 3, std::memory_order_relaxed);      // bit 0: the one-launch decoder forms, bit 1: the backward sweep pipeline
  if (!on) g_dev[cur_dev()].fused_disabled.store(false, std::memory_order_relaxed);   // tests re-arm the one-launch form
  return MMT_OK;
}

// The pipeline's STASH instantiation (include/mmt_amd.h: mmt_decoder_forward_stash): static conditions + what the device says now
bool mmt_cluster_decoder_stash_ok(const mmt_decoder *dec, int map_dim, int B, int T, int ph, bool drop) {
  if (dec->d_model != 256 || dec->nhead != 8 || dec->d_ff != 512 || dec->n_out != 32) return false;
  if (dec->has_map_fusion) {      // a single Linear (fusion_hist_map_layer = 1) from cat[output_fc, map feature] (<= 64 wide) to 32
    if (dec->map_fusion.n_hidden != 0 || map_dim < 1 || dec->n_out + map_dim > 64 || dec->map_fusion.out.k != dec->n_out + map_dim ||
        dec->map_fusion.out.n != 32 || !pipe_wide_chain(dec, false))
      return false;
  } else if (map_dim != 0) return false;
  if (dec->head.n_hidden != 1 || dec->head.hidden[0].n != 32 || dec->head.hidden[0].k != 32 || !dec->head.hidden_ln[0].g) return false;
  if (B < 1 || B > PIPE_MAX_B || T < 1 || T > KMAX || ph < 1 || ph > KMAX) return false;
  const int dev = cur_dev(), ncu = dev_ncu(dev), NC = (B + 3) / 4;
  if (cluster_rows(dec, B) != 4 || !pipe_applies(dec, NC, ncu) || g_dev[dev].fused_disabled.load(std::memory_order_relaxed)) return false;
  const PipeGeom pg = pipe_geom(dec->nlayers, dec->nhead, NC, ncu);
  return one_launch_fits(true, true, pg.grid, ncu, dec->d_model, drop, dec->has_map_fusion != 0, true);
}
void mmt_cluster_decoder_stash_layout(const mmt_decoder *dec, int map_dim, int B, int T, int ph, mmt_decoder_stash_layout *out) {
  *out = stash_off(B, T, ph, dec->d_model, dec->d_ff, dec->nhead, dec->n_out, dec->nlayers, dec->has_map_fusion ? map_dim : 0);
}

// Same contract as mmt_decoder_forward (arguments already validated there).
bool mmt_cluster_decoder_dropout_ok(const mmt_decoder *dec, int B) {
  return cluster_rows(dec, B) == 4 && mfma4_fits(dec);
}

int mmt_cluster_decoder_forward(const mmt_decoder *dec, const float *memory, const uint8_t *pad_mask, const float *start_pos,
                                const float *map_feat, int map_dim, const float *mem_feat, int mem_feat_dim,
                                const float *tgt, int B, int T, int ph, float *pred_st, float *h_out, float *workspace,
                                int allow_fused, void *stream, const mmt_decoder_dropout *drop, float *stash) {      // 0: launch per phase, 1: fused kernel, 2: pipeline, 3: best
  hipStream_t st = (hipStream_t)stream;
  const int d = dec->d_model, H = dec->nhead, L = dec->nlayers;
  const int RC = cluster_rows(dec, B), NC = (B + RC - 1) / RC;
  if (drop) {
    if (!mmt_cluster_decoder_dropout_ok(dec, B) || tgt)
      return mmt_set_error(MMT_EINVAL, "mmt_decoder_forward_dropout: only the autoregressive decode on the 4-agent-cluster kernels "
                                       "(head dimension 32, lin4 weight copies, batch small enough for 4-agent clusters) takes keep masks");
    // the launch-per-phase kernels and the pipeline are instantiated with the mask reads, the fused kernel is not
    allow_fused = (allow_fused == 2 || allow_fused == 3) ? 2 : 0;
  }
  float *memkv = workspace;
  float *selfkv = memkv + (size_t)L * B * T * 2 * d;
  float *extra = selfkv + (size_t)L * B * ph * 2 * d;
  float *part[2], *xres[2];
  part[0] = extra;
  part[1] = part[0] + (size_t)NC * H * 16 * d;
  xres[0] = part[1] + (size_t)NC * H * 16 * d;
  xres[1] = xres[0] + (size_t)NC * 16 * d;
  float *fused = xres[1] + (size_t)NC * 16 * d;
  float *after = fused + (dec->has_mem_fusion ? (size_t)B * T * d : 0) + 64;
  unsigned *sync = reinterpret_cast<unsigned *>((reinterpret_cast<uintptr_t>(after) + 127) & ~(uintptr_t)127);
  float *x0buf = reinterpret_cast<float *>(sync + (size_t)(NC + 1) * 32);
  // One launch for all phases when every workgroup of the decode is resident at once (one per CU: the phase kernels need
  // > 80 KB of LDS) and the 4x4x1 path applies; else a launch per phase.  MMT_NO_FUSED_DECODER=1 forces the latter.
  const int dev = cur_dev(), ncu = dev_ncu(dev);
  const bool m4 = RC == 4 && mfma4_fits(dec);
  // One-launch forms: all their workgroups must be resident at once (one per CU by their LDS footprint), checked against the
  // occupancy query, not the CU count alone.  A device whose one-launch decode has timed out before (mmt_decoder_status) is not
  // trusted again.  The weight-stationary pipeline is preferred; the fused kernel needs NC*H <= #CUs.
  int one_launch = ONE_LAUNCH_NONE;
  PipeGeom pg = {0, 0, 0, nullptr};
  if (allow_fused != 0 && m4 && !dec_env().no_fused && !g_dev[dev].fused_disabled.load(std::memory_order_relaxed)) {
    if (d == 256 && dec->d_ff == 64 * H) pg = pipe_geom(L, H, NC, ncu);      // the geometry the pipeline kernel is compiled for
    if (allow_fused != 1 && !dec_env().no_pipe && pg.R >= 1 && one_launch_fits(true, d > 128, pg.grid, ncu, d, drop != nullptr, pipe_wide_chain(dec, tgt != nullptr), stash != nullptr)) one_launch = ONE_LAUNCH_PIPE;
    else if (allow_fused != 2 && NC * H <= ncu && one_launch_fits(false, d > 128, NC * H, ncu, d)) one_launch = ONE_LAUNCH_FUSED;
  }
  if (stash && one_launch != ONE_LAUNCH_PIPE)
    return mmt_set_error(MMT_EINVAL, "mmt_decoder_forward_stash: the pipeline decode does not run here (geometry, occupancy, a device "
                                     "whose one-launch decode timed out, or a kernel family forced by mmt_decoder_select)");
  const bool fuse_phases = one_launch != ONE_LAUNCH_NONE;
  pg.x0buf = x0buf;
  mmt_note_decoder_path(one_launch == ONE_LAUNCH_PIPE ? MMT_PATH_CLUSTER_PIPELINE : one_launch == ONE_LAUNCH_FUSED ? MMT_PATH_CLUSTER_FUSED
                        : m4 ? MMT_PATH_CLUSTER_LAUNCHES4 : MMT_PATH_CLUSTER_LAUNCHES16);

  const float *src = memory;
  const int rows = B * T;
  if (dec->has_mem_fusion) {
    FuseArgs f;
    f.rows = rows;
    f.d = d;
    f.T = T;
    f.mf = mem_feat_dim;
    f.ld = ((std::max(mlp_width(dec->mem_fusion), d + mem_feat_dim) + 15) & ~15) + MMT_PAD;
    f.memory = memory;
    f.mem_feat = mem_feat;
    f.fused = fused;
    f.M = dec->mem_fusion;
    const size_t lds = (size_t)16 * 3 * f.ld * sizeof(float);
    if (lds > 160 * 1024) return mmt_set_error(MMT_EINVAL, "mmt_decoder_forward: memory fusion tile needs %zu B of LDS", lds);
    (void)hipFuncSetAttribute((const void *)mmt_dec_fuse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mmt_dec_fuse_kernel, dim3((rows + 15) / 16), dim3(CNT), lds, st, f);
    CL_CHECK("mmt_decoder_forward (memory fusion)");
    src = fused;
  }
  {
    CkvArgs c;
    c.rows = rows;
    c.d = d;
    c.L = L;
    c.B = B;
    c.T = T;
    c.ldx = d + MMT_PAD;
    c.ldb = 2 * d + MMT_PAD;
    c.src = src;
    c.memkv = memkv;
    c.sync = sync;
    c.nsync = fuse_phases ? (NC + 1) * 32 : 0;
    const bool pipe = one_launch == ONE_LAUNCH_PIPE;
    c.fill[0] = reinterpret_cast<unsigned *>(part[0]);
    c.nfill[0] = pipe ? (int)((size_t)NC * 16 * d * (2 * H + 2)) : 0;
    c.fill[1] = reinterpret_cast<unsigned *>(x0buf);
    c.nfill[1] = pipe ? NC * 4 * d : 0;
    for (int l = 0; l < L; ++l) c.ckv[l] = dec->layer[l].ckv;
    const size_t lds = (size_t)16 * (c.ldx + c.ldb) * sizeof(float);
    hipLaunchKernelGGL(mmt_dec_ckv_kernel, dim3((rows + 15) / 16, L), dim3(CNT), lds, st, c);
    CL_CHECK("mmt_decoder_forward (cross K/V)");
  }
  ClArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B;
  a.T = T;
  a.ph = ph;
  a.d = d;
  a.H = H;
  a.RC = RC;
  a.n_in = dec->n_in;
  a.n_out = dec->n_out;
  a.FS16 = dec->d_ff / H / 16;
  a.map_dim = map_dim;
  a.has_map_fusion = (!tgt && dec->has_map_fusion) ? 1 : 0;
  a.tagged = m4 ? 1 : 0;
  a.input_fc = dec->input_fc;
  a.output_fc = dec->output_fc;
  a.pe = dec->pe;
  if (!tgt) {
    auto push = [&](const mmt_mlp &M) {
      for (int i = 0; i < M.n_hidden; ++i) {
        a.chain[a.n_chain] = M.hidden[i];
        a.chain_ln[a.n_chain++] = M.hidden_ln[i];
      }
      a.chain[a.n_chain++] = M.out;       // chain_ln stays NULL: no LayerNorm after an Mlp's last Linear
    };
    if (dec->has_map_fusion) push(dec->map_fusion);
    push(dec->head);
  }
  a.start_pos = start_pos;
  a.tgt = tgt;
  a.map_feat = map_feat;
  a.pred_st = pred_st;
  a.h_out = h_out;
  a.pad_mask = pad_mask;
  const int dh = d / H;
  const bool cpw2 = d > 128;
  if (m4) {
    // (CPW still sizes the K-split of the head's small linears, which stay on 16x16x4 tiles)
    return cpw2 ? launch_phases<32, 2, true>(dec, a, B, ph, part, xres, memkv, selfkv, sync, one_launch, pg, st, drop, stash)
                : launch_phases<32, 1, true>(dec, a, B, ph, part, xres, memkv, selfkv, sync, one_launch, pg, st, drop, stash);
  }
  if (dh == 32)
    return cpw2 ? launch_phases<32, 2, false>(dec, a, B, ph, part, xres, memkv, selfkv, sync, ONE_LAUNCH_NONE, pg, st)
                : launch_phases<32, 1, false>(dec, a, B, ph, part, xres, memkv, selfkv, sync, ONE_LAUNCH_NONE, pg, st);
  return cpw2 ? launch_phases<16, 2, false>(dec, a, B, ph, part, xres, memkv, selfkv, sync, ONE_LAUNCH_NONE, pg, st)
              : launch_phases<16, 1, false>(dec, a, B, ph, part, xres, memkv, selfkv, sync, ONE_LAUNCH_NONE, pg, st);
}


// ------------------------------------------------------------------------------- backward sweep pipeline (decoder_bwd_pipe.cuh)
static bool sweep_pipe_static_ok(const mmt_sweep_pipe *a) {
  if (!a) return false;
  if (a->d != 256 || a->H != 8 || a->d_ff != 64 * a->H || a->n_out != 32 || a->n_pos != 2) return false;
  if (a->nlayers < 1 || a->nlayers > MMT_MAX_LAYERS || 3 * a->nlayers / 8 + 1 > (int)TAG_MASK + 1) return false;      // tag generations: see pipe_applies
  if (a->T < 1 || a->T > KMAX || a->ph < 1 || a->ph > KMAX || a->B < 1 || a->B > PIPE_MAX_B) return false;
  if (a->w_fuse ? (!a->h_gu || a->map_dim < 1 || a->gho_ld != a->n_out + a->map_dim || a->gho_ld > 64) : a->gho_ld != a->n_out) return false;
  return true;
}
static bool sweep_has_dropout(const mmt_sweep_pipe *a) {
  if (a->k_pe) return true;
  for (int l = 0; l < a->nlayers; ++l) {
    const mmt_sweep_layer &W = a->layer[l];
    if (W.k_ln[0] || W.k_ln[1] || W.k_ln[2] || W.k_ff || W.k_att[0] || W.k_att[1]) return true;
  }
  return false;
}
extern "C" int mmt_sweep_pipe_supported(const mmt_sweep_pipe *a) {
  if (!sweep_pipe_static_ok(a) || dec_env().no_fused || dec_env().no_pipe || getenv("MMT_NO_PIPELINE_SWEEP")) return 0;
  const int dev = cur_dev(), ncu = dev_ncu(dev);
  if (g_dev[dev].fused_disabled.load(std::memory_order_relaxed)) return 0;
  const PipeGeom pg = pipe_geom(a->nlayers, a->H, (a->B + 3) / 4, ncu);
  if (pg.R < 1 || pg.grid > ncu) return 0;
  int nb = 0;
  const size_t lds = bwd_lds_floats(a->d) * sizeof(float);
  const bool drop = sweep_has_dropout(a);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, drop ? (const void *)mmt_bwd_pipe4_kernel<32, 256, 64, true>
                                                         : (const void *)mmt_bwd_pipe4_kernel<32, 256, 64, false>, CNT, lds) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return nb >= 1;
}
extern "C" size_t mmt_sweep_pipe_workspace_floats(const mmt_sweep_pipe *a) {
  if (!sweep_pipe_static_ok(a)) return 0;
  const size_t NC = (a->B + 3) / 4, d = a->d;
  return 2 * NC * a->H * 16 * d + 2 * NC * 16 * d + NC * 4 * d + 64;
}
extern "C" int mmt_sweep_pipe_run(const mmt_sweep_pipe *a, float *ws, void *stream) {
  if (!sweep_pipe_static_ok(a) || !ws) return mmt_set_error(MMT_EINVAL, "mmt_sweep_pipe_run: unsupported geometry or no workspace");
  if (!mmt_sweep_pipe_supported(a)) return mmt_set_error(MMT_EINVAL, "mmt_sweep_pipe_run: the pipeline does not fit this device (or is disabled)");
  hipStream_t st = (hipStream_t)stream;
  const int dev = cur_dev(), ncu = dev_ncu(dev);
  BwdPipeArgs f;
  f.a = *a;
  const size_t NC = (a->B + 3) / 4, d = a->d;
  const PipeGeom pg = pipe_geom(a->nlayers, a->H, (int)NC, ncu);
  float *p = ws;
  f.sync = reinterpret_cast<unsigned *>(p);
  p += 64;
  const size_t npart = NC * a->H * 16 * d, nres = NC * 16 * d;
  f.part[0] = p, f.part[1] = p + npart;
  p += 2 * npart;
  f.xres[0] = p, f.xres[1] = p + nres;
  p += 2 * nres;
  f.x0buf = p;
  p += NC * 4 * d;
  f.NC = (int)NC, f.R = pg.R, f.NHW = pg.NHW;
  f.test_stall = (g_test_stall.load(std::memory_order_relaxed) >> 1) & 1;
  // every hand-off word starts as the 0xFFFFFFFF no tag matches; the time-out word as 0
  hipLaunchKernelGGL(mmt_bwd_pipe_fill_kernel, dim3(512), dim3(256), 0, st, ws, (size_t)(p - (ws + 64)) / 4);
  CL_CHECK("mmt_bwd_pipe_fill_kernel");
  const size_t lds = bwd_lds_floats(a->d) * sizeof(float);
  if (sweep_has_dropout(a)) hipLaunchKernelGGL((mmt_bwd_pipe4_kernel<32, 256, 64, true>), dim3(pg.grid), dim3(CNT), lds, st, f);
  else hipLaunchKernelGGL((mmt_bwd_pipe4_kernel<32, 256, 64, false>), dim3(pg.grid), dim3(CNT), lds, st, f);
  CL_CHECK("mmt_bwd_pipe4_kernel");
  hipLaunchKernelGGL(mmt_bwd_pipe_check_kernel, dim3(1), dim3(64), 0, st, (const unsigned *)f.sync, a->poison);
  CL_CHECK("mmt_bwd_pipe_check_kernel");
  return MMT_OK;
}
