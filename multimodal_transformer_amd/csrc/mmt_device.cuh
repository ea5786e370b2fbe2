// Device-side building blocks shared by the encoder / decoder / ViT kernels (gfx950 only).
//
// Execution model: one workgroup (NW wave64s) owns a tile of MT = 16*MB activation rows that
// lives in LDS for the whole kernel; weights are streamed straight from L2/Infinity-Cache into
// VGPRs in "MFMA streaming order" (see include/mmt_amd.h) and consumed by
// v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain).  No tensor leaves LDS between
// the ops of a layer; residual + LayerNorm run as an in-LDS epilogue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mmt_amd.h"

#include "mmt_gfx950.cuh"

#define MMT_LN_EPS 1e-5f
#define MMT_PAD 8   // row padding (floats) that makes the ds_read_b128 A-operand reads conflict free

// MMT_OPAQUE_TID (decoder_cluster.hip): the thread index goes through an empty volatile asm, so that nothing derived from it is
// loop invariant.  The fused decoder kernel inlines five phase bodies into one loop; left alone, LICM hoists every per-lane
// offset of every body into the loop preheader and the kernel spills hundreds of registers.
__device__ __forceinline__ int thread_idx() {
  int t = threadIdx.x;
#ifdef MMT_OPAQUE_TID
  asm volatile("" : "+v"(t));
#endif
  return t;
}
__device__ __forceinline__ int lane_id() { return thread_idx() & 63; }
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(thread_idx() >> 6); }   // SGPR: keeps wave-level control flow scalar
__device__ __forceinline__ int n_waves() { return blockDim.x >> 6; }

__device__ __forceinline__ int round16(int v) { return (v + 15) & ~15; }

// ---------------------------------------------------------------------------------------------
// Weight streaming registers: one "group" = U consecutive 16-k chunks of SL slabs (SL*U KiB per wave).
#ifndef MMT_WSL
#define MMT_WSL 2
#endif
#ifndef MMT_WU
#define MMT_WU 4
#endif
#ifndef MMT_NT
#define MMT_NT 512
#endif
constexpr int WSL = MMT_WSL;   // slabs in flight per wave (independent MFMA accumulator chains)
constexpr int WU = MMT_WU;     // chunks per group
struct WRegs {
  f32x4 w[WSL][WU];
};

struct NextLin {          // the linear that follows in program order (its first group is prefetched)
  const float *w;
  int K16, N16;
};
__device__ __forceinline__ NextLin next_of(const mmt_linear &L) {
  return NextLin{L.w, round16(L.k) >> 4, round16(L.n) >> 4};
}
__device__ __forceinline__ NextLin no_next() { return NextLin{nullptr, 0, 0}; }
__device__ __forceinline__ bool fast_geom(int K16) { return K16 >= WU && (K16 % WU) == 0; }

__device__ __forceinline__ void wload(WRegs &r, const f32x4 *W4, int K16, int N16, int s0, int c0) {
#pragma unroll
  for (int s = 0; s < WSL; ++s) {
    const int slab = min(s0 + s, N16 - 1);    // a dead slab re-reads the last one (in bounds, discarded)
    const f32x4 *p = W4 + ((size_t)slab * K16 + c0) * 64;
#pragma unroll
    for (int u = 0; u < WU; ++u) r.w[s][u] = p[(size_t)u * 64];
  }
}

// Y[r][n] = epi(r, n, sum_k X[r][k] * W[n][k])  for r < 16*MB, n < N16*16.
// X: LDS, row stride ldx floats, columns [0, 16*K16) must be finite (zero padded).
// W: packed weights.  Each wave owns WSL consecutive 16-column slabs at a time ("slab set").
// A operand lane (i = lane&15, q = lane>>4) holds X[i][16c + 4q + e], B operand lane (j, q) holds
// W[16s + j][16c + 4q + e]; MFMA step e sums the four q's, four steps cover the 16 k of a chunk.
// C/D layout: col = lane&15, row = 4*(lane>>4) + reg.
// The wave's weight reads form one flat stream (slab set major, k minor) that is software-pipelined one
// group ahead ACROSS slab sets, and -- through `pre` / `next` -- across the barrier-separated phases of a
// layer: the first group of the next linear is requested before this one's last MFMAs issue, so the
// L2/Infinity-Cache latency hides behind the epilogue, the barrier and the LayerNorm/attention phase.
//   PRE: `pre` already holds this linear's first group (the previous call was given it as `next`).
//   XR: one more row, X[16*MB], is multiplied on the VALU from the very weight registers the MFMAs use (lane (j, q) holds
//   W[16s + j][16c + 4q .. +3]: four FMAs per weight quad, the four q's summed across the lane groups at the flush).  A 17th
//   row (the ViT's cls token) then costs 4 VALU FMAs per weight quad instead of a second, 15/16 empty MFMA row tile.
template <int MB, bool PRE, class Epi, bool XR = false>
__device__ __forceinline__ void tile_linear(const float *__restrict__ X, int ldx, const float *__restrict__ Wp,
                                            int K16, int N16, Epi epi, WRegs &pre, NextLin next) {
  const int lane = lane_id();
  const int nw = n_waves();
  const int first = wave_id() * WSL;
  const float *xrow = X + (lane & 15) * ldx + (lane >> 4) * 4;
  const f32x4 *W4 = reinterpret_cast<const f32x4 *>(Wp) + lane;
  const f32x4 *N4 = reinterpret_cast<const f32x4 *>(next.w) + lane;
  const bool do_next = next.w != nullptr && fast_geom(next.K16);
  const int nfirst = min(first, max(next.N16 - 1, 0));
  f32x4 acc[WSL][MB];
  float xacc[XR ? WSL : 1];
  const float *xextra = X + MB * 16 * ldx + (lane >> 4) * 4;
#pragma unroll
  for (int s = 0; s < WSL; ++s)
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[s][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (XR) {
#pragma unroll
    for (int s = 0; s < WSL; ++s) xacc[s] = 0.f;
  }

  // Epilogue operands (bias, residual) are requested before the last MFMAs of a slab set so that the
  // flush itself is add + ds_write with nothing to wait for.
  typename Epi::template Pre<MB> pb[WSL];
  auto prep = [&](int s0) {
#pragma unroll
    for (int s = 0; s < WSL; ++s) epi.prep(pb[s], (lane >> 4) * 4, min(s0 + s, N16 - 1) * 16 + (lane & 15));
  };
  auto flush = [&](int s0) {
#pragma unroll
    for (int s = 0; s < WSL; ++s) {
      if (s0 + s < N16) {
        const int col = (s0 + s) * 16 + (lane & 15);
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) epi.store(m * 16 + (lane >> 4) * 4 + r, col, acc[s][m][r], pb[s], m, r);
      }
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[s][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (XR) {
        float v = xacc[s];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (s0 + s < N16 && lane < 16) epi.store_row(MB * 16, (s0 + s) * 16 + lane, v, pb[s]);
        xacc[s] = 0.f;
      }
    }
  };

  if (fast_geom(K16)) {
    const int nsets = first < N16 ? (N16 - first + nw * WSL - 1) / (nw * WSL) : 0;
    const int gps = K16 / WU;
    const int total = nsets * gps;
    WRegs cur;
    if (PRE) cur = pre;
    else if (total > 0) wload(cur, W4, K16, N16, first, 0);
    int set = 0, kg = 0;
    for (int g = 0; g < total; ++g) {
      int kg1 = kg + 1, set1 = set;
      if (kg1 == gps) {
        kg1 = 0;
        ++set1;
      }
      WRegs nxt = cur;
#ifndef MMT_EXP_NOLOAD
      if (g + 1 < total) wload(nxt, W4, K16, N16, first + set1 * nw * WSL, kg1 * WU);
      else if (do_next) wload(nxt, N4, next.K16, next.N16, nfirst, 0);
#endif
      if (kg == gps - 1) prep(first + set * nw * WSL);
#pragma unroll
      for (int u = 0; u < WU; ++u) {
        f32x4 xv[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m)
          xv[m] = *reinterpret_cast<const f32x4 *>(xrow + m * 16 * ldx + (kg * WU + u) * 16);
#ifdef MMT_EXP_HALFMFMA
#define MMT_E_END 2
#else
#define MMT_E_END 4
#endif
#pragma unroll
        for (int e = 0; e < MMT_E_END; ++e)
#pragma unroll
          for (int s = 0; s < WSL; ++s)
#pragma unroll
            for (int m = 0; m < MB; ++m)
              acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[m][e], cur.w[s][u][e], acc[s][m], 0, 0, 0);
        if constexpr (XR) {
          const f32x4 xe = *reinterpret_cast<const f32x4 *>(xextra + (kg * WU + u) * 16);
#pragma unroll
          for (int s = 0; s < WSL; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) xacc[s] = fmaf(xe[e], cur.w[s][u][e], xacc[s]);
        }
      }
      if (kg == gps - 1) flush(first + set * nw * WSL);
      cur = nxt;
      kg = kg1;
      set = set1;
    }
    if (total == 0 && do_next) wload(cur, N4, next.K16, next.N16, nfirst, 0);
    pre = cur;
  } else {
    // small / odd K (input_fc, the 32-wide MLPs): plain loop, no pipelining
    for (int s0 = first; s0 < N16; s0 += nw * WSL) {
      prep(s0);
      for (int c = 0; c < K16; ++c) {
        f32x4 wv[WSL];
#pragma unroll
        for (int s = 0; s < WSL; ++s) wv[s] = W4[((size_t)min(s0 + s, N16 - 1) * K16 + c) * 64];
        f32x4 xv[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m) xv[m] = *reinterpret_cast<const f32x4 *>(xrow + m * 16 * ldx + c * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int s = 0; s < WSL; ++s)
#pragma unroll
            for (int m = 0; m < MB; ++m)
              acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[m][e], wv[s][e], acc[s][m], 0, 0, 0);
        if constexpr (XR) {
          const f32x4 xe = *reinterpret_cast<const f32x4 *>(xextra + c * 16);
#pragma unroll
          for (int s = 0; s < WSL; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) xacc[s] = fmaf(xe[e], wv[s][e], xacc[s]);
        }
      }
      flush(s0);
    }
    if (do_next) wload(pre, N4, next.K16, next.N16, nfirst, 0);
  }
}

// Convenience: Y = X W^T + b [+relu] [+= residual already in Y]
enum { EPI_STORE = 0, EPI_RELU = 1, EPI_ADD = 2, EPI_GELU = 3 };

template <int MODE>
struct LdsEpi {
  const float *__restrict__ bias;
  float *__restrict__ Y;
  int n, ldy;
  template <int MB>
  struct Pre {
    float b;
    float res[MB][4];
  };
  // row0 = first of the lane's 4 rows inside a 16-row block, col = the lane's output column
  template <int MB>
  __device__ __forceinline__ void prep(Pre<MB> &p, int row0, int col) const {
    p.b = (bias != nullptr && col < n) ? bias[col] : 0.f;
    if (MODE == EPI_ADD) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) p.res[m][r] = col < n ? Y[(m * 16 + row0 + r) * ldy + col] : 0.f;
    }
  }
  template <int MB>
  __device__ __forceinline__ void store(int r, int c, float v, const Pre<MB> &p, int m, int ri) const {
    if (c < n) {
      v += p.b;
      if (MODE == EPI_RELU) v = fmaxf(v, 0.f);
      if (MODE == EPI_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));   // nn.GELU() (exact erf form)
      if (MODE == EPI_ADD) v += p.res[m][ri];
      Y[r * ldy + c] = v;
    }
  }
  // the extra (VALU) row of tile_linear<.., XR = true>: lane's column c, bias already in p.b
  template <int MB>
  __device__ __forceinline__ void store_row(int r, int c, float v, const Pre<MB> &p) const {
    if (c < n) {
      v += p.b;
      if (MODE == EPI_RELU) v = fmaxf(v, 0.f);
      if (MODE == EPI_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
      if (MODE == EPI_ADD) v += Y[r * ldy + c];
      Y[r * ldy + c] = v;
    }
  }
};

// Y[0..16*MB] = the MFMA rows, Y[16*MB] = one more row on the VALU (tile_linear's XR form)
template <int MB, int MODE>
__device__ __forceinline__ void linear_to_lds_xrow(const float *X, int ldx, const mmt_linear &L, float *Y, int ldy) {
  LdsEpi<MODE> epi{L.b, Y, L.n, ldy};
  WRegs scratch;
  tile_linear<MB, false, LdsEpi<MODE>, true>(X, ldx, L.w, round16(L.k) >> 4, round16(L.n) >> 4, epi, scratch, no_next());
}

template <int MB, int MODE, bool PRE = false>
__device__ __forceinline__ void linear_to_lds(const float *X, int ldx, const mmt_linear &L, float *Y, int ldy,
                                              WRegs &pre, NextLin next) {
  LdsEpi<MODE> epi{L.b, Y, L.n, ldy};
  tile_linear<MB, PRE>(X, ldx, L.w, round16(L.k) >> 4, round16(L.n) >> 4, epi, pre, next);
}

template <int MB, int MODE>
__device__ __forceinline__ void linear_to_lds(const float *X, int ldx, const mmt_linear &L, float *Y, int ldy) {
  WRegs scratch;
  linear_to_lds<MB, MODE, false>(X, ldx, L, Y, ldy, scratch, no_next());
}

// ---------------------------------------------------------------------------------------------
// In-place LayerNorm over the first n columns of rows [0, nrows) of an LDS tile; one wave per row.
template <bool RELU>
__device__ __forceinline__ void layer_norm_rows(float *X, int ldx, int nrows, int n, const mmt_norm &N) {
  const int lane = lane_id();
  for (int r = wave_id(); r < nrows; r += n_waves()) {
    float *x = X + r * ldx;
    float s = 0.f;
    for (int c = lane; c < n; c += 64) s += x[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mu = s / (float)n;
    float v = 0.f;
    for (int c = lane; c < n; c += 64) {
      float d = x[c] - mu;
      v += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const float rstd = 1.0f / sqrtf(v / (float)n + MMT_LN_EPS);
    for (int c = lane; c < n; c += 64) {
      float y = (x[c] - mu) * rstd * N.g[c] + N.b[c];
      if (RELU) y = fmaxf(y, 0.f);
      x[c] = y;
    }
  }
}

// LayerNorm of rows [0, nrows) of src into dst (different LDS tiles, same column count).
__device__ __forceinline__ void layer_norm_rows_to(const float *S, int lds, float *Dst, int ldd, int nrows, int n,
                                                   const mmt_norm &N) {
  const int lane = lane_id();
  for (int r = wave_id(); r < nrows; r += n_waves()) {
    const float *x = S + r * lds;
    float s = 0.f;
    for (int c = lane; c < n; c += 64) s += x[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mu = s / (float)n;
    float v = 0.f;
    for (int c = lane; c < n; c += 64) {
      float d = x[c] - mu;
      v += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const float rstd = 1.0f / sqrtf(v / (float)n + MMT_LN_EPS);
    for (int c = lane; c < n; c += 64) Dst[r * ldd + c] = (x[c] - mu) * rstd * N.g[c] + N.b[c];
  }
}

// n_hidden x (Linear, LN, ReLU) + Linear on a 16-row LDS tile; cur/alt are ping-pong regions.
__device__ __forceinline__ void mlp_stack_tile(const mmt_mlp &M, float *cur, float *alt, int ld, float *out,
                                               int ldout, int nrows) {
  for (int i = 0; i < M.n_hidden; ++i) {
    linear_to_lds<1, EPI_STORE>(cur, ld, M.hidden[i], alt, ld);
    __syncthreads();
    layer_norm_rows<true>(alt, ld, nrows, M.hidden[i].n, M.hidden_ln[i]);
    __syncthreads();
    float *t = cur;
    cur = alt;
    alt = t;
  }
  linear_to_lds<1, EPI_STORE>(cur, ld, M.out, out, ldout);
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// One attention "group" = 16 lanes serving one (query row, head); DH = head dim (16, 32 or 64).
//   q      : pointer to the query's head slice (DH floats, LDS)
//   kbase  : pointer to key 0's head slice, keys are kstride floats apart (LDS or global)
//   vbase  : same for values
//   nkeys  : keys 0..nkeys-1 are candidates (<= 32); key j is masked out when (maskbits >> j) & 1
//   out    : head slice of the output (DH floats); may alias q
// Lane j owns keys j and j+16 for the scores and DH/16 contiguous output dims for P.V.  All K loads
// of a lane and all V loads of the first 16 keys are issued before anything is consumed, so a group
// costs one memory round trip instead of one per key (the KV cache lives in L2, ~500+ cycles away).
template <int DH>
__device__ __forceinline__ void attend_group(const float *q, const float *kbase, const float *vbase,
                                             int kstride, int nkeys, uint32_t maskbits, float scale, float *out) {
  constexpr int DPL = DH / 16;      // output dims per lane
  constexpr int NQ = DH / 4;
  typedef float vrow __attribute__((ext_vector_type(DPL)));
  const int l16 = threadIdx.x & 15;
  const float *vmine = vbase + l16 * DPL;
  // V of keys 0..15, my dims (keys >= nkeys: not loaded)
  float vr[16][DPL];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j < nkeys) {
      if constexpr (DPL == 1) {
        vr[j][0] = vmine[(size_t)j * kstride];
      } else {
        const vrow t = *reinterpret_cast<const vrow *>(vmine + (size_t)j * kstride);
#pragma unroll
        for (int e = 0; e < DPL; ++e) vr[j][e] = t[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < DPL; ++e) vr[j][e] = 0.f;
    }
  }
  f32x4 qv[NQ];
#pragma unroll
  for (int c = 0; c < NQ; ++c) qv[c] = *reinterpret_cast<const f32x4 *>(q + 4 * c);
  float sc[2] = {-INFINITY, -INFINITY};
  const int nhalf = nkeys > 16 ? 2 : 1;
  for (int h = 0; h < nhalf; ++h) {
    const int j = l16 + 16 * h;
    if (j < nkeys && !((maskbits >> j) & 1u)) {
      const float *k = kbase + (size_t)j * kstride;
      f32x4 kr[NQ];
#pragma unroll
      for (int c = 0; c < NQ; ++c) kr[c] = *reinterpret_cast<const f32x4 *>(k + 4 * c);
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < NQ; ++c) {
        a = fmaf(qv[c][0], kr[c][0], a);
        a = fmaf(qv[c][1], kr[c][1], a);
        a = fmaf(qv[c][2], kr[c][2], a);
        a = fmaf(qv[c][3], kr[c][3], a);
      }
      if (h == 0) sc[0] = a * scale; else sc[1] = a * scale;
    }
  }
  float mx = fmaxf(sc[0], sc[1]);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
  float p0 = 0.f, p1 = 0.f;
  if (mx > -INFINITY) {   // a fully masked query row yields zeros (reference: undefined / NaN)
    p0 = expf(sc[0] - mx);
    p1 = expf(sc[1] - mx);
  }
  float sum = p0 + p1;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
  p0 *= inv;
  p1 *= inv;
  float o_acc[DPL];
#pragma unroll
  for (int e = 0; e < DPL; ++e) o_acc[e] = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float pj = __shfl(p0, j, 16);
#pragma unroll
    for (int e = 0; e < DPL; ++e) o_acc[e] = pj > 0.f ? fmaf(pj, vr[j][e], o_acc[e]) : o_acc[e];
  }
  if (nkeys > 16) {       // rare: history / horizon beyond 16 tokens
    for (int j = 16; j < nkeys; ++j) {
      const float pj = __shfl(p1, j - 16, 16);
      if (pj > 0.f) {
#pragma unroll
        for (int e = 0; e < DPL; ++e) o_acc[e] = fmaf(pj, vmine[(size_t)j * kstride + e], o_acc[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < DPL; ++e) out[l16 * DPL + e] = o_acc[e];
}
