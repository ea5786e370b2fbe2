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

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MMT_LN_EPS 1e-5f
#define MMT_PAD 8   // row padding (floats) that makes the ds_read_b128 A-operand reads conflict free

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ int n_waves() { return blockDim.x >> 6; }

__device__ __forceinline__ int round16(int v) { return (v + 15) & ~15; }

// ---------------------------------------------------------------------------------------------
// Y[r][n] = epi(r, n, sum_k X[r][k] * W[n][k])  for r < 16*MB, n < N16*16.
// X: LDS, row stride ldx floats, columns [0, 16*K16) must be finite (zero padded).
// W: packed weights.  Each wave owns SL consecutive 16-column slabs at a time.
// A operand lane (i = lane&15, q = lane>>4) holds X[i][16c + 4q + e], B operand lane (j, q) holds
// W[16s + j][16c + 4q + e]; MFMA step e sums the four q's, four steps cover the 16 k of a chunk.
// C/D layout: col = lane&15, row = 4*(lane>>4) + reg.
template <int MB, int SL, class Epi>
__device__ __forceinline__ void tile_linear(const float *__restrict__ X, int ldx,
                                            const float *__restrict__ Wp, int K16, int N16, Epi epi) {
  const int lane = lane_id();
  const int nw = n_waves();
  const float *xrow = X + (lane & 15) * ldx + (lane >> 4) * 4;
  for (int s0 = wave_id() * SL; s0 < N16; s0 += nw * SL) {
    f32x4 acc[SL][MB];
#pragma unroll
    for (int s = 0; s < SL; ++s)
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[s][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 *wp[SL];
    bool live[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) {
      live[s] = (s0 + s) < N16;
      int slab = live[s] ? (s0 + s) : s0;   // wave-uniform; a dead slab re-reads slab s0 (in bounds)
      wp[s] = reinterpret_cast<const f32x4 *>(Wp) + (size_t)slab * K16 * 64 + lane;
    }
    constexpr int U = 4;
    int c = 0;
    if (K16 >= U) {
      f32x4 wcur[SL][U], wnxt[SL][U];
#pragma unroll
      for (int s = 0; s < SL; ++s)
#pragma unroll
        for (int u = 0; u < U; ++u) wcur[s][u] = wp[s][(size_t)u * 64];
      for (; c + U <= K16; c += U) {
        const bool more = (c + 2 * U) <= K16;
        if (more) {
#pragma unroll
          for (int s = 0; s < SL; ++s)
#pragma unroll
            for (int u = 0; u < U; ++u) wnxt[s][u] = wp[s][(size_t)(c + U + u) * 64];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          f32x4 xv[MB];
#pragma unroll
          for (int m = 0; m < MB; ++m)
            xv[m] = *reinterpret_cast<const f32x4 *>(xrow + m * 16 * ldx + (c + u) * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int s = 0; s < SL; ++s)
#pragma unroll
              for (int m = 0; m < MB; ++m)
                acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[m][e], wcur[s][u][e], acc[s][m], 0, 0, 0);
        }
        if (more) {
#pragma unroll
          for (int s = 0; s < SL; ++s)
#pragma unroll
            for (int u = 0; u < U; ++u) wcur[s][u] = wnxt[s][u];
        }
      }
    }
    for (; c < K16; ++c) {   // remainder chunks (K16 not a multiple of 4)
      f32x4 wv[SL];
#pragma unroll
      for (int s = 0; s < SL; ++s) wv[s] = wp[s][(size_t)c * 64];
      f32x4 xv[MB];
#pragma unroll
      for (int m = 0; m < MB; ++m) xv[m] = *reinterpret_cast<const f32x4 *>(xrow + m * 16 * ldx + c * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int s = 0; s < SL; ++s)
#pragma unroll
          for (int m = 0; m < MB; ++m)
            acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[m][e], wv[s][e], acc[s][m], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < SL; ++s) {
      if (!live[s]) continue;
      const int col = (s0 + s) * 16 + (lane & 15);
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) epi(m * 16 + (lane >> 4) * 4 + r, col, acc[s][m][r]);
    }
  }
}

// Convenience: Y = X W^T + b [+relu] [+= residual already in Y]
enum { EPI_STORE = 0, EPI_RELU = 1, EPI_ADD = 2 };

template <int MB, int MODE>
__device__ __forceinline__ void linear_to_lds(const float *X, int ldx, const mmt_linear &L, float *Y, int ldy) {
  const float *bias = L.b;
  const int n = L.n;
  auto epi = [=](int r, int c, float v) {
    if (c < n) {
      if (bias) v += bias[c];
      if (MODE == EPI_RELU) v = fmaxf(v, 0.f);
      float *dst = Y + r * ldy + c;
      if (MODE == EPI_ADD) v += *dst;
      *dst = v;
    }
  };
  tile_linear<MB, 2>(X, ldx, L.w, round16(L.k) >> 4, round16(L.n) >> 4, epi);
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

// ---------------------------------------------------------------------------------------------
// One attention "group" = 16 lanes serving one (query row, head).
//   q      : pointer to the query's head slice (dh floats, LDS)
//   kbase  : pointer to key 0's head slice, keys are kstride floats apart (LDS or global)
//   vbase  : same for values
//   nkeys  : keys 0..nkeys-1 are candidates; key j is masked out when (maskbits >> j) & 1
//   out    : head slice of the output (dh floats); may alias q
// Softmax over <= 32 keys is held in registers: lane j owns keys j and j+16.
__device__ __forceinline__ void attend_group(const float *q, const float *kbase, const float *vbase,
                                             int kstride, int nkeys, uint32_t maskbits, int dh, float scale,
                                             float *out) {
  const int l16 = threadIdx.x & 15;
  float sc[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int j = l16 + 16 * h;
    float s = -INFINITY;
    if (j < nkeys && !((maskbits >> j) & 1u)) {
      const float *k = kbase + (size_t)j * kstride;
      float a = 0.f;
      for (int c = 0; c < dh; c += 4) {
        const f32x4 kv = *reinterpret_cast<const f32x4 *>(k + c);
        const f32x4 qv = *reinterpret_cast<const f32x4 *>(q + c);
        a = fmaf(qv[0], kv[0], a);
        a = fmaf(qv[1], kv[1], a);
        a = fmaf(qv[2], kv[2], a);
        a = fmaf(qv[3], kv[3], a);
      }
      s = a * scale;
    }
    sc[h] = s;
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
  // P.V : lane owns output dims l16, l16+16, ... (< dh)
  float o_acc[4] = {0.f, 0.f, 0.f, 0.f};   // dh <= 64
  for (int j = 0; j < nkeys; ++j) {
    const float pj = __shfl(j < 16 ? p0 : p1, j & 15, 16);
    if (!(pj > 0.f)) continue;   // masked keys may hold non-finite values: never touch them
    const float *v = vbase + (size_t)j * kstride;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = l16 + 16 * t;
      if (c < dh) o_acc[t] = fmaf(pj, v[c], o_acc[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = l16 + 16 * t;
    if (c < dh) out[c] = o_acc[t];
  }
}
