// Training-step building blocks for gfx950: whole-batch fp32 kernels that the host
// orchestration (multimodal_transformer_amd/training.py) chains into forward-with-stash, the reverse-time
// backward sweep through the KV-cached autoregressive decoder, deferred weight-gradient GEMMs over all
// (step, agent) rows, and one fused clip-by-value + AdamW update.  Everything is launched on the caller's
// stream and is capturable in a HIP graph (no allocation, no host sync).
//
//   mmt_gemm            C = epilogue(A x B)   arbitrary strides: forward (X W^T), input grads (dY W) and
//                       weight grads (dY^T X) share one fp32-MFMA kernel (v_mfma_f32_16x16x4_f32)
//   mmt_colsum          bias / LayerNorm parameter gradients: sum_m a[m][n] (* b[m][n])
//   mmt_layernorm_fwd/bwd   residual add + LayerNorm (+ReLU), stashing xhat / rstd
//   mmt_attention_fwd/bwd   short-sequence multi-head attention with arbitrary strides: encoder (causal, T x T),
//                       decoder self-attention on the KV cache (1 x s), cross-attention (1 x T, key padding)
//   mmt_l2_loss_bwd     d/dpos of mean_b mean_t ||pos - y||_2                      (mgcvae.py:614-620)
//   mmt_adamw_multi     clip_grad_value_ + AdamW over a table of tensors            (train.py:218-253)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <stdint.h>
#include "../../include/mmt_amd.h"

int mmt_set_error(int code, const char *fmt, ...);   // transformer.hip

#include "mmt_gfx950.cuh"

namespace {

// ------------------------------------------------------------------------------------------ panel GEMM
// The GEMMs of a training step are many and small (M = 128 rows per decoder step, K = 256..512) or tall in K (weight
// gradients: K = ph*B rows).  BT x BT output tile per workgroup, K walked in 64-deep panels: the next panel's global
// loads (16-byte, branch-free, predicated by out-of-range buffer offsets) are in flight while the current one is
// multiplied out of LDS, one barrier per panel.  An operand whose unit stride is K ("KC") sits in LDS as [row][k]
// and feeds the MFMA with one ds_read_b128; one whose unit stride is the row index sits as [k][row] and feeds it
// with four conflict-free ds_read_b32.  Covers X W^T (KC,KC), dY W (KC,rows) and dY^T X (rows,rows).
// SPLIT (weight gradients, both operands row-major in the reduction dimension): blockIdx.z owns a 64-aligned slice of the
// K range and writes its partial to C + z*part_stride; the workgroups of the first column block also reduce the row sums
// of their A tile (asum: the bias gradient) from the registers the tile passes through anyway.
// epilogue of every GEMM kernel below; v = the accumulated product of output (m, n)
__device__ __forceinline__ float gemm_epi(const mmt_gemm_args &g, float v, int m, int n) {
  if (g.bias) v += g.bias[n];
  if (g.drop) {
    const float keep = g.drop[(size_t)m * g.ld_drop + n] ? g.drop_scale : 0.f;
    if (g.drop_after_add) {
      if (g.addend) v += g.addend[(size_t)m * g.ld_add + n];
      v *= keep;
    } else {
      if (g.relu) v = fmaxf(v, 0.f);
      v *= keep;
      if (g.addend) v += g.addend[(size_t)m * g.ld_add + n];
    }
  } else {
    if (g.addend) v += g.addend[(size_t)m * g.ld_add + n];
    if (g.relu) v = fmaxf(v, 0.f);
  }
  if (g.mask) {
    if (!(g.mask[(size_t)m * g.ld_mask + n] > 0.f)) v = 0.f;
    else if (g.mask_scale != 0.f) v *= g.mask_scale;
  }
  return v;
}
// the same epilogue on operands requested at kernel entry (the row-block kernel: a load at the end of a 5 us kernel is a memory
// round trip on the sweep's critical path); absent operands were "loaded" out of range and read as zero
struct EpiPre {
  float bias, add, mask;
  uint32_t keep;
};
__device__ __forceinline__ EpiPre gemm_epi_load(const mmt_gemm_args &g, int m, int n, bool ok) {
  EpiPre p;
  p.bias = bl1(mk_rsrc(g.bias), (g.bias && ok) ? (unsigned)n * 4u : OOB);
  p.add = bl1(mk_rsrc(g.addend), (g.addend && ok) ? (unsigned)(m * (int)g.ld_add + n) * 4u : OOB);
  p.mask = bl1(mk_rsrc(g.mask), (g.mask && ok) ? (unsigned)(m * (int)g.ld_mask + n) * 4u : OOB);
  p.keep = __builtin_amdgcn_raw_buffer_load_b8(mk_rsrc(g.drop), (g.drop && ok) ? (unsigned)(m * (int)g.ld_drop + n) : OOB, 0, 0);
  return p;
}
__device__ __forceinline__ float gemm_epi_pre(const mmt_gemm_args &g, float v, const EpiPre &p) {
#pragma clang fp contract(off)      // keep * x and + addend stay two roundings (as torch's separate dropout and add kernels)
  if (g.bias) v += p.bias;
  if (g.drop) {
    const float keep = (p.keep & 0xffu) ? g.drop_scale : 0.f;
    if (g.drop_after_add) {
      if (g.addend) v += p.add;
      v *= keep;
    } else {
      if (g.relu) v = fmaxf(v, 0.f);
      v *= keep;
      if (g.addend) v += p.add;
    }
  } else {
    if (g.addend) v += p.add;
    if (g.relu) v = fmaxf(v, 0.f);
  }
  if (g.mask) {
    if (!(p.mask > 0.f)) v = 0.f;
    else if (g.mask_scale != 0.f) v *= g.mask_scale;
  }
  return v;
}
__device__ __forceinline__ float keep_of(uint32_t bytes, int e, float scale) { return ((bytes >> (8 * e)) & 0xffu) ? scale : 0.f; }

// (bx, by, bz) = the tile this workgroup owns: blockIdx of a single-job launch, or what the block map of a group launch says
template <int BT, bool AK, bool BK, bool SPLIT>
__device__ __forceinline__ void gemm_panel_body(mmt_gemm_args g, const int bx, const int by, const int bz, float *smem) {
  constexpr int KC = 64, LDK = KC + 4, LDR = BT + 4;
  constexpr int ASZ = AK ? BT * LDK : KC * LDR, BSZ = BK ? BT * LDK : KC * LDR;
  constexpr int NI = BT / 16;      // 16-byte loads per thread per operand per panel
  constexpr int WT = BT / 32;      // 16x16 MFMA tiles per wave per dimension
  constexpr int R4 = BT / 4;
  float *As = smem, *Bs = smem + 2 * ASZ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = by * BT, n0 = bx * BT;
  const int wm = (wave >> 1) * (BT / 2), wn = (wave & 1) * (BT / 2);
  const rsrc_t ra = mk_rsrc(g.A), rb = mk_rsrc(g.B);
  f32x4 ar[NI], br[NI];
  int kbeg = 0, kend = g.K;
  if constexpr (SPLIT) {
    const int per = (((g.K + g.splitk - 1) / g.splitk) + KC - 1) / KC * KC;
    kbeg = min(g.K, bz * per);
    kend = min(g.K, kbeg + per);
    g.C += (size_t)bz * g.part_stride;
  }
  f32x4 rsum = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_sum = SPLIT && g.asum && bx == 0;
  auto gload = [&](int k0) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      if (AK) {
        const int gm = m0 + (tid >> 4) + 16 * it, gk = k0 + 4 * (tid & 15);
        ar[it] = bl4(ra, (gm < g.M && gk < g.K) ? (unsigned)(gm * (int)g.sam + gk) * 4u : OOB);
      } else {
        const int gk = k0 + tid / R4 + (256 / R4) * it, gm = m0 + 4 * (tid % R4);
        ar[it] = bl4(ra, (gm < g.M && gk < kend) ? (unsigned)(gk * (int)g.sak + gm) * 4u : OOB);
      }
      if (BK) {
        const int gn = n0 + (tid >> 4) + 16 * it, gk = k0 + 4 * (tid & 15);
        br[it] = bl4(rb, (gn < g.N && gk < g.K) ? (unsigned)(gn * (int)g.sbn + gk) * 4u : OOB);
      } else {
        const int gk = k0 + tid / R4 + (256 / R4) * it, gn = n0 + 4 * (tid % R4);
        br[it] = bl4(rb, (gn < g.N && gk < kend) ? (unsigned)(gk * (int)g.sbk + gn) * 4u : OOB);
      }
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      if (AK) *reinterpret_cast<f32x4 *>(As + buf * ASZ + ((tid >> 4) + 16 * it) * LDK + 4 * (tid & 15)) = ar[it];
      else *reinterpret_cast<f32x4 *>(As + buf * ASZ + (tid / R4 + (256 / R4) * it) * LDR + 4 * (tid % R4)) = ar[it];
      if constexpr (SPLIT && !AK) rsum += ar[it];       // out-of-range elements arrived as zeros
      if (BK) *reinterpret_cast<f32x4 *>(Bs + buf * BSZ + ((tid >> 4) + 16 * it) * LDK + 4 * (tid & 15)) = br[it];
      else *reinterpret_cast<f32x4 *>(Bs + buf * BSZ + (tid / R4 + (256 / R4) * it) * LDR + 4 * (tid % R4)) = br[it];
    }
  };
  f32x4 acc[WT][WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nc = (kend - kbeg + KC - 1) / KC;
  gload(kbeg);
  sstore(0);
  __syncthreads();
  for (int c = 0; c < nc; ++c) {
    if (c + 1 < nc) gload(kbeg + (c + 1) * KC);
    const float *A = As + (c & 1) * ASZ, *B = Bs + (c & 1) * BSZ;
    const int kkmax = min(4, (kend - kbeg - c * KC + 15) >> 4);
    for (int kk = 0; kk < kkmax; ++kk) {
      float a[WT][4], b[WT][4];
#pragma unroll
      for (int i = 0; i < WT; ++i) {
        if (AK) {
          const f32x4 v = *reinterpret_cast<const f32x4 *>(A + (wm + 16 * i + (lane & 15)) * LDK + 16 * kk + 4 * (lane >> 4));
#pragma unroll
          for (int e = 0; e < 4; ++e) a[i][e] = v[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) a[i][e] = A[(16 * kk + 4 * (lane >> 4) + e) * LDR + wm + 16 * i + (lane & 15)];
        }
        if (BK) {
          const f32x4 v = *reinterpret_cast<const f32x4 *>(B + (wn + 16 * i + (lane & 15)) * LDK + 16 * kk + 4 * (lane >> 4));
#pragma unroll
          for (int e = 0; e < 4; ++e) b[i][e] = v[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) b[i][e] = B[(16 * kk + 4 * (lane >> 4) + e) * LDR + wn + 16 * i + (lane & 15)];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
          for (int j = 0; j < WT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
    if (c + 1 < nc) sstore((c + 1) & 1);
    __syncthreads();
  }
  // Epilogue: the operands of ALL of the thread's outputs are requested first (branch-free, out-of-range offsets for absent ones),
  // then applied.  With a load per output under the bounds branches the compiler waited for each one in turn: up to 4 dependent
  // round trips per output, 16 outputs per thread - most of the run time of the all-steps GEMMs of a training step.
  // (offsets fit 32 bits: the dispatcher sends only operands below 5e8 floats here)
  EpiPre pre[WT][WT][4];
  float oldc[WT][WT][4];
  const rsrc_t rc = mk_rsrc(g.C);
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      const int n = n0 + wn + 16 * j + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + 16 * i + 4 * (lane >> 4) + r;
        const bool ok = n < g.N && m < g.M;
        pre[i][j][r] = gemm_epi_load(g, m, n, ok);
        oldc[i][j][r] = bl1(rc, (ok && g.accumulate) ? (unsigned)(m * (int)g.ldc + n) * 4u : OOB);
      }
    }
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      const int n = n0 + wn + 16 * j + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + 16 * i + 4 * (lane >> 4) + r;
        if (n < g.N && m < g.M) {
          const float v = gemm_epi_pre(g, acc[i][j][r], pre[i][j][r]);
          g.C[(size_t)m * g.ldc + n] = g.accumulate ? (oldc[i][j][r] + v) : v;
        }
      }
    }
  if constexpr (SPLIT && !AK) {
    if (do_sum) {      // thread (tid % R4) holds rows 4*(tid%R4)..+3 of the tile summed over its k rows: 256/R4 partials each
      __syncthreads();
      float *ps = smem;                         // [256 / R4][BT]
      *reinterpret_cast<f32x4 *>(ps + (tid / R4) * BT + 4 * (tid % R4)) = rsum;
      __syncthreads();
      if (tid < BT && m0 + tid < g.M) {
        float t = 0.f;
        for (int r = 0; r < 256 / R4; ++r) t += ps[r * BT + tid];
        g.asum[(size_t)bz * g.part_stride + m0 + tid] = t;
      }
    }
  }
}

template <int BT, bool AK, bool BK, bool SPLIT = false>
__global__ __launch_bounds__(256) void mmt_gemm_panel_kernel(mmt_gemm_args g) {
  extern __shared__ __align__(16) float smem[];
  gemm_panel_body<BT, AK, BK, SPLIT>(g, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// Group launch: workgroup i works on tile map[i] = (job, bx, by, bz) of the split-K weight-gradient GEMM jobs[job] (same body, same
// split, same sums as that job's own launch would give).  The deferred weight gradients of a training step are ~70 independent
// GEMMs of 64..1536 tiles each: one launch of all their tiles fills the chip where 70 launches each leave most of it idle.
struct GroupBlock {
  int32_t job, bx, by, bz;
};
template <int BT>
__global__ __launch_bounds__(256) void mmt_gemm_group_kernel(const mmt_gemm_args *__restrict__ jobs, const GroupBlock *__restrict__ map) {
  extern __shared__ __align__(16) float smem[];
  const GroupBlock m = map[blockIdx.x];
  const int j = __builtin_amdgcn_readfirstlane(m.job);
  gemm_panel_body<BT, false, false, true>(jobs[j], __builtin_amdgcn_readfirstlane(m.bx), __builtin_amdgcn_readfirstlane(m.by),
                                          __builtin_amdgcn_readfirstlane(m.bz), smem);
}

template <int BT, bool AK, bool BK>
void launch_panel(const mmt_gemm_args &g, hipStream_t st) {
  constexpr int KC = 64, LDK = KC + 4, LDR = BT + 4;
  constexpr size_t lds = 2 * ((AK ? BT * LDK : KC * LDR) + (BK ? BT * LDK : KC * LDR)) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)mmt_gemm_panel_kernel<BT, AK, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((mmt_gemm_panel_kernel<BT, AK, BK>), dim3((g.N + BT - 1) / BT, (g.M + BT - 1) / BT), dim3(256), lds, st, g);
}

void launch_split(const mmt_gemm_args &g, hipStream_t st) {
  constexpr int BT = 32, KC = 64, LDR = BT + 4;
  constexpr size_t lds = 2 * (KC * LDR + KC * LDR) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)mmt_gemm_panel_kernel<BT, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((mmt_gemm_panel_kernel<BT, false, false, true>), dim3((g.N + BT - 1) / BT, (g.M + BT - 1) / BT, std::max(g.splitk, 1)),
                     dim3(256), lds, st, g);
}

// ------------------------------------------------------------------------------------------ row-block GEMM (small M)
// The backward sweep is a chain of ~70 dependent GEMMs per decoder step with M = B rows: latency, not throughput.  A
// workgroup owns a 16 x 32 output tile and the WHOLE K range (K <= 768): every global load of the kernel is requested
// in one burst straight into MFMA operand layout (no LDS staging), the K range is dealt over the 4 waves in 16-wide
// chunks (chunk c -> wave c % 4) and the 4 partial tiles are summed through LDS in wave order.  M/16 x N/32
// workgroups: 4x the panel kernel's at the same shape.
// LN = true: the A operand is the gradient at a LayerNorm output; its LayerNorm backward (row statistics over the K
// columns: the workgroup holds whole rows) is applied in registers before the multiply, and the workgroups of the
// first column block also store it (ln_dx) for the residual path and the deferred weight-gradient GEMMs.
// lane j of every 16-lane row, to all lanes of that row (j a compile-time constant after unrolling)
__device__ __forceinline__ float row16_bcast(float v, int j) {
  const int x = __builtin_bit_cast(int, v);
  int r;
  switch (j & 15) {
#define MMT_BC(J) case J: r = __builtin_amdgcn_update_dpp(0, x, 0x150 + J, 0xF, 0xF, false); break;
    MMT_BC(0) MMT_BC(1) MMT_BC(2) MMT_BC(3) MMT_BC(4) MMT_BC(5) MMT_BC(6) MMT_BC(7)
    MMT_BC(8) MMT_BC(9) MMT_BC(10) MMT_BC(11) MMT_BC(12) MMT_BC(13) MMT_BC(14)
#undef MMT_BC
    default: r = __builtin_amdgcn_update_dpp(0, x, 0x15F, 0xF, 0xF, false); break;
  }
  return __builtin_bit_cast(float, r);
}
template <int DPL>
__device__ __forceinline__ void ld_dpl(float (&dst)[DPL], rsrc_t r, unsigned off) {
  if constexpr (DPL == 2) {        // one 8-byte load (the callers' offsets are even numbers of floats from 16-byte aligned bases)
    // (bit_cast of whatever 8-byte type the builtin returns: assigning it to a 2 x u32 vector compiles as a splat of its low
    // half, and the load shrinks to one dword)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    dst[0] = v[0];
    dst[1] = v[1];
  } else {
#pragma unroll
    for (int e = 0; e < DPL; ++e) dst[e] = bl1(r, off == OOB ? OOB : off + 4u * e);
  }
}
// ATT = true (head dim 32, one query per agent: the decoder's per-step sweep): the 16 x 32 tile is the gradient at the output of
// head blockIdx.x of an attention for 16 agents, and is consumed in place by that attention's backward (dq =, dK / dV +=, as
// mmt_attention_bwd with Lq = 1) instead of being written out: 16 lanes per agent, the K / dK / dV rows requested together with
// the GEMM operands.  out_proj's input gradient never goes to memory.
struct RowsAttn {
  mmt_attn_args a;
  float *dq, *dk, *dv;
  long dq_bs, dk_bs, dk_rs;
};
// NT: 16-column tiles per workgroup (2 = the 16 x 32 tile above; 1 = 16 x 16: twice the workgroups, half the B operand loads
// and MFMAs per wave; at K > 256 a wave of the 16 x 32 form has more loads in flight than the 63 its counter can track)
template <int NCH, bool BK, bool LN, bool ATT = false, int NT = 2>
__global__ __launch_bounds__(256) void mmt_gemm_rows_kernel(mmt_gemm_args g, RowsAttn att) {
  static_assert(!ATT || NT == 2, "a head is 32 columns wide");
  __shared__ float red[4][NT][16 * 17];
  __shared__ float stat[4][16][2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16 * NT, nch = g.K >> 4;
  const int m = m0 + l15;
  const bool mok = m < g.M;
  const rsrc_t ra = mk_rsrc(g.A), rb = mk_rsrc(g.B);
  f32x4 a[NCH], b[NT][NCH];
  f32x4 xh[LN ? NCH : 1], gg[LN ? NCH : 1];
  float rstd = 0.f;               // (first of all loads: everything behind the row statistics is scaled by it)
  if constexpr (LN) rstd = bl1(mk_rsrc(g.ln_rstd), mok ? (unsigned)m * 4u : OOB);
  uint32_t kbs[LN ? NCH : 1];     // keep bytes of ln_drop: requested with everything else (a load behind the row statistics
                                  // would put a memory round trip into the middle of every kernel of the sweep)
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = 4 * i + w, kk = 16 * c + 4 * kq;
    const bool cok = c < nch;
    a[i] = bl4(ra, (cok && mok) ? (unsigned)(m * (int)g.sam + kk) * 4u : OOB);
    if constexpr (LN) {
      xh[i] = bl4(mk_rsrc(g.ln_xhat), (cok && mok) ? (unsigned)(m * (int)g.ld_ln + kk) * 4u : OOB);
      gg[i] = bl4(mk_rsrc(g.ln_g), cok ? (unsigned)kk * 4u : OOB);
      kbs[i] = __builtin_amdgcn_raw_buffer_load_b32(mk_rsrc(g.ln_drop), (g.ln_drop && cok && mok) ? (unsigned)(m * (int)g.ld_ln_drop + kk) : OOB, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = n0 + 16 * t + l15;
      const bool ok = cok && n < g.N;
      if constexpr (BK) {
        b[t][i] = bl4(rb, ok ? (unsigned)(n * (int)g.sbn + kk) * 4u : OOB);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) b[t][i][e] = bl1(rb, ok ? (unsigned)((kk + e) * (int)g.sbk + n) * 4u : OOB);
      }
    }
  }
  EpiPre pre[ATT ? 1 : NT];
  if constexpr (!ATT) {
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const int mm = m0 + (tid >> 4), n = n0 + 16 * it + (tid & 15);
      pre[it] = gemm_epi_load(g, mm, n, mm < g.M && n < g.N);
    }
  }
  // (the loads are consumed in the order LayerNorm operands -> GEMM operands -> attention operands and return in issue order:
  // keep the compiler from hoisting the ~110 attention loads in front of the operands the first stage waits for)
  __builtin_amdgcn_sched_barrier(0);
  // attention operands of (agent m0 + tid/16, head blockIdx.x), lane l16 = tid % 16
  const int ab = m0 + (tid >> 4), l16 = tid & 15;
  float kr[ATT ? 16 : 1][2], dkr[ATT ? 16 : 1][2], dvr[ATT ? 16 : 1][2], qd[2] = {0.f, 0.f}, pk = 0.f;
  uint32_t pkeep_b = 0;      // keep byte of the probability this lane owns (a.pdrop)
  f32x4 vrow[ATT ? 8 : 1];
  int Lk = 0;
  if constexpr (ATT) {
    const mmt_attn_args &a = att.a;
    const int h = blockIdx.x;
    Lk = ab < g.M ? a.Lk : 0;
    const rsrc_t rk = mk_rsrc(a.k), rv = mk_rsrc(a.v), rdk = mk_rsrc(att.dk), rdv = mk_rsrc(att.dv);
    const unsigned kb = (unsigned)(ab * (int)a.k_bs + h * 32 + l16 * 2), vb = (unsigned)(ab * (int)a.v_bs + h * 32);
    const unsigned db = (unsigned)(ab * (int)att.dk_bs + h * 32 + l16 * 2);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      ld_dpl<2>(kr[j], rk, j < Lk ? (kb + (unsigned)(j * (int)a.k_rs)) * 4u : OOB);
      ld_dpl<2>(dkr[j], rdk, j < Lk ? (db + (unsigned)(j * (int)att.dk_rs)) * 4u : OOB);
      ld_dpl<2>(dvr[j], rdv, j < Lk ? (db + (unsigned)(j * (int)att.dk_rs)) * 4u : OOB);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) vrow[c] = bl4(rv, l16 < Lk ? (vb + (unsigned)(l16 * (int)a.v_rs + 4 * c)) * 4u : OOB);
    ld_dpl<2>(qd, mk_rsrc(a.q), Lk > 0 ? (unsigned)(ab * (int)a.q_bs + h * 32 + l16 * 2) * 4u : OOB);
    pk = bl1(mk_rsrc(a.probs), l16 < Lk ? (unsigned)((ab * a.H + h) * (int)a.p_hs + l16) * 4u : OOB);
    pkeep_b = __builtin_amdgcn_raw_buffer_load_b8(mk_rsrc(a.pdrop), (a.pdrop && l16 < Lk) ? (unsigned)((ab * a.H + h) * (int)a.p_hs + l16) : OOB, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (LN) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dg = a[i][e] * gg[i][e];
        a[i][e] = dg;
        s1 += dg;
        s2 = fmaf(dg, xh[i][e], s2);
      }
    s1 = xor32_sum(xor16_sum(s1));           // over the four k sub-columns of the row inside this wave
    s2 = xor32_sum(xor16_sum(s2));
    if (kq == 0) {
      stat[w][l15][0] = s1;
      stat[w][l15][1] = s2;
    }
    __syncthreads();
    const float inv = 1.0f / (float)g.K;
    const float m1 = (((stat[0][l15][0] + stat[1][l15][0]) + stat[2][l15][0]) + stat[3][l15][0]) * inv;
    const float m2 = (((stat[0][l15][1] + stat[1][l15][1]) + stat[2][l15][1]) + stat[3][l15][1]) * inv;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) a[i][e] = rstd * (a[i][e] - m1 - xh[i][e] * m2);
      const int c = 4 * i + w;
      if (blockIdx.x == 0 && g.ln_dx && c < nch && mok)
        *reinterpret_cast<f32x4 *>(g.ln_dx + (size_t)m * g.ld_ln + 16 * c + 4 * kq) = a[i];
      if (g.ln_drop) {     // the sub-layer's output was dropped before the residual add: its gradient is keep * A'
        const uint32_t kb = kbs[i];          // (lanes beyond the rows / chunks loaded nothing: zeros)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[i][e] *= keep_of(kb, e, g.ln_drop_scale);
        if (blockIdx.x == 0 && g.ln_dxm && c < nch && mok)
          *reinterpret_cast<f32x4 *>(g.ln_dxm + (size_t)m * g.ld_ln + 16 * c + 4 * kq) = a[i];
      }
    }
  }
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (4 * i + w < nch) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[t][i][e], acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][t][(4 * kq + r) * 17 + l15] = acc[t][r];
  __syncthreads();
  if constexpr (ATT) {
    __shared__ float ct[16][33];                  // dO of head blockIdx.x for the 16 agents
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int o = tid + 256 * it, tt = o >> 8, row = (o & 255) >> 4, col = o & 15, ix = row * 17 + col;
      ct[row][16 * tt + col] = ((red[0][tt][ix] + red[1][tt][ix]) + red[2][tt][ix]) + red[3][tt][ix];
    }
    __syncthreads();
    const mmt_attn_args &a = att.a;
    const float pkeep = a.pdrop ? ((pkeep_b & 0xffu) ? a.pdrop_scale : 0.f) : 1.f;
    const float *grow = ct[tid >> 4];
    float dp = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      dp = fmaf(grow[4 * c + 0], vrow[c][0], dp);
      dp = fmaf(grow[4 * c + 1], vrow[c][1], dp);
      dp = fmaf(grow[4 * c + 2], vrow[c][2], dp);
      dp = fmaf(grow[4 * c + 3], vrow[c][3], dp);
    }
    const float dP = pk > 0.f ? dp * pkeep : 0.f;     // (pkeep = 1 without dropout on the probabilities)
    const float dot = row16_sum(pk * dP);
    const float ds = pk * (dP - dot) * a.scale;
    const float gd[2] = {grow[l16 * 2], grow[l16 * 2 + 1]};
    float dqa[2] = {0.f, 0.f};
    const float pkd = pk * pkeep;
#pragma unroll
    for (int j = 0; j < 16; ++j) {          // masked / absent keys carry p = ds = 0: exact no-ops
      const float pj = row16_bcast(pkd, j), dsj = row16_bcast(ds, j);      // (DPP row_newbcast: no trip through the LDS crossbar)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        dqa[e] = fmaf(dsj, kr[j][e], dqa[e]);
        dkr[j][e] = fmaf(dsj, qd[e], dkr[j][e]);
        dvr[j][e] = fmaf(pj, gd[e], dvr[j][e]);
      }
    }
    if (ab < g.M) {
      const int h = blockIdx.x;
      float *dQ = att.dq + (size_t)ab * att.dq_bs + h * 32 + l16 * 2;
      dQ[0] = dqa[0];
      dQ[1] = dqa[1];
      float *dK = att.dk + (size_t)ab * att.dk_bs + h * 32 + l16 * 2, *dV = att.dv + (size_t)ab * att.dk_bs + h * 32 + l16 * 2;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j < Lk) {
          dK[(size_t)j * att.dk_rs] = dkr[j][0];
          dK[(size_t)j * att.dk_rs + 1] = dkr[j][1];
          dV[(size_t)j * att.dk_rs] = dvr[j][0];
          dV[(size_t)j * att.dk_rs + 1] = dvr[j][1];
        }
      }
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    const int o = tid + 256 * it, t = o >> 8, row = (o & 255) >> 4, col = o & 15;
    const int mm = m0 + row, n = n0 + 16 * t + col;
    if (mm >= g.M || n >= g.N) continue;
    const int ix = row * 17 + col;
    const float v = gemm_epi_pre(g, ((red[0][t][ix] + red[1][t][ix]) + red[2][t][ix]) + red[3][t][ix], pre[ATT ? 0 : it]);
    float *cp = g.C + (size_t)mm * g.ldc + n;
    *cp = g.accumulate ? (*cp + v) : v;
  }
}

template <bool BK, bool LN>
void launch_rows(const mmt_gemm_args &g, hipStream_t st) {
  const dim3 grid((g.N + 31) / 32, (g.M + 15) / 16), grid1((g.N + 15) / 16, (g.M + 15) / 16), blk(256);
  const RowsAttn none = {};
  // 16 x 16 tiles where that still fits the chip in one round (the sweep's GEMMs at <= 256 rows: 4.06 against 5.64 us at K = 768,
  // 3.19 / 4.27 at K = 512, 2.35 / 3.05 at K = 256, back to back in a graph)
  static const int nt1_mode = getenv("MMT_ROWS_NT1") ? atoi(getenv("MMT_ROWS_NT1")) : 2;      // 0: never, 1: K > 256, 2: every K (measured best)
  const bool nt1 = (nt1_mode == 2 || (nt1_mode == 1 && !LN && g.K > 256)) && (long)grid1.x * grid1.y <= 256;
  if (g.K <= 256) {
    if (nt1) hipLaunchKernelGGL((mmt_gemm_rows_kernel<4, BK, LN, false, 1>), grid1, blk, 0, st, g, none);
    else hipLaunchKernelGGL((mmt_gemm_rows_kernel<4, BK, LN>), grid, blk, 0, st, g, none);
  } else if (g.K <= 512) {
    if (nt1) hipLaunchKernelGGL((mmt_gemm_rows_kernel<8, BK, LN, false, 1>), grid1, blk, 0, st, g, none);
    else hipLaunchKernelGGL((mmt_gemm_rows_kernel<8, BK, LN>), grid, blk, 0, st, g, none);
  } else {
    if (nt1) hipLaunchKernelGGL((mmt_gemm_rows_kernel<12, BK, LN, false, 1>), grid1, blk, 0, st, g, none);
    else hipLaunchKernelGGL((mmt_gemm_rows_kernel<12, BK, LN>), grid, blk, 0, st, g, none);
  }
}

// ------------------------------------------------------------------------------------------ thin GEMM
// A handful of outputs (M*N <= 2048) of a long reduction (weight gradients of the 2-column linears: K = ph*B rows):
// one wave per output element, lanes stride the K range, fixed-order wave sum.  Any strides.
__global__ __launch_bounds__(256) void mmt_gemm_thin_kernel(mmt_gemm_args g) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= g.M * g.N) return;
  const int m = o / g.N, n = o - m * g.N;
  const float *A = g.A + (size_t)m * g.sam, *B = g.B + (size_t)n * g.sbn;
  // epilogue operands and the old output first, the reduction's loads eight per lane at a time (a rolled loop waits for every
  // load before it issues the next, and the epilogue's loads came behind the wave sum: six dependent round trips at K = 256)
  const EpiPre pre = gemm_epi_load(g, m, n, lane == 0);
  const float old = (lane == 0 && g.accumulate) ? g.C[(size_t)m * g.ldc + n] : 0.f;
  float s = 0.f;
  for (int k0 = lane; k0 < g.K; k0 += 512) {
    float av[8], bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = k0 + 64 * i;
      const bool in = k < g.K;
      av[i] = in ? A[(size_t)k * g.sak] : 0.f;
      bv[i] = in ? B[(size_t)k * g.sbk] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s = (k0 + 64 * i < g.K) ? fmaf(av[i], bv[i], s) : s;
  }
  s = wave64_sum(s);
  if (lane == 0) {
    const float v = gemm_epi_pre(g, s, pre);
    g.C[(size_t)m * g.ldc + n] = g.accumulate ? (old + v) : v;
  }
}

// ------------------------------------------------------------------------------------------ GEMM with a handful of k (K <= 8)
// The head's last Linear has 2 outputs: its input gradient is a K = 2 product per step of the sweep (12 dependent launches per
// training step, 6.8 us each on the generic tile kernel below).  One output per thread, every operand requested at once, the
// same k-ordered chain of fused multiply-adds the matrix-core kernels run.
__global__ __launch_bounds__(256) void mmt_gemm_smallk_kernel(mmt_gemm_args g) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  const bool ok = o < g.M * g.N;
  const int m = ok ? o / g.N : 0, n = ok ? o - m * g.N : 0;
  const rsrc_t ra = mk_rsrc(g.A), rb = mk_rsrc(g.B);
  float a[8], b[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = bl1(ra, (ok && k < g.K) ? (unsigned)(m * (int)g.sam + k * (int)g.sak) * 4u : OOB);
    b[k] = bl1(rb, (ok && k < g.K) ? (unsigned)(k * (int)g.sbk + n * (int)g.sbn) * 4u : OOB);
  }
  const EpiPre pre = gemm_epi_load(g, m, n, ok);
  const float old = bl1(mk_rsrc(g.C), (ok && g.accumulate) ? (unsigned)(m * (int)g.ldc + n) * 4u : OOB);
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc = k < g.K ? fmaf(a[k], b[k], acc) : acc;
  if (ok) {
    const float v = gemm_epi_pre(g, acc, pre);
    g.C[(size_t)m * g.ldc + n] = g.accumulate ? old + v : v;
  }
}

// ------------------------------------------------------------------------------------------ GEMM
constexpr int GBM = 64, GBN = 64, GBK = 16, GLD = GBK + 4;   // LDS row stride 20 floats (16-B aligned rows)

__global__ __launch_bounds__(256) void mmt_gemm_kernel(mmt_gemm_args g) {
  __shared__ __align__(16) float As[GBM * GLD];
  __shared__ __align__(16) float Bs[GBN * GLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;     // each wave: 32 x 32 = 2 x 2 MFMA tiles
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < g.K; k0 += GBK) {
    // stage A[m0:m0+64, k0:k0+16] and B[k0:k0+16, n0:n0+64] (as Bs[n][k]); pick the thread->element map that
    // walks the unit-stride dimension of each operand
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m, k;
      if (g.sak == 1) {
        k = tid & 15;
        m = (tid >> 4) + 16 * i;
      } else {
        m = tid & 63;
        k = (tid >> 6) + 4 * i;
      }
      const int gm = m0 + m, gk = k0 + k;
      As[m * GLD + k] = (gm < g.M && gk < g.K) ? g.A[(size_t)gm * g.sam + (size_t)gk * g.sak] : 0.f;
      int n, kb;
      if (g.sbk == 1) {
        kb = tid & 15;
        n = (tid >> 4) + 16 * i;
      } else {
        n = tid & 63;
        kb = (tid >> 6) + 4 * i;
      }
      const int gn = n0 + n, gkb = k0 + kb;
      Bs[n * GLD + kb] = (gn < g.N && gkb < g.K) ? g.B[(size_t)gkb * g.sbk + (size_t)gn * g.sbn] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 16) {
      f32x4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const f32x4 *>(&As[(wm + 16 * i + (lane & 15)) * GLD + kk + 4 * (lane >> 4)]);
        b[i] = *reinterpret_cast<const f32x4 *>(&Bs[(wn + 16 * i + (lane & 15)) * GLD + kk + 4 * (lane >> 4)]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn + 16 * j + (lane & 15);
      if (n >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + 16 * i + 4 * (lane >> 4) + r;
        if (m >= g.M) continue;
        const float v = gemm_epi(g, acc[i][j][r], m, n);
        float *c = g.C + (size_t)m * g.ldc + n;
        *c = g.accumulate ? (*c + v) : v;
      }
    }
}

// LayerNorm parameter gradients in one pass: gw[n] = sum_m dy[m][n] * xhat[m][n], gb[n] = sum_m dy[m][n].  The rows are
// dealt over gridDim.y slices (slice z -> out + z*part_stride, summed by the caller like the split weight gradients).
// (bx, bz of nz) = column block and row slice: blockIdx / gridDim.y of a single-job launch, or a group launch's block map
__device__ __forceinline__ void ln_param_grads_body(const float *__restrict__ dy, const float *__restrict__ xhat, int M, int N, long ld,
                                                    float *__restrict__ gw, float *__restrict__ gb, long part_stride, const int bx,
                                                    const int bz, const int nz) {
  __shared__ float pw[64][17], pb[64][17];
  __shared__ float qw[4][16], qb[4][16];
  const int cl = threadIdx.x & 15, c = bx * 16 + cl, rl = threadIdx.x >> 4;
  const int per = (M + nz - 1) / nz, mb = bz * per, me = min(M, mb + per);
  float sw = 0.f, sb = 0.f;
  if (c < N) {
#pragma unroll 4
    for (int m = mb + rl; m < me; m += 64) {
      const float d = dy[(size_t)m * ld + c];
      sw = fmaf(d, xhat[(size_t)m * ld + c], sw);
      sb += d;
    }
  }
  pw[rl][cl] = sw;
  pb[rl][cl] = sb;
  __syncthreads();
  if (rl < 4) {
    float tw = 0.f, tb = 0.f;
    for (int r = 0; r < 16; ++r) {
      tw += pw[rl * 16 + r][cl];
      tb += pb[rl * 16 + r][cl];
    }
    qw[rl][cl] = tw;
    qb[rl][cl] = tb;
  }
  __syncthreads();
  if (rl == 0 && c < N) {
    gw[(size_t)bz * part_stride + c] = (qw[0][cl] + qw[1][cl]) + (qw[2][cl] + qw[3][cl]);
    gb[(size_t)bz * part_stride + c] = (qb[0][cl] + qb[1][cl]) + (qb[2][cl] + qb[3][cl]);
  }
}
__global__ __launch_bounds__(1024) void mmt_ln_param_grads_kernel(const float *__restrict__ dy, const float *__restrict__ xhat, int M,
                                                                  int N, long ld, float *__restrict__ gw, float *__restrict__ gb,
                                                                  long part_stride) {
  ln_param_grads_body(dy, xhat, M, N, ld, gw, gb, part_stride, blockIdx.x, blockIdx.y, gridDim.y);
}
// group launch (as mmt_gemm_group_kernel): workgroup i reduces block map[i] = (job, bx, -, bz) of jobs[job]
__global__ __launch_bounds__(1024) void mmt_ln_grad_group_kernel(const mmt_ln_grad_job *__restrict__ jobs, const GroupBlock *__restrict__ map) {
  const GroupBlock m = map[blockIdx.x];
  const mmt_ln_grad_job J = jobs[__builtin_amdgcn_readfirstlane(m.job)];
  ln_param_grads_body(J.dy, J.xhat, J.M, J.N, J.ld, J.gw, J.gb, J.part_stride, __builtin_amdgcn_readfirstlane(m.bx),
                      __builtin_amdgcn_readfirstlane(m.bz), J.splitk);
}

// out[n] (+)= sum_m a[m][n] * (b ? b[m][n] : 1) [only where mask[m][n] > 0]; deterministic order.
// 16 columns x 64 row-lanes per workgroup: N/16 workgroups, M/64 independent loads per thread.
__global__ __launch_bounds__(1024) void mmt_colsum_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                          const float *__restrict__ mask, int M, int N, long lda,
                                                          long ldb, long ldm, float *__restrict__ out, int accumulate) {
  __shared__ float part[64][17];
  __shared__ float part2[4][16];
  const int cl = threadIdx.x & 15, c = blockIdx.x * 16 + cl, rl = threadIdx.x >> 4;
  float s = 0.f;
  if (c < N) {
#pragma unroll 4
    for (int m = rl; m < M; m += 64) {
      float v = a[(size_t)m * lda + c];
      if (b) v *= b[(size_t)m * ldb + c];
      if (mask && !(mask[(size_t)m * ldm + c] > 0.f)) v = 0.f;
      s += v;
    }
  }
  part[rl][cl] = s;
  __syncthreads();
  if (rl < 4) {           // 64 partials per column: 4 lanes x 16 in order, then the 4 lane sums in order
    float t = 0.f;
    for (int r = 0; r < 16; ++r) t += part[rl * 16 + r][cl];
    part2[rl][cl] = t;
  }
  __syncthreads();
  if (rl == 0 && c < N) {
    const float t = (part2[0][cl] + part2[1][cl]) + (part2[2][cl] + part2[3][cl]);
    out[c] = accumulate ? out[c] + t : t;
  }
}

// ------------------------------------------------------------------------------------------ LayerNorm
// y = [relu](LN(x + res) * g + b); stashes xhat (M,N) and rstd (M).  One wave per row.
__global__ __launch_bounds__(256) void mmt_ln_fwd_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                         int M, int N, const float *__restrict__ g,
                                                         const float *__restrict__ b, int relu, float *__restrict__ y,
                                                         float *__restrict__ xhat, float *__restrict__ rstd_out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float *xr = x + (size_t)row * N, *rr = res ? res + (size_t)row * N : nullptr;
  float s = 0.f;
  for (int c = lane; c < N; c += 64) s += xr[c] + (rr ? rr[c] : 0.f);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mu = s / (float)N;
  float v = 0.f;
  for (int c = lane; c < N; c += 64) {
    const float d = xr[c] + (rr ? rr[c] : 0.f) - mu;
    v += d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const float rstd = 1.0f / sqrtf(v / (float)N + 1e-5f);
  for (int c = lane; c < N; c += 64) {
    const float h = (xr[c] + (rr ? rr[c] : 0.f) - mu) * rstd;
    float o = h * g[c] + b[c];
    if (relu) o = fmaxf(o, 0.f);
    y[(size_t)row * N + c] = o;
    xhat[(size_t)row * N + c] = h;
  }
  if (lane == 0) rstd_out[row] = rstd;
}

// The same two kernels with the row in registers (N = 64 * NPL columns): every load of a row is requested at once and the row is
// read once instead of three times (the generic forms above walk it per pass: three dependent memory round trips in a kernel
// that 31 + 24 launches of a training step sit behind).  Same element-to-lane assignment, same order of additions: same bits.
// Contraction is switched off in both and every fused multiply-add is written out, in the places where the compiler fuses in the
// generic forms: a contraction the compiler is free to pick differently here (it did: d0*d0 + d1*d1 fused the other way round, the
// constant 1/N folded into the subtraction) moves a result by an ulp, and the full-size gradient test against the reference's
// autograd sits on exactly those ulps after 12 feedback steps.
template <int NPL>
__global__ __launch_bounds__(256) void mmt_ln_fwd_regs_kernel(const float *__restrict__ x, const float *__restrict__ res, int M,
                                                              const float *__restrict__ g, const float *__restrict__ b, int relu,
                                                              float *__restrict__ y, float *__restrict__ xhat,
                                                              float *__restrict__ rstd_out) {
#pragma clang fp contract(off)
  constexpr int N = 64 * NPL;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const size_t o = (size_t)row * N;
  float v[NPL], gg[NPL], bb[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) v[i] = x[o + lane + 64 * i];
  if (res) {
#pragma unroll
    for (int i = 0; i < NPL; ++i) v[i] += res[o + lane + 64 * i];
  }
#pragma unroll
  for (int i = 0; i < NPL; ++i) gg[i] = g[lane + 64 * i], bb[i] = b[lane + 64 * i];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) s += v[i];
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) s += __shfl_xor(s, k, 64);
  const float mu = s * (1.0f / (float)N);        // (N is a power of two: the quotient, exactly)
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const float d = v[i] - mu;
    q = fmaf(d, d, q);
  }
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) q += __shfl_xor(q, k, 64);
  const float rstd = 1.0f / sqrtf(q * (1.0f / (float)N) + 1e-5f);
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const float h = (v[i] - mu) * rstd;
    float out = fmaf(h, gg[i], bb[i]);
    if (relu) out = fmaxf(out, 0.f);
    y[o + lane + 64 * i] = out;
    xhat[o + lane + 64 * i] = h;
  }
  if (lane == 0) rstd_out[row] = rstd;
}
// HALF: N = 32 (the head's hidden width): lanes 0..31 hold the row, the others zeros (as the generic form's idle lanes do)
template <int NPL, bool HALF = false>
__global__ __launch_bounds__(256) void mmt_ln_bwd_regs_kernel(const float *__restrict__ dy, const float *__restrict__ xhat,
                                                              const float *__restrict__ rstd, const float *__restrict__ g,
                                                              const float *__restrict__ y_relu, int M, float *__restrict__ dx,
                                                              int accumulate, float *__restrict__ dy_masked,
                                                              const uint8_t *__restrict__ drop, float drop_scale,
                                                              float *__restrict__ dxm) {
#pragma clang fp contract(off)
  static_assert(!HALF || NPL == 1, "HALF is the 32-column form");
  constexpr int N = HALF ? 32 : 64 * NPL;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const size_t o = (size_t)row * N;
  float d[NPL], xh[NPL], gg[NPL], old[NPL];
  uint8_t kp[NPL];
  const float rs = rstd[row];
  const bool act = !HALF || lane < 32;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const size_t c = o + lane + 64 * i;
    d[i] = act ? dy[c] : 0.f;
    xh[i] = act ? xhat[c] : 0.f;
    gg[i] = act ? g[lane + 64 * i] : 0.f;
    old[i] = (act && accumulate) ? dx[c] : 0.f;
    kp[i] = (act && drop) ? drop[c] : (uint8_t)1;
  }
  if (y_relu) {
#pragma unroll
    for (int i = 0; i < NPL; ++i)
      if (act && !(y_relu[o + lane + 64 * i] > 0.f)) d[i] = 0.f;
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    if (dy_masked && act) dy_masked[o + lane + 64 * i] = d[i];
    const float dg = d[i] * gg[i];
    s1 += dg;
    s2 += dg * xh[i];
  }
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) {
    s1 += __shfl_xor(s1, k, 64);
    s2 += __shfl_xor(s2, k, 64);
  }
  const float m1 = s1 * (1.0f / (float)N), m2 = s2 * (1.0f / (float)N);
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const size_t c = o + lane + 64 * i;
    const float v = rs * fmaf(-xh[i], m2, fmaf(d[i], gg[i], -m1));
    if (act) {
      dx[c] = accumulate ? old[i] + v : v;
      if (drop) dxm[c] = kp[i] ? v * drop_scale : 0.f;
    }
  }
}

// dx (+)= rstd * (dyg - mean(dyg) - xhat * mean(dyg * xhat)), dyg = dy * g [masked where y <= 0 if relu];
// the masked dy is written to dy_masked (may alias dy) for the deferred parameter-gradient reduction.
__global__ __launch_bounds__(256) void mmt_ln_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ xhat,
                                                         const float *__restrict__ rstd, const float *__restrict__ g,
                                                         const float *__restrict__ y_relu, int M, int N,
                                                         float *__restrict__ dx, int accumulate,
                                                         float *__restrict__ dy_masked, const uint8_t *__restrict__ drop,
                                                         float drop_scale, float *__restrict__ dxm) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const size_t o = (size_t)row * N;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < N; c += 64) {
    float d = dy[o + c];
    if (y_relu && !(y_relu[o + c] > 0.f)) d = 0.f;
    if (dy_masked) dy_masked[o + c] = d;
    const float dg = d * g[c];
    s1 += dg;
    s2 += dg * xhat[o + c];
  }
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) {
    s1 += __shfl_xor(s1, k, 64);
    s2 += __shfl_xor(s2, k, 64);
  }
  const float m1 = s1 / (float)N, m2 = s2 / (float)N, rs = rstd[row];
  for (int c = lane; c < N; c += 64) {
    float d = dy_masked ? dy_masked[o + c] : dy[o + c];
    if (!dy_masked && y_relu && !(y_relu[o + c] > 0.f)) d = 0.f;
    const float v = rs * (d * g[c] - m1 - xhat[o + c] * m2);
    dx[o + c] = accumulate ? dx[o + c] + v : v;
    if (drop) dxm[o + c] = drop[o + c] ? v * drop_scale : 0.f;
  }
}

// ------------------------------------------------------------------------------------------ attention
// 16 lanes per (batch, head, query).  Keys 0..nk-1 with nk = causal ? min(i + 1 + causal_offset, Lk) : Lk (<= 32);
// key j masked when pad[b*Lk + j].  All operands addressed by (batch stride, row stride) in floats; head slices are
// 16-byte aligned (dh % 16 == 0, strides % 4 == 0: checked on the host).  Lane j scores keys j and j+16 (row
// reductions over the DPP row), then owns dh/16 output dims for P.V.
template <int DH>
__global__ __launch_bounds__(256) void mmt_attn_fwd_kernel(mmt_attn_args a) {
  constexpr int NQ = DH / 4, DPL = DH / 16;
  const int grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, l16 = threadIdx.x & 15;
  const bool live = grp < a.B * a.H * a.Lq;
  const int gq = live ? grp : 0;
  const int i = gq % a.Lq, h = (gq / a.Lq) % a.H, b = gq / (a.Lq * a.H);
  const float *q = a.q + (size_t)b * a.q_bs + (size_t)i * a.q_rs + h * DH;
  const float *K = a.k + (size_t)b * a.k_bs + h * DH, *V = a.v + (size_t)b * a.v_bs + h * DH;
  const int nk = !live ? 0 : a.causal ? min(i + 1 + a.causal_offset, a.Lk) : a.Lk;
  f32x4 qv[NQ];
#pragma unroll
  for (int c = 0; c < NQ; ++c) qv[c] = *reinterpret_cast<const f32x4 *>(q + 4 * c);
  float sc[2] = {-INFINITY, -INFINITY};
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int j = l16 + 16 * hf;
    if (j < nk && !(a.pad && a.pad[(size_t)b * a.Lk + j])) {
      const float *kj = K + (size_t)j * a.k_rs;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < NQ; ++c) {
        const f32x4 kv = *reinterpret_cast<const f32x4 *>(kj + 4 * c);
        acc = fmaf(qv[c][0], kv[0], acc);
        acc = fmaf(qv[c][1], kv[1], acc);
        acc = fmaf(qv[c][2], kv[2], acc);
        acc = fmaf(qv[c][3], kv[3], acc);
      }
      sc[hf] = acc * a.scale;
    }
  }
  const float mx = row16_max(fmaxf(sc[0], sc[1]));
  float p0 = 0.f, p1 = 0.f;
  if (mx > -INFINITY) {
    p0 = expf(sc[0] - mx);
    p1 = expf(sc[1] - mx);
  }
  const float sum = row16_sum(p0 + p1);
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
  p0 *= inv;
  p1 *= inv;
  if (live) {
    float *P = a.probs + (size_t)(b * a.H + h) * a.p_hs + (size_t)i * a.p_rs;
    if (l16 < a.Lk) P[l16] = p0;
    if (l16 + 16 < a.Lk) P[l16 + 16] = p1;
    if (a.pdrop) {       // the stash keeps the softmax; the output uses keep * softmax
      const uint8_t *D = a.pdrop + (size_t)(b * a.H + h) * a.p_hs + (size_t)i * a.p_rs;
      p0 *= (l16 < a.Lk && D[l16]) ? a.pdrop_scale : 0.f;
      p1 *= (l16 + 16 < a.Lk && D[l16 + 16]) ? a.pdrop_scale : 0.f;
    }
  }
  float o[DPL];
#pragma unroll
  for (int e = 0; e < DPL; ++e) o[e] = 0.f;
  const int nkw = min(nk, 16);
  for (int j = 0; j < 16; ++j) {
    const float pj = __shfl(p0, j, 16);
    if (j < nkw && pj > 0.f) {
#pragma unroll
      for (int e = 0; e < DPL; ++e) o[e] = fmaf(pj, V[(size_t)j * a.v_rs + l16 * DPL + e], o[e]);
    }
  }
  for (int j = 16; j < 32; ++j) {
    const float pj = __shfl(p1, j - 16, 16);
    if (j < nk && pj > 0.f) {
#pragma unroll
      for (int e = 0; e < DPL; ++e) o[e] = fmaf(pj, V[(size_t)j * a.v_rs + l16 * DPL + e], o[e]);
    }
  }
  if (live) {
    float *out = a.o + (size_t)b * a.o_bs + (size_t)i * a.o_rs + h * DH + l16 * DPL;
#pragma unroll
    for (int e = 0; e < DPL; ++e) out[e] = o[e];
  }
}

// The same for <= 16 keys with every operand requested at kernel entry (the form above loads the pad byte, then the key row under
// that branch, then - behind the softmax - the keep byte and one value row per key under a branch on its probability: a dozen
// dependent memory round trips, 9.9 us per launch at 1536 x 8 queries).  Same arithmetic in the same order: same bits.
template <int DH>
__global__ __launch_bounds__(256) void mmt_attn_fwd16_kernel(mmt_attn_args a) {
  constexpr int NQ = DH / 4, DPL = DH / 16;
  const int grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, l16 = threadIdx.x & 15;
  const bool live = grp < a.B * a.H * a.Lq;
  const int gq = live ? grp : 0;
  const int i = gq % a.Lq, h = (gq / a.Lq) % a.H, b = gq / (a.Lq * a.H);
  const int nk = !live ? 0 : a.causal ? min(i + 1 + a.causal_offset, a.Lk) : a.Lk;
  const rsrc_t rq = mk_rsrc(a.q), rk = mk_rsrc(a.k), rv = mk_rsrc(a.v);
  const bool kin = l16 < nk;
  const unsigned qo = (unsigned)(b * (int)a.q_bs + i * (int)a.q_rs + h * DH), ko = (unsigned)(b * (int)a.k_bs + l16 * (int)a.k_rs + h * DH);
  const unsigned vo = (unsigned)(b * (int)a.v_bs + h * DH + l16 * DPL), po = (unsigned)((b * a.H + h) * (int)a.p_hs + i * (int)a.p_rs + l16);
  f32x4 qv[NQ], kv[NQ];
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    qv[c] = bl4(rq, live ? (qo + 4u * c) * 4u : OOB);
    kv[c] = bl4(rk, kin ? (ko + 4u * c) * 4u : OOB);
  }
  const unsigned padb = __builtin_amdgcn_raw_buffer_load_b8(mk_rsrc(a.pad), (a.pad && kin) ? (unsigned)(b * a.Lk + l16) : OOB, 0, 0);
  const unsigned pdb = __builtin_amdgcn_raw_buffer_load_b8(mk_rsrc(a.pdrop), (a.pdrop && live && l16 < a.Lk) ? po : OOB, 0, 0);
  float vv[16][DPL];
#pragma unroll
  for (int j = 0; j < 16; ++j) ld_dpl<DPL>(vv[j], rv, j < nk ? (vo + (unsigned)(j * (int)a.v_rs)) * 4u : OOB);
  float sc = -INFINITY;
  if (kin && !(padb & 0xffu)) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      acc = fmaf(qv[c][0], kv[c][0], acc);
      acc = fmaf(qv[c][1], kv[c][1], acc);
      acc = fmaf(qv[c][2], kv[c][2], acc);
      acc = fmaf(qv[c][3], kv[c][3], acc);
    }
    sc = acc * a.scale;
  }
  const float mx = row16_max(sc);
  float p0 = mx > -INFINITY ? expf(sc - mx) : 0.f;
  const float sum = row16_sum(p0);
  p0 *= sum > 0.f ? 1.0f / sum : 0.f;
  if (live && l16 < a.Lk) a.probs[po] = p0;
  if (a.pdrop) p0 *= (l16 < a.Lk && (pdb & 0xffu)) ? a.pdrop_scale : 0.f;     // the stash keeps the softmax; the output uses keep * softmax
  float o[DPL];
#pragma unroll
  for (int e = 0; e < DPL; ++e) o[e] = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float pj = row16_bcast(p0, j);
    const bool on = j < nk && pj > 0.f;
#pragma unroll
    for (int e = 0; e < DPL; ++e) o[e] = on ? fmaf(pj, vv[j][e], o[e]) : o[e];
  }
  if (live) {
    float *out = a.o + (size_t)b * a.o_bs + (size_t)i * a.o_rs + h * DH + l16 * DPL;
#pragma unroll
    for (int e = 0; e < DPL; ++e) out[e] = o[e];
  }
}

// 16 lanes per (batch, head), queries in sequence: dq (=), dk +=, dv += from dO and the stashed probabilities.
// Per query: lanes = keys for dP_j = dO . v_j and the softmax Jacobian, then lanes = dims for the three updates.
template <int DH>
__global__ __launch_bounds__(256) void mmt_attn_bwd_kernel(mmt_attn_args a, const float *__restrict__ dO, long do_bs, long do_rs,
                                                           float *__restrict__ dq, long dq_bs, long dq_rs,
                                                           float *__restrict__ dk, float *__restrict__ dv, long dk_bs,
                                                           long dk_rs) {
  constexpr int NQ = DH / 4, DPL = DH / 16;
  const int grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, l16 = threadIdx.x & 15;
  const bool live = grp < a.B * a.H;
  const int gq = live ? grp : 0;
  const int h = gq % a.H, b = gq / a.H;
  const float *K = a.k + (size_t)b * a.k_bs + h * DH, *V = a.v + (size_t)b * a.v_bs + h * DH;
  float *dK = dk + (size_t)b * dk_bs + h * DH + l16 * DPL, *dV = dv + (size_t)b * dk_bs + h * DH + l16 * DPL;
  const int Lk = live ? a.Lk : 0;
  for (int i = 0; i < a.Lq; ++i) {
    const float *P = a.probs + (size_t)(b * a.H + h) * a.p_hs + (size_t)i * a.p_rs;
    const float *q = a.q + (size_t)b * a.q_bs + (size_t)i * a.q_rs + h * DH;
    const float *g = dO + (size_t)b * do_bs + (size_t)i * do_rs + h * DH;
    f32x4 gv[NQ];
#pragma unroll
    for (int c = 0; c < NQ; ++c) gv[c] = *reinterpret_cast<const f32x4 *>(g + 4 * c);
    const uint8_t *D = a.pdrop ? a.pdrop + (size_t)(b * a.H + h) * a.p_hs + (size_t)i * a.p_rs : nullptr;
    float p[2] = {0.f, 0.f}, dP[2] = {0.f, 0.f}, pd[2] = {0.f, 0.f};
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int j = l16 + 16 * hf;
      if (j < Lk) p[hf] = P[j];
      const float keep = (D && j < Lk) ? (D[j] ? a.pdrop_scale : 0.f) : 1.f;
      pd[hf] = p[hf] * keep;
      if (p[hf] > 0.f) {
        const float *vj = V + (size_t)j * a.v_rs;
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < NQ; ++c) {
          const f32x4 vv = *reinterpret_cast<const f32x4 *>(vj + 4 * c);
          t = fmaf(gv[c][0], vv[0], t);
          t = fmaf(gv[c][1], vv[1], t);
          t = fmaf(gv[c][2], vv[2], t);
          t = fmaf(gv[c][3], vv[3], t);
        }
        dP[hf] = t * keep;
      }
    }
    const float dot = row16_sum(fmaf(p[0], dP[0], p[1] * dP[1]));
    const float ds0 = p[0] * (dP[0] - dot) * a.scale, ds1 = p[1] * (dP[1] - dot) * a.scale;
    float qd[DPL], gd[DPL], dqa[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) {
      qd[e] = q[l16 * DPL + e];
      gd[e] = g[l16 * DPL + e];
      dqa[e] = 0.f;
    }
    for (int j = 0; j < 32; ++j) {
      const float pj = __shfl(j < 16 ? p[0] : p[1], j & 15, 16), dsj = __shfl(j < 16 ? ds0 : ds1, j & 15, 16);
      const float pdj = __shfl(j < 16 ? pd[0] : pd[1], j & 15, 16);
      if (j < Lk && pj > 0.f) {
        const float *kj = K + (size_t)j * a.k_rs + l16 * DPL;
        float *dkj = dK + (size_t)j * dk_rs, *dvj = dV + (size_t)j * dk_rs;
#pragma unroll
        for (int e = 0; e < DPL; ++e) {
          dqa[e] = fmaf(dsj, kj[e], dqa[e]);
          dkj[e] = fmaf(dsj, qd[e], dkj[e]);
          dvj[e] = fmaf(pdj, gd[e], dvj[e]);
        }
      }
    }
    if (live) {
      float *dqi = dq + (size_t)b * dq_bs + (size_t)i * dq_rs + h * DH + l16 * DPL;
#pragma unroll
      for (int e = 0; e < DPL; ++e) dqi[e] = dqa[e];
    }
  }
}

// Same contract for Lk <= 16 and head dim 16 / 32: everything a (batch, head) group reads more than once -- its K rows,
// the dK / dV rows it accumulates into, the V row of "its" key -- is requested up front with predicated buffer loads
// and lives in registers across the queries; the only memory traffic inside the query loop is q, dO and the
// probabilities.  (The general kernel above walks the keys with a load -> update -> store chain per key: one memory
// round trip each, ~10 us for 12 keys.)
template <int DH>
__global__ __launch_bounds__(256) void mmt_attn_bwd16_kernel(mmt_attn_args a, const float *__restrict__ dO, long do_bs, long do_rs,
                                                             float *__restrict__ dq, long dq_bs, long dq_rs,
                                                             float *__restrict__ dk, float *__restrict__ dv, long dk_bs,
                                                             long dk_rs) {
  constexpr int NQ = DH / 4, DPL = DH / 16;
  const int grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, l16 = threadIdx.x & 15;
  const bool live = grp < a.B * a.H;
  const int gq = live ? grp : 0;
  const int h = gq % a.H, b = gq / a.H;
  const int Lk = live ? a.Lk : 0;
  // wave-uniform descriptors (a per-lane base would make every load a readfirstlane loop); the group's base goes into the offset
  const rsrc_t rk = mk_rsrc(a.k), rv = mk_rsrc(a.v), rdk = mk_rsrc(dk), rdv = mk_rsrc(dv);
  const unsigned kb = (unsigned)(b * (int)a.k_bs + h * DH + l16 * DPL), vb = (unsigned)(b * (int)a.v_bs + h * DH);
  const unsigned db = (unsigned)(b * (int)dk_bs + h * DH + l16 * DPL);
  float kr[16][DPL], dkr[16][DPL], dvr[16][DPL];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    ld_dpl<DPL>(kr[j], rk, j < Lk ? (kb + (unsigned)(j * (int)a.k_rs)) * 4u : OOB);
    ld_dpl<DPL>(dkr[j], rdk, j < Lk ? (db + (unsigned)(j * (int)dk_rs)) * 4u : OOB);
    ld_dpl<DPL>(dvr[j], rdv, j < Lk ? (db + (unsigned)(j * (int)dk_rs)) * 4u : OOB);
  }
  f32x4 vrow[NQ];                    // V row of key l16
#pragma unroll
  for (int c = 0; c < NQ; ++c) vrow[c] = bl4(rv, l16 < Lk ? (vb + (unsigned)(l16 * (int)a.v_rs + 4 * c)) * 4u : OOB);
  float *dQ = dq + (size_t)b * dq_bs + h * DH + l16 * DPL;
  for (int i = 0; i < a.Lq; ++i) {
    const float *P = a.probs + (size_t)(b * a.H + h) * a.p_hs + (size_t)i * a.p_rs;
    const float *q = a.q + (size_t)b * a.q_bs + (size_t)i * a.q_rs + h * DH;
    const float *g = dO + (size_t)b * do_bs + (size_t)i * do_rs + h * DH;
    const float p = l16 < Lk ? P[l16] : 0.f;
    const float keep = (a.pdrop && l16 < Lk) ? (a.pdrop[(size_t)(b * a.H + h) * a.p_hs + (size_t)i * a.p_rs + l16] ? a.pdrop_scale : 0.f) : 1.f;
    float t = 0.f;
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const f32x4 gv = *reinterpret_cast<const f32x4 *>(g + 4 * c);
      t = fmaf(gv[0], vrow[c][0], t);
      t = fmaf(gv[1], vrow[c][1], t);
      t = fmaf(gv[2], vrow[c][2], t);
      t = fmaf(gv[3], vrow[c][3], t);
    }
    const float dP = p > 0.f ? t * keep : 0.f;
    const float dot = row16_sum(p * dP);
    const float ds = p * (dP - dot) * a.scale;
    const float pdrop = p * keep;
    float qd[DPL], gd[DPL], dqa[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) {
      qd[e] = q[l16 * DPL + e];
      gd[e] = g[l16 * DPL + e];
      dqa[e] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {       // masked / absent keys carry p = ds = 0: the updates are exact no-ops
      const float pj = row16_bcast(pdrop, j), dsj = row16_bcast(ds, j);
#pragma unroll
      for (int e = 0; e < DPL; ++e) {
        dqa[e] = fmaf(dsj, kr[j][e], dqa[e]);
        dkr[j][e] = fmaf(dsj, qd[e], dkr[j][e]);
        dvr[j][e] = fmaf(pj, gd[e], dvr[j][e]);
      }
    }
    if (live) {
#pragma unroll
      for (int e = 0; e < DPL; ++e) dQ[(size_t)i * dq_rs + e] = dqa[e];
    }
  }
  float *dK = dk + (size_t)b * dk_bs + h * DH + l16 * DPL, *dV = dv + (size_t)b * dk_bs + h * DH + l16 * DPL;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j < Lk) {
#pragma unroll
      for (int e = 0; e < DPL; ++e) {
        dK[(size_t)j * dk_rs + e] = dkr[j][e];
        dV[(size_t)j * dk_rs + e] = dvr[j][e];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ convolution as GEMM (training)
// The map CNN's training path (forward with stash + backward): every conv layer -- and the final flatten + Linear, which is a
// convolution whose kernel covers the whole feature map -- is im2col + the GEMMs above.  Activations stay as the GEMM leaves
// them: rows (b, y, x), one column per channel ("NHWC"); the column order of a patch row is (c, ky, kx) = the flattening of a
// Conv2d weight [Cout][Cin][kh][kw], so weights are used in place.
//   cols[(b*OH + oy)*OW + ox][(c*kh + ky)*kw + kx] = f(in[b][c][oy*s + ky][ox*s + kx]),  f(v) = alpha*v + beta then leaky_relu(slope)
// (alpha, beta: the map*2-1 rescale of the first layer; slope: the previous layer's activation applied on load, 1 = none).
__global__ __launch_bounds__(256) void mmt_im2col_kernel(const float *__restrict__ in, int B, int C, int H, int W, int kh, int kw,
                                                        int stride, int nhwc, long ld_in, float alpha, float beta, float slope,
                                                        float *__restrict__ cols, long ld_cols, int OH, int OW) {
  const int Kc = C * kh * kw;
  const long total = (long)B * OH * OW * Kc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int col = (int)(i % Kc);
    const long row = i / Kc;
    const int kx = col % kw, ky = (col / kw) % kh, c = col / (kw * kh);
    const int ox = (int)(row % OW), oy = (int)((row / OW) % OH), b = (int)(row / ((long)OW * OH));
    const int y = oy * stride + ky, x = ox * stride + kx;
    float v = nhwc ? in[((long)(b * H + y) * W + x) * ld_in + c] : in[(((long)b * C + c) * H + y) * W + x];
    v = fmaf(alpha, v, beta);
    cols[row * ld_cols + col] = v > 0.f ? v : slope * v;
  }
}
// Gather form of the transposed scatter (deterministic): the gradient at input pixel (b, y, x, c) collects every patch entry
// that read it, and is multiplied by the derivative of the activation that was applied on load (z = the stored pre-activation).
__global__ __launch_bounds__(256) void mmt_col2im_kernel(const float *__restrict__ dcols, int B, int C, int H, int W, int kh, int kw,
                                                        int stride, const float *__restrict__ z, float slope,
                                                        float *__restrict__ din, long ld_dcols, long ld_din, int OH, int OW) {
  const long total = (long)B * H * W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long pix = i / C;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    float acc = 0.f;
    for (int ky = 0; ky < kh; ++ky) {
      const int ty = y - ky;
      if (ty < 0 || ty % stride) continue;
      const int oy = ty / stride;
      if (oy >= OH) continue;
      for (int kx = 0; kx < kw; ++kx) {
        const int tx = x - kx;
        if (tx < 0 || tx % stride) continue;
        const int ox = tx / stride;
        if (ox >= OW) continue;
        acc += dcols[((long)(b * OH + oy) * OW + ox) * ld_dcols + (c * kh + ky) * kw + kx];
      }
    }
    if (z) acc *= z[pix * ld_din + c] > 0.f ? 1.0f : slope;
    din[pix * ld_din + c] = acc;
  }
}

// ------------------------------------------------------------------------------------------ loss / optimizer
// dpos[t][b][:] = (pos - y) / ||pos - y|| / (B * ph)   with pos laid out [ph][B][2], y laid out [B][ph][2]
// (with loss_sum: ONE block, fixed summation order - the same bits on every run, like mmt_finalize)
__global__ __launch_bounds__(1024) void mmt_l2_loss_bwd_kernel(const float *__restrict__ pos, const float *__restrict__ y, int B, int ph,
                                                               float *__restrict__ dpos, float *__restrict__ loss_sum) {
  __shared__ float wsum[16];
  float contrib = 0.f;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < B * ph; idx += gridDim.x * blockDim.x) {
    const int b = idx % B, t = idx / B;
    const float dx = pos[(size_t)idx * 2] - y[((size_t)b * ph + t) * 2];
    const float dy = pos[(size_t)idx * 2 + 1] - y[((size_t)b * ph + t) * 2 + 1];
    const float n = sqrtf(dx * dx + dy * dy);
    const float w = n > 0.f ? 1.0f / (n * (float)B * (float)ph) : 0.f;
    dpos[(size_t)idx * 2] = dx * w;
    dpos[(size_t)idx * 2 + 1] = dy * w;
    contrib += n / ((float)B * (float)ph);
  }
  if (loss_sum) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) contrib += __shfl_xor(contrib, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = contrib;
    __syncthreads();
    if (threadIdx.x == 0) {
      float total = 0.f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) total += wsum[w];
      *loss_sum += total;
    }
  }
}

// state[0] = step count (float), state[1] = learning rate of group 0 (both live on the device so that a captured
// graph stays valid while the host scheduler changes them).
__global__ void mmt_adamw_kernel(const mmt_adamw_tensor *__restrict__ table, int ntensors, const float *__restrict__ state,
                                 float beta1, float beta2, float eps, float weight_decay, float clip, float grad_scale,
                                 const float *__restrict__ guard) {
  const int t = blockIdx.y;
  if (t >= ntensors) return;
  if (guard && !isfinite(guard[0])) return;        // a step whose (all-reduced) loss is NaN / inf must not touch the weights
  const mmt_adamw_tensor T = table[t];
  const float step = state[0], lr = state[1] * T.lr_scale;
  const float bc1 = 1.0f - powf(beta1, step), bc2 = 1.0f - powf(beta2, step);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)T.numel; i += (size_t)gridDim.x * blockDim.x) {
    float g = T.grad[i] * grad_scale;
    if (clip > 0.f) g = fminf(fmaxf(g, -clip), clip);            // nn.utils.clip_grad_value_
    float p = T.param[i];
    p *= 1.0f - lr * weight_decay;                                // decoupled weight decay
    const float m = beta1 * T.exp_avg[i] + (1.0f - beta1) * g;
    const float v = beta2 * T.exp_avg_sq[i] + (1.0f - beta2) * g * g;
    T.exp_avg[i] = m;
    T.exp_avg_sq[i] = v;
    const float denom = sqrtf(v) / sqrtf(bc2) + eps;
    p -= (lr / bc1) * (m / denom);
    T.param[i] = p;
  }
}

// x0 = nan_to_num(x, nan=0) over (rows, n) and pad[row] = isnan(x[row][1])
__global__ void mmt_history_prepare_kernel(const float *__restrict__ x, int rows, int n, float *__restrict__ x0,
                                           uint8_t *__restrict__ pad) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * n) return;
  const float v = x[idx];
  x0[idx] = (v != v) ? 0.f : v;
  if (idx % n == 1) pad[idx / n] = (v != v) ? 1 : 0;
}

// y[o][j][i] = softmax_j x[o][j][i]  (nn.Softmax over one axis of a contiguous tensor; the reference's
// Mlp(mode='classification') applies it with dim=None, i.e. over dim 0 of its 3-D input, MLP.py:18-21)
__global__ void mmt_softmax_axis_kernel(const float *__restrict__ x, long outer, int n, long inner, float *__restrict__ y) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= outer * inner) return;
  const long o = idx / inner, i = idx - o * inner;
  const float *xp = x + o * n * inner + i;
  float *yp = y + o * n * inner + i;
  float mx = -INFINITY;
  for (int j = 0; j < n; ++j) mx = fmaxf(mx, xp[j * inner]);
  float sum = 0.f;
  for (int j = 0; j < n; ++j) sum += expf(xp[j * inner] - mx);
  const float inv = 1.0f / sum;
  for (int j = 0; j < n; ++j) yp[j * inner] = expf(xp[j * inner] - mx) * inv;
}

// y[o][t][i] = sum_{u <= t} x[o][u][i]   (torch.cumsum over one axis; mgcvae.py:352,370)
__global__ void mmt_cumsum_axis_kernel(const float *__restrict__ x, long outer, int n, long inner, float *__restrict__ y) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= outer * inner) return;
  const long o = idx / inner, i = idx - o * inner;
  float acc = 0.f;
  for (int j = 0; j < n; ++j) {
    acc += x[(o * n + j) * inner + i];
    y[(o * n + j) * inner + i] = acc;
  }
}

// out[r] = || a[r][:] - b[rb][:] ||_2 with rb = (r / (rep * inner)) * inner + r % inner: b is broadcast over a middle axis of
// length rep (L2_norm(lane_pred (bs,L,T,2), labels (bs,1,T,2)), model_utils.py:154-160; rep = 1: equal shapes)
__global__ void mmt_l2_norm_kernel(const float *__restrict__ a, const float *__restrict__ b, long rows, int n, long rep, long inner,
                                   float *__restrict__ out) {
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
    const long rb = (r / (rep * inner)) * inner + r % inner;
    float s = 0.f;
    for (int c = 0; c < n; ++c) {
      const float d = a[r * n + c] - b[rb * n + c];
      s += d * d;
    }
    out[r] = sqrtf(s);
  }
}

__global__ void mmt_add_inplace_kernel(float *__restrict__ y, const float *__restrict__ x, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += x[i];
}

// ---- ViT map encoder training pieces (vit_pytorch 0.33.2: patch rearrange, nn.GELU, cls / position tokens) ----
// img (B,C,S,S) -> patches (B*(S/P)^2, ld): row (b, hy, wx), column (p1*P + p2)*C + c   ('b c (h p1) (w p2) -> b (h w) (p1 p2 c)')
__global__ void mmt_patchify_kernel(const float *__restrict__ img, int B, int C, int S, int P, float alpha, float beta,
                                    float *__restrict__ out, long ld) {
  const int n = S / P, pd = P * P * C;
  const long total = (long)B * n * n * pd;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int col = (int)(i % pd);
    const long row = i / pd;
    const int c = col % C, p2 = (col / C) % P, p1 = col / (C * P);
    const int wx = (int)(row % n), hy = (int)((row / n) % n);
    const long b = row / ((long)n * n);
    out[row * ld + col] = alpha * img[((b * C + c) * S + hy * P + p1) * (long)S + wx * P + p2] + beta;
  }
}
// exact (erf) GELU, nn.GELU's default
__global__ void mmt_gelu_fwd_kernel(const float *__restrict__ z, float *__restrict__ y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = 0.5f * z[i] * (1.0f + erff(z[i] * 0.70710678118654752f));
}
__global__ void mmt_gelu_bwd_kernel(const float *__restrict__ z, const float *__restrict__ dy, float *__restrict__ dz, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float v = z[i];
    dz[i] = dy[i] * (0.5f * (1.0f + erff(v * 0.70710678118654752f)) + v * 0.39894228040143268f * expf(-0.5f * v * v));
  }
}
// dst[b][r][0..N) (+)= src[b][r][0..N): views with batch / row strides (a stride of 0 broadcasts the source)
__global__ void mmt_copy3_kernel(const float *__restrict__ src, long sbs, long srs, float *__restrict__ dst, long dbs, long drs,
                                 int B, int R, int N, int accumulate) {
  const long total = (long)B * R * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % N);
    const long br = i / N;
    const int r = (int)(br % R);
    const long b = br / R;
    const float v = src[b * sbs + r * srs + c];
    float *d = dst + b * dbs + r * drs + c;
    *d = accumulate ? *d + v : v;
  }
}

// ------------------------------------------------------------------------------------------ dropout masks
// Philox4x32-10 (Salmon et al., SC'11): counter (element index / 4, mask stream, step), key = seed.
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0, c[1] = lo1, c[2] = n2, c[3] = lo0;
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
}
__global__ __launch_bounds__(256) void mmt_dropout_mask_kernel(uint32_t *__restrict__ keep4, long n4, uint32_t threshold,
                                                               const uint64_t *__restrict__ rng, uint32_t stream_id) {
  const uint64_t seed = rng[0], step = rng[1];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    uint32_t c[4] = {(uint32_t)i, (uint32_t)((uint64_t)i >> 32) ^ (stream_id << 8), (uint32_t)step, (uint32_t)(step >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint32_t out = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) out |= (c[e] >= threshold ? 1u : 0u) << (8 * e);
    keep4[i] = out;
  }
}
__global__ void mmt_rng_advance_kernel(uint64_t *rng) {
  if (threadIdx.x == 0 && blockIdx.x == 0) rng[1] += 1;
}
__global__ __launch_bounds__(256) void mmt_dropout_apply_kernel(const float *__restrict__ x, const uint8_t *__restrict__ keep, float scale,
                                                                float *__restrict__ y, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = keep[i] ? x[i] * scale : 0.f;
}

__global__ void mmt_lr_step_kernel(float *state, float gamma, const float *guard, float *skipped) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (guard && !isfinite(guard[0])) {            // the optimizer kernel skipped this step: count it, leave step and lr alone
      if (skipped) skipped[0] += 1.0f;
      return;
    }
    state[0] += 1.0f;
    state[1] *= gamma;
  }
}

}  // namespace

// diagnostic switches, read once: MMT_TRAIN_GENERIC bit 0 = generic LayerNorm forward, 1 = generic LayerNorm backward, 2 = generic
// attention forward (A/B of the register-resident / burst forms against the forms they replaced)
static int train_generic() {
  static const int v = getenv("MMT_TRAIN_GENERIC") ? atoi(getenv("MMT_TRAIN_GENERIC")) : 0;
  return v;
}
static int launch_ok(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mmt_set_error(MMT_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return MMT_OK;
}

// the kernels that request their epilogue operands through buffer descriptors address them with 32-bit byte offsets
static bool epi32_ok(const mmt_gemm_args *g) {
  const double M = (double)g->M;
  return g->ldc >= 0 && M * (double)g->ldc + g->N < 5e8 && (!g->addend || (g->ld_add >= 0 && M * (double)g->ld_add + g->N < 5e8)) &&
         (!g->mask || (g->ld_mask >= 0 && M * (double)g->ld_mask + g->N < 5e8)) &&
         (!g->drop || (g->ld_drop >= 0 && M * (double)g->ld_drop + g->N < 2e9));
}

extern "C" int mmt_gemm(const mmt_gemm_args *g, void *stream) {
  if (!g || !g->A || !g->B || !g->C) return mmt_set_error(MMT_EINVAL, "mmt_gemm: null pointer");
  if (g->M <= 0 || g->N <= 0 || g->K <= 0) return mmt_set_error(MMT_EINVAL, "mmt_gemm: empty problem %dx%dx%d", g->M, g->N, g->K);
  // panel kernel: 16-byte aligned operands whose unit stride is K or the row index, all offsets within 2^31 bytes
  const bool ak = g->sak == 1 && g->sam % 4 == 0, ar = g->sam == 1 && g->sak % 4 == 0 && g->M % 4 == 0;
  const bool bk = g->sbk == 1 && g->sbn % 4 == 0, br = g->sbn == 1 && g->sbk % 4 == 0 && g->N % 4 == 0;
  const bool aligned = (((uintptr_t)g->A | (uintptr_t)g->B) & 15) == 0 && g->K % 4 == 0;
  const double amax = (double)g->M * (double)(ak ? g->sam : 1) + (double)g->K * (double)(ak ? 1 : g->sak);
  const double bmax = (double)g->N * (double)(bk ? g->sbn : 1) + (double)g->K * (double)(bk ? 1 : g->sbk);
  hipStream_t st = (hipStream_t)stream;
  if (g->drop && (g->ld_drop < g->N || g->accumulate || (g->drop_after_add && g->relu)))
    return mmt_set_error(MMT_EINVAL, "mmt_gemm: dropout needs ld_drop >= N, no accumulate, and relu only before the addend");
  if (g->ln_drop && (!g->ln_xhat || ((uintptr_t)g->ln_drop & 3) || (g->ld_ln_drop & 3) || ((uintptr_t)g->ln_dxm & 15)))
    return mmt_set_error(MMT_EINVAL, "mmt_gemm: ln_drop goes with the fused LayerNorm backward, 4-byte aligned keep rows");
  if (g->splitk > 1 || g->asum) {
    if (!(aligned && ar && br && amax < 5e8 && bmax < 5e8) || g->bias || g->addend || g->mask || g->relu || g->accumulate ||
        g->ln_xhat || g->drop || g->splitk > 16 || (g->splitk > 1 && g->part_stride <= 0))
      return mmt_set_error(MMT_EINVAL, "mmt_gemm: splitk / asum need A and B with unit stride along their rows (M, N "
                                       "multiples of 4, 16-byte aligned), no epilogue, splitk <= 16");
    launch_split(*g, st);
    return launch_ok("mmt_gemm");
  }
  // row-block kernel: few rows, whole K per workgroup (the dependent chain of the backward sweep)
  const bool rows_ok = aligned && ak && g->K % 16 == 0 && g->K <= 768 && g->M <= 256 && amax < 5e8 && bmax < 5e8 && epi32_ok(g) &&
                       (bk || (g->sbn == 1));
  if (g->ln_xhat) {
    if (!rows_ok || !g->ln_rstd || !g->ln_g || g->ld_ln % 4 || ((uintptr_t)g->ln_xhat & 15) || ((uintptr_t)g->ln_dx & 15) ||
        ((uintptr_t)g->ln_g & 15))
      return mmt_set_error(MMT_EINVAL, "mmt_gemm: the fused LayerNorm backward needs M <= 256, K <= 768 a multiple of 16, "
                                       "16-byte aligned rows");
    bk ? launch_rows<true, true>(*g, st) : launch_rows<false, true>(*g, st);
    return launch_ok("mmt_gemm");
  }
  if (rows_ok && g->N >= 32) {
    bk ? launch_rows<true, false>(*g, st) : launch_rows<false, false>(*g, st);
    return launch_ok("mmt_gemm");
  }
  if (aligned && (ak || ar) && (bk || br) && !(ar && !ak && bk) && amax < 5e8 && bmax < 5e8 && epi32_ok(g)) {
    const bool big = ((g->M + 63) / 64) * ((g->N + 63) / 64) >= 128;
    if (ak && bk) big ? launch_panel<64, true, true>(*g, st) : launch_panel<32, true, true>(*g, st);
    else if (ak) big ? launch_panel<64, true, false>(*g, st) : launch_panel<32, true, false>(*g, st);
    else big ? launch_panel<64, false, false>(*g, st) : launch_panel<32, false, false>(*g, st);
    return launch_ok("mmt_gemm");
  }
  if (g->K <= 8 && (long)g->M * g->N <= (1 << 20) && !g->ln_xhat && std::abs((double)g->M * (double)g->sam) + 8.0 * std::abs((double)g->sak) < 5e8 &&
      std::abs((double)g->N * (double)g->sbn) + 8.0 * std::abs((double)g->sbk) < 5e8 && (double)g->M * (double)g->ldc < 5e8 &&
      g->sam >= 0 && g->sak >= 0 && g->sbk >= 0 && g->sbn >= 0 && (!g->addend || (double)g->M * (double)g->ld_add < 5e8) &&
      (!g->mask || (double)g->M * (double)g->ld_mask < 5e8) && (!g->drop || (double)g->M * (double)g->ld_drop < 2e9)) {
    hipLaunchKernelGGL(mmt_gemm_smallk_kernel, dim3((unsigned)(((long)g->M * g->N + 255) / 256)), dim3(256), 0, st, *g);
    return launch_ok("mmt_gemm");
  }
  if ((long)g->M * g->N <= 2048 && g->K >= 256) {
    hipLaunchKernelGGL(mmt_gemm_thin_kernel, dim3((g->M * g->N + 3) / 4), dim3(256), 0, st, *g);
    return launch_ok("mmt_gemm");
  }
  dim3 grid((g->N + GBN - 1) / GBN, (g->M + GBM - 1) / GBM);
  hipLaunchKernelGGL(mmt_gemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, *g);
  return launch_ok("mmt_gemm");
}

// dO = A' @ B (as mmt_gemm with the fused LayerNorm backward when g->ln_xhat is set, no epilogue, C unused), consumed by the
// backward of an attention with one query per batch element: dq (=), dk/dv (+=) as mmt_attention_bwd with d_o = dO.
extern "C" int mmt_gemm_attention_bwd(const mmt_gemm_args *g, const mmt_attn_args *a, float *dq, long dq_bs, float *dk, float *dv,
                                      long dk_bs, long dk_rs, void *stream) {
  if (!g || !a || !g->A || !g->B || !dq || !dk || !dv) return mmt_set_error(MMT_EINVAL, "mmt_gemm_attention_bwd: null pointer");
  const bool ak = g->sak == 1 && g->sam % 4 == 0;
  const bool aligned = (((uintptr_t)g->A | (uintptr_t)g->B) & 15) == 0;
  const bool shape = ak && g->sbn == 1 && g->K % 16 == 0 && g->K <= 256 && g->M <= 256 && g->M == a->B && a->Lq == 1 && a->dh == 32 &&
                     g->N == a->H * 32 && a->Lk >= 1 && a->Lk <= 16;
  const bool plain = !g->bias && !g->addend && !g->mask && !g->relu && !g->accumulate && g->splitk <= 1 && !g->asum && !g->drop &&
                     (!g->ln_drop || (g->ln_xhat && !((uintptr_t)g->ln_drop & 3) && !(g->ld_ln_drop & 3) && !((uintptr_t)g->ln_dxm & 15)));
  const bool attn_ok = a->q && a->k && a->v && a->probs && (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)dk |
                                                             (uintptr_t)dv | (uintptr_t)dq) & 15) == 0 &&
                       ((a->q_bs | a->k_bs | a->k_rs | a->v_bs | a->v_rs | dk_bs | dk_rs | dq_bs) & 3) == 0 && a->p_hs >= a->Lk &&
                       (double)a->B * (double)std::max(std::max(a->k_bs, a->v_bs), (int64_t)dk_bs) +
                               (double)a->Lk * (double)std::max(std::max(a->k_rs, a->v_rs), (int64_t)dk_rs) < 5e8 &&
                       (double)a->B * a->H * (double)a->p_hs < 5e8;
  const bool ln_ok = !g->ln_xhat || (g->ln_rstd && g->ln_g && g->ld_ln % 4 == 0 && (((uintptr_t)g->ln_xhat | (uintptr_t)g->ln_dx |
                                                                                    (uintptr_t)g->ln_g) & 15) == 0);
  if (!aligned || !shape || !plain || !attn_ok || !ln_ok)
    return mmt_set_error(MMT_EINVAL, "mmt_gemm_attention_bwd: needs the row-block GEMM shape (M = batch <= 256, K <= 256 a multiple "
                                     "of 16, B operand with unit column stride), head dim 32, one query, <= 16 keys, no epilogue");
  RowsAttn t;
  t.a = *a;
  t.dq = dq, t.dk = dk, t.dv = dv;
  t.dq_bs = dq_bs, t.dk_bs = dk_bs, t.dk_rs = dk_rs;
  const dim3 grid(a->H, (g->M + 15) / 16), blk(256);
  hipStream_t st = (hipStream_t)stream;
#define ROWS_ATT(NCHV)                                                                                        \
  do {                                                                                                        \
    if (g->ln_xhat) hipLaunchKernelGGL((mmt_gemm_rows_kernel<NCHV, false, true, true>), grid, blk, 0, st, *g, t); \
    else hipLaunchKernelGGL((mmt_gemm_rows_kernel<NCHV, false, false, true>), grid, blk, 0, st, *g, t);          \
  } while (0)
  ROWS_ATT(4);
#undef ROWS_ATT
  return launch_ok("mmt_gemm_attention_bwd");
}

extern "C" int mmt_im2col(const float *in, int B, int C, int H, int W, int kh, int kw, int stride, int nhwc, long ld_in, float alpha,
                          float beta, float slope, float *cols, long ld_cols, void *stream) {
  if (!in || !cols || B <= 0 || C <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || H < kh || W < kw || (nhwc && ld_in < C) ||
      ld_cols < (long)C * kh * kw)
    return mmt_set_error(MMT_EINVAL, "mmt_im2col: bad arguments");
  const int OH = (H - kh) / stride + 1, OW = (W - kw) / stride + 1;
  const long total = (long)B * OH * OW * C * kh * kw;
  hipLaunchKernelGGL(mmt_im2col_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 65536)), dim3(256), 0, (hipStream_t)stream, in, B, C,
                     H, W, kh, kw, stride, nhwc, ld_in, alpha, beta, slope, cols, ld_cols, OH, OW);
  return launch_ok("mmt_im2col");
}

extern "C" int mmt_col2im(const float *dcols, long ld_dcols, int B, int C, int H, int W, int kh, int kw, int stride, const float *z,
                          float slope, float *din, long ld_din, void *stream) {
  if (!dcols || !din || B <= 0 || C <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || H < kh || W < kw || ld_dcols < (long)C * kh * kw ||
      ld_din < C)
    return mmt_set_error(MMT_EINVAL, "mmt_col2im: bad arguments");
  const int OH = (H - kh) / stride + 1, OW = (W - kw) / stride + 1;
  const long total = (long)B * H * W * C;
  hipLaunchKernelGGL(mmt_col2im_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 65536)), dim3(256), 0, (hipStream_t)stream, dcols, B,
                     C, H, W, kh, kw, stride, z, slope, din, ld_dcols, ld_din, OH, OW);
  return launch_ok("mmt_col2im");
}

extern "C" int mmt_ln_param_grads(const float *dy, const float *xhat, int M, int N, long ld, float *gw, float *gb, int splitk,
                                  long part_stride, void *stream) {
  if (!dy || !xhat || !gw || !gb || M <= 0 || N <= 0 || splitk < 1 || splitk > 64 || (splitk > 1 && part_stride <= 0))
    return mmt_set_error(MMT_EINVAL, "mmt_ln_param_grads: bad arguments");
  hipLaunchKernelGGL(mmt_ln_param_grads_kernel, dim3((N + 15) / 16, splitk), dim3(1024), 0, (hipStream_t)stream, dy, xhat, M, N, ld,
                     gw, gb, part_stride);
  return launch_ok("mmt_ln_param_grads");
}

// ---- group launches: the table is [njobs job structs][nblocks GroupBlock], built on the host by *_plan (which validates every
// job as the single-job entry point would), copied to device memory by the caller once and launched any number of times
static bool split_job_ok(const mmt_gemm_args *g) {
  if (!g->A || !g->B || !g->C || g->M <= 0 || g->N <= 0 || g->K <= 0) return false;
  const bool ar = g->sam == 1 && g->sak % 4 == 0 && g->M % 4 == 0, br = g->sbn == 1 && g->sbk % 4 == 0 && g->N % 4 == 0;
  const bool aligned = (((uintptr_t)g->A | (uintptr_t)g->B) & 15) == 0 && g->K % 4 == 0;
  const double amax = (double)g->M + (double)g->K * (double)g->sak, bmax = (double)g->N + (double)g->K * (double)g->sbk;
  return aligned && ar && br && amax < 5e8 && bmax < 5e8 && !g->bias && !g->addend && !g->mask && !g->relu && !g->accumulate &&
         !g->ln_xhat && !g->drop && g->splitk <= 16 && (g->splitk <= 1 || g->part_stride > 0);
}
template <class Job, class Blocks>
static int group_plan(const char *who, const Job *jobs, int njobs, void *table_host, size_t capacity, size_t *bytes, int *nblocks,
                      Blocks blocks_of) {
  if (!jobs || njobs <= 0 || !bytes || !nblocks) return mmt_set_error(MMT_EINVAL, "%s: null pointer / no jobs", who);
  long total = 0;
  for (int j = 0; j < njobs; ++j) {
    int nx, ny, nz;
    if (!blocks_of(jobs[j], nx, ny, nz)) return mmt_set_error(MMT_EINVAL, "%s: job %d is not one the group kernel takes", who, j);
    total += (long)nx * ny * nz;
  }
  if (total > (1l << 30)) return mmt_set_error(MMT_EINVAL, "%s: too many workgroups", who);
  *nblocks = (int)total;
  *bytes = (size_t)njobs * sizeof(Job) + (size_t)total * sizeof(GroupBlock);
  if (!table_host) return MMT_OK;
  if (capacity < *bytes) return mmt_set_error(MMT_EINVAL, "%s: table needs %zu bytes, %zu given", who, *bytes, capacity);
  memcpy(table_host, jobs, (size_t)njobs * sizeof(Job));
  GroupBlock *map = reinterpret_cast<GroupBlock *>(static_cast<char *>(table_host) + (size_t)njobs * sizeof(Job));
  // Placement (speed only: every block computes the same tile wherever it runs).  Workgroups are dealt round-robin over the 8
  // XCDs, each with its own L2: with the jobs' tiles simply concatenated every XCD ends up reading every operand panel of every
  // job (measured: 1.3 GB per training step at the fabric for ~160 MB of operands).  So the jobs are dealt over 8 lanes -
  // largest first, each to the lane with the fewest tiles so far - and table entry 8 i + x is the i-th tile of lane x: the
  // tiles of one job run on one XCD and find its panels in that L2.
  constexpr int NX = 8;
  std::vector<int> order(njobs), ntile(njobs);
  for (int j = 0; j < njobs; ++j) {
    int nx, ny, nz;
    blocks_of(jobs[j], nx, ny, nz);
    order[j] = j, ntile[j] = nx * ny * nz;
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ntile[a] > ntile[b]; });
  std::vector<std::vector<GroupBlock>> lane(NX);
  for (int j : order) {
    int best = 0;
    for (int x = 1; x < NX; ++x)
      if (lane[x].size() < lane[best].size()) best = x;
    int nx, ny, nz;
    blocks_of(jobs[j], nx, ny, nz);
    for (int z = 0; z < nz; ++z)
      for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) lane[best].push_back(GroupBlock{j, x, y, z});
  }
  size_t pos[NX] = {0};
  for (long left = total; left > 0;)
    for (int x = 0; x < NX && left > 0; ++x)
      if (pos[x] < lane[x].size()) *map++ = lane[x][pos[x]++], --left;
  return MMT_OK;
}
constexpr int GROUP_BT = 32;
extern "C" int mmt_gemm_group_plan(const mmt_gemm_args *jobs, int njobs, void *table_host, size_t capacity, size_t *bytes, int *nblocks) {
  return group_plan("mmt_gemm_group_plan", jobs, njobs, table_host, capacity, bytes, nblocks,
                    [](const mmt_gemm_args &g, int &nx, int &ny, int &nz) {
                      if (!split_job_ok(&g)) return false;
                      nx = (g.N + GROUP_BT - 1) / GROUP_BT, ny = (g.M + GROUP_BT - 1) / GROUP_BT, nz = std::max(g.splitk, 1);
                      return true;
                    });
}
extern "C" int mmt_gemm_group(const void *table_dev, int njobs, int nblocks, void *stream) {
  if (!table_dev || njobs <= 0 || nblocks <= 0 || ((uintptr_t)table_dev & 15))
    return mmt_set_error(MMT_EINVAL, "mmt_gemm_group: null / misaligned table or nothing to do");
  constexpr int KC = 64, LDR = GROUP_BT + 4;
  constexpr size_t lds = 2 * (KC * LDR + KC * LDR) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)mmt_gemm_group_kernel<GROUP_BT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const mmt_gemm_args *jobs = static_cast<const mmt_gemm_args *>(table_dev);
  hipLaunchKernelGGL((mmt_gemm_group_kernel<GROUP_BT>), dim3(nblocks), dim3(256), lds, (hipStream_t)stream, jobs,
                     reinterpret_cast<const GroupBlock *>(jobs + njobs));
  return launch_ok("mmt_gemm_group");
}
extern "C" int mmt_ln_grad_group_plan(const mmt_ln_grad_job *jobs, int njobs, void *table_host, size_t capacity, size_t *bytes,
                                      int *nblocks) {
  return group_plan("mmt_ln_grad_group_plan", jobs, njobs, table_host, capacity, bytes, nblocks,
                    [](const mmt_ln_grad_job &J, int &nx, int &ny, int &nz) {
                      if (!J.dy || !J.xhat || !J.gw || !J.gb || J.M <= 0 || J.N <= 0 || J.splitk < 1 || J.splitk > 64 ||
                          (J.splitk > 1 && J.part_stride <= 0))
                        return false;
                      nx = (J.N + 15) / 16, ny = 1, nz = J.splitk;
                      return true;
                    });
}
extern "C" int mmt_ln_grad_group(const void *table_dev, int njobs, int nblocks, void *stream) {
  if (!table_dev || njobs <= 0 || nblocks <= 0 || ((uintptr_t)table_dev & 15))
    return mmt_set_error(MMT_EINVAL, "mmt_ln_grad_group: null / misaligned table or nothing to do");
  const mmt_ln_grad_job *jobs = static_cast<const mmt_ln_grad_job *>(table_dev);
  hipLaunchKernelGGL(mmt_ln_grad_group_kernel, dim3(nblocks), dim3(1024), 0, (hipStream_t)stream, jobs,
                     reinterpret_cast<const GroupBlock *>(jobs + njobs));
  return launch_ok("mmt_ln_grad_group");
}

extern "C" int mmt_colsum(const float *a, const float *b, const float *mask, int M, int N, long lda, long ldb, long ldm,
                          float *out, int accumulate, void *stream) {
  if (!a || !out || M <= 0 || N <= 0) return mmt_set_error(MMT_EINVAL, "mmt_colsum: bad arguments");
  hipLaunchKernelGGL(mmt_colsum_kernel, dim3((N + 15) / 16), dim3(1024), 0, (hipStream_t)stream, a, b, mask, M, N, lda, ldb,
                     ldm, out, accumulate);
  return launch_ok("mmt_colsum");
}

extern "C" int mmt_layernorm_fwd(const float *x, const float *res, int M, int N, const float *g, const float *b, int relu,
                                 float *y, float *xhat, float *rstd, void *stream) {
  if (!x || !g || !b || !y || !xhat || !rstd || M <= 0 || N <= 0) return mmt_set_error(MMT_EINVAL, "mmt_layernorm_fwd: bad arguments");
  const dim3 grid((M + 3) / 4), blk(256);
  hipStream_t st = (hipStream_t)stream;
  // (y / xhat may alias x or res element for element - each lane reads its elements before it writes them)
  if (train_generic() & 1) hipLaunchKernelGGL(mmt_ln_fwd_kernel, grid, blk, 0, st, x, res, M, N, g, b, relu, y, xhat, rstd);
  else if (N == 256) hipLaunchKernelGGL(mmt_ln_fwd_regs_kernel<4>, grid, blk, 0, st, x, res, M, g, b, relu, y, xhat, rstd);
  else if (N == 128) hipLaunchKernelGGL(mmt_ln_fwd_regs_kernel<2>, grid, blk, 0, st, x, res, M, g, b, relu, y, xhat, rstd);
  else if (N == 64) hipLaunchKernelGGL(mmt_ln_fwd_regs_kernel<1>, grid, blk, 0, st, x, res, M, g, b, relu, y, xhat, rstd);
  else if (N == 512) hipLaunchKernelGGL(mmt_ln_fwd_regs_kernel<8>, grid, blk, 0, st, x, res, M, g, b, relu, y, xhat, rstd);
  else hipLaunchKernelGGL(mmt_ln_fwd_kernel, grid, blk, 0, st, x, res, M, N, g, b, relu, y, xhat, rstd);
  return launch_ok("mmt_layernorm_fwd");
}

extern "C" int mmt_layernorm_bwd(const float *dy, const float *xhat, const float *rstd, const float *g, const float *y_relu,
                                 int M, int N, float *dx, int accumulate, float *dy_masked, const uint8_t *drop, float drop_scale,
                                 float *dx_dropped, void *stream) {
  if (!dy || !xhat || !rstd || !g || !dx || M <= 0 || N <= 0) return mmt_set_error(MMT_EINVAL, "mmt_layernorm_bwd: bad arguments");
  if ((drop != nullptr) != (dx_dropped != nullptr) || (drop && accumulate))
    return mmt_set_error(MMT_EINVAL, "mmt_layernorm_bwd: drop and dx_dropped go together, and not with accumulate");
  const dim3 grid((M + 3) / 4), blk(256);
  hipStream_t st = (hipStream_t)stream;
#define LN_BWD_REGS(NPL) hipLaunchKernelGGL(mmt_ln_bwd_regs_kernel<NPL>, grid, blk, 0, st, dy, xhat, rstd, g, y_relu, M, dx, accumulate, dy_masked, drop, drop_scale, dx_dropped)
  if (train_generic() & 2) hipLaunchKernelGGL(mmt_ln_bwd_kernel, grid, blk, 0, st, dy, xhat, rstd, g, y_relu, M, N,
                     dx, accumulate, dy_masked, drop, drop_scale, dx_dropped);
  else if (N == 256) LN_BWD_REGS(4);
  else if (N == 128) LN_BWD_REGS(2);
  else if (N == 64) LN_BWD_REGS(1);
  else if (N == 512) LN_BWD_REGS(8);
  else if (N == 32) hipLaunchKernelGGL((mmt_ln_bwd_regs_kernel<1, true>), grid, blk, 0, st, dy, xhat, rstd, g, y_relu, M, dx, accumulate, dy_masked, drop, drop_scale, dx_dropped);
  else
#undef LN_BWD_REGS
  hipLaunchKernelGGL(mmt_ln_bwd_kernel, grid, blk, 0, st, dy, xhat, rstd, g, y_relu, M, N,
                     dx, accumulate, dy_masked, drop, drop_scale, dx_dropped);
  return launch_ok("mmt_layernorm_bwd");
}

static int check_attn(const mmt_attn_args *a, const char *who) {
  if (!a || !a->q || !a->k || !a->v || !a->probs) return mmt_set_error(MMT_EINVAL, "%s: null pointer", who);
  if (a->B <= 0 || a->H <= 0 || a->Lq <= 0 || a->Lk <= 0 || a->Lk > 32 || (a->dh != 16 && a->dh != 32 && a->dh != 64))
    return mmt_set_error(MMT_EINVAL, "%s: need B,H,Lq > 0, 1 <= Lk <= 32, head dim 16, 32 or 64", who);
  if (a->p_rs < a->Lk || a->p_hs < (int64_t)(a->Lq - 1) * a->p_rs + a->Lk)
    return mmt_set_error(MMT_EINVAL, "%s: probs strides overlap (need p_rs >= Lk, p_hs >= (Lq-1)*p_rs + Lk)", who);
  if (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v) & 15 || (a->q_bs | a->q_rs | a->k_bs | a->k_rs | a->v_bs | a->v_rs) & 3)
    return mmt_set_error(MMT_EINVAL, "%s: q/k/v head slices must be 16-byte aligned (strides multiples of 4 floats)", who);
  return MMT_OK;
}

extern "C" int mmt_attention_fwd(const mmt_attn_args *a, void *stream) {
  if (int rc = check_attn(a, "mmt_attention_fwd")) return rc;
  if (!a->o) return mmt_set_error(MMT_EINVAL, "mmt_attention_fwd: null output");
  const int total = a->B * a->H * a->Lq;                 // groups of 16 lanes
  const dim3 grid((total + 15) / 16), blk(256);
  hipStream_t st = (hipStream_t)stream;
  // <= 16 keys and offsets within a buffer descriptor's range: every operand in one burst
  const double qmax = (double)a->B * (double)a->q_bs + (double)a->Lq * (double)a->q_rs;
  const double kmax = (double)a->B * (double)std::max(a->k_bs, a->v_bs) + (double)a->Lk * (double)std::max(a->k_rs, a->v_rs);
  const double pmax = (double)a->B * a->H * (double)a->p_hs + (double)a->Lq * (double)a->p_rs;
  const bool burst = !(train_generic() & 4) && a->Lk <= 16 && qmax < 5e8 && kmax < 5e8 && pmax < 5e8 && a->q_bs >= 0 && a->q_rs >= 0 && a->k_bs >= 0 &&
                     a->k_rs >= 0 && a->v_bs >= 0 && a->v_rs >= 0;
  if (burst && a->dh == 16) hipLaunchKernelGGL(mmt_attn_fwd16_kernel<16>, grid, blk, 0, st, *a);
  else if (burst && a->dh == 32) hipLaunchKernelGGL(mmt_attn_fwd16_kernel<32>, grid, blk, 0, st, *a);
  else if (burst) hipLaunchKernelGGL(mmt_attn_fwd16_kernel<64>, grid, blk, 0, st, *a);
  else if (a->dh == 16) hipLaunchKernelGGL(mmt_attn_fwd_kernel<16>, grid, blk, 0, st, *a);
  else if (a->dh == 32) hipLaunchKernelGGL(mmt_attn_fwd_kernel<32>, grid, blk, 0, st, *a);
  else hipLaunchKernelGGL(mmt_attn_fwd_kernel<64>, grid, blk, 0, st, *a);
  return launch_ok("mmt_attention_fwd");
}

extern "C" int mmt_attention_bwd(const mmt_attn_args *a, const float *d_o, long do_bs, long do_rs, float *dq, long dq_bs,
                                 long dq_rs, float *dk, float *dv, long dk_bs, long dk_rs, void *stream) {
  if (int rc = check_attn(a, "mmt_attention_bwd")) return rc;
  if (!d_o || !dq || !dk || !dv) return mmt_set_error(MMT_EINVAL, "mmt_attention_bwd: null gradient pointer");
  if (((uintptr_t)d_o & 15) || ((do_bs | do_rs) & 3))
    return mmt_set_error(MMT_EINVAL, "mmt_attention_bwd: d_o head slices must be 16-byte aligned");
  const int total = a->B * a->H;                         // groups of 16 lanes
  const dim3 grid((total + 15) / 16), blk(256);
  hipStream_t st = (hipStream_t)stream;
#define BWD(KERNEL, DHV) hipLaunchKernelGGL(KERNEL<DHV>, grid, blk, 0, st, *a, d_o, do_bs, do_rs, dq, dq_bs, dq_rs, dk, dv, dk_bs, dk_rs)
  // register-resident variant: few keys, 8-byte aligned per-lane slices, offsets within the buffer descriptors' range
  const bool small = a->Lk <= 16 && a->dh <= 32 && ((uintptr_t)dk & 15) == 0 && ((uintptr_t)dv & 15) == 0 && (dk_bs & 3) == 0 &&
                     (dk_rs & 3) == 0 && ((uintptr_t)a->k & 7) == 0 && ((a->k_bs | a->k_rs) & 1) == 0 &&
                     (double)a->B * (double)std::max(std::max(a->k_bs, a->v_bs), (int64_t)dk_bs) +
                             (double)a->Lk * (double)std::max(std::max(a->k_rs, a->v_rs), (int64_t)dk_rs) < 5e8;
  if (small && a->dh == 16) BWD(mmt_attn_bwd16_kernel, 16);
  else if (small) BWD(mmt_attn_bwd16_kernel, 32);
  else if (a->dh == 16) BWD(mmt_attn_bwd_kernel, 16);
  else if (a->dh == 32) BWD(mmt_attn_bwd_kernel, 32);
  else BWD(mmt_attn_bwd_kernel, 64);
#undef BWD
  return launch_ok("mmt_attention_bwd");
}

extern "C" int mmt_l2_loss_bwd(const float *pos, const float *y, int B, int ph, float *dpos, float *loss_sum, void *stream) {
  if (!pos || !y || !dpos || B <= 0 || ph <= 0) return mmt_set_error(MMT_EINVAL, "mmt_l2_loss_bwd: bad arguments");
  hipLaunchKernelGGL(mmt_l2_loss_bwd_kernel, dim3(loss_sum ? 1 : (B * ph + 255) / 256), dim3(loss_sum ? 1024 : 256), 0,
                     (hipStream_t)stream, pos, y, B, ph, dpos, loss_sum);
  return launch_ok("mmt_l2_loss_bwd");
}

extern "C" int mmt_adamw_multi(const mmt_adamw_tensor *table_dev, int ntensors, const float *state_dev, float beta1, float beta2,
                               float eps, float weight_decay, float clip, float grad_scale, const float *guard, void *stream) {
  if (!table_dev || !state_dev || ntensors <= 0) return mmt_set_error(MMT_EINVAL, "mmt_adamw_multi: bad arguments");
  hipLaunchKernelGGL(mmt_adamw_kernel, dim3(64, ntensors), dim3(256), 0, (hipStream_t)stream, table_dev, ntensors, state_dev,
                     beta1, beta2, eps, weight_decay, clip, grad_scale, guard);
  return launch_ok("mmt_adamw_multi");
}

extern "C" int mmt_history_prepare(const float *x_st, int rows, int n_in, float *x0, uint8_t *pad, void *stream) {
  if (!x_st || !x0 || !pad || rows <= 0 || n_in < 2) return mmt_set_error(MMT_EINVAL, "mmt_history_prepare: bad arguments");
  hipLaunchKernelGGL(mmt_history_prepare_kernel, dim3((rows * n_in + 255) / 256), dim3(256), 0, (hipStream_t)stream, x_st, rows,
                     n_in, x0, pad);
  return launch_ok("mmt_history_prepare");
}

extern "C" int mmt_dropout_mask(uint8_t *keep, long n, float p, const uint64_t *rng_state, uint32_t stream_id, void *stream) {
  if (!keep || !rng_state || n <= 0 || (n & 3) || ((uintptr_t)keep & 3) || !(p >= 0.f && p < 1.f) || stream_id >= (1u << 24))
    return mmt_set_error(MMT_EINVAL, "mmt_dropout_mask: need n > 0 a multiple of 4, a 4-byte aligned buffer, 0 <= p < 1");
  const uint32_t threshold = (uint32_t)std::min(4294967295.0, (double)p * 4294967296.0);
  const long n4 = n >> 2;
  hipLaunchKernelGGL(mmt_dropout_mask_kernel, dim3((unsigned)std::min<long>((n4 + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<uint32_t *>(keep), n4, threshold, rng_state, stream_id);
  return launch_ok("mmt_dropout_mask");
}
extern "C" int mmt_rng_advance(uint64_t *rng_state, void *stream) {
  if (!rng_state) return mmt_set_error(MMT_EINVAL, "mmt_rng_advance: null state");
  hipLaunchKernelGGL(mmt_rng_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rng_state);
  return launch_ok("mmt_rng_advance");
}
extern "C" int mmt_dropout_apply(const float *x, const uint8_t *keep, float scale, float *y, long n, void *stream) {
  if (!x || !keep || !y || n <= 0) return mmt_set_error(MMT_EINVAL, "mmt_dropout_apply: bad arguments");
  hipLaunchKernelGGL(mmt_dropout_apply_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, x,
                     keep, scale, y, n);
  return launch_ok("mmt_dropout_apply");
}

extern "C" int mmt_lr_step(float *state_dev, float gamma, const float *guard, float *skipped, void *stream) {
  if (!state_dev) return mmt_set_error(MMT_EINVAL, "mmt_lr_step: null state");
  hipLaunchKernelGGL(mmt_lr_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state_dev, gamma, guard, skipped);
  return launch_ok("mmt_lr_step");
}

extern "C" int mmt_softmax_axis(const float *x, long outer, int n, long inner, float *y, void *stream) {
  if (!x || !y || outer <= 0 || n <= 0 || inner <= 0) return mmt_set_error(MMT_EINVAL, "mmt_softmax_axis: bad arguments");
  const long total = outer * inner;
  hipLaunchKernelGGL(mmt_softmax_axis_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, outer, n,
                     inner, y);
  return launch_ok("mmt_softmax_axis");
}

extern "C" int mmt_cumsum_axis(const float *x, long outer, int n, long inner, float *y, void *stream) {
  if (!x || !y || outer <= 0 || n <= 0 || inner <= 0) return mmt_set_error(MMT_EINVAL, "mmt_cumsum_axis: bad arguments");
  const long total = outer * inner;
  hipLaunchKernelGGL(mmt_cumsum_axis_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, outer, n,
                     inner, y);
  return launch_ok("mmt_cumsum_axis");
}

extern "C" int mmt_l2_norm(const float *a, const float *b, long rows, int n, long rep, long inner, float *out, void *stream) {
  if (!a || !b || !out || rows <= 0 || n <= 0 || rep <= 0 || inner <= 0 || rows % (rep * inner))
    return mmt_set_error(MMT_EINVAL, "mmt_l2_norm: bad arguments (rows %ld must be a multiple of rep %ld x inner %ld)", rows, rep, inner);
  hipLaunchKernelGGL(mmt_l2_norm_kernel, dim3((unsigned)std::min<long>((rows + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, a,
                     b, rows, n, rep, inner, out);
  return launch_ok("mmt_l2_norm");
}

extern "C" int mmt_patchify(const float *img, int B, int C, int S, int P, float alpha, float beta, float *out, long ld, void *stream) {
  if (!img || !out || B <= 0 || C <= 0 || S <= 0 || P <= 0 || S % P || ld < (long)P * P * C)
    return mmt_set_error(MMT_EINVAL, "mmt_patchify: bad arguments (image %d, patch %d, row stride %ld)", S, P, ld);
  const long total = (long)B * S * S * C;
  hipLaunchKernelGGL(mmt_patchify_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, img,
                     B, C, S, P, alpha, beta, out, ld);
  return launch_ok("mmt_patchify");
}
extern "C" int mmt_gelu_fwd(const float *z, float *y, long n, void *stream) {
  if (!z || !y || n <= 0) return mmt_set_error(MMT_EINVAL, "mmt_gelu_fwd: bad arguments");
  hipLaunchKernelGGL(mmt_gelu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, y, n);
  return launch_ok("mmt_gelu_fwd");
}
extern "C" int mmt_gelu_bwd(const float *z, const float *dy, float *dz, long n, void *stream) {
  if (!z || !dy || !dz || n <= 0) return mmt_set_error(MMT_EINVAL, "mmt_gelu_bwd: bad arguments");
  hipLaunchKernelGGL(mmt_gelu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, dy, dz, n);
  return launch_ok("mmt_gelu_bwd");
}
extern "C" int mmt_copy3(const float *src, long sbs, long srs, float *dst, long dbs, long drs, int B, int R, int N, int accumulate,
                         void *stream) {
  if (!src || !dst || B <= 0 || R <= 0 || N <= 0 || sbs < 0 || srs < 0 || dbs < 0 || drs < N)
    return mmt_set_error(MMT_EINVAL, "mmt_copy3: bad arguments");
  const long total = (long)B * R * N;
  hipLaunchKernelGGL(mmt_copy3_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, src, sbs,
                     srs, dst, dbs, drs, B, R, N, accumulate);
  return launch_ok("mmt_copy3");
}

extern "C" int mmt_add_inplace(float *y, const float *x, long n, void *stream) {
  if (!x || !y || n <= 0) return mmt_set_error(MMT_EINVAL, "mmt_add_inplace: bad arguments");
  hipLaunchKernelGGL(mmt_add_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, x, n);
  return launch_ok("mmt_add_inplace");
}
