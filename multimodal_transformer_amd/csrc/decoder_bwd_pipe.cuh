// Reverse-time sweep of the training step as ONE weight-stationary pipeline launch: the backward twin of mmt_dec_pipe4_kernel.
// (Included by decoder_cluster.hip inside its anonymous namespace: it re-uses the 4x4x1 linears, the tagged hand-off and the
// bounded waits of the forward pipeline.)
//
// What it replaces: training.TrainEngine._sweep -- per (step, layer) six dependent row-block kernels (432 launches per step at
// ph = 12, L = 6; 2.56 of the step's 5.3 ms).  The chain through the fed-back positions (mgcvae.py:432-446; train.py:248
// loss.backward()) is inherently sequential -- ph x (3L + 1) dependent hops -- so, as in the forward, a workgroup is bound for
// the whole sweep to one (layer, sub-layer, slice), keeps the weights of ITS slice of dX = dY W in registers, and the 4-agent
// clusters flow through the stages in reverse layer and reverse time order:
//   stage C' (feed forward, 2 slices of H):  dy = sum of the parts + residual;  gf2 = LayerNorm3 backward(dy);
//                                            gf1[:, slice] = relu'(f1) * (gf2 W2[:, slice]);  part = gf1[:, slice] W1[slice, :]
//   stage B' (cross attention, head g):      go2 = LayerNorm2 backward;  dO_h = go2 Wo[:, head];  one-query attention backward
//                                            (dq -> gcq, dK / dV of the memory accumulated by this workgroup, which owns the head);
//                                            part = dq_h Wq[head, :]
//   stage A' (self attention, head g):       go1 = LayerNorm1 backward;  dO_h = go1 Wo[:, head];  attention backward over the keys
//                                            of steps <= s (dK_j / dV_j accumulated in place: complete when step j is reached);
//                                            part = [dq | gk_s | gv_s] Win[head rows, :]
//   head workgroups:                         gx[0][s+1] = sum of the parts + residual;  G[s+1] = dL/dpos[s+1] + G[s+2] +
//                                            gx[0][s+1] W_in (the position feedback);  decoder/MLP backward;  gx[L][s] = gho W_out
// Every stage also writes what the deferred weight-gradient GEMMs read (dy / dx of each LayerNorm, gf1, gcq, gqkv, the head's
// tape): the same buffers TrainEngine._sweep fills.  Phase p' = s' (3L) + k' with s' = ph - 1 - s and k' = 3 (L - 1 - layer) +
// {C': 0, B': 1, A': 2}; hand-off slots, tags and time-outs are those of the forward pipeline.
//
// Basic / Lane configurations (the head chain is decoder/MLP with one hidden layer); anything else stays on the launch-per-kernel
// sweep (mmt_sweep_pipe_supported).  Training-mode dropout: the DROP instantiation reads the step's keep bytes (requested before a
// visit's poll) where the reference's modules drop -- dropout1/2/3 (the sub-layer branch sees keep * scale * dx, the residual dx),
// the feed-forward hidden units, the attention probabilities, the positional encoding.

constexpr int BK_C = 0, BK_B = 1, BK_A = 2;

struct BwdPipeArgs {
  mmt_sweep_pipe a;                            // the caller's pointers (include/mmt_amd.h)
  float *part[2], *xres[2], *x0buf;
  unsigned *sync;                              // [0]: time-out word
  int NC, R, NHW, test_stall;
};

struct BwdLds {
  float *xs, *t1, *t2, *qs, *ps, *pp, *Ks, *Vs;
  int *abort;
};
constexpr int BW_LD1 = 3 * 32 + 8;             // row stride of t1 / t2 (<= 96 columns)
constexpr int BW_LDQ = 32 + 4;
constexpr int BW_MAXPOS = 2 * 4 * BW_LDQ / 8;  // clusters whose running position gradient a head workgroup keeps in LDS (its qs tile)
__host__ __device__ __forceinline__ size_t bwd_lds_floats(int d) {
  return (size_t)4 * (d + MMT_PAD) + 2 * 4 * BW_LD1 + 2 * 4 * BW_LD1 + 2 * 4 * BW_LDQ + 128 + 128 + 64 + 2 * (2 * 4 * KMAX * (32 + 4));
}
__device__ __forceinline__ BwdLds bwd_lds(float *smem, int d) {
  BwdLds L;
  L.xs = smem;                                 // [4][d + PAD]  dx rows (input of the first linear)
  L.t1 = L.xs + 4 * (d + MMT_PAD);             // [2][4][LD1]   dO of a head / hidden-slice gradient / gho
  L.t2 = L.t1 + 2 * 4 * BW_LD1;                // [2][4][LD1]   dq of a head (B') / dq | gk | gv (A')
  L.qs = L.t2 + 2 * 4 * BW_LD1;                // [2][4][LDQ]   the query rows
  L.ps = L.qs + 2 * 4 * BW_LDQ;                // [2][64]       scale * dS per (agent, key)
  L.pp = L.ps + 128;                           // [2][64]       P per (agent, key)
  L.abort = reinterpret_cast<int *>(L.pp + 128);
  L.Ks = L.pp + 128 + 64;                      // [2][4][KMAX][LDK]
  L.Vs = L.Ks + 2 * 4 * KMAX * (32 + 4);
  return L;
}

#ifndef BW_PROBES
#define BW_PROBES 1        // (3 probes in flight measured slower: 3.96 against 3.86 ms per training step)
#endif
#ifndef BW_PRE
#define BW_PRE 0
#endif
#ifndef BW_PROBE_SLEEP
#define BW_PROBE_SLEEP 3
#endif
// slot of phase q (as the forward pipeline: 2 buffers x 4 row blocks of the 16-row cluster slots)
__device__ __forceinline__ size_t bwd_rowblock(unsigned q, int d) { return (size_t)4 * ((q >> 1) & 3u) * d; }

// ---- hand-off: poll the previous phase's rows until every word carries its tag (poll_rows without the ClArgs block) ------------
// X0: one full row from x0buf (the head's gx[L][s]); else residual row + the parts at stride GS
// the loads of one full attempt
template <bool X0, int GS>
__device__ __forceinline__ void bwd_poll_issue(Red4Regs &R, const float *res_base, const float *part_base, int d, int H) {
  const int row = wave_id(), q = lane_id();
  // parts of a cluster: [16 rows][H][d] -- the H parts of a row are 4 H d contiguous bytes (with [H][16][d] the 8 loads of a lane
  // were 16 KB apart: a 9-row fetch took 3.2 us where a 5-row one took 1.4)
  const unsigned co = q < (d >> 2) ? (unsigned)q * 16u : OOB, gstride = (unsigned)d * 4u;
  R.res = bl4x<AUX_SC1>(mk_rsrc(res_base + (size_t)row * d), co);
  if constexpr (!X0) {
    const rsrc_t pp = mk_rsrc(part_base + (size_t)row * H * d);
#pragma unroll
    for (int g = 0; g < 8 / GS; ++g) R.p[g] = bl4x<AUX_SC1>(pp, g * GS < H ? co : OOB, g * GS < H ? g * GS * gstride : 0u);
  }
}
// PRE: the first attempt's loads were issued by the caller (bwd_poll_issue) AHEAD of the visit's other requests -- vector memory
// returns in order, and behind a dozen loads that come from HBM the first look at the rows was 2 us late even when they were there
template <bool X0, int GS, bool PRE = false>
__device__ __forceinline__ bool bwd_poll_rows(Red4Regs &R, const float *res_base, const float *part_base, int d, int H, unsigned tag,
                                              const PipeSync &ps, int phase, int cluster, int visit = 0) {
  (void)visit;
  const int row = wave_id(), q = lane_id();
  const bool live = q < (d >> 2);
  const unsigned probe = q < 4 ? (unsigned)q * 16u : OOB;
  const rsrc_t res = mk_rsrc(res_base + (size_t)row * d);
  const rsrc_t pp = mk_rsrc(part_base + (size_t)row * H * d);
  unsigned long long t0 = 0;
  for (int tries = 0;; ++tries) {
    asm volatile("" ::: "memory");
    if (!(PRE && tries == 0)) bwd_poll_issue<X0, GS>(R, res_base, part_base, d, H);
    bool ok = !live || tag_ok(R.res, tag);
    if constexpr (!X0) {
#pragma unroll
      for (int g = 0; g < 8 / GS; ++g) ok = ok && (g * GS >= H || !live || tag_ok(R.p[g], tag));
#pragma unroll
      for (int g = 8 / GS; g < 8; ++g) R.p[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int g = 0; g < 8; ++g) R.p[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#ifdef MMT_PROFILE
    if (tries == 0) PSTAMP(visit, 9);
    if (__all(ok)) {
      const int v_ = (visit >> 12) * 16 + (visit & 4095);
      if (thread_idx() == 0 && blockIdx.x < PPR_WGS && (visit & 4095) < 16 && v_ < PPR_VISITS) g_pprof[blockIdx.x][v_][10] = (unsigned long long)tries;
    }
#endif
    if (__all(ok)) return true;
    if (!t0) t0 = wall_clock64();
    if (tries > 0) {
      __builtin_amdgcn_s_sleep(1);
      if ((tries & 63) == 0 && wall_clock64() - t0 > PIPE_WAIT_TICKS) {
        if (q == 0) pipe_give_up(ps, phase + 1000 * row + 500000, cluster);
        return false;
      }
      continue;
    }
    // thin polling: 64 bytes of the part that arrives last -- slice 0's: that workgroup also writes the residual row and the two
    // stash rows of the stage (measured with the last slice's part as the probe: the full fetch behind a matching probe usually met
    // a missing part 0 and was repeated two or three times, 0.85 us each)
#if BW_PROBES <= 1
    for (unsigned n = 0;; ++n) {
      __builtin_amdgcn_s_sleep(2);
      asm volatile("" ::: "memory");
      const f32x4 v = X0 ? bl4x<AUX_SC1>(res, probe) : bl4x<AUX_SC1>(pp, probe, 0u);
      if (__all(q >= 4 || tag_ok(v, tag))) break;
      if ((n & 15u) == 15u) {
        const bool late = wall_clock64() - t0 > PIPE_WAIT_TICKS;
        if (late || __hip_atomic_load(ps.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          if (q == 0) {
            if (late) pipe_give_up(ps, phase + 1000 * row + 500000 + (X0 ? 100000 : 0), cluster);
            else *ps.abort = 1;
          }
          return false;
        }
      }
    }
#else
    // BW_PROBES probes are kept in flight: a probe takes a memory round trip (past the L2: ~0.7 us), and every workgroup of a stage
    // notices its input at a random phase of its probe period -- the consumer of 8 parts waits for the last of 8 such phases
    {
      auto probe_load = [&]() -> f32x4 { return X0 ? bl4x<AUX_SC1>(res, probe) : bl4x<AUX_SC1>(pp, probe, 0u); };
      f32x4 pv[BW_PROBES];
#pragma unroll
      for (int i = 0; i < BW_PROBES; ++i) {
        asm volatile("" ::: "memory");
        pv[i] = probe_load();
      }
      bool out = false;
      for (unsigned n = 0; !out; ++n) {
#pragma unroll
        for (int i = 0; i < BW_PROBES; ++i) {
          if (out) break;
          const f32x4 v = pv[i];
          asm volatile("" ::: "memory");
          pv[i] = probe_load();              // (issued before v is looked at: the others stay in flight)
          asm volatile("" ::: "memory");
          if (__all(q >= 4 || tag_ok(v, tag))) out = true;
          else __builtin_amdgcn_s_sleep(BW_PROBE_SLEEP);
        }
        if (!out && (n & 7u) == 7u) {
          const bool late = wall_clock64() - t0 > PIPE_WAIT_TICKS;
          if (late || __hip_atomic_load(ps.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            if (q == 0) {
              if (late) pipe_give_up(ps, phase + 1000 * row + 500000 + (X0 ? 100000 : 0), cluster);
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

// ---- LayerNorm backward of a row on one wave (lane q: columns 4q..4q+3) ------------------------------------------------------
struct BwdRow {
  f32x4 xhat;
  float rstd;
};
__device__ __forceinline__ void bwd_row_issue(BwdRow &X, const float *xhat_s, const float *rstd_s, int ag, int B, int d) {
  const int q = lane_id();
  const bool ok = ag < B && q < (d >> 2);
  X.xhat = bl4(mk_rsrc(xhat_s), ok ? (unsigned)(ag * d + 4 * q) * 4u : OOB);
  X.rstd = bl1(mk_rsrc(rstd_s), ag < B ? (unsigned)ag * 4u : OOB);
}
// dy = residual + parts (fixed order); dx = rstd (dy g - mean(dy g) - xhat mean(dy g xhat)); dx -> xs row, tagged residual row for the
// next stage and the two stash rows (write: slice 0 of the stage)
// keep_scale > 0 (training-mode dropout on the sub-layer's output, dropout1/2/3): the sub-layer branch sees dxm = keep * scale * dx
// (xs row and the dx stash row, which feeds the weight gradients); the residual path keeps dx
__device__ __forceinline__ void bwd_ln_finish(const Red4Regs &R, const BwdRow &X, const f32x4 &gam, float *xs_row, bool write,
                                              float *xres_out_row, unsigned tag_out, float *gdy_row, float *gdx_row, bool in_batch,
                                              int d, unsigned keep = 0u, float keep_scale = 0.f) {
  const int q = lane_id();
  const bool live = q < (d >> 2);
  f32x4 dy = R.res;
#pragma unroll
  for (int g = 0; g < 8; ++g) dy += R.p[g];
  f32x4 gg, dx;
#pragma unroll
  for (int e = 0; e < 4; ++e) gg[e] = dy[e] * gam[e];
  const float inv_d = 1.0f / (float)d;
  const float m1 = wave64_total((gg[0] + gg[1]) + (gg[2] + gg[3])) * inv_d;
  const float m2 = wave64_total((gg[0] * X.xhat[0] + gg[1] * X.xhat[1]) + (gg[2] * X.xhat[2] + gg[3] * X.xhat[3])) * inv_d;
#pragma unroll
  for (int e = 0; e < 4; ++e) dx[e] = X.rstd * (gg[e] - m1 - X.xhat[e] * m2);
  f32x4 dxm = dx;
  if (keep_scale > 0.f) dxm = dx * keep4w(keep, keep_scale);
  if (live) {
    *reinterpret_cast<f32x4 *>(xs_row + 4 * q) = dxm;
    if (write) {
      bs4x<AUX_SC1>(mk_rsrc(xres_out_row), (unsigned)q * 16u, tag4(dx, tag_out));
      if (in_batch) {
        *reinterpret_cast<f32x4 *>(gdy_row + 4 * q) = dy;
        *reinterpret_cast<f32x4 *>(gdx_row + 4 * q) = dxm;
      }
    }
  }
}

// ---- first linear with the K parts summed: lane (DPP row r = b >> 2, in-row block bb = b & 3, j) gets row r of column
// (w * cgw + bb) * 4 + j (valid for bb < cgw); same products and sums as gemm1_4s
template <int DH, int PHASE, int D, int FS>
__device__ __forceinline__ float bwd_gemm1_tot(const MainRegs4<DH, PHASE> &M, const float *xs, int ldx) {
  typedef Geom4<PHASE> G4;
  typedef Lin4S<(PHASE == PH_C ? FS : DH) / 4, D / 4> L1;
  static_assert(G4::NPART == 1 && L1::q <= G4::QMAX1, "one-part first linear");
  constexpr int KQ = D / 4;
  const int lane = lane_id(), b = lane >> 2, j = lane & 3;
  const int kq0 = (b >> L1::lg_cgw) * L1::q;
  const float *xr = xs + j * ldx;
  f32x4 xv[L1::q];
#pragma unroll
  for (int t = 0; t < L1::q; ++t) xv[t] = *reinterpret_cast<const f32x4 *>(xr + 4 * min(kq0 + t, KQ - 1));
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < L1::q; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[t][e], M.w1[0][t][e], acc, 0, 0, 0);
  return kparts_sum_rows<L1::cgw>(acc);
}

// the stationary slices of a stage: w1 = first linear (may be NULL: none), w2 = second linear, slice g; no bias in a backward
template <int DH, int PHASE>
__device__ __forceinline__ void bwd_weights_issue(MainRegs4<DH, PHASE> &M, const float *w1, const float *w2, int g, int d, int FS) {
  typedef Geom4<PHASE> G4;
  static_assert(G4::NPART == 1, "bwd stages use one-part linears");
  const int w = wave_id(), lane = lane_id();
  const unsigned lane_off = lane * 16u;
  M.b1 = 0.f;
  M.pad = 0;
  if (w1) {
    const Lin4 L1 = lin4((PHASE == PH_C ? FS : DH) / 4, d / 4);
    const rsrc_t r1 = mk_rsrc(w1);
#pragma unroll
    for (int t = 0; t < G4::QMAX1; ++t)
      M.w1[0][t] = bl4s(r1, t < L1.q ? lane_off : OOB, t < L1.q ? (unsigned)(((g * L1.q + t) * 8 + w) * 1024) : 0u);
  }
  {
    const Lin4 L2 = lin4(d / 4, (PHASE == PH_C ? FS : DH) / 4);
    const rsrc_t r2 = mk_rsrc(w2);
#pragma unroll
    for (int t = 0; t < G4::QMAX2; ++t)
      M.w2[t] = bl4s(r2, t < L2.q ? lane_off : OOB, t < L2.q ? (unsigned)(((g * L2.q + t) * 8 + w) * 1024) : 0u);
  }
}

// ---- one-query attention backward of one head for the 4 agents of a cluster, on ONE wave: lane = (agent ar, key j) -------------
// in : dO rows (do_ + ar * ldo), K / V staging tiles (rows beyond the keys are zeros), P = this lane's probability (0 beyond the keys)
// out: ps[lane] = scale * dS, pp[lane] = P, dq rows (dq + ar * ldq)
// att_scale > 0 (dropout on the probabilities): the forward used keep * att_scale * P, so dP = keep * att_scale * (dO . V_j) and
// the value gradient takes the dropped probability (pp)
template <int DH>
__device__ __forceinline__ void attend4_bwd(int na, int nkeys, float P, const float *do_, int ldo, const float *Ks, const float *Vs,
                                            float *ps, float *pp, float *dq, int ldq, unsigned att_keep = 1u, float att_scale = 0.f) {
  constexpr int LDK = DH + 4, DPL = DH / 16;
  const int lane = lane_id(), ar = lane >> 4, l16 = lane & 15;
  const bool valid = ar < na && l16 < nkeys;
  float dP = 0.f;
  {
    const float *o = do_ + ar * ldo, *v = Vs + (ar * KMAX + l16) * LDK;
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) {
      const f32x4 ov = *reinterpret_cast<const f32x4 *>(o + 4 * c), vv = *reinterpret_cast<const f32x4 *>(v + 4 * c);
      dP = fmaf(ov[0], vv[0], dP);
      dP = fmaf(ov[1], vv[1], dP);
      dP = fmaf(ov[2], vv[2], dP);
      dP = fmaf(ov[3], vv[3], dP);
    }
  }
  const float Pv = valid ? P : 0.f;
  float Pd = Pv;                                   // the probability the forward multiplied the value by
  if (att_scale > 0.f) {
    const float kf = (att_keep & 0xffu) ? att_scale : 0.f;
    dP *= kf;
    Pd = Pv * kf;
  }
  const float sum = row16_sum(Pv * dP);
  const float ds = Pv * (dP - sum) * (1.0f / sqrtf((float)DH));
  ps[lane] = ds;
  pp[lane] = Pd;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  typedef float vrow __attribute__((ext_vector_type(DPL)));
  float o[DPL];
#pragma unroll
  for (int e = 0; e < DPL; ++e) o[e] = 0.f;
  f32x4 pr[KMAX / 4];
  vrow kk[KMAX];
#pragma unroll
  for (int c = 0; c < KMAX / 4; ++c) pr[c] = *reinterpret_cast<const f32x4 *>(ps + ar * 16 + 4 * c);
#pragma unroll
  for (int j = 0; j < KMAX; ++j) kk[j] = *reinterpret_cast<const vrow *>(Ks + (ar * KMAX + j) * LDK + l16 * DPL);
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    const float pj = pr[j >> 2][j & 3];
#pragma unroll
    for (int e = 0; e < DPL; ++e) o[e] = fmaf(pj, kk[j][e], o[e]);
  }
#pragma unroll
  for (int e = 0; e < DPL; ++e) dq[ar * ldq + l16 * DPL + e] = o[e];
}

// ---- stages -------------------------------------------------------------------------------------------------------------------
// Stage kind KIND of `layer`, slices / heads g0 .. g0 + NH - 1, replica `replica`.
template <int DH, int KIND, int NH, int D, int FS, bool DROP>
__device__ __forceinline__ void bwd_stage(const BwdPipeArgs &f, int layer, int g0, int replica, const BwdLds &S, const PipeSync &ps) {
  constexpr int PHASE = KIND == BK_C ? PH_C : PH_B;          // register-tile geometry of the linears
  constexpr int NM = KIND == BK_A ? 3 : NH;                  // A': one head, three second linears (q | k | v rows of in_proj)
  constexpr int LDK = DH + 4, LD1 = BW_LD1, LDQ = BW_LDQ;
  static_assert(KIND != BK_A || NH == 1, "self-attention stage: one head per workgroup");
  static_assert(D == 256, "one float4 per lane and row");
  const mmt_sweep_pipe &A = f.a;
  const mmt_sweep_layer &W = A.layer[layer];
  const int d = D, B = A.B, T = A.T, ph = A.ph, H = A.H, ldx = d + MMT_PAD, tid = thread_idx(), lane = lane_id(), wv = wave_id();
  const int L = A.nlayers, NP = 3 * L, kst = 3 * (L - 1 - layer) + KIND;
  const bool loadx = KIND == BK_C && layer == L - 1;         // the first stage of a step takes the head's full row
  // ---- once: the weight slices, the LayerNorm weights of the reducing lanes, zeroed staging tiles
  MainRegs4<DH, PHASE> M[NM];
#pragma unroll
  for (int h = 0; h < NM; ++h) M[h] = {};
  if constexpr (KIND == BK_A) {
    bwd_weights_issue<DH, PHASE>(M[0], W.w1[BK_A], W.w2[BK_A], g0, d, FS);
    bwd_weights_issue<DH, PHASE>(M[1], nullptr, W.w2k, g0, d, FS);
    bwd_weights_issue<DH, PHASE>(M[2], nullptr, W.w2v, g0, d, FS);
  } else {
#pragma unroll
    for (int h = 0; h < NH; ++h) bwd_weights_issue<DH, PHASE>(M[h], W.w1[KIND], W.w2[KIND], g0 + h, d, FS);
  }
  const f32x4 gam = bl4(mk_rsrc(W.gamma[KIND]), (wv < 4 && lane < (d >> 2)) ? (unsigned)lane * 16u : OOB);
  for (int i = tid; i < 2 * (2 * 4 * KMAX * LDK); i += CNT) S.Ks[i] = 0.f;       // Ks and Vs are adjacent
  for (int i = tid; i < 2 * 4 * LD1; i += CNT) S.t2[i] = 0.f;
  __syncthreads();
  const bool red_wave = wv < 4;
#pragma clang loop unroll(disable)
  for (int sp = 0; sp < ph; ++sp) {
    const int s = ph - 1 - sp;
    const unsigned p = (unsigned)(sp * NP + kst);
    const unsigned tag_out = tag_of_phase(p), tag_in = tag_of_phase(p - 1u);
    float *part_out = f.part[p & 1] + bwd_rowblock(p, d) * H, *xres_out = f.xres[p & 1] + bwd_rowblock(p, d);
    const float *part_in = f.part[(p + 1) & 1] + bwd_rowblock(p - 1u, d) * H, *xres_in = f.xres[(p + 1) & 1] + bwd_rowblock(p - 1u, d);
    const size_t srow = (size_t)s * B;             // first row of step s in a step-major (ph, B, .) stash
    int li = 0;
#pragma clang loop unroll(disable)
    for (int m = replica; m < f.NC; m += f.R, ++li) {
      if (f.test_stall && sp == 1 && layer == 1 && KIND == BK_B && g0 == 0 && m == replica) return;   // its successors must time out
      const int a0 = m * 4, na = min(4, B - a0);
      const int visit = sp * 4096 + li;
      (void)visit;
      PSTAMP(visit, 0);
      // ---- 0. everything that does not depend on the previous phase is requested before the poll.  (BW_PRE: the first look at the
      // handed-off rows issued ahead of these requests - vector memory returns in order - measured SLOWER, 3.85 against 3.82 ms per
      // training step: the stage that waits gains nothing, and the rows rarely are there at the first look)
      Red4Regs R;
      if (BW_PRE && red_wave) {
        if (loadx) bwd_poll_issue<true, 1>(R, f.x0buf + (size_t)m * 4 * d, nullptr, d, H);
        else if constexpr (KIND == BK_B) bwd_poll_issue<false, 2>(R, xres_in + (size_t)m * 16 * d, part_in + (size_t)m * H * 16 * d, d, H);
        else bwd_poll_issue<false, 1>(R, xres_in + (size_t)m * 16 * d, part_in + (size_t)m * H * 16 * d, d, H);
      }
      __builtin_amdgcn_sched_barrier(0);           // (the compiler must not move the requests below ahead of these)
      BwdRow X;
      if (red_wave) bwd_row_issue(X, W.xhat[KIND] + srow * d, W.rstd[KIND] + srow, a0 + wv, B, d);
      unsigned k_ln = 0u, k_att = 1u;
      float s_ln = 0.f, s_ff = 1.f, s_att = 0.f;
      if constexpr (DROP) {      // this visit's keep bytes: requested before the poll like everything else that is not handed over
        const uint8_t *kl = W.k_ln[KIND];
        if (kl) s_ln = W.s_ln[KIND];
        if (red_wave) k_ln = __builtin_amdgcn_raw_buffer_load_b32(mk_rsrc(kl), (kl && lane < (d >> 2) && a0 + wv < B)
                                                                    ? (unsigned)((srow + a0 + wv) * d + 4 * lane) : OOB, 0, 0);
        if constexpr (KIND == BK_C) {
          if (W.k_ff) s_ff = W.s_ff;
        } else {
          const uint8_t *ka = W.k_att[KIND == BK_B ? 0 : 1];
          if (ka) {
            s_att = W.s_att[KIND == BK_B ? 0 : 1];
            const int ar = lane >> 4, l16 = lane & 15, Lk = KIND == BK_B ? T : ph, nk = KIND == BK_B ? T : s + 1;
            k_att = __builtin_amdgcn_raw_buffer_load_b8(mk_rsrc(ka), (wv < NH && ar < na && l16 < nk)
                                                            ? (unsigned)((((a0 + ar) * H + g0 + wv) * ph + s) * Lk + l16) : OOB, 0, 0);
          }
        }
      }
      float f1v[NH];
#pragma unroll
      for (int h = 0; h < NH; ++h) f1v[h] = 0.f;
      f32x4 kvr[NH][DH / 16], oldr[NH][DH / 16], qr = (f32x4){0.f, 0.f, 0.f, 0.f};
      float Pl = 0.f;
      const int nkeys = KIND == BK_B ? T : s + 1;
      if constexpr (KIND == BK_C) {
        constexpr int cgw = (FS / 4) >> 3;
        const int b = lane >> 2, j = lane & 3, r = b >> 2, bb = b & 3;
#pragma unroll
        for (int h = 0; h < NH; ++h)
          f1v[h] = bl1(mk_rsrc(W.f1 + srow * (size_t)(FS * H)), (bb < cgw && r < na) ? (unsigned)((a0 + r) * (FS * H) + (g0 + h) * FS + (wv * cgw + bb) * 4 + j) * 4u : OOB);
      } else {
        // the accumulator rows are read back by the thread that stored them; with one cluster per replica the previous visit's stores
        // go to the very same addresses: let them leave first (otherwise the next visit is to another cluster)
        if (f.NC <= f.R) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const int g = g0 + h;
#pragma unroll
          for (int it = 0; it < DH / 16; ++it) {
            int ag, j, kv, c4;
            kv_unit<DH>(tid + CNT * it, ag, j, kv, c4);
            const bool ok = ag < na && j < nkeys;
            if constexpr (KIND == BK_B) {
              const unsigned off = (unsigned)(((a0 + ag) * T + j) * 2 * d + kv * d + g * DH + 4 * c4) * 4u;
              kvr[h][it] = bl4(mk_rsrc(W.mkv), ok ? off : OOB);
              oldr[h][it] = (f32x4){0.f, 0.f, 0.f, 0.f};      // (the accumulator rows are read behind the attention: below)
            } else {
              const unsigned off = (unsigned)((j * B + a0 + ag) * 3 * d + (1 + kv) * d + g * DH + 4 * c4) * 4u;
              kvr[h][it] = bl4(mk_rsrc(W.qkv), ok ? off : OOB);
              oldr[h][it] = bl4(mk_rsrc(W.gqkv), ok ? off : OOB);
            }
          }
        }
        if (wv < NH) {       // the attention wave of head g0 + wv: its probabilities (lane = agent, key) and query rows (lanes 0..31)
          const int g = g0 + wv, ar = lane >> 4, l16 = lane & 15;
          const bool okp = ar < na && l16 < nkeys;
          const int ar8 = lane >> 3, c4 = lane & 7;
          const bool okq = lane < 32 && ar8 < na;
          if constexpr (KIND == BK_B) {
            Pl = bl1(mk_rsrc(W.p2), okp ? (unsigned)((((a0 + ar) * H + g) * ph + s) * T + l16) * 4u : OOB);
            qr = bl4(mk_rsrc(W.cq + srow * d), okq ? (unsigned)((a0 + ar8) * d + g * DH + 4 * c4) * 4u : OOB);
          } else {
            Pl = bl1(mk_rsrc(W.p1), okp ? (unsigned)((((a0 + ar) * H + g) * ph + s) * ph + l16) * 4u : OOB);
            qr = bl4(mk_rsrc(W.qkv + srow * 3 * d), okq ? (unsigned)((a0 + ar8) * 3 * d + g * DH + 4 * c4) * 4u : OOB);
          }
        }
      }
      // ---- 1. the previous phase's rows
      if (red_wave) {
        if (loadx) (void)bwd_poll_rows<true, 1, BW_PRE != 0>(R, f.x0buf + (size_t)m * 4 * d, nullptr, d, H, (unsigned)sp & TAG_MASK, ps, (int)p, m, visit);
        else if constexpr (KIND == BK_B) (void)bwd_poll_rows<false, 2, BW_PRE != 0>(R, xres_in + (size_t)m * 16 * d, part_in + (size_t)m * H * 16 * d, d, H, tag_in, ps, (int)p, m, visit);
        else (void)bwd_poll_rows<false, 1, BW_PRE != 0>(R, xres_in + (size_t)m * 16 * d, part_in + (size_t)m * H * 16 * d, d, H, tag_in, ps, (int)p, m, visit);
      }
      PSTAMP(visit, 1);
      // ---- 2. LayerNorm backward -> xs; K / V rows and the queries -> staging tiles
      if (red_wave) {
        const size_t ro = (srow + a0 + wv) * d;
        bwd_ln_finish(R, X, gam, S.xs + wv * ldx, g0 == 0, xres_out + ((size_t)m * 16 + wv) * d, tag_out, W.gdy[KIND] + ro, W.gdx[KIND] + ro,
                      a0 + wv < B, d, k_ln, s_ln);
      }
      if constexpr (KIND != BK_C) {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
          for (int it = 0; it < DH / 16; ++it) {
            int ag, j, kv, c4;
            kv_unit<DH>(tid + CNT * it, ag, j, kv, c4);
            // (rows j >= nkeys of agents < na may hold an earlier visit's keys: zero them, attend4_bwd reads whole tiles)
            if (ag < 4) *reinterpret_cast<f32x4 *>((kv ? S.Vs : S.Ks) + ((h * 4 + ag) * KMAX + j) * LDK + 4 * c4) = kvr[h][it];
          }
        if (wv < NH && lane < 32) *reinterpret_cast<f32x4 *>(S.qs + (wv * 4 + (lane >> 3)) * LDQ + 4 * (lane & 7)) = qr;
      }
      PSTAMP(visit, 7);
      __syncthreads();
      if (*ps.abort) return;
      PSTAMP(visit, 2);
      // ---- 3. first linear
      if constexpr (KIND == BK_C) {
        constexpr int cgw = (FS / 4) >> 3;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const float tot = bwd_gemm1_tot<DH, PHASE, D, FS>(M[h], S.xs, ldx);
          const int b = lane >> 2, j = lane & 3, r = b >> 2, bb = b & 3;
          if (bb < cgw) {
            const int col = (wv * cgw + bb) * 4 + j;
            // relu'(hidden) (rows beyond the batch: f1v = 0); with dropout on the hidden units the stash holds keep * scale * relu(.),
            // whose derivative is scale where it is positive
            const float v = f1v[h] > 0.f ? (DROP ? tot * s_ff : tot) : 0.f;
            S.t1[(h * 4 + r) * LD1 + col] = v;
            if (r < na) W.gf1[(srow + a0 + r) * (size_t)(FS * H) + (g0 + h) * FS + col] = v;
          }
        }
      } else {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const float tot = bwd_gemm1_tot<DH, PHASE, D, FS>(M[h], S.xs, ldx);
          const int b = lane >> 2, j = lane & 3, r = b >> 2, bb = b & 3;
          if (bb < 1) S.t1[(h * 4 + r) * LD1 + wv * 4 + j] = tot;              // dO of head g0 + h
        }
      }
      __syncthreads();
      PSTAMP(visit, 3);
      auto kv_grad_units = [&]() {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const int g = g0 + h;
#pragma unroll
          for (int it = 0; it < DH / 16; ++it) {
            int ag, j, kv, c4;
            kv_unit<DH>(tid + CNT * it, ag, j, kv, c4);
            if (ag < na && j < nkeys) {
              const float wgt = (kv ? S.pp : S.ps)[h * 64 + ag * 16 + j];
              const f32x4 src = kv ? *reinterpret_cast<const f32x4 *>(S.t1 + (h * 4 + ag) * LD1 + 4 * c4)
                                   : *reinterpret_cast<const f32x4 *>(S.qs + (h * 4 + ag) * LDQ + 4 * c4);
              f32x4 nv = oldr[h][it];
#pragma unroll
              for (int e = 0; e < 4; ++e) nv[e] = fmaf(wgt, src[e], nv[e]);
              if constexpr (KIND == BK_B) {
                *reinterpret_cast<f32x4 *>(W.gmkv + (size_t)((a0 + ag) * T + j) * 2 * d + kv * d + g * DH + 4 * c4) = nv;
              } else if constexpr (KIND == BK_A) {
                *reinterpret_cast<f32x4 *>(W.gqkv + (size_t)(j * B + a0 + ag) * 3 * d + (1 + kv) * d + g * DH + 4 * c4) = nv;
                if (j == s) *reinterpret_cast<f32x4 *>(S.t2 + ag * LD1 + (1 + kv) * DH + 4 * c4) = nv;     // row s is complete now
              }
            }
          }
        }
      };
      // ---- 4. attention backward (wave h: head g0 + h), then the K / V gradient units on every thread
      if constexpr (KIND != BK_C) {
        if (wv < NH) {
          const int h = wv;
          attend4_bwd<DH>(na, nkeys, Pl, S.t1 + h * 4 * LD1, LD1, S.Ks + h * 4 * KMAX * LDK, S.Vs + h * 4 * KMAX * LDK, S.ps + h * 64,
                          S.pp + h * 64, S.t2 + h * 4 * LD1, LD1, k_att, s_att);
          // dq -> stash (gcq / the q columns of gqkv): lane (ar, l16) holds DH/16 = 2 consecutive columns
          const int ar = lane >> 4, l16 = lane & 15, g = g0 + h;
          if (ar < na) {
            const float2 v2 = *reinterpret_cast<const float2 *>(S.t2 + (h * 4 + ar) * LD1 + l16 * 2);
            if constexpr (KIND == BK_B) *reinterpret_cast<float2 *>(W.gcq + (srow + a0 + ar) * d + g * DH + l16 * 2) = v2;
            else *reinterpret_cast<float2 *>(W.gqkv + (srow + a0 + ar) * 3 * d + g * DH + l16 * 2) = v2;
          }
        }
        __syncthreads();
        // dK_j += scale dS_j q, dV_j += P_j dO: the workgroup owns its heads' rows (read-modify-write, no atomics).  The self-attention
        // stage needs row s before its second linear; the cross-attention stage does this behind the hand-off (below)
        PSTAMP(visit, 8);
        if constexpr (KIND == BK_B) {      // the memory K / V gradient rows this visit adds to: needed behind the hand-off only
#pragma unroll
          for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int it = 0; it < DH / 16; ++it) {
              int ag, j, kv, c4;
              kv_unit<DH>(tid + CNT * it, ag, j, kv, c4);
              oldr[h][it] = bl4(mk_rsrc(W.gmkv), (ag < na && j < nkeys) ? (unsigned)(((a0 + ag) * T + j) * 2 * d + kv * d + (g0 + h) * DH + 4 * c4) * 4u : OOB);
            }
        }
        if constexpr (KIND == BK_A) kv_grad_units();
        if constexpr (KIND == BK_A) __syncthreads();
      }
      PSTAMP(visit, 4);
      // ---- 5. second linear: one tagged part per head / slice, write-through
      {
        typedef Lin4S<D / 4, (PHASE == PH_C ? FS : DH) / 4> L2;
        const int b = lane >> 2, j = lane & 3;
        f32x4 acc;
        if constexpr (KIND == BK_C) {
          // two feed-forward slices per workgroup: ONE pre-summed part (in the slot of the even slice) - half the rows for the
          // cross-attention stage to fetch; that stage is the sweep's busiest, so it has one head per workgroup and this one two slices
          static_assert(NH == 2, "feed-forward stage: two slices, one pre-summed part");
          acc = tag4(gemm2_4s_acc<DH, PHASE, D, FS>(M[0], S.t1, LD1), tag_out) + tag4(gemm2_4s_acc<DH, PHASE, D, FS>(M[1], S.t1 + 4 * LD1, LD1), tag_out);
        } else if constexpr (KIND == BK_B) {
          static_assert(NH == 1, "cross-attention stage: one head per workgroup");
          acc = gemm2_4s_acc<DH, PHASE, D, FS>(M[0], S.t2, LD1);
        } else {
          acc = gemm2_4s_acc<DH, PHASE, D, FS>(M[0], S.t2, LD1);
          acc += gemm2_4s_acc<DH, PHASE, D, FS>(M[1], S.t2 + DH, LD1);
          acc += gemm2_4s_acc<DH, PHASE, D, FS>(M[2], S.t2 + 2 * DH, LD1);
        }
        quad_transpose(acc, j);                      // lane j of a quad: row j, the quad's 4 columns -> one 16-byte store
        acc = tag4(acc, tag_out);
        // part g0 of row j of the cluster: [16 rows][H][d]
        bs4x<AUX_SC1>(mk_rsrc(part_out + (size_t)m * 16 * H * d + (size_t)g0 * d), b < L2::cgw ? (unsigned)(j * H * D + (wv * L2::cgw + b) * 4) * 4u : OOB, acc);
        if constexpr (KIND == BK_B) kv_grad_units();      // off the hop: the part is on its way
      }
      PSTAMP(visit, 5);
      // (the next visit's barrier behind its poll orders this visit's LDS reads before the next writes of xs / t1 / tiles ... except
      // the staging tiles, which are written before that barrier: one more barrier here)
      __syncthreads();
      PSTAMP(visit, 6);
    }
  }
}

// ---- head workgroups: close step s + 1 (the position feedback) and open step s (decoder/MLP backward, output_fc backward) ------
template <int DH, int D, int FS, bool DROP>
__device__ __forceinline__ void bwd_head(const BwdPipeArgs &f, int replica, int jh, const BwdLds &S, const PipeSync &ps) {
  constexpr int LD1 = BW_LD1;
  const mmt_sweep_pipe &A = f.a;
  const int d = D, B = A.B, ph = A.ph, H = A.H, L = A.nlayers, NP = 3 * L, NO = A.n_out, lane = lane_id(), wv = wave_id(), tid = thread_idx();
  // once: output_fc^T as the second linear of a PH_B tile (K = n_out = DH), W_in for the lane's 4 columns, the head MLP's columns
  MainRegs4<DH, PH_B> Mo = {};
  bwd_weights_issue<DH, PH_B>(Mo, nullptr, A.w_ofc, 0, d, FS);
  const bool red_wave = wv < 4;
  f32x4 win[2];       // W_in[n][c], n = 4 lane .. 4 lane + 3, c = 0, 1 (natural (d, 2) layout: 8 consecutive floats)
  win[0] = bl4(mk_rsrc(A.w_in), (red_wave && lane < (d >> 2)) ? (unsigned)lane * 32u : OOB);
  win[1] = bl4(mk_rsrc(A.w_in), (red_wave && lane < (d >> 2)) ? (unsigned)lane * 32u + 16u : OOB);
  const int c = lane & 31;
  // lane c < 32 of a red wave owns column c of the 32-wide head MLP: Wout[:, c] (2), gamma[c], W0[:, c] (32: gho[c] = sum_k ghz[k] W0[k][c])
  const float wo0 = bl1(mk_rsrc(A.w_hout), (red_wave && c < NO) ? (unsigned)c * 4u : OOB);
  const float wo1 = bl1(mk_rsrc(A.w_hout), (red_wave && c < NO) ? (unsigned)(NO + c) * 4u : OOB);
  const float hgam = bl1(mk_rsrc(A.h_gamma), (red_wave && c < NO) ? (unsigned)c * 4u : OOB);
  float w0c[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) w0c[k] = bl1(mk_rsrc(A.w_h0), (red_wave && c < NO && k < NO) ? (unsigned)(k * NO + c) * 4u : OOB);
  // map configurations: lane c' < n_out + map_dim (<= 64) owns column c' of fusion/hist_map's Linear: W_f[:, c']
  const bool fused_head = A.w_fuse != nullptr;
  const int GL = A.gho_ld;
  float wfc[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) wfc[k] = bl1(mk_rsrc(A.w_fuse), (fused_head && red_wave && lane < GL && k < NO) ? (unsigned)(k * GL + lane) * 4u : OOB);
  for (int i = tid; i < 2 * 4 * LD1; i += CNT) S.t1[i] = 0.f;
  __syncthreads();
  // head visit sp = 0 .. ph: sp < ph opens step s = ph - 1 - sp; sp > 0 first closes step s + 1
#pragma clang loop unroll(disable)
  for (int sp = 0; sp <= ph; ++sp) {
    const int s = ph - 1 - sp;                       // -1 on the last visit: only gx[0][0] is reduced
    const unsigned p = (unsigned)(sp * NP);          // the phase the step's first stage runs as; this visit consumes phase p - 1
    const float *part_in = f.part[(p + 1) & 1] + bwd_rowblock(p - 1u, d) * H, *xres_in = f.xres[(p + 1) & 1] + bwd_rowblock(p - 1u, d);
    int li = 0;
#pragma clang loop unroll(disable)
    for (int m = replica + jh * f.R; m < f.NC; m += f.R * f.NHW, ++li) {
      const int a0 = m * 4, na = min(4, B - a0), ag = a0 + wv;
      const bool mine = red_wave && ag < B;
      const int visit = sp * 4096 + li;
      (void)visit;
      PSTAMP(visit, 0);
      // ---- 0. prefetch: the head tape of step s (independent of the previous phase)
      unsigned k_pe = 0u;
      if constexpr (DROP) {      // PositionalEncoding's dropout: the gradient at the embedding of step s + 1
        if (red_wave && sp > 0)
          k_pe = __builtin_amdgcn_raw_buffer_load_b32(mk_rsrc(A.k_pe), (A.k_pe && mine && lane < (d >> 2))
                                                          ? (unsigned)(((s + 1) * B + ag) * d + 4 * lane) : OOB, 0, 0);
      }
      float huv = 0.f, hxh = 0.f, hrs = 0.f, dl0 = 0.f, dl1 = 0.f;
      if (red_wave && s >= 0) {
        dl0 = bl1(mk_rsrc(A.G), mine ? (unsigned)(((s + 1) * B + ag) * 2 + 0) * 4u : OOB);
        dl1 = bl1(mk_rsrc(A.G), mine ? (unsigned)(((s + 1) * B + ag) * 2 + 1) * 4u : OOB);
        huv = bl1(mk_rsrc(A.h_hu), (mine && c < NO && lane < 32) ? (unsigned)((s * B + ag) * NO + c) * 4u : OOB);
        hxh = bl1(mk_rsrc(A.h_xhat), (mine && c < NO && lane < 32) ? (unsigned)((s * B + ag) * NO + c) * 4u : OOB);
        hrs = bl1(mk_rsrc(A.h_rstd), mine ? (unsigned)(s * B + ag) * 4u : OOB);
      }
      // G rows: G[t] = dL/dpos[t] (t = 1..ph).  Opening step s needs the total gradient at pos[s + 1]:
      //   Gt[s + 1] = G[s + 1] + Gt[s + 2] + gx[0][s + 1] W_in      (the last two terms only for s + 1 < ph)
      float gt0 = 0.f, gt1 = 0.f;                    // wave-uniform: the row's total gradient at pos[s + 1]
      Red4Regs R;
      if (sp > 0) {
        if (red_wave) (void)bwd_poll_rows<false, 1>(R, xres_in + (size_t)m * 16 * d, part_in + (size_t)m * H * 16 * d, d, H, tag_of_phase(p - 1u), ps, (int)p, m);
      }
      PSTAMP(visit, 1);
      __syncthreads();
      if (*ps.abort) return;
      PSTAMP(visit, 2);
      if (red_wave) {
        float d0 = 0.f, d1 = 0.f;
        if (sp > 0) {        // gx[0][s + 1] = residual + parts; its product with W_in
          f32x4 gx = R.res;
#pragma unroll
          for (int g = 0; g < 8; ++g) gx += R.p[g];
          if constexpr (DROP) {
            if (A.k_pe) gx = gx * keep4w(k_pe, A.s_pe);
          }
          if (mine && lane < (d >> 2)) *reinterpret_cast<f32x4 *>(A.gx0 + ((size_t)(s + 1) * B + ag) * d + 4 * lane) = gx;
          float t0 = 0.f, t1 = 0.f;
          if (lane < (d >> 2)) {
            t0 = (gx[0] * win[0][0] + gx[1] * win[0][2]) + (gx[2] * win[1][0] + gx[3] * win[1][2]);
            t1 = (gx[0] * win[0][1] + gx[1] * win[0][3]) + (gx[2] * win[1][1] + gx[3] * win[1][3]);
          }
          d0 = wave64_total(t0);
          d1 = wave64_total(t1);
        }
        if (s >= 0) {
          // rows of A.G: [t] for t = 0..ph; dL/dpos[s + 1] sits in row s + 1 (requested before the poll); the total of step s + 2
          // was left in LDS by this very wave at its previous visit of this cluster
          float n0 = 0.f, n1 = 0.f;
          if (sp > 0) {
            if (li < BW_MAXPOS) {
              n0 = S.qs[(li * 4 + wv) * 2 + 0];
              n1 = S.qs[(li * 4 + wv) * 2 + 1];
            } else {       // (more clusters per head workgroup than the LDS table holds: from memory, past this CU's L1)
              const rsrc_t rG = mk_rsrc(A.G);
              n0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rG, mine ? (unsigned)(((s + 2) * B + ag) * 2 + 0) * 4u : OOB, 0, AUX_SC1));
              n1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rG, mine ? (unsigned)(((s + 2) * B + ag) * 2 + 1) * 4u : OOB, 0, AUX_SC1));
            }
          }
          gt0 = dl0 + (n0 + d0);
          gt1 = dl1 + (n1 + d1);
          if (lane == 0) {
            if (li < BW_MAXPOS) S.qs[(li * 4 + wv) * 2 + 0] = gt0, S.qs[(li * 4 + wv) * 2 + 1] = gt1;
            if (mine && sp > 0) {
              A.G[((size_t)(s + 1) * B + ag) * 2 + 0] = gt0;
              A.G[((size_t)(s + 1) * B + ag) * 2 + 1] = gt1;
            }
          }
          // ---- decoder/MLP backward of step s for this row (MLP.py:12-24): lanes 0..31 = columns
          const float ghu = gt0 * wo0 + gt1 * wo1;
          const float ghum = huv > 0.f ? ghu : 0.f;
          const float gg = (lane < 32 && c < NO) ? ghum * hgam : 0.f;
          const float inv_n = 1.0f / (float)NO;
          const float m1 = wave64_total(gg) * inv_n, m2 = wave64_total(gg * hxh) * inv_n;
          const float ghz = (lane < 32 && c < NO) ? hrs * (gg - m1 - hxh * m2) : 0.f;
          float gho = 0.f;           // gradient at decoder/MLP's input
#pragma unroll
          for (int k = 0; k < 32; ++k)
            gho = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ghz), k)), w0c[k], gho);
          if (lane < 32 && c < NO && mine) {
            const size_t o = ((size_t)s * B + ag) * NO + c;
            A.h_ghum[o] = ghum;
            A.h_ghz[o] = ghz;
          }
          if (fused_head) {          // ... which is fusion/hist_map's output: one more Linear back to cat[output_fc(x), encoded_map]
            const float gu = (lane < 32 && c < NO) ? gho : 0.f;
            if (lane < 32 && c < NO && mine) A.h_gu[((size_t)s * B + ag) * NO + c] = gu;
            float gc = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k)
              gc = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gu), k)), wfc[k], gc);
            if (lane < GL && mine) A.h_gho[((size_t)s * B + ag) * GL + lane] = gc;
            gho = gc;                // lanes < n_out: the gradient at output_fc's output
          } else if (lane < 32 && c < NO && mine) {
            A.h_gho[((size_t)s * B + ag) * NO + c] = gho;
          }
          if (lane < 32 && c < NO) S.t1[wv * LD1 + c] = mine ? gho : 0.f;
        }
      }
      PSTAMP(visit, 3);
      __syncthreads();
      PSTAMP(visit, 4);
      if (s >= 0) {
        // ---- gx[L][s] = gho W_out: the 4 rows go to x0buf, tagged with the step, for the last layer's feed-forward stage
        gemm2_4s<DH, PH_B, D, FS, AUX_SC1>(Mo, S.t1, LD1, f.x0buf + (size_t)m * 4 * d, (unsigned)sp & TAG_MASK);
      }
      PSTAMP(visit, 5);
      __syncthreads();
      PSTAMP(visit, 6);
      (void)na;
    }
  }
}

template <int DH, int D, int FS, bool DROP>
__global__ __launch_bounds__(CNT) void mmt_bwd_pipe4_kernel(BwdPipeArgs f) {
  extern __shared__ __align__(16) float smem[];
  const BwdLds S = bwd_lds(smem, f.a.d);
  const int H = f.a.H, L = f.a.nlayers, per_layer = 2 * H + H / 2, per_rep = L * per_layer + f.NHW;
  const int replica = blockIdx.x / per_rep, rb = blockIdx.x - replica * per_rep;
  PipeSync ps;
  ps.tmo = (gu32 *)f.sync;
  ps.abort = S.abort;
  if (thread_idx() == 0) ps.abort[0] = 0, ps.abort[1] = 0;
  __syncthreads();
  if (rb < L * per_layer) {
    const int layer = rb / per_layer, q = rb - layer * per_layer;
    if (q < H) bwd_stage<DH, BK_A, 1, D, FS, DROP>(f, layer, q, replica, S, ps);
    else if (q < 2 * H) bwd_stage<DH, BK_B, 1, D, FS, DROP>(f, layer, q - H, replica, S, ps);
    else bwd_stage<DH, BK_C, 2, D, FS, DROP>(f, layer, 2 * (q - 2 * H), replica, S, ps);
  } else {
    bwd_head<DH, D, FS, DROP>(f, replica, rb - L * per_layer, S, ps);
  }
}

// before the sweep: every hand-off word becomes the 0xFFFFFFFF that no tag matches, the time-out word 0.  (A kernel, not
// hipMemsetAsync: replayed from a captured graph the memset nodes were not ordered before the sweep kernel - the fill then
// overwrote rows that had just been handed off and their consumers timed out.)
__global__ __launch_bounds__(256) void mmt_bwd_pipe_fill_kernel(float *ws, size_t n4) {
  const u32x4 ones = (u32x4){0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  u32x4 *dst = reinterpret_cast<u32x4 *>(ws + 64);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = ones;
  if (blockIdx.x == 0 && threadIdx.x < 64) reinterpret_cast<unsigned *>(ws)[threadIdx.x] = 0u;
}

// behind the sweep: a sweep that gave up poisons the step's loss (the optimizer kernels are guarded by it) and raises the decoder's
// sticky fault word (mmt_decoder_status reports it)
__global__ void mmt_bwd_pipe_check_kernel(const unsigned *sync, float *poison) {
  if (thread_idx() == 0 && sync[0] != 0u) {
    if (poison) *poison = __builtin_nanf("");
    atomicOr(&g_decoder_fault, 4u);
  }
}
