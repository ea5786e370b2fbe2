// Context encoders for gfx950: CNN map encoder (K4) and lane encoder (K6).
//
// K4 mmt_cnn_kernel : one workgroup per agent; the raster is read once from HBM (the only HBM
//                     traffic: 120 KB/agent), `map*2-1` is applied on the fly, every feature map of
//                     the 4 conv + leaky_relu(0.2) stages stays in LDS, then flatten + Linear.
//                                                   (map_encoder.py:31-36, mgcvae.py:288-290)
// K6 mmt_lane_kernel: Conv1d + leaky_relu + Linear on the raw (B*L, C, S) view.
//                                                   (Lane_encoder.py:29-33, mgcvae.py:301-306)
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <stdint.h>
#include "../../include/mmt_amd.h"

namespace {

constexpr int CB = 5;   // output channels accumulated per work item

__device__ __forceinline__ float leaky02(float v) { return v > 0.f ? v : 0.2f * v; }

// One conv + leaky_relu stage.  `in` is LDS ([Ci][Hi][Wi]) or, for the first stage, the global raster
// (AFFINE: value = 2*m - 1).  Work item = (channel block, output pixel); pixel is the fast index so
// neighbouring lanes read neighbouring input columns.
template <bool AFFINE>
__device__ __forceinline__ void conv_stage(const float *__restrict__ in, int Ci, int Hi, int Wi,
                                           const float *__restrict__ w, const float *__restrict__ b, int Co, int k,
                                           int s, float *__restrict__ out, int Ho, int Wo) {
  const int npix = Ho * Wo;
  const int nblk = (Co + CB - 1) / CB;
  for (int it = threadIdx.x; it < npix * nblk; it += blockDim.x) {
    const int cb = it / npix, p = it - cb * npix;
    const int oy = p / Wo, ox = p - oy * Wo;
    const int co0 = cb * CB;
    float acc[CB];
#pragma unroll
    for (int j = 0; j < CB; ++j) acc[j] = (co0 + j < Co) ? b[co0 + j] : 0.f;
    for (int ci = 0; ci < Ci; ++ci) {
      const float *ip = in + ((size_t)ci * Hi + oy * s) * Wi + ox * s;
      const float *wp = w + ((size_t)co0 * Ci + ci) * k * k;
      for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
          float v = ip[ky * Wi + kx];
          if (AFFINE) v = v * 2.0f - 1.0f;
#pragma unroll
          for (int j = 0; j < CB; ++j)
            if (co0 + j < Co) acc[j] = fmaf(v, wp[(size_t)j * Ci * k * k + ky * k + kx], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < CB; ++j)
      if (co0 + j < Co) out[(size_t)(co0 + j) * npix + p] = leaky02(acc[j]);
  }
}

struct CnnShape {
  int H[MMT_MAX_CONV + 1], W[MMT_MAX_CONV + 1], C[MMT_MAX_CONV + 1];
  int off[MMT_MAX_CONV + 1];   // LDS float offset of stage i's OUTPUT (stage 0 input stays in HBM)
  int flat;
};

__global__ __launch_bounds__(512) void mmt_cnn_kernel(mmt_cnn N, CnnShape S, const float *__restrict__ maps, int B,
                                                      float *__restrict__ out) {
  extern __shared__ __align__(16) float smem[];
  const int b = blockIdx.x;
  const float *img = maps + (size_t)b * S.C[0] * S.H[0] * S.W[0];
  for (int i = 0; i < N.nconv; ++i) {
    const float *src = i == 0 ? img : smem + S.off[i];
    float *dst = smem + S.off[i + 1];
    if (i == 0 && N.input_affine)
      conv_stage<true>(src, S.C[0], S.H[0], S.W[0], N.conv_w[0], N.conv_b[0], S.C[1], N.ksize[0], N.stride[0], dst,
                       S.H[1], S.W[1]);
    else
      conv_stage<false>(src, S.C[i], S.H[i], S.W[i], N.conv_w[i], N.conv_b[i], S.C[i + 1], N.ksize[i], N.stride[i],
                        dst, S.H[i + 1], S.W[i + 1]);
    __syncthreads();
  }
  // flatten (channel-major, as torch.flatten of NCHW) + Linear: 16 lanes per output feature
  const float *x = smem + S.off[N.nconv];
  const int l16 = threadIdx.x & 15;
  for (int o = threadIdx.x >> 4; o < N.n_out; o += blockDim.x >> 4) {
    const float *wr = N.fc_w + (size_t)o * S.flat;
    float a = 0.f;
    for (int i = l16; i < S.flat; i += 16) a = fmaf(wr[i], x[i], a);
#pragma unroll
    for (int m = 8; m > 0; m >>= 1) a += __shfl_xor(a, m, 16);
    if (l16 == 0) out[(size_t)b * N.n_out + o] = a + N.fc_b[o];
  }
}


// ------------------------------------------------------------------------------------------ K4 on MFMA
// Implicit-GEMM form of the reference architecture (config.json map_encoder.cnn_param: 3x100x100 -> conv 10,k5,s2 ->
// 20,k5,s2 -> 10,k5,s1 -> 1,k3,s1 -> Linear(256,n_out)), one workgroup per agent, everything after the raster in LDS.
// GEMM view of a conv stage: D[16 pixels][16 out channels] += A[16 pixels][4 k] B[4 k][16 channels] with
// v_mfma_f32_16x16x4_f32 (exact fp32).  K is ordered (tap, channel) with the channel fastest and padded to 4, so the
// four k of one MFMA are four input channels of ONE tap: lane (pixel i, q) reads in[ch(4m+q)][tap] = one ds_read_b32 at
// `lane base + compile-time offset` (no im2col buffer, no address table); a padding lane re-reads the last channel
// against a zero weight.  The B operands of a stage (<= 150 per lane) sit in VGPRs, loaded once per stage from the
// lane-ordered pack produced by mmt_pack_conv.  The raster is streamed in bands of 8 output rows whose loads (three
// 16-byte requests per thread) are in flight while the previous band is multiplied.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CI, int CO, int KS>
struct ConvPack {
  static constexpr int G = (CI + 3) / 4, NT = (CO + 15) / 16, STEPS = KS * KS * G;
};

__global__ void mmt_pack_conv_kernel(const float *__restrict__ w, int Co, int Ci, int ks, float *__restrict__ dst, int total) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int G = (Ci + 3) / 4, NT = (Co + 15) / 16;
  const int lane = idx & 63, nt = (idx >> 6) % NT, step = (idx >> 6) / NT;
  const int tap = step / G, m = step - tap * G, j = lane & 15, q = lane >> 4;
  const int co = nt * 16 + j, ci = 4 * m + q;
  dst[idx] = (co < Co && ci < Ci) ? w[((size_t)co * Ci + ci) * ks * ks + tap] : 0.f;
}

// One conv + leaky_relu(0.2) stage over output pixels [0, npix) of this call.
//   in : LDS, [CI][in_cs] with rows of WI floats; pixel p -> (oy, ox) = (p / WO, p % WO) reads rows oy*ST.., cols ox*ST..
//   out: LDS, channel stride out_cs, pixel index out_p0 + p
template <int CI, int CO, int KS, int ST, int WI, int WO>
__device__ __forceinline__ void conv_mfma(const float *__restrict__ in, int in_cs, const float *__restrict__ wreg_src,
                                          const float *__restrict__ bias, float *__restrict__ out, int out_cs, int out_p0,
                                          int npix, const float *wreg) {
  typedef ConvPack<CI, CO, KS> P;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  int choff[P::G];
#pragma unroll
  for (int m = 0; m < P::G; ++m) choff[m] = (4 * m + q < CI ? 4 * m + q : CI - 1) * in_cs;
  float bv[P::NT];
#pragma unroll
  for (int nt = 0; nt < P::NT; ++nt) bv[nt] = (nt * 16 + i < CO) ? bias[nt * 16 + i] : 0.f;
  const int ntile = (npix + 15) >> 4;
  for (int tile = wave; tile < ntile; tile += nw) {
    const int p = min(tile * 16 + i, npix - 1);
    const int oy = p / WO, ox = p - oy * WO;
    const float *base = in + oy * ST * WI + ox * ST;
    f32x4 acc[P::NT];
#pragma unroll
    for (int nt = 0; nt < P::NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < KS * KS; ++tap) {
#pragma unroll
      for (int m = 0; m < P::G; ++m) {
        const float a = base[choff[m] + (tap / KS) * WI + (tap % KS)];
#pragma unroll
        for (int nt = 0; nt < P::NT; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[(tap * P::G + m) * P::NT + nt], acc[nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int nt = 0; nt < P::NT; ++nt) {
      const int ch = nt * 16 + i;                     // D layout: column = lane & 15 = channel, rows 4q..4q+3 = pixels
      if (ch < CO) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pp = tile * 16 + 4 * q + r;
          if (pp < npix) out[ch * out_cs + out_p0 + pp] = leaky02(acc[nt][r] + bv[nt]);
        }
      }
    }
  }
}

template <int CI, int CO, int KS>
__device__ __forceinline__ void load_wreg(const float *__restrict__ wp, float *wreg) {
  typedef ConvPack<CI, CO, KS> P;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int t = 0; t < P::STEPS * P::NT; ++t) wreg[t] = wp[t * 64 + lane];
}

constexpr int R_H = 100, R_W = 100, R_C = 3;             // raster
constexpr int R_H1 = 48, R_H2 = 22, R_H3 = 18, R_H4 = 16;  // feature map sides
constexpr int R_BAND = 8, R_BROWS = (R_BAND - 1) * 2 + 5;  // output rows per band, raster rows it needs (19)

__global__ __launch_bounds__(512) void mmt_cnn_mfma_kernel(mmt_cnn N, const float *__restrict__ maps, int B,
                                                           float *__restrict__ out) {
  extern __shared__ __align__(16) float smem[];
  float *o1 = smem;                              // [10][48*48]   (later: conv3 output [10][18*18])
  float *o2 = o1 + 10 * R_H1 * R_H1;             // [20][22*22]
  float *band = o2 + 20 * R_H2 * R_H2;           // [3][19][100]  (later: conv4 output [256])
  const int tid = threadIdx.x, b = blockIdx.x;
  const float *img = maps + (size_t)b * R_C * R_H * R_W;
  const float sc = N.input_affine ? 2.0f : 1.0f, sh = N.input_affine ? -1.0f : 0.0f;
  constexpr int BAND4 = R_C * R_BROWS * (R_W / 4);       // 16-byte pieces of a band (1425)
  constexpr int NLD = (BAND4 + 511) / 512;
  f32x4 pre[NLD];
  auto band_load = [&](int bi) {
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int u = tid + 512 * it;
      pre[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (u < BAND4) {
        const int c = u / (R_BROWS * (R_W / 4)), rem = u - c * (R_BROWS * (R_W / 4)), row = rem / (R_W / 4), x4 = rem - row * (R_W / 4);
        pre[it] = *reinterpret_cast<const f32x4 *>(img + ((size_t)c * R_H + bi * R_BAND * 2 + row) * R_W + 4 * x4);
      }
    }
  };
  auto band_store = [&]() {
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int u = tid + 512 * it;
      if (u < BAND4) *reinterpret_cast<f32x4 *>(band + 4 * u) = pre[it] * sc + sh;     // map*2-1 (mgcvae.py:290)
    }
  };
  {
    float wreg[ConvPack<3, 10, 5>::STEPS * ConvPack<3, 10, 5>::NT];
    load_wreg<3, 10, 5>(N.conv_wp[0], wreg);
    band_load(0);
    for (int bi = 0; bi < R_H1 / R_BAND; ++bi) {
      band_store();
      __syncthreads();
      if (bi + 1 < R_H1 / R_BAND) band_load(bi + 1);
      conv_mfma<3, 10, 5, 2, R_W, R_H1>(band, R_BROWS * R_W, nullptr, N.conv_b[0], o1, R_H1 * R_H1, bi * R_BAND * R_H1,
                                        R_BAND * R_H1, wreg);
      __syncthreads();
    }
  }
  {
    float wreg[ConvPack<10, 20, 5>::STEPS * ConvPack<10, 20, 5>::NT];
    load_wreg<10, 20, 5>(N.conv_wp[1], wreg);
    conv_mfma<10, 20, 5, 2, R_H1, R_H2>(o1, R_H1 * R_H1, nullptr, N.conv_b[1], o2, R_H2 * R_H2, 0, R_H2 * R_H2, wreg);
    __syncthreads();
  }
  {
    float wreg[ConvPack<20, 10, 5>::STEPS * ConvPack<20, 10, 5>::NT];
    load_wreg<20, 10, 5>(N.conv_wp[2], wreg);
    conv_mfma<20, 10, 5, 1, R_H2, R_H3>(o2, R_H2 * R_H2, nullptr, N.conv_b[2], o1, R_H3 * R_H3, 0, R_H3 * R_H3, wreg);
    __syncthreads();
  }
  {
    float wreg[ConvPack<10, 1, 3>::STEPS * ConvPack<10, 1, 3>::NT];
    load_wreg<10, 1, 3>(N.conv_wp[3], wreg);
    conv_mfma<10, 1, 3, 1, R_H3, R_H4>(o1, R_H3 * R_H3, nullptr, N.conv_b[3], band, R_H4 * R_H4, 0, R_H4 * R_H4, wreg);
    __syncthreads();
  }
  // flatten + Linear(256, n_out): 16 lanes per output feature
  const int l16 = tid & 15;
  for (int o = tid >> 4; o < N.n_out; o += 32) {
    const float *wr = N.fc_w + (size_t)o * (R_H4 * R_H4);
    float a = 0.f;
    for (int i = l16; i < R_H4 * R_H4; i += 16) a = fmaf(wr[i], band[i], a);
#pragma unroll
    for (int m = 8; m > 0; m >>= 1) a += __shfl_xor(a, m, 16);
    if (l16 == 0) out[(size_t)b * N.n_out + o] = a + N.fc_b[o];
  }
}

__global__ void mmt_lane_kernel(mmt_lane N, const float *__restrict__ lanes, int n_rows, float *__restrict__ out) {
  const int width = (N.n_samples - N.ksize) / N.stride + 1;
  const int flat = N.out_ch * width;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * N.n_out) return;
  const int row = idx / N.n_out, o = idx - row * N.n_out;
  const float *x = lanes + (size_t)row * N.in_ch * N.n_samples;   // raw view: [in_ch][n_samples]
  float a = N.fc_b[o];
  for (int f = 0; f < flat; ++f) {
    const int oc = f / width, j = f - oc * width;
    float c = N.conv_b[oc];
    for (int ci = 0; ci < N.in_ch; ++ci)
      for (int kk = 0; kk < N.ksize; ++kk)
        c = fmaf(N.conv_w[(oc * N.in_ch + ci) * N.ksize + kk], x[ci * N.n_samples + j * N.stride + kk], c);
    a = fmaf(N.fc_w[(size_t)o * flat + f], leaky02(c), a);
  }
  out[idx] = a;
}

}  // namespace

extern "C" const char *mmt_last_error(void);
int mmt_set_error(int code, const char *fmt, ...);   // transformer.hip

static int cnn_shape(const mmt_cnn *c, CnnShape *S, size_t *lds_floats, char *why, size_t nwhy) {
  if (!c || c->nconv <= 0 || c->nconv > MMT_MAX_CONV) {
    snprintf(why, nwhy, "nconv outside 1..%d", MMT_MAX_CONV);
    return -1;
  }
  S->C[0] = c->in_ch;
  S->H[0] = c->in_h;
  S->W[0] = c->in_w;
  size_t off = 0;
  S->off[0] = 0;
  for (int i = 0; i < c->nconv; ++i) {
    if (c->ksize[i] <= 0 || c->stride[i] <= 0 || c->out_ch[i] <= 0 || S->H[i] < c->ksize[i] || S->W[i] < c->ksize[i]) {
      snprintf(why, nwhy, "conv %d: bad kernel/stride/channels for a %dx%d input", i, S->H[i], S->W[i]);
      return -1;
    }
    S->C[i + 1] = c->out_ch[i];
    S->H[i + 1] = (S->H[i] - c->ksize[i]) / c->stride[i] + 1;
    S->W[i + 1] = (S->W[i] - c->ksize[i]) / c->stride[i] + 1;
    S->off[i + 1] = (int)off;
    off += ((size_t)S->C[i + 1] * S->H[i + 1] * S->W[i + 1] + 3) & ~(size_t)3;
  }
  S->flat = S->C[c->nconv] * S->H[c->nconv] * S->W[c->nconv];
  *lds_floats = off;
  return 0;
}

extern "C" size_t mmt_cnn_workspace_floats(const mmt_cnn *, int) { return 0; }   // everything lives in LDS

extern "C" int mmt_cnn_forward(const mmt_cnn *cnn, const float *maps, int B, float *out, float *, void *stream) {
  if (!cnn || !maps || !out) return mmt_set_error(MMT_EINVAL, "mmt_cnn_forward: null pointer");
  if (B <= 0) return mmt_set_error(MMT_EINVAL, "mmt_cnn_forward: empty batch (B=%d)", B);
  CnnShape S;
  size_t lds_floats = 0;
  char why[128];
  if (cnn_shape(cnn, &S, &lds_floats, why, sizeof(why))) return mmt_set_error(MMT_EINVAL, "mmt_cnn_forward: %s", why);
  // the reference architecture with lane-ordered weights: implicit-GEMM MFMA kernel
  const bool ref_arch = cnn->nconv == 4 && cnn->in_ch == 3 && cnn->in_h == 100 && cnn->in_w == 100 && cnn->out_ch[0] == 10 &&
                        cnn->out_ch[1] == 20 && cnn->out_ch[2] == 10 && cnn->out_ch[3] == 1 && cnn->ksize[0] == 5 &&
                        cnn->ksize[1] == 5 && cnn->ksize[2] == 5 && cnn->ksize[3] == 3 && cnn->stride[0] == 2 &&
                        cnn->stride[1] == 2 && cnn->stride[2] == 1 && cnn->stride[3] == 1 && cnn->conv_wp[0] && cnn->conv_wp[1] &&
                        cnn->conv_wp[2] && cnn->conv_wp[3] && ((uintptr_t)maps & 15) == 0;
  if (ref_arch) {
    const size_t lds_m = (size_t)(10 * R_H1 * R_H1 + 20 * R_H2 * R_H2 + R_C * R_BROWS * R_W) * sizeof(float);
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void *)mmt_cnn_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
      attr = true;
    }
    hipLaunchKernelGGL(mmt_cnn_mfma_kernel, dim3(B), dim3(512), lds_m, (hipStream_t)stream, *cnn, maps, B, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mmt_set_error(MMT_ELAUNCH, "mmt_cnn_forward: %s", hipGetErrorString(e));
    return MMT_OK;
  }
  const size_t lds = lds_floats * sizeof(float);
  if (lds > 160 * 1024) return mmt_set_error(MMT_EINVAL, "mmt_cnn_forward: feature maps need %zu B of LDS (> 160 KiB)", lds);
  (void)hipFuncSetAttribute((const void *)mmt_cnn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(mmt_cnn_kernel, dim3(B), dim3(512), lds, (hipStream_t)stream, *cnn, S, maps, B, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mmt_set_error(MMT_ELAUNCH, "mmt_cnn_forward: %s", hipGetErrorString(e));
  return MMT_OK;
}

extern "C" size_t mmt_packed_conv_floats(int Co, int Ci, int ks) {
  return (size_t)ks * ks * ((Ci + 3) / 4) * ((Co + 15) / 16) * 64;
}

extern "C" int mmt_pack_conv(const float *w, int Co, int Ci, int ks, float *dst, void *stream) {
  if (!w || !dst || Co <= 0 || Ci <= 0 || ks <= 0) return mmt_set_error(MMT_EINVAL, "mmt_pack_conv: bad arguments");
  const int total = (int)mmt_packed_conv_floats(Co, Ci, ks);
  hipLaunchKernelGGL(mmt_pack_conv_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, Co, Ci, ks, dst, total);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mmt_set_error(MMT_ELAUNCH, "mmt_pack_conv: %s", hipGetErrorString(e));
  return MMT_OK;
}

extern "C" int mmt_lane_forward(const mmt_lane *lane, const float *lanes, int n_rows, float *out, void *stream) {
  if (!lane || !lanes || !out) return mmt_set_error(MMT_EINVAL, "mmt_lane_forward: null pointer");
  if (n_rows <= 0) return mmt_set_error(MMT_EINVAL, "mmt_lane_forward: empty input (n_rows=%d)", n_rows);
  if (lane->ksize <= 0 || lane->stride <= 0 || lane->n_samples < lane->ksize || lane->in_ch <= 0 || lane->out_ch <= 0 ||
      lane->n_out <= 0)
    return mmt_set_error(MMT_EINVAL, "mmt_lane_forward: bad conv shape");
  const int total = n_rows * lane->n_out;
  hipLaunchKernelGGL(mmt_lane_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, *lane, lanes, n_rows, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mmt_set_error(MMT_ELAUNCH, "mmt_lane_forward: %s", hipGetErrorString(e));
  return MMT_OK;
}
