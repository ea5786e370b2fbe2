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
  const size_t lds = lds_floats * sizeof(float);
  if (lds > 160 * 1024) return mmt_set_error(MMT_EINVAL, "mmt_cnn_forward: feature maps need %zu B of LDS (> 160 KiB)", lds);
  (void)hipFuncSetAttribute((const void *)mmt_cnn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(mmt_cnn_kernel, dim3(B), dim3(512), lds, (hipStream_t)stream, *cnn, S, maps, B, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mmt_set_error(MMT_ELAUNCH, "mmt_cnn_forward: %s", hipGetErrorString(e));
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
