// GMM2D for gfx950: the 2-D Gaussian mixture of Transformer/model/components/gmm2d.py:7-217 (named by the north
// star, unused by the live Transformer path: SURVEY section 8 a22 / f3) as five small HBM-bound kernels.
// R = product of the leading dims, N = components.  Everything is fp32 like the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mmt_amd.h"

int mmt_set_error(int code, const char *fmt, ...);   // transformer.hip

namespace {

constexpr int GMM_NMAX = 64;

// gmm2d.py:39-66 (__init__): normalised log_pis, sigmas, clamped 1 - rho^2, Cholesky factor L, covariance
__global__ void mmt_gmm2d_prepare_kernel(const float *__restrict__ log_pis, const float *__restrict__ log_sigmas,
                                         const float *__restrict__ corrs, long R, int N, float *__restrict__ norm_log_pis,
                                         float *__restrict__ sigmas, float *__restrict__ omr, float *__restrict__ L,
                                         float *__restrict__ cov) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  float mx = -INFINITY;
  for (int n = 0; n < N; ++n) mx = fmaxf(mx, fmaxf(log_pis[r * N + n], -1e5f));
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += expf(fmaxf(log_pis[r * N + n], -1e5f) - mx);
  const float lse = mx + logf(s);
  for (int n = 0; n < N; ++n) {
    const long i = r * N + n;
    norm_log_pis[i] = fmaxf(log_pis[i], -1e5f) - lse;
    const float sx = expf(log_sigmas[2 * i]), sy = expf(log_sigmas[2 * i + 1]), rho = corrs[i];
    const float o = fminf(fmaxf(1.0f - rho * rho, 1e-5f), 1.0f);
    sigmas[2 * i] = sx;
    sigmas[2 * i + 1] = sy;
    omr[i] = o;
    L[4 * i] = sx;
    L[4 * i + 1] = 0.f;
    L[4 * i + 2] = sy * rho;
    L[4 * i + 3] = sy * sqrtf(o);
    const float c = rho * (sx * sy);
    cov[4 * i] = sx * sx;
    cov[4 * i + 1] = c;
    cov[4 * i + 2] = c;
    cov[4 * i + 3] = sy * sy;
  }
}

__device__ __forceinline__ float comp_log_p(float vx, float vy, float mx_, float my_, float sx, float sy, float lsx, float lsy,
                                            float rho, float o) {
  const float dx = vx - mx_, dy = vy - my_;
  const float qx = dx / sx, qy = dy / sy;
  const float en = (qx * qx + qy * qy) - 2.0f * rho * (dx * dy) / (sx * sy);
  return -(3.67575407f + logf(o) + 2.0f * (lsx + lsy) + en / o) / 2.0f;     // 2*log(2*pi) = 3.6757541
}

// gmm2d.py:136-163 log_prob (logsumexp over components) and :107-133 reg_los (per-component L2 distance)
__global__ void mmt_gmm2d_log_prob_kernel(const float *__restrict__ norm_log_pis, const float *__restrict__ mus,
                                          const float *__restrict__ log_sigmas, const float *__restrict__ sigmas,
                                          const float *__restrict__ corrs, const float *__restrict__ omr,
                                          const float *__restrict__ value, long R, int N, float *__restrict__ logp,
                                          float *__restrict__ lreg) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float vx = value[2 * r], vy = value[2 * r + 1];
  float t[GMM_NMAX];
  float mx = -INFINITY;
  for (int n = 0; n < N; ++n) {
    const long i = r * N + n;
    const float v = norm_log_pis[i] + comp_log_p(vx, vy, mus[2 * i], mus[2 * i + 1], sigmas[2 * i], sigmas[2 * i + 1],
                                                 log_sigmas[2 * i], log_sigmas[2 * i + 1], corrs[i], omr[i]);
    t[n] = v;
    mx = fmaxf(mx, v);
    if (lreg) {
      const float dx = vx - mus[2 * i], dy = vy - mus[2 * i + 1];
      lreg[i] = sqrtf(dx * dx + dy * dy + 1e-8f);
    }
  }
  if (logp) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += expf(t[n] - mx);
    logp[r] = mx + logf(s);
  }
}

// gmm2d.py:79-105 rsample: mu_c + L_c z_c for the drawn component c.  The normal noise z (S,R,N,2) and either the
// component indices comp (S,R) or uniforms u (S,R) (inverse CDF over softmax(log_pis)) are supplied by the caller.
__global__ void mmt_gmm2d_rsample_kernel(const float *__restrict__ norm_log_pis, const float *__restrict__ mus,
                                         const float *__restrict__ L, const float *__restrict__ z,
                                         const int32_t *__restrict__ comp, const float *__restrict__ u, long S, long R, int N,
                                         float *__restrict__ out, int32_t *__restrict__ comp_out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= S * R) return;
  const long r = idx % R;
  int c;
  if (comp) c = comp[idx];
  else {
    const float uu = u[idx];
    float acc = 0.f;
    c = N - 1;
    for (int n = 0; n < N; ++n) {
      acc += expf(norm_log_pis[r * N + n]);
      if (uu < acc) {
        c = n;
        break;
      }
    }
  }
  c = min(max(c, 0), N - 1);
  const long i = r * N + c;
  const float z0 = z[(idx * N + c) * 2], z1 = z[(idx * N + c) * 2 + 1];
  out[2 * idx] = mus[2 * i] + (L[4 * i] * z0 + L[4 * i + 1] * z1);
  out[2 * idx + 1] = mus[2 * i + 1] + (L[4 * i + 2] * z0 + L[4 * i + 3] * z1);
  if (comp_out) comp_out[idx] = c;
}

// gmm2d.py:176-204 mode: arg max of log_prob over the mesh arange(x_min, x_max, 0.01) x arange(y_min, y_max, 0.01)
// spanned by the component means ('ij' order, first maximum wins).  One workgroup per distribution.
__global__ __launch_bounds__(256) void mmt_gmm2d_mode_kernel(const float *__restrict__ norm_log_pis, const float *__restrict__ mus,
                                                             const float *__restrict__ log_sigmas, const float *__restrict__ sigmas,
                                                             const float *__restrict__ corrs, const float *__restrict__ omr,
                                                             int N, float *__restrict__ out) {
  __shared__ float p[GMM_NMAX][8];
  __shared__ float best_v[256];
  __shared__ long best_i[256];
  const long r = blockIdx.x;
  const int tid = threadIdx.x;
  for (int n = tid; n < N; n += 256) {
    const long i = r * N + n;
    p[n][0] = norm_log_pis[i];
    p[n][1] = mus[2 * i];
    p[n][2] = mus[2 * i + 1];
    p[n][3] = sigmas[2 * i];
    p[n][4] = sigmas[2 * i + 1];
    p[n][5] = log_sigmas[2 * i] + log_sigmas[2 * i + 1];
    p[n][6] = corrs[i];
    p[n][7] = omr[i];
  }
  __syncthreads();
  float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
  for (int n = 0; n < N; ++n) {
    xmin = fminf(xmin, p[n][1]);
    xmax = fmaxf(xmax, p[n][1]);
    ymin = fminf(ymin, p[n][2]);
    ymax = fmaxf(ymax, p[n][2]);
  }
  // torch.arange(start, end, 0.01) on float tensors: size and values computed in double, then cast to float
  const long nx = (long)ceil(((double)xmax - (double)xmin) / 0.01), ny = (long)ceil(((double)ymax - (double)ymin) / 0.01);
  float bv = -INFINITY;
  long bi = -1;
  for (long g = tid; g < nx * ny; g += 256) {
    const float x = (float)((double)xmin + 0.01 * (double)(g / ny)), y = (float)((double)ymin + 0.01 * (double)(g % ny));
    float mx = -INFINITY, t[GMM_NMAX];
    for (int n = 0; n < N; ++n) {
      const float dx = x - p[n][1], dy = y - p[n][2];
      const float qx = dx / p[n][3], qy = dy / p[n][4];
      const float en = (qx * qx + qy * qy) - 2.0f * p[n][6] * (dx * dy) / (p[n][3] * p[n][4]);
      t[n] = p[n][0] - (3.67575407f + logf(p[n][7]) + 2.0f * p[n][5] + en / p[n][7]) / 2.0f;
      mx = fmaxf(mx, t[n]);
    }
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += expf(t[n] - mx);
    const float v = mx + logf(s);
    if (v > bv) {       // strided visit order is increasing in g: the first maximum of this thread is kept
      bv = v;
      bi = g;
    }
  }
  best_v[tid] = bv;
  best_i[tid] = bi;
  __syncthreads();
  if (tid == 0) {
    float v = -INFINITY;
    long gi = -1;
    for (int k = 0; k < 256; ++k)
      if (best_i[k] >= 0 && (best_v[k] > v || (best_v[k] == v && best_i[k] < gi))) {
        v = best_v[k];
        gi = best_i[k];
      }
    if (gi < 0) {       // empty grid (all means equal along an axis): the reference would fail on an empty argmax
      out[2 * r] = xmin;
      out[2 * r + 1] = ymin;
    } else {
      out[2 * r] = (float)((double)xmin + 0.01 * (double)(gi / ny));
      out[2 * r + 1] = (float)((double)ymin + 0.01 * (double)(gi % ny));
    }
  }
}

// gmm2d.py:68-77 from_log_pis_mus_cov_mats: covariance matrices (R*N,2,2) -> log_sigmas (R*N,2), corrs (R*N)
__global__ void mmt_gmm2d_from_cov_kernel(const float *__restrict__ cov, long RN, float *__restrict__ log_sigmas,
                                          float *__restrict__ corrs) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= RN) return;
  const float s1 = sqrtf(fmaxf(cov[4 * i], 1e-8f)), s2 = sqrtf(fmaxf(cov[4 * i + 3], 1e-8f));
  log_sigmas[2 * i] = logf(s1);
  log_sigmas[2 * i + 1] = logf(s2);
  corrs[i] = cov[4 * i + 1] / (s1 * s2);
}

int ok_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mmt_set_error(MMT_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return MMT_OK;
}

}  // namespace

extern "C" int mmt_gmm2d_prepare(const float *log_pis, const float *log_sigmas, const float *corrs, long R, int N,
                                 float *norm_log_pis, float *sigmas, float *one_minus_rho2, float *L, float *cov, void *stream) {
  if (!log_pis || !log_sigmas || !corrs || !norm_log_pis || !sigmas || !one_minus_rho2 || !L || !cov || R <= 0 || N <= 0)
    return mmt_set_error(MMT_EINVAL, "mmt_gmm2d_prepare: bad arguments");
  hipLaunchKernelGGL(mmt_gmm2d_prepare_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, log_pis,
                     log_sigmas, corrs, R, N, norm_log_pis, sigmas, one_minus_rho2, L, cov);
  return ok_launch("mmt_gmm2d_prepare");
}

extern "C" int mmt_gmm2d_log_prob(const float *norm_log_pis, const float *mus, const float *log_sigmas, const float *sigmas,
                                  const float *corrs, const float *one_minus_rho2, const float *value, long R, int N,
                                  float *logp, float *lreg, void *stream) {
  if (!norm_log_pis || !mus || !log_sigmas || !sigmas || !corrs || !one_minus_rho2 || !value || (!logp && !lreg) || R <= 0)
    return mmt_set_error(MMT_EINVAL, "mmt_gmm2d_log_prob: bad arguments");
  if (N <= 0 || N > GMM_NMAX) return mmt_set_error(MMT_EINVAL, "mmt_gmm2d_log_prob: 1 <= components <= %d", GMM_NMAX);
  hipLaunchKernelGGL(mmt_gmm2d_log_prob_kernel, dim3((unsigned)((R + 127) / 128)), dim3(128), 0, (hipStream_t)stream,
                     norm_log_pis, mus, log_sigmas, sigmas, corrs, one_minus_rho2, value, R, N, logp, lreg);
  return ok_launch("mmt_gmm2d_log_prob");
}

extern "C" int mmt_gmm2d_rsample(const float *norm_log_pis, const float *mus, const float *L, const float *z, const int32_t *comp,
                                 const float *u, long S, long R, int N, float *out, int32_t *comp_out, void *stream) {
  if (!norm_log_pis || !mus || !L || !z || (!comp && !u) || !out || S <= 0 || R <= 0 || N <= 0)
    return mmt_set_error(MMT_EINVAL, "mmt_gmm2d_rsample: bad arguments (need z and either comp or u)");
  hipLaunchKernelGGL(mmt_gmm2d_rsample_kernel, dim3((unsigned)((S * R + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     norm_log_pis, mus, L, z, comp, u, S, R, N, out, comp_out);
  return ok_launch("mmt_gmm2d_rsample");
}

extern "C" int mmt_gmm2d_mode(const float *norm_log_pis, const float *mus, const float *log_sigmas, const float *sigmas,
                              const float *corrs, const float *one_minus_rho2, long R, int N, float *out, void *stream) {
  if (!norm_log_pis || !mus || !log_sigmas || !sigmas || !corrs || !one_minus_rho2 || !out || R <= 0)
    return mmt_set_error(MMT_EINVAL, "mmt_gmm2d_mode: bad arguments");
  if (N <= 0 || N > GMM_NMAX) return mmt_set_error(MMT_EINVAL, "mmt_gmm2d_mode: 1 <= components <= %d", GMM_NMAX);
  hipLaunchKernelGGL(mmt_gmm2d_mode_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, norm_log_pis, mus, log_sigmas,
                     sigmas, corrs, one_minus_rho2, N, out);
  return ok_launch("mmt_gmm2d_mode");
}

extern "C" int mmt_gmm2d_from_cov(const float *cov, long RN, float *log_sigmas, float *corrs, void *stream) {
  if (!cov || !log_sigmas || !corrs || RN <= 0) return mmt_set_error(MMT_EINVAL, "mmt_gmm2d_from_cov: bad arguments");
  hipLaunchKernelGGL(mmt_gmm2d_from_cov_kernel, dim3((unsigned)((RN + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cov, RN,
                     log_sigmas, corrs);
  return ok_launch("mmt_gmm2d_from_cov");
}
