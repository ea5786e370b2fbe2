// Probe: operand / result layout of v_mfma_f32_4x4x1_16b_f32 on gfx950 (build: hipcc --offload-arch=gfx950 -O2).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out, long long *cyc) {
  const int l = threadIdx.x;
  const float a = 1.0f + (l & 3) + 10.0f * (l >> 2);        // A[block][row i = l&3]
  const float b = 100.0f * (1 + (l & 3)) + 1000.0f * (l >> 2);   // B[block][col j = l&3]
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
  // throughput: 256 dependent-free MFMAs on 4 accumulators
  f32x4 acc[4] = {c, c, c, c};
  long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < 64; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 0, 0, 0);
  long long t1 = clock64();
  if (l == 0) cyc[0] = t1 - t0;
  out[256 + l] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
int main() {
  float *d; long long *c;
  hipMalloc(&d, 4096); hipMalloc(&c, 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c);
  float h[256]; long long hc;
  hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost); hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
  // expectation if D_b[i][j] = A_b[i]*B_b[j] with lane 4b+j holding column j, reg r = row r
  int ok = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const int blk = l >> 2, j = l & 3;
    const float want = (1.0f + r + 10.0f * blk) * (100.0f * (1 + j) + 1000.0f * blk);
    if (h[l * 4 + r] != want) { ok = 0; if (l < 8) printf("lane %d reg %d got %g want %g\n", l, r, h[l * 4 + r], want); }
  }
  printf("layout D[lane=4b+j][reg=i] = A_b[i]*B_b[j]: %s; 256 MFMAs took %lld cycles (%.1f per MFMA)\n", ok ? "YES" : "NO", hc, hc / 256.0);
  return 0;
}
