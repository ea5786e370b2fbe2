// Probe: in which order does v_mfma_f32_16x16x4_f32 add its four products to the accumulator?  D[i][j] = C[i][j] + sum over the
// lane groups q = 0..3 of A[i][q] * B[q][j].  Candidates, each evaluated in fp32 on the host: a chain of fused multiply-adds in
// q order, the reverse chain, pairwise, and the exactly rounded sum (fp64).  (build: hipcc --offload-arch=gfx950 -O2)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, const float *C, float *D) {      // A[16][4], B[4][16], C/D[16][16]
  const int l = threadIdx.x, i = l & 15, q = l >> 4;
  f32x4 c;
  for (int r = 0; r < 4; ++r) c[r] = C[(4 * q + r) * 16 + i];      // col = lane&15, row = 4*(lane>>4) + reg
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 4 + q], B[q * 16 + i], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + i] = c[r];
}
int main() {
  float hA[64], hB[64], hC[256], hD[256];
  int hits[4] = {0, 0, 0, 0}, total = 0;
  float *A, *B, *C, *D;
  hipMalloc(&A, 256); hipMalloc(&B, 256); hipMalloc(&C, 1024); hipMalloc(&D, 1024);
  srand(1);
  for (int trial = 0; trial < 200; ++trial) {
    for (int x = 0; x < 64; ++x) hA[x] = (rand() / (float)RAND_MAX - 0.5f) * 4.f, hB[x] = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    for (int x = 0; x < 256; ++x) hC[x] = (rand() / (float)RAND_MAX - 0.5f) * (trial & 1 ? 4.f : 0.f);
    hipMemcpy(A, hA, 256, hipMemcpyHostToDevice); hipMemcpy(B, hB, 256, hipMemcpyHostToDevice); hipMemcpy(C, hC, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, C, D);
    hipMemcpy(hD, D, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      const float c = hC[i * 16 + j];
      float p[4]; for (int q = 0; q < 4; ++q) p[q] = 0;
      float fw = c, bw = c; double ex = c;
      for (int q = 0; q < 4; ++q) fw = fmaf(hA[i * 4 + q], hB[q * 16 + j], fw);
      for (int q = 3; q >= 0; --q) bw = fmaf(hA[i * 4 + q], hB[q * 16 + j], bw);
      for (int q = 0; q < 4; ++q) ex += (double)hA[i * 4 + q] * (double)hB[q * 16 + j];
      const float pw = (fmaf(hA[i * 4 + 1], hB[16 + j], hA[i * 4] * hB[j]) + fmaf(hA[i * 4 + 3], hB[48 + j], hA[i * 4 + 2] * hB[32 + j])) + c;
      const float got = hD[i * 16 + j];
      hits[0] += got == fw; hits[1] += got == bw; hits[2] += got == pw; hits[3] += got == (float)ex; ++total;
    }
  }
  printf("of %d results: forward fma chain %d, reverse chain %d, pairwise %d, exactly rounded sum %d\n", total, hits[0], hits[1], hits[2], hits[3]);
  return 0;
}
