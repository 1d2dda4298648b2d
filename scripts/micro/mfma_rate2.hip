// GEMM-shaped MFMA phase without memory: per "K step" a wave issues the MFMAs of a 128x64 tile from DISTINCT operand
// registers (as gemm2's inner loop does), 2 waves per SIMD (8 waves per 512-thread block, 1 block per CU).
//   A: 16x16x32 bf16, 8 A-fragments x 4 B-fragments, 32 accumulators (128 VGPRs), 2 chunks -> 64 MFMAs / step
//   B: 32x32x16 bf16, 4 A-fragments x 2 B-fragments,  8 accumulators (128 VGPRs), 4 k-halves -> 32 MFMAs / step
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ inline bf16x8 mk(int s) { bf16x8 v; for (int i = 0; i < 8; ++i) v[i] = (__bf16)((threadIdx.x + s * 7 + i) * 0.001f); return v; }

__global__ __launch_bounds__(512) void kA(float* out, int steps, int barrier) {
  bf16x8 a0[8], b0[4], a1[8], b1[4];
  for (int i = 0; i < 8; ++i) { a0[i] = mk(i); a1[i] = mk(20 + i); }
  for (int j = 0; j < 4; ++j) { b0[j] = mk(40 + j); b1[j] = mk(50 + j); }
  f32x4 acc[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  for (int s = 0; s < steps; ++s) {
    if (barrier) __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
    // keep the operands "live and changing" so nothing is hoisted: cheap VALU tweak of one register per step
    a0[0][0] = (__bf16)((float)a0[0][0] + 1.0f);
  }
  float t = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 512 + threadIdx.x] = t;
}
__global__ __launch_bounds__(512) void kB(float* out, int steps, int barrier) {
  bf16x8 a[4][4], b[4][2];      // [k-quarter][fragment]
  for (int q = 0; q < 4; ++q) { for (int i = 0; i < 4; ++i) a[q][i] = mk(q * 4 + i); for (int j = 0; j < 2; ++j) b[q][j] = mk(40 + q * 2 + j); }
  f32x16 acc[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int s = 0; s < steps; ++s) {
    if (barrier) __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
    a[0][0][0] = (__bf16)((float)a[0][0][0] + 1.0f);
  }
  float t = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) t += acc[i][j][0] + acc[i][j][15];
  out[blockIdx.x * 512 + threadIdx.x] = t;
}
template <typename K> double run(K kern, int steps, int barrier, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, 10, barrier);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, steps, barrier);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  // per step per wave: 128 x 64 x 64 MACs
  return 256.0 * 8 * steps * 2.0 * 128 * 64 * 64 / (ms * 1e-3) / 1e12;
}
int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  for (int barrier : {0, 1}) {
    printf("barrier per step: %d\n", barrier);
    printf("  16x16x32 x64 per step : %8.1f TFLOP/s\n", run(kA, 20000, barrier, d));
    printf("  32x32x16 x32 per step : %8.1f TFLOP/s\n", run(kB, 20000, barrier, d));
  }
  return 0;
}
