// MFMA issue-rate microbenchmark (gfx950): hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
// Every wave runs a long chain of independent MFMAs on register operands (no memory), 8 or 16 accumulators in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K> double run(K kern, int blocks, int threads, int iters, double flop_per_wave_iter, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * threads / 64;
  return waves * iters * flop_per_wave_iter / (ms * 1e-3) / 1e12;
}
int main() {
  float* d; hipMalloc(&d, 4096 * 512 * 4);
  const int iters = 20000;
  for (int wpb : {4, 8}) {       // waves per block: 1 or 2 per SIMD when one block per CU... use many blocks: occupancy decides
    const int threads = wpb * 64, blocks = 256 * 2;
    printf("threads/block %d, blocks %d\n", threads, blocks);
    printf("  16x16x32 bf16, 8 acc : %8.1f TFLOP/s\n", run(k16<8>, blocks, threads, iters, 8 * 2.0 * 16 * 16 * 32, d));
    printf("  16x16x32 bf16, 16 acc: %8.1f TFLOP/s\n", run(k16<16>, blocks, threads, iters, 16 * 2.0 * 16 * 16 * 32, d));
    printf("  32x32x16 bf16, 4 acc : %8.1f TFLOP/s\n", run(k32<4>, blocks, threads, iters, 4 * 2.0 * 32 * 32 * 16, d));
    printf("  32x32x16 bf16, 8 acc : %8.1f TFLOP/s\n", run(k32<8>, blocks, threads, iters, 8 * 2.0 * 32 * 32 * 16, d));
  }
  return 0;
}
