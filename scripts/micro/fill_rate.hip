// L2/HBM -> CU fill-rate microbenchmark (gfx950): per CU one 512-thread block streams "K steps" of 64 KiB
//   mode 0: global_load_lds_dwordx4 (LDS-DMA), 8 x 1 KiB pieces per wave per step, wait all, barrier, repeat  (= gemm2's fill)
//   mode 1: same but two steps in flight (wait for the older one only)
//   mode 2: global_load_dwordx4 into VGPRs (8 per thread per step), results xor-ed into a register (no LDS write)
// Prints aggregate TB/s and bytes/clk/CU (2.4 GHz nominal).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ inline void dma8(const char* p, unsigned lds, size_t stride) {
  const char* p0 = p; const char* p1 = p + stride; const char* p2 = p + 2 * stride; const char* p3 = p + 3 * stride;
  const char* p4 = p + 4 * stride; const char* p5 = p + 5 * stride; const char* p6 = p + 6 * stride; const char* p7 = p + 7 * stride;
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %9\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t" "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, off\n\t" "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t"
      "global_load_lds_dwordx4 %3, off\n\t" "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t"
      "global_load_lds_dwordx4 %4, off\n\t" "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t"
      "global_load_lds_dwordx4 %5, off\n\t" "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t"
      "global_load_lds_dwordx4 %6, off\n\t" "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t"
      "global_load_lds_dwordx4 %7, off\n\t" "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t"
      "global_load_lds_dwordx4 %8, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7), "s"(lds)
      : "memory", "scc");
}

__global__ __launch_bounds__(512) void fill(const char* __restrict__ buf, size_t window, int steps, int mode, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  // every block walks its own 64 KiB-per-step stream through the window (so data comes from L2 / MALL / HBM, not L1)
  size_t off = ((size_t)blockIdx.x * 64 * 1024 * 37) % window;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int s = 0; s < steps; ++s) {
    const char* base = buf + off + (size_t)wave * 8192 + lane * 16;      // a wave covers 8 KiB = 8 pieces of 1 KiB
    if (mode == 0 || mode == 1) {
      dma8(base, lds_base + ((s & 1) * 65536 + wave * 8192), 1024);
      if (mode == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __syncthreads();
    } else {
      uint4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *(const uint4*)(base + i * 1024);
#pragma unroll
      for (int i = 0; i < 8; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
      __syncthreads();
    }
    off += 65536 * 256;                       // next step: far away (other blocks' streams in between)
    if (off + 65536 > window) off -= window - 65536 > off ? 0 : (window - 65536);
    if (off + 65536 > window) off = ((size_t)blockIdx.x * 65536) % (window - 65536);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w ^ ((unsigned*)smem)[lane];
}

int main() {
  const size_t sizes[] = {(size_t)24 << 20, (size_t)192 << 20, (size_t)4 << 30};
  char* buf; hipMalloc(&buf, sizes[2]); hipMemset(buf, 1, sizes[2]);
  unsigned* out; hipMalloc(&out, 4096);
  hipFuncSetAttribute((const void*)fill, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (size_t window : sizes) {
    for (int mode = 0; mode < 3; ++mode) {
      const int steps = 4000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(fill, dim3(256), dim3(512), 131072, 0, buf, window, 50, mode, out);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(fill, dim3(256), dim3(512), 131072, 0, buf, window, steps, mode, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = 256.0 * steps * 65536;
      printf("window %5zu MiB mode %d: %7.2f TB/s  %6.1f B/clk/CU  %6.2f us/step\n", window >> 20, mode, bytes / (ms * 1e-3) / 1e12,
             bytes / 256 / (ms * 1e-3 * 2.4e9), ms * 1e3 / steps);
    }
  }
  return 0;
}
