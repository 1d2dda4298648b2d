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

// GEMM-shaped fill (no MFMA): M x K (row-major bf16, mode 3) or the same data stored tile-major [tm][kstep][256 rows][128 B]
// (mode 4); W (N x K) likewise.  256x256 tiles, XCD-aware bijective tile order as in gemm2, two LDS stages.
__global__ __launch_bounds__(512) void gemm_fill(const char* __restrict__ A, const char* __restrict__ W, int M, int N, int K, int tile_major,
                                                 unsigned* out, int depth2 = 0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tiles_n = N / 256, nwg = gridDim.x;
  int bid = blockIdx.x;
  { const int q = nwg >> 3, r = nwg & 7; const int xcd = bid & 7, idx = bid >> 3; bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int nk = K / 64;
  const size_t S = (size_t)K * 2;                       // row stride in bytes
  const int lrow = lane >> 3, lcol = lane & 7;
  for (int kt = 0; kt < nk; ++kt) {
    const char *pa, *pw; size_t stride;
    if (tile_major) {                                     // 32 KiB contiguous per (tile row block, k step): wave = 4 KiB, piece = 1 KiB
      pa = A + ((size_t)tm * nk + kt) * 32768 + (size_t)wave * 4096 + lane * 16;
      pw = W + ((size_t)tn * nk + kt) * 32768 + (size_t)wave * 4096 + lane * 16;
      stride = 1024;
    } else {                                              // 8 rows x 128 B per piece, rows S bytes apart
      pa = A + ((size_t)tm * 256 + wave * 32 + lrow) * S + (size_t)kt * 128 + lcol * 16;
      pw = W + ((size_t)tn * 256 + wave * 32 + lrow) * S + (size_t)kt * 128 + lcol * 16;
      stride = 8 * S;
    }
    const unsigned dst = lds_base + (kt & 1) * 65536 + wave * 4096;
    const char* a0 = pa; const char* a1 = pa + stride; const char* a2 = pa + 2 * stride; const char* a3 = pa + 3 * stride;
    const char* w0 = pw; const char* w1 = pw + stride; const char* w2 = pw + 2 * stride; const char* w3 = pw + 3 * stride;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t" "s_mov_b32 m0, %9\n\t" "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t" "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, off\n\t" "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, off\n\t" "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, off\n\t" "s_add_u32 m0, m0, 0x7400\n\t" "s_nop 0\n\t"
        "global_load_lds_dwordx4 %5, off\n\t" "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t"
        "global_load_lds_dwordx4 %6, off\n\t" "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t"
        "global_load_lds_dwordx4 %7, off\n\t" "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t"
        "global_load_lds_dwordx4 %8, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "s"(dst) : "memory", "scc");
    if (depth2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // the previous step has landed, this one stays in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) out[blockIdx.x] = ((unsigned*)smem)[lane];
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
  {
    const int M = 90112, N = 1024;
    for (int K : {1024, 4096, 19456}) {
      char *A, *Wm; hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&Wm, (size_t)N * K * 2);
      hipMemset(A, 1, (size_t)M * K * 2); hipMemset(Wm, 1, (size_t)N * K * 2);
      for (int depth2 = 0; depth2 < 2; ++depth2)
      for (int tile_major = 0; tile_major < 2; ++tile_major) {
        const int tiles = (M / 256) * (N / 256);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(gemm_fill, dim3(tiles), dim3(512), 131072, 0, A, Wm, M, N, K, tile_major, out, depth2);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(gemm_fill, dim3(tiles), dim3(512), 131072, 0, A, Wm, M, N, K, tile_major, out, depth2);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        const double bytes = (double)tiles * (K / 64) * 65536;
        printf("GEMM-shaped fill (%d step%s in flight) M=%d N=%d K=%5d %s: %7.3f ms  %6.2f TB/s of fill  (a GEMM bound by it: %6.0f TFLOP/s)\n", depth2 + 1, depth2 ? "s" : " ", M, N, K,
               tile_major ? "tile-major" : "row-major ", ms, bytes / (ms * 1e-3) / 1e12, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
      }
      hipFree(A); hipFree(Wm);
    }
  }
  return 0;
}
