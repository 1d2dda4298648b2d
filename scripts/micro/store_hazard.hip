// Does a 16-byte global store read its data registers "at issue" on gfx950 -- i.e. is it safe to overwrite them (by an LDS
// read, or by a VALU instruction one wait state later) right behind the store?  Round 4's full-tile GEMM epilogue (stores as
// inline asm, invisible to hipcc's waitcnt pass) produced a few wrong elements in one kernel; this probe isolates the question.
//
// Every wave writes ITER x 1 KiB: iteration i puts the pattern (tag_i, lane, i, wave) into v[0:3], stores it with an inline-asm
// global_store_dwordx4, and then -- depending on MODE -- overwrites the same registers
//   mode 0: not at all (control: fresh registers for the poison)
//   mode 1: by a ds_read_b128 of a poison pattern, immediately
//   mode 2: by four v_mov_b32 of the poison after s_nop 0 (one wait state); modes 4 / 5 / 6: after s_nop 1 / 2 / 3
//   mode 3: like 1, after GAP extra global loads per iteration are put in flight (a busy memory pipe, as in the epilogue with
//           a residual ring)
// The host then counts poisoned / wrong words.  build: hipcc --offload-arch=gfx950 -O2 -o store_hazard store_hazard.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned* out, const unsigned* junk, int iters, int gap, unsigned* sink) {
  __shared__ u32x4 poison[64 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * 8 + wave;
  poison[threadIdx.x] = (u32x4){0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu};
  __syncthreads();
  const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)&poison[threadIdx.x];
  unsigned acc = 0;
  for (int i = 0; i < iters; ++i) {
    u32x4 v = {0xA5000000u | (unsigned)i, (unsigned)lane, (unsigned)gw, 0x5A5A0000u | (unsigned)(i & 0xffff)};
    unsigned* q = out + ((size_t)gw * iters + i) * 256 + lane * 4;
    if (MODE == 3) {
      for (int g = 0; g < gap; ++g) {
        unsigned t;
        const unsigned* jp = junk + ((size_t)(gw * 37 + i * 131 + g * 8191) % (1u << 22)) * 64 + lane;
        asm volatile("global_load_dword %0, %1, off" : "=v"(t) : "v"(jp) : "memory");
        acc += 0;      // never waited for inside the loop: the loads just occupy the memory pipe
        (void)t;
      }
    }
    // explicit registers v[40:43]: filled by four moves, stored, then (mode-dependent) overwritten
    if (MODE == 0) {
      asm volatile(
          "v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %4\n\t"
          "global_store_dwordx4 %0, v[40:43], off\n\t"
          "s_nop 0"
          ::"v"(q), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])
          : "memory", "v40", "v41", "v42", "v43");
    } else if (MODE == 1 || MODE == 3) {
      asm volatile(
          "v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %4\n\t"
          "global_store_dwordx4 %0, v[40:43], off\n\t"
          "s_nop 0\n\t"
          "ds_read_b128 v[40:43], %5\n\t"
          "s_waitcnt lgkmcnt(0)"
          ::"v"(q), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(lds)
          : "memory", "v40", "v41", "v42", "v43");
    } else {
      asm volatile(
          "v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %4\n\t"
          "global_store_dwordx4 %0, v[40:43], off\n\t"
          "s_nop %5\n\t"
          "v_mov_b32 v40, 0xDEADBEEF\n\tv_mov_b32 v41, 0xDEADBEEF\n\tv_mov_b32 v42, 0xDEADBEEF\n\tv_mov_b32 v43, 0xDEADBEEF"
          ::"v"(q), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "n"(MODE == 2 ? 0 : MODE - 3)
          : "memory", "v40", "v41", "v42", "v43");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345u) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 256;
  const int blocks = argc > 2 ? atoi(argv[2]) : 1024;
  const int gap = argc > 3 ? atoi(argv[3]) : 2;
  const size_t words = (size_t)blocks * 8 * iters * 256;
  unsigned *out, *junk, *sink;
  hipMalloc(&out, words * 4);
  hipMalloc(&junk, ((size_t)1 << 22) * 64 * 4 + 4096);
  hipMalloc(&sink, 64);
  std::vector<unsigned> h(words);
  for (int mode = 0; mode < 7; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(out, 0, words * 4);
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(512), 0, 0, out, junk, iters, gap, sink);
      else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(512), 0, 0, out, junk, iters, gap, sink);
      else if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(512), 0, 0, out, junk, iters, gap, sink);
      else if (mode == 3) continue;       // (the loads-in-flight variant wrote its own never-awaited load results over live registers: void)
      else if (mode == 4) hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(512), 0, 0, out, junk, iters, gap, sink);
      else if (mode == 5) hipLaunchKernelGGL(probe<5>, dim3(blocks), dim3(512), 0, 0, out, junk, iters, gap, sink);
      else hipLaunchKernelGGL(probe<6>, dim3(blocks), dim3(512), 0, 0, out, junk, iters, gap, sink);
      if (hipDeviceSynchronize() != hipSuccess) { printf("mode %d: launch failed\n", mode); return 1; }
      hipMemcpy(h.data(), out, words * 4, hipMemcpyDeviceToHost);
      size_t poison = 0, wrong = 0;
      for (size_t w = 0; w < words; w += 4) {
        const size_t rec = w / 256;            // (gw, i)
        const unsigned i = (unsigned)(rec % iters), gw = (unsigned)(rec / iters), lane = (unsigned)((w % 256) / 4);
        const unsigned e0 = 0xA5000000u | i, e1 = lane, e2 = gw, e3 = 0x5A5A0000u | (i & 0xffff);
        for (int k = 0; k < 4; ++k) {
          const unsigned got = h[w + k], exp = k == 0 ? e0 : (k == 1 ? e1 : (k == 2 ? e2 : e3));
          if (got == 0xDEADBEEFu) ++poison;
          else if (got != exp) ++wrong;
        }
      }
      printf("mode %d rep %d: %zu words, poisoned %zu, otherwise wrong %zu\n", mode, rep, words, poison, wrong);
    }
  }
  return 0;
}
