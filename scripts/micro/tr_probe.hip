// Probe of ds_read_b64_tr_b16 on gfx950: LDS holds u16 value = its own element index; every lane passes its own byte
// address; the 4 u16 each lane receives are printed as (source element index), for two address patterns.
//   hipcc -O2 --offload-arch=gfx950 -o tr_probe tr_probe.hip && ./tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const int* addr_elems, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) uint16_t*)lds + addr_elems[threadIdx.x] * 2;
  uint2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = r.x & 0xffff; out[threadIdx.x * 4 + 1] = r.x >> 16;
  out[threadIdx.x * 4 + 2] = r.y & 0xffff; out[threadIdx.x * 4 + 3] = r.y >> 16;
}
int main() {
  int h_addr[64]; uint16_t h_out[256];
  int* d_addr; uint16_t* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int pat = 0; pat < 3; ++pat) {
    for (int l = 0; l < 64; ++l) {
      if (pat == 0) h_addr[l] = l * 4;                              // lane l -> its own consecutive 8-byte chunk
      if (pat == 1) h_addr[l] = (l & 15) * 64 + (l >> 4) * 4;       // 16 rows of 64 elements, lane group picks a 4-element column block
      if (pat == 2) h_addr[l] = (l % 4) * 72 * 0 + (l / 16) * 1024 + ((l % 16) / 4) * 72 + (l % 4) * 4;   // [4 rows stride 72][16 cols] block per 16-lane group
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("pattern %d\n", pat);
    for (int l = 0; l < 64; ++l) printf("lane %2d addr %4d -> %4d %4d %4d %4d\n", l, h_addr[l], h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
  }
  return 0;
}
