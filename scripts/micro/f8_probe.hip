// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3, unit block scales) on gfx950: checks the operand layout
// assumed by the fp8 GEMM -- lane l holds row (l & 31), k = 32*(l >> 5) + j for byte j of its 32 bytes; C/D as every 32x32
// MFMA: col = l & 31, row = (r & 3) + 8*(r >> 2) + 4*(l >> 5) -- against a host reference, then a second run with a
// non-unit scale on A to see which byte of the scale register applies.
//   hipcc -O2 --offload-arch=gfx950 -o f8_probe f8_probe.hip && ./f8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// e4m3fn encode of small values exactly representable (host)
static uint8_t enc(float v) {
  if (v == 0.f) return 0;
  uint8_t s = v < 0 ? 0x80 : 0; v = fabsf(v);
  int e; float m = frexpf(v, &e);            // v = m * 2^e, m in [0.5,1)
  int E = e - 1 + 7;                          // exponent field for 1.xxx * 2^(e-1)
  float frac = m * 2.f - 1.f;                 // [0,1)
  int M = (int)lrintf(frac * 8.f);
  if (M == 8) { M = 0; E += 1; }
  if (E <= 0) { M = (int)lrintf(v / ldexpf(1.f, -9)); return s | (uint8_t)M; }     // subnormal: units of 2^-9
  return s | (uint8_t)((E << 3) | M);
}
static float dec(uint8_t b) {
  int s = b >> 7, E = (b >> 3) & 15, M = b & 7;
  float v = E == 0 ? ldexpf((float)M, -9) : ldexpf(1.f + M / 8.f, E - 7);
  return s ? -v : v;
}

__global__ void k(const uint8_t* A, const uint8_t* B, float* C, int scale_a) {
  const int l = threadIdx.x;
  i32x8 a, b;
  const int* pa = (const int*)(A + (l & 31) * 64 + (l >> 5) * 32);
  const int* pb = (const int*)(B + (l & 31) * 64 + (l >> 5) * 32);
  for (int j = 0; j < 8; ++j) { a[j] = pa[j]; b[j] = pb[j]; }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, 0x7f7f7f7f);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

int main() {
  static uint8_t hA[32 * 64], hB[32 * 64];
  static float fA[32 * 64], fB[32 * 64], ref[32 * 32], got[32 * 32];
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (int)((s >> 16) % 9) - 4; };
  for (int i = 0; i < 32 * 64; ++i) { float a = rnd() * 0.5f, b = rnd() * 0.25f; hA[i] = enc(a); hB[i] = enc(b); fA[i] = dec(hA[i]); fB[i] = dec(hB[i]); }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double t = 0; for (int kk = 0; kk < 64; ++kk) t += (double)fA[i * 64 + kk] * fB[j * 64 + kk]; ref[i * 32 + j] = (float)t; }
  uint8_t *dA, *dB; float* dC;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(got));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  const int scales[5] = {0x7f7f7f7f, 0x7f7f7f80, 0x7f7f807f, 0x7f807f7f, (int)0x807f7f7fu};
  for (int t = 0; t < 5; ++t) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, scales[t]);
    hipMemcpy(got, dC, sizeof(got), hipMemcpyDeviceToHost);
    double md = 0, ratio = 0; int nz = 0;
    for (int i = 0; i < 1024; ++i) { md = fmax(md, fabs(got[i] - ref[i])); if (ref[i] != 0) { ratio += got[i] / ref[i]; ++nz; } }
    printf("scale_a=%08x  max|got-ref|=%g  mean(got/ref)=%g   got[0..3]=%g %g %g %g ref=%g %g %g %g\n", scales[t], md, ratio / nz, got[0], got[1], got[2], got[3], ref[0], ref[1], ref[2], ref[3]);
  }
  return 0;
}
