// Shared device/host helpers for librvb (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

namespace rvb {

typedef uint16_t bf16_t;  // raw bfloat16 bits (storage type of the bf16 compute mode)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__host__ __device__ inline float bf16_to_f32(bf16_t v) {
  union { uint32_t u; float f; } x;
  x.u = ((uint32_t)v) << 16;
  return x.f;
}
// round-to-nearest-even (matches torch's float->bfloat16 conversion).  Device code uses the gfx950 conversion instruction
// (v_cvt_pk_bf16_f32, RNE; one instruction per PAIR through pack2_bf16 below instead of ~8 VALU + a divergent NaN branch
// per value -- the software form was a third of the attention kernel's instruction stream)
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __bf16 r = (__bf16)f;
  return __builtin_bit_cast(bf16_t, r);
#else
  union { uint32_t u; float f; } x;
  x.f = f;
  if ((x.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((x.u >> 16) | 0x40);  // NaN
  x.u += 0x7fffu + ((x.u >> 16) & 1u);
  return (bf16_t)(x.u >> 16);
#endif
}
// two floats -> two bf16 packed in one dword (lo in bits 0..15)
__device__ inline uint32_t pack2_bf16(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const f32x2_t v = {lo, hi};
  const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return __builtin_bit_cast(uint32_t, r);
}

typedef uint8_t fp8_t;    // raw OCP e4m3 ("e4m3fn": no infinities, max 448) bits

// four floats -> four fp8 bytes in one dword (a in bits 0..7), round-to-nearest-even, saturating at +-448
__device__ inline uint32_t pack4_fp8(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (uint32_t)r;
}
// how many of eight / four values lie outside what e4m3 can hold once scaled (pack4_fp8 clips them to +-448): the fp8 mode's
// saturation counters (rvb_get_fp8_saturation)
__device__ inline unsigned fp8_clipped(float a, float b, float c, float d) {
  return (unsigned)(fabsf(a) > 448.f) + (unsigned)(fabsf(b) > 448.f) + (unsigned)(fabsf(c) > 448.f) + (unsigned)(fabsf(d) > 448.f);
}
// host-side e4m3 encode (weights) / decode (tests), same rounding
__host__ __device__ inline uint8_t f32_to_fp8_host(float f) {
  if (f != f) return 0x7f;
  const uint8_t sgn = f < 0.f ? 0x80 : 0;
  float v = f < 0.f ? -f : f;
  if (v >= 448.f) return sgn | 0x7e;
  if (v < 0.0009765625f) return sgn;                       // below half of the smallest subnormal (2^-9 / 2): zero
  int e;
  const float m = frexpf(v, &e);                             // v = m * 2^e, m in [0.5, 1)
  int E = e - 1 + 7;
  if (E <= 0) {                                              // subnormal: multiples of 2^-9
    const int M = (int)nearbyintf(ldexpf(v, 9));
    return sgn | (uint8_t)(M >= 8 ? 0x08 : M);
  }
  int M = (int)nearbyintf((m * 2.f - 1.f) * 8.f);
  if (M == 8) { M = 0; E += 1; }
  if (E > 15 || (E == 15 && M == 7)) return sgn | 0x7e;
  return sgn | (uint8_t)((E << 3) | M);
}
__host__ __device__ inline float fp8_to_f32_host(uint8_t b) {
  const int E = (b >> 3) & 15, M = b & 7;
  const float v = E == 0 ? ldexpf((float)M, -9) : ldexpf(1.f + M / 8.f, E - 7);
  return (b & 0x80) ? -v : v;
}

template <typename T> struct Cvt;
template <> struct Cvt<float> {
  __host__ __device__ static inline float to_f32(float v) { return v; }
  __host__ __device__ static inline float from_f32(float v) { return v; }
};
template <> struct Cvt<fp8_t> {     // only so that generic code instantiates; fp8 stores go through pack4_fp8
  __host__ __device__ static inline float to_f32(fp8_t v) { return fp8_to_f32_host(v); }
  __host__ __device__ static inline fp8_t from_f32(float v) { return f32_to_fp8_host(v); }
};
template <> struct Cvt<bf16_t> {
  __host__ __device__ static inline float to_f32(bf16_t v) { return bf16_to_f32(v); }
  __host__ __device__ static inline bf16_t from_f32(float v) { return f32_to_bf16(v); }
};

// One 16-byte vector of T per lane per operand = one "K chunk" of 64 bytes per matrix row:
//   bf16 : 8 elems/lane, one v_mfma_f32_16x16x32_bf16       (K chunk = 32)
//   f32  : 4 elems/lane, four v_mfma_f32_16x16x4_f32        (K chunk = 16)
// A operand: lane l holds row (l&15), bytes [(l>>4)*16, +16) of the chunk; B operand likewise
// for column (l&15).  The k <-> (lane group, element) assignment is the same for A and B, so the
// contraction is exact whatever k order the hardware uses inside one instruction.
// C/D: lane l holds C[(l>>4)*4 + r][l&15], r = 0..3.
template <typename T> struct Mma16;
template <> struct Mma16<bf16_t> {
  static constexpr int KC = 32;   // elements of K per chunk
  static constexpr int VE = 8;    // elements per 16-byte vector
  __device__ static inline void run(const uint4& a, const uint4& b, f32x4_t& c) {
    union U { uint4 u; bf16x8_t v; };
    U ua, ub;
    ua.u = a; ub.u = b;
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, c, 0, 0, 0);
  }
};
template <> struct Mma16<float> {
  static constexpr int KC = 16;
  static constexpr int VE = 4;
  __device__ static inline void run(const uint4& a, const uint4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// reductions over the four lanes {l, l^16, l^32, l^48} (the 16-lane rows of a wave) with gfx950's row-swap instructions:
// v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of the second, so with both
// operands = v the two results hold v[l] and v[l^16] between them; v_permlane32_swap likewise for the 32-lane halves.
// VALU only -- __shfl_xor(.., 16 / 32) goes through ds_bpermute (an LDS round trip on the softmax's critical path).
__device__ inline float rows_max(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ inline float rows_sum(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// 64-lane sum without LDS: four DPP steps inside every 16-lane row (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then the
// row swaps above.  About 10 VALU instructions against six ds_bpermute round trips for the __shfl_xor form (wave_sum).  The
// summation tree differs from wave_sum's, i.e. the result can differ in the last bit: used where that is allowed (bf16 engine).
template <int CTRL> __device__ inline float dpp_f32(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
__device__ inline float wave_sum_dpp(float v) {
  v += dpp_f32<0xB1>(v);       // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);       // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);      // row_half_mirror
  v += dpp_f32<0x140>(v);      // row_mirror
  return rows_sum(v);
}

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------ status codes (include/rvb.h)
enum { OK = 0, E_ARG = -1, E_HIP = -2, E_STATE = -3, E_NOMEM = -4, E_UNSUPPORTED = -5, E_TIMEOUT = -7 /* -6: corrupt audio data (audio.cpp) */ };

}  // namespace rvb

#define RVB_HIP_CHECK(expr)                                                             \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      rvb::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                \
      return rvb::E_HIP;                                                                \
    }                                                                                   \
  } while (0)

namespace rvb {
void set_error(const std::string& msg);  // thread-local last error (engine.hip)
const char* last_error();
// Tuning / diagnosis switches (kernel variants, A/B legs of scripts/, the measured-slower alternatives the tuning log keeps):
// the PRODUCT library never reads them -- lab_env() returns nullptr there and every switch takes its default.  Only
// librvb_test.so (engine.hip compiled with -DRVB_TEST_API; reverb_amd._lib: RVB_LAB=1) consults the environment.
const char* lab_env(const char* name);
}
