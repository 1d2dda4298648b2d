// Large-shape MFMA GEMM for gfx950 (same contract as gemm.hip, used when K is a multiple of the
// 128-byte K step):  256x256 output tile, 8 waves (2 along M x 4 along N, 128x64 per wave =
// 8x4 MFMA 16x16 fragments), operands go HBM -> LDS directly with `global_load_lds_dwordx4`
// (LDS-DMA, no VGPR round trip, no ds_write pass), two LDS stages of 64 KiB, one barrier per K step,
// fragments double-buffered in registers:
//
//     MFMAs of k32 slice 0 of tile t | ds_read slice 1 ;
//     wait(own DMA of tile t+1) ; barrier ; issue DMA of tile t+2 into the stage tile t just left ;
//     MFMAs of slice 1 of tile t | ds_read slice 0 of tile t+1      <- the DMA lands underneath both
//
// The LDS image is lane-linear (8 rows x 128 B per wave-instruction); bank conflicts of the
// 128-byte rows are removed by an XOR swizzle applied to the SOURCE column and to the read:
// LDS[row][c] = G[row][c ^ ((row>>1)&7)]  (16-byte columns), which makes every 16-lane group of a
// ds_read_b128 touch 16 distinct 16-byte slots of the 256-byte bank row.
// Out-of-range rows are clamped (their results are never stored); K tails are not supported here
// (gemm.hip handles them).
#include "common.h"
#include "kernels.h"

#include <cstdlib>
#include <type_traits>

namespace rvb {

static constexpr int B2M = 256, B2N = 256;
static constexpr int ROW2 = 128;                         // bytes of K per tile row
static constexpr int STAGE2 = (B2M + B2N) * ROW2;        // 64 KiB
static constexpr int GEMM2_LDS = 2 * STAGE2;             // 128 KiB
static constexpr int GEMM2_DEFAULT_FLAGS = 0;
// tile order: GROUP_M_AUTO = 8-row groups for short K, row-major for K >= 2048 (profiles/r03_gemm_bench_switches.txt: with the
// phase-interleaved loop ffn2 runs 1031 vs 889 and embed 1314 vs 1181 TFLOP/s row-major, ffn1 / qkv 901 vs 858 / 1021 vs 967 grouped)
static constexpr int GROUP_M_AUTO = -2, GEMM2_DEFAULT_GROUP_M = GROUP_M_AUTO;

__device__ inline float act_apply2(float v, int act) {
  if (act == ACT_SILU) return v / (1.0f + expf(-v));
  if (act == ACT_RELU) return fmaxf(v, 0.0f);
  return v;
}

__device__ inline void glds16(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// MMA32 (bf16 only): v_mfma_f32_32x32x16_bf16 instead of v_mfma_f32_16x16x32_bf16 -- the same LDS bytes per flop (a lane's
// 16-byte vector is 8 k-values of one of 32 rows instead of one of 16), half the MFMA instructions, and the shape whose
// issue rate reaches the 2.5 PFLOP/s peak (16x16x32 tops out ~13 % lower, MI355X_MICROARCH.md / cdna_hip_programming.md 3).
//
// T = fp8_t (one byte, OCP e4m3): the same tile geometry with twice the K per 128-byte row, multiplied by
// v_mfma_scale_f32_32x32x64_f8f6f4 at unit block scales (twice the bf16 MFMA rate, half the fill and fragment bytes per
// flop).  Scaling is per tensor for the activations (a_scale, calibrated) and per output channel for the weights (w_scale):
// C = act(a_scale * w_scale[n] * sum_k A8[m][k] W8[n][k] + bias[n]) ...; OutT = fp8_t divides by out_scale and saturates.
typedef __attribute__((ext_vector_type(8))) int i32x8_t;

// One 16-byte global store the compiler's waitcnt bookkeeping does not see (gemm2p_kernel's full-tile epilogue explains why).
// s_nop 1: gfx950 wants TWO wait states between a store of more than 8 bytes and a VALU write of its data registers -- hipcc
// places them for the stores it knows (one unrelated VALU instruction + s_nop 0 in its own code), its hazard recognizer does
// not look into inline asm, and with one (s_nop 0) 0.11 % of the stored words carried the NEXT value of the register on a
// busy chip (scripts/micro/store_hazard.hip, profiles/r04_store_hazard.txt; in the first form of this epilogue: 16-128
// wrong elements per million in one kernel whose register allocation put a v_add / v_mad right behind a store).  An LDS
// read into the same registers right behind the store is safe (0 of 2.7e8 words).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ inline void store16_hidden(void* q, unsigned a, unsigned b, unsigned c, unsigned d) {
  const u32x4_t v = {a, b, c, d};
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(q), "v"(v) : "memory");
}
// ... and its 8-byte sibling (ACT_GLU: a lane's eight columns are four gated values)
__device__ inline void store8_hidden(void* q, unsigned a, unsigned b) {
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const u32x2_t v = {a, b};
  asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" ::"v"(q), "v"(v) : "memory");
}
// a * sigmoid(b) as glu_dw_kernel's bf16 staging computes it
__device__ inline float glu_gate(float a, float b) { return a * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * b)); }

template <typename T, typename OutT, bool CONV, bool MMA32>
__global__ __launch_bounds__(512) void gemm2_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool F8 = std::is_same<T, fp8_t>::value;
  constexpr int VE = 16 / (int)sizeof(T);
  constexpr int BKE = ROW2 / (int)sizeof(T);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  if (p.dbg && tid == 0) {
    p.dbg[blockIdx.x * 6 + 0] = wall_clock64();
    p.dbg[blockIdx.x * 6 + 4] = __builtin_amdgcn_s_getreg(63492);      // HW_ID
    p.dbg[blockIdx.x * 6 + 5] = __builtin_amdgcn_s_getreg(63508);      // XCC_ID
  }

  const int tiles_n = (p.N + B2N - 1) / B2N;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  if (p.group_m > 1) {
    // grouped order inside the XCD's contiguous run: walk `group_m` row tiles down, then the next column, so that the
    // ~32 tiles an XCD runs at a time form a group_m x (32 / group_m) patch (for N = 4096: 8 + 4 operand panels per
    // wave of tiles instead of 2 + 16) and successive patches keep their A panels in that XCD's L2
    const int tiles_m = (p.M + B2M - 1) / B2M;
    const int per_group = p.group_m * tiles_n;
    const int g = bid / per_group;
    const int first_m = g * p.group_m;
    const int gsz = min(tiles_m - first_m, p.group_m);
    const int in_g = bid - g * per_group;
    tm = first_m + in_g % gsz;
    tn = in_g / gsz;
  } else {
    tm = bid / tiles_n; tn = bid - tm * tiles_n;
  }
  const int m0 = tm * B2M, n0 = tn * B2N;

  const T* __restrict__ A = (const T*)p.A;
  const T* __restrict__ W = (const T*)p.W;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  // ---- DMA source pointers: wave w stages rows [32w, 32w+32) of A and of W, 8 rows per instruction
  const int lrow = lane >> 3, lcol = lane & 7;
  const T* a_src[4];
  const T* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + lrow;
    const int gcol = (lcol ^ ((row >> 1) & 7)) * VE;     // swizzled 16-byte source column
    int m = m0 + row;
    if (m >= p.M) m = p.M - 1;
    if (CONV) {
      const int tf = p.cT2 * p.cF2;
      const int b = m / tf;
      const int rem = m - b * tf;
      const int t2 = rem / p.cF2, f2 = rem - t2 * p.cF2;
      a_src[i] = A + (((size_t)b * p.cT1 + 2 * t2) * p.cF1 + 2 * f2) * (size_t)p.cC + gcol;
    } else {
      a_src[i] = A + (size_t)m * p.lda + gcol;
    }
    int n = n0 + row;
    if (n >= p.N) n = p.N - 1;
    w_src[i] = W + (size_t)n * p.ldw + gcol;
  }

  auto issue = [&](int kt, int buf) {
    size_t koff = (size_t)kt * BKE;
    if (CONV) {   // the whole 128-byte K step lies inside one (kh,kw) tap: cC % BKE == 0
      const int k0 = kt * BKE;
      const int kk = k0 / p.cC;
      const int cin = k0 - kk * p.cC;
      const int kh = kk / 3, kw = kk - kh * 3;
      koff = ((size_t)kh * p.cF1 + kw) * p.cC + cin;
    }
    // eight 1-KiB LDS-DMA pieces (4 of A, 4 of W) in ONE asm statement: hipcc's waitcnt pass does not
    // see them (it would otherwise drain vmcnt(0) before every ds_read that follows a DMA issue), so
    // the explicit `s_waitcnt vmcnt(0)` at the top of the K loop is what orders them.  M0 carries the
    // wave-uniform LDS destination; it is saved and restored because the compiler owns it.
    const unsigned ldsA = lds_base + (unsigned)(buf * STAGE2 + (wave * 32) * ROW2);
    const T* pa0 = a_src[0] + koff; const T* pa1 = a_src[1] + koff; const T* pa2 = a_src[2] + koff; const T* pa3 = a_src[3] + koff;
    const size_t wk = (size_t)kt * BKE;
    const T* pw0 = w_src[0] + wk; const T* pw1 = w_src[1] + wk; const T* pw2 = w_src[2] + wk; const T* pw3 = w_src[3] + wk;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %9\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, off\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, off\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, off\n\t"
        "s_add_u32 m0, m0, 0x7400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %5, off\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %6, off\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %7, off\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %8, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(pa0), "v"(pa1), "v"(pa2), "v"(pa3), "v"(pw0), "v"(pw1), "v"(pw2), "v"(pw3), "s"(ldsA)
        : "memory", "scc");
  };

  constexpr bool M32 = (MMA32 && sizeof(T) == 2) || F8;     // 32x32 accumulator blocks
  // 16x16 fragments: acc[i][j] = rows 16i.., cols 16j.. of the wave's 128x64 tile (C/D: col = lane&15, row = 4*(lane>>4)+r).
  // 32x32 blocks (M32): acc32[bi][bj] = rows 32bi.., cols 32bj.. (C/D: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
  f32x4_t acc[M32 ? 1 : 8][M32 ? 1 : 4];
  f32x16_t acc32[M32 ? 4 : 1][M32 ? 2 : 1];
  if constexpr (M32) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }

  // fragment addressing.  16x16x32: lane = (row 0..15, 16-byte column group 0..3 of the 64-byte k32 slice).
  // 32x32x16: lane = (row 0..31, half g = lane>>5); MFMA t (0/1) of a k32 slice takes 16-byte columns 2t + g.  Which k a
  // (lane group, element) slot carries is free as long as A and B agree (common.h), and the source swizzle
  // (row>>1)&7 keeps either read pattern conflict-free: a ds_read_b128 lane group always holds every swizzle class twice,
  // once on an even and once on an odd row.
  const int frow = M32 ? (lane & 31) : (lane & 15), lgrp = M32 ? (lane >> 5) : (lane >> 4);
  const int swz = (frow >> 1) & 7;
  int roff[2];     // byte offset of this lane's 16-byte vector inside a row, per 64-byte chunk
  roff[0] = ((0 * 4 + lgrp) ^ swz) << 4;
  roff[1] = ((1 * 4 + lgrp) ^ swz) << 4;
  int roff_t1[2];  // M32 bf16: second MFMA of the slice (columns 2 + g)
  roff_t1[0] = ((0 * 4 + 2 + lgrp) ^ swz) << 4;
  roff_t1[1] = ((1 * 4 + 2 + lgrp) ^ swz) << 4;
  if constexpr (F8) {   // one 32x32x64 MFMA per k64 slice: the lane's 32 bytes = 16-byte columns 2g, 2g + 1 of the slice
    roff[0] = ((0 * 4 + 2 * lgrp) ^ swz) << 4; roff_t1[0] = ((0 * 4 + 2 * lgrp + 1) ^ swz) << 4;
    roff[1] = ((1 * 4 + 2 * lgrp) ^ swz) << 4; roff_t1[1] = ((1 * 4 + 2 * lgrp + 1) ^ swz) << 4;
  }

  // ---- main loop: k32 slices q = 2*kt + s stream through two register buffers of fragments.  Block u multiplies
  // slice u while slice u+1 is read (two MFMAs, one ds_read, ... so the reads leave early and land under the MFMAs);
  // a stage is "entered" (own DMA awaited, barrier, the other buffer refilled with the stage after it) between two
  // blocks, so there is MFMA work on both sides of every barrier.  Measured against the plain
  // wait-barrier-refill-read-multiply loop in scripts/micro/gemm_lab.hip: +7..11 % on the engine's shapes.
  const int nk = p.K / BKE;
  // bf16 16x16 path: a residual with no activation in front of it starts out in the accumulators (see the prologue)
  const bool res_acc = sizeof(T) == 2 && !M32 && !F8 && p.res != nullptr && p.act == ACT_NONE && p.alpha != 0.f &&
                       (p.N & 3) == 0 && (p.ldres & 3) == 0 && ((size_t)p.res & 15) == 0;      // 16-byte residual vectors, whole inside N
  if constexpr (F8) {
    // fp8: the plain loop.  A k64 slice is eight 64-cycle MFMAs per wave, and the SIMD's other wave multiplies while this
    // one waits for its 12 fragment reads; the register-pipelined form of the bf16 path spills here (fragments are
    // 8-register tuples: 36 scratch accesses per K step).
    const int a_off = (wr * 128 + frow) * ROW2;
    const int b_off = B2M * ROW2 + (wc * 64 + frow) * ROW2;
    if (p.prio && wave >= 4) __builtin_amdgcn_s_setprio(1);
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of stage kt have landed
      __syncthreads();                                   // ... and so have everybody else's
      if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
      const char* st = smem + cur * STAGE2;
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        i32x8_t a8[4], b8[2];
#pragma unroll
        for (int bi = 0; bi < 4; ++bi) {
          const uint4 lo = *(const uint4*)(st + a_off + bi * 32 * ROW2 + roff[sl]), hi = *(const uint4*)(st + a_off + bi * 32 * ROW2 + roff_t1[sl]);
          a8[bi] = (i32x8_t){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        }
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) {
          const uint4 lo = *(const uint4*)(st + b_off + bj * 32 * ROW2 + roff[sl]), hi = *(const uint4*)(st + b_off + bj * 32 * ROW2 + roff_t1[sl]);
          b8[bj] = (i32x8_t){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        }
#pragma unroll
        for (int bi = 0; bi < 4; ++bi)
#pragma unroll
          for (int bj = 0; bj < 2; ++bj)
            acc32[bi][bj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[bi], b8[bj], acc32[bi][bj], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
    }
    if (p.prio && wave >= 4) __builtin_amdgcn_s_setprio(0);
  } else if constexpr (sizeof(T) == 2) {
    const int nq = 2 * nk;
    uint4 fa[2][8], fb[2][4];
    const int a_off = (wr * 128 + frow) * ROW2;
    const int b_off = B2M * ROW2 + (wc * 64 + frow) * ROW2;
    auto read_slice = [&](auto bufc, int q) __attribute__((always_inline)) {
      constexpr int buf = decltype(bufc)::value;
      const char* st = smem + ((q >> 1) & 1) * STAGE2;
      const int ro = roff[q & 1];
      if constexpr (M32) {
        // fa[2*bi + t] = A rows 32bi + (lane&31), MFMA t of the slice; fb[2*bj + t] likewise for W rows (output columns)
        const int r1 = roff_t1[q & 1];
#pragma unroll
        for (int bi = 0; bi < 4; ++bi) {
          fa[buf][2 * bi] = *(const uint4*)(st + a_off + bi * 32 * ROW2 + ro);
          fa[buf][2 * bi + 1] = *(const uint4*)(st + a_off + bi * 32 * ROW2 + r1);
        }
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) {
          fb[buf][2 * bj] = *(const uint4*)(st + b_off + bj * 32 * ROW2 + ro);
          fb[buf][2 * bj + 1] = *(const uint4*)(st + b_off + bj * 32 * ROW2 + r1);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[buf][i] = *(const uint4*)(st + a_off + i * 16 * ROW2 + ro);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[buf][j] = *(const uint4*)(st + b_off + j * 16 * ROW2 + ro);
      }
    };
    auto mma_slice = [&](auto bufc) __attribute__((always_inline)) {
      constexpr int buf = decltype(bufc)::value;
      if constexpr (M32) {
        union U { uint4 u; bf16x8_t v; };
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int bi = 0; bi < 4; ++bi)
#pragma unroll
            for (int bj = 0; bj < 2; ++bj) {
              U ua, ub;
              ua.u = fa[buf][2 * bi + t]; ub.u = fb[buf][2 * bj + t];
              acc32[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc32[bi][bj], 0, 0, 0);
            }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) Mma16<T>::run(fa[buf][i], fb[buf][j], acc[i][j]);
      }
    };
    auto block = [&](auto mbufc, int q) __attribute__((always_inline)) {     // multiply buffer mbuf, read slice q into the other
      constexpr int mb = decltype(mbufc)::value;
      read_slice(std::integral_constant<int, 1 - mb>(), q);
      mma_slice(mbufc);
      if constexpr (M32) {
#pragma unroll
        for (int r = 0; r < 12; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 16 MFMAs of twice the length: one per fragment read
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 12; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);     // MFMA first: they depend on the previous block's reads only
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // one ds_read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      }
    };
    auto enter_stage = [&](int kt) __attribute__((always_inline)) {          // kt >= 1
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of stage kt have landed
      __syncthreads();                                   // ... everybody's have; nobody reads stage kt-1 any more
      if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);      // into the buffer stage kt-1 just left
    };
    std::integral_constant<int, 0> b0;
    std::integral_constant<int, 1> b1;
    if (p.prio && wave >= 4) __builtin_amdgcn_s_setprio(1);   // the later-dispatched half loses every arbitration otherwise
    bool stage1_issued = false;
    if constexpr (!M32) {
      if (res_acc) {
        // C = res + alpha * (A.W^T + bias) = alpha * (res / alpha + A.W^T + bias): the fp32 residual tile starts out in
        // the accumulators instead of being added in the epilogue -- there, every slab of every wave waited a full memory
        // latency for its residual rows before it could store (17 of the 24.5 us a K = 1024 tile spent in its epilogue,
        // scripts/gemm_timeline.py).  The wave's 128 x 64 sub-tile is read as 32 row-contiguous 16-byte vectors per lane
        // (4 rows x 256 B per instruction, all in flight together under stage 0's DMA) INTO the accumulator registers,
        // and each 16-row slab is then turned into the MFMA C layout through the not-yet-used second LDS stage
        // (accumulator-layout loads straight from memory are 64-byte scatters: 50 us per tile, measured).
        const float inv_alpha = 1.0f / p.alpha;
        const int lr = lane >> 4, lc = (lane & 15) * 4;
        const int colr = min(n0 + wc * 64 + lc, p.N - 4);
#pragma unroll
        for (int q = 0; q < 32; ++q) {
          const int row = min(m0 + wr * 128 + (q >> 2) * 16 + (q & 3) * 4 + lr, p.M - 1);
          acc[q >> 2][q & 3] = *(const f32x4_t*)(p.res + (size_t)row * p.ldres + colr);
        }
        issue(0, 0);      // behind the residual in the (in-order) vmcnt queue: slab i below waits for its own four loads only
        constexpr int TROW = 64 * 4 + 16;
        char* tb = smem + STAGE2 + wave * (16 * TROW);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
          for (int q = 0; q < 4; ++q) *(f32x4_t*)(tb + (q * 4 + lr) * TROW + lc * 4) = acc[i][q];
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = *(const float*)(tb + (lr * 4 + r) * TROW + (j * 16 + (lane & 15)) * 4) * inv_alpha;
          __builtin_amdgcn_wave_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stage 0's DMA pieces
      }
    }
    if (!res_acc) {
      issue(0, 0);
      if (nk > 1) {
        issue(1, 1);
        stage1_issued = true;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // stage 0 (the older group of 8 pieces)
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    if (nk > 1 && !stage1_issued) issue(1, 1);             // the residual's transposition used stage 1 as scratch until here
    if (p.dbg && tid == 0) p.dbg[blockIdx.x * 6 + 1] = wall_clock64();
    read_slice(b0, 0);
    block(b0, 1);
    if (nk > 1) {
      enter_stage(1);
      for (int u = 1; u <= nq - 5; u += 2) {
        block(b1, u + 1);
        block(b0, u + 2);
        enter_stage((u + 3) >> 1);
      }
      block(b1, nq - 2);
      block(b0, nq - 1);
    }
    mma_slice(b1);
    if (p.prio && wave >= 4) __builtin_amdgcn_s_setprio(0);
  } else {
    // f32 (parity mode, four 16x16x4 MFMAs per fragment pair): the plain loop -- the double-buffered fragments do not
    // fit next to 128 accumulator registers there
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of tile kt have landed
      __syncthreads();                                   // ... and so have everybody else's
      const char* sA = smem + cur * STAGE2 + (wr * 128 + frow) * ROW2;
      const char* sB = smem + cur * STAGE2 + B2M * ROW2 + (wc * 64 + frow) * ROW2;
      if (kt + 1 < nk) issue(kt + 1, cur ^ 1);   // other stage: last read one K step ago, every wave is past the barrier
      uint4 a0[8], b0[4], a1[8], b1[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) a0[i] = *(const uint4*)(sA + i * 16 * ROW2 + roff[0]);
#pragma unroll
      for (int j = 0; j < 4; ++j) b0[j] = *(const uint4*)(sB + j * 16 * ROW2 + roff[0]);
#pragma unroll
      for (int i = 0; i < 8; ++i) a1[i] = *(const uint4*)(sA + i * 16 * ROW2 + roff[1]);
#pragma unroll
      for (int j = 0; j < 4; ++j) b1[j] = *(const uint4*)(sB + j * 16 * ROW2 + roff[1]);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mma16<T>::run(a0[i], b0[j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mma16<T>::run(a1[i], b1[j], acc[i][j]);
    }
  }

  if (p.dbg && tid == 0) p.dbg[blockIdx.x * 6 + 2] = wall_clock64();
  // ---- epilogue ----
  // Accumulator fragments hold 4 rows x 1 column per lane: stored directly they make 2-byte/4-byte
  // scattered writes (32-64 B runs).  Each wave instead transposes one 16x64 slab at a time through
  // the idle LDS stage so that a lane owns 16 consecutive columns of one row: residual reads and
  // output writes become 16-byte vectors, 4 lanes covering a 64-column row segment.
  // bf16 mode uses the hardware exp for SiLU (the result is rounded to bf16 anyway).
  OutT* __restrict__ C = (OutT*)p.C;
  const int crow = (lane >> 4) * 4;
  const int ccol = lane & 15;
  const bool full = (m0 + B2M <= p.M) && (n0 + B2N <= p.N);
  const bool vec_ok = ((p.ldc * (int)sizeof(OutT)) % 16 == 0) && (((size_t)p.C & 15) == 0) &&
                      (p.res == nullptr || ((p.ldres % 4) == 0 && ((size_t)p.res & 15) == 0));
  if (vec_ok) {
    constexpr int SROW = 64 * 4 + 16;                      // padded slab row (bytes)
    char* slab = smem + ((nk & 1) ? STAGE2 : 0) + wave * (16 * SROW);   // the stage NOT used by the last K step
    // Read-back mapping: the lanes of one store instruction cover whole rows of the slab's 64 columns -- 16 lanes x 4 fp32,
    // 8 lanes x 8 bf16 or 4 lanes x 16 fp8, i.e. 256 / 128 / 64 contiguous bytes per row and instruction (and the same for
    // the fp32 residual reads).  Sixteen consecutive columns per lane whatever the type -- the first form of this epilogue
    // -- made every fp32 instruction touch 64 different 64-byte segments, 16 bytes each: the fp32 + residual epilogue of a
    // tile then took as long as its K = 1024 main loop (24.5 us, scripts/gemm_timeline.py).
    constexpr int CPL = sizeof(OutT) == 4 ? 4 : (sizeof(OutT) == 2 ? 8 : 16);   // columns per lane
    constexpr int LPR = 64 / CPL;                                                // lanes per slab row
    constexpr int RPP = 64 / LPR;                                                // rows per pass
    constexpr int NQ = 16 / RPP;                                                 // passes per slab
    const int orow = lane / LPR;                           // row (within a pass) this lane finishes
    const int ocol = (lane % LPR) * CPL;                   // first of its columns
    const int col0 = n0 + wc * 64 + ocol;
    float biasv[CPL], scv[F8 ? CPL : 1];
#pragma unroll
    for (int e = 0; e < CPL; ++e) biasv[e] = (p.bias && col0 + e < p.N) ? p.bias[col0 + e] : 0.0f;
    if constexpr (F8) {
#pragma unroll
      for (int e = 0; e < CPL; ++e) scv[e] = col0 + e < p.N ? p.a_scale * p.w_scale[col0 + e] : 0.0f;
    }
    const bool seg_full = col0 + CPL <= p.N;
    auto finish_v = [&](auto actf) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_wave_barrier();
        if constexpr (M32) {
          // rows [16i, 16i+16) of the wave tile = half h = i&1 of block row bi = i>>1: registers 8h .. 8h+7
          const int bi = i >> 1, h = i & 1;
          const int r32 = 4 * (lane >> 5), c32 = lane & 31;
#pragma unroll
          for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                *(float*)(slab + (8 * qq + r32 + r) * SROW + (bj * 32 + c32) * 4) = acc32[bi][bj][8 * h + 4 * qq + r];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              *(float*)(slab + (crow + r) * SROW + (j * 16 + ccol) * 4) = acc[i][j][r];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int srow = q * RPP + orow;
          float v[CPL];
#pragma unroll
          for (int c4 = 0; c4 < CPL / 4; ++c4) {
            const float4 t = *(const float4*)(slab + srow * SROW + (ocol + c4 * 4) * 4);
            v[c4 * 4 + 0] = t.x; v[c4 * 4 + 1] = t.y; v[c4 * 4 + 2] = t.z; v[c4 * 4 + 3] = t.w;
          }
          const int row = m0 + wr * 128 + i * 16 + srow;
          if (row >= p.M || col0 >= p.N) continue;
          if constexpr (F8) {
#pragma unroll
            for (int e = 0; e < CPL; ++e) v[e] *= scv[e];
          }
#pragma unroll
          for (int e = 0; e < CPL; ++e) v[e] = actf(v[e] + biasv[e]) * p.alpha;
          OutT* cp = C + (size_t)row * p.ldc + col0;
          if (seg_full) {
            if (p.res && !res_acc) {
              const float4* rp = (const float4*)(p.res + (size_t)row * p.ldres + col0);
#pragma unroll
              for (int c4 = 0; c4 < CPL / 4; ++c4) {
                const float4 t = rp[c4];
                v[c4 * 4 + 0] += t.x; v[c4 * 4 + 1] += t.y; v[c4 * 4 + 2] += t.z; v[c4 * 4 + 3] += t.w;
              }
            }
            if constexpr (sizeof(OutT) == 1) {
              const float qs = p.out_inv_scale;
              if (p.sat) {      // saturation counter of this fp8 tensor (rvb_get_fp8_saturation)
                // common path: one maximum over the lane's 16 magnitudes (v_max3 with |.| modifiers) and one compare; the exact
                // count and the atomic only when something clips (per-element compares here cost 1.5 ms per hour of audio)
                float mx = fmaxf(fabsf(v[0]), fabsf(v[1]));
#pragma unroll
                for (int e2 = 2; e2 < 16; ++e2) mx = fmaxf(mx, fabsf(v[e2]));
                if (mx * qs > 448.f) {
                  const unsigned ns = fp8_clipped(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs) + fp8_clipped(v[4] * qs, v[5] * qs, v[6] * qs, v[7] * qs) +
                                      fp8_clipped(v[8] * qs, v[9] * qs, v[10] * qs, v[11] * qs) + fp8_clipped(v[12] * qs, v[13] * qs, v[14] * qs, v[15] * qs);
                  atomicAdd(p.sat, ns);
                }
              }
              *(uint4*)cp = make_uint4(pack4_fp8(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs), pack4_fp8(v[4] * qs, v[5] * qs, v[6] * qs, v[7] * qs),
                                       pack4_fp8(v[8] * qs, v[9] * qs, v[10] * qs, v[11] * qs), pack4_fp8(v[12] * qs, v[13] * qs, v[14] * qs, v[15] * qs));
            } else if constexpr (sizeof(OutT) == 2) {
              uint4 o0;
              o0.x = pack2_bf16(v[0], v[1]);
              o0.y = pack2_bf16(v[2], v[3]);
              o0.z = pack2_bf16(v[4], v[5]);
              o0.w = pack2_bf16(v[6], v[7]);
              *(uint4*)cp = o0;
            } else {
              *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
            }
          } else {            // ragged last column segment of the matrix
            for (int e = 0; e < CPL && col0 + e < p.N; ++e) {
              float o = v[e];
              if (p.res && !res_acc) o += p.res[(size_t)row * p.ldres + col0 + e];
              if constexpr (sizeof(OutT) == 1) cp[e] = (OutT)(pack4_fp8(o * p.out_inv_scale, 0.f, 0.f, 0.f) & 0xffu);
              else cp[e] = Cvt<OutT>::from_f32(o);
            }
          }
        }
      }
    };
    // Full tiles: the stores are inline asm and the residual comes from a ring of vectors requested RING slabs ahead -- see
    // finish_fast in gemm2p_kernel (here the generic loop even loaded the residual inside the pass that stores it: one
    // exposed memory round trip and one drain of the store queue per pass, 16-32 per tile)
    constexpr int RING = sizeof(OutT) == 4 ? 3 : 2, RV = NQ * (CPL / 4);
    auto finish_fast = [&](auto actf, auto resc) {
      constexpr bool HR = decltype(resc)::value;
      float4 rring[HR ? RING : 1][HR ? RV : 1];
      auto res_issue_f = [&](int i, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const float4* rp = (const float4*)(p.res + (size_t)(m0 + wr * 128 + i * 16 + q * RPP + orow) * p.ldres + col0);
#pragma unroll
          for (int c4 = 0; c4 < CPL / 4; ++c4) rring[HR ? slot : 0][HR ? q * (CPL / 4) + c4 : 0] = rp[c4];
        }
      };
      if constexpr (HR) {
#pragma unroll
        for (int i = 0; i < RING; ++i) res_issue_f(i, i);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_wave_barrier();
        if constexpr (M32) {
          const int bi = i >> 1, h = i & 1;
          const int r32 = 4 * (lane >> 5), c32 = lane & 31;
#pragma unroll
          for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                *(float*)(slab + (8 * qq + r32 + r) * SROW + (bj * 32 + c32) * 4) = acc32[bi][bj][8 * h + 4 * qq + r];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              *(float*)(slab + (crow + r) * SROW + (j * 16 + ccol) * 4) = acc[i][j][r];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int srow = q * RPP + orow;
          float v[CPL];
#pragma unroll
          for (int c4 = 0; c4 < CPL / 4; ++c4) {
            const float4 t = *(const float4*)(slab + srow * SROW + (ocol + c4 * 4) * 4);
            v[c4 * 4 + 0] = t.x; v[c4 * 4 + 1] = t.y; v[c4 * 4 + 2] = t.z; v[c4 * 4 + 3] = t.w;
          }
          if constexpr (F8) {
#pragma unroll
            for (int e = 0; e < CPL; ++e) v[e] *= scv[e];
          }
#pragma unroll
          for (int e = 0; e < CPL; ++e) v[e] = actf(v[e] + biasv[e]) * p.alpha;
          if constexpr (HR) {
#pragma unroll
            for (int c4 = 0; c4 < CPL / 4; ++c4) {
              const float4 t = rring[i % RING][q * (CPL / 4) + c4];
              v[c4 * 4 + 0] += t.x; v[c4 * 4 + 1] += t.y; v[c4 * 4 + 2] += t.z; v[c4 * 4 + 3] += t.w;
            }
          }
          OutT* cp = C + (size_t)(m0 + wr * 128 + i * 16 + srow) * p.ldc + col0;
          if constexpr (sizeof(OutT) == 1) {
            const float qs = p.out_inv_scale;
            if (p.sat) {
              float mx = fmaxf(fabsf(v[0]), fabsf(v[1]));
#pragma unroll
              for (int e2 = 2; e2 < 16; ++e2) mx = fmaxf(mx, fabsf(v[e2]));
              if (mx * qs > 448.f) {
                const unsigned ns = fp8_clipped(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs) + fp8_clipped(v[4] * qs, v[5] * qs, v[6] * qs, v[7] * qs) +
                                    fp8_clipped(v[8] * qs, v[9] * qs, v[10] * qs, v[11] * qs) + fp8_clipped(v[12] * qs, v[13] * qs, v[14] * qs, v[15] * qs);
                atomicAdd(p.sat, ns);
              }
            }
            store16_hidden(cp, pack4_fp8(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs), pack4_fp8(v[4] * qs, v[5] * qs, v[6] * qs, v[7] * qs),
                           pack4_fp8(v[8] * qs, v[9] * qs, v[10] * qs, v[11] * qs), pack4_fp8(v[12] * qs, v[13] * qs, v[14] * qs, v[15] * qs));
          } else if constexpr (sizeof(OutT) == 2) {
            store16_hidden(cp, pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
          } else {
            store16_hidden(cp, __float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
          }
        }
        if constexpr (HR) {
          if (i + RING < 8) res_issue_f(i + RING, i % RING);
        }
      }
    };
    const bool hr = p.res != nullptr && !res_acc;
    const bool fast = full && (p.fast_epilogue & (hr ? 2 : 1));
    auto run = [&](auto actf) __attribute__((always_inline)) {
      if (fast) {
        if (hr) finish_fast(actf, std::true_type()); else finish_fast(actf, std::false_type());
      } else {
        finish_v(actf);
      }
    };
    if (p.act == ACT_SILU) {
      // raw v_exp_f32 / v_rcp_f32 (no denormal fix-ups): 5 VALU ops per element, the result is rounded to bf16 anyway
      if constexpr (sizeof(T) <= 2) run([](float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x)); });
      else run([](float x) { return x / (1.0f + expf(-x)); });
    } else if (p.act == ACT_RELU) {
      run([](float x) { return fmaxf(x, 0.0f); });
    } else {
      run([](float x) { return x; });
    }
    if (p.dbg && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); p.dbg[blockIdx.x * 6 + 3] = wall_clock64(); }
    return;
  }
  // unaligned output / residual rows: element-wise stores
  if constexpr (M32) {
    auto finish32 = [&](auto actf) {
#pragma unroll
      for (int bi = 0; bi < 4; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) {
          const int col = n0 + wc * 64 + bj * 32 + (lane & 31);
          const float bvv = (p.bias && col < p.N) ? p.bias[col] : 0.0f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wr * 128 + bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row >= p.M || col >= p.N) continue;
            float av = acc32[bi][bj][r];
            if constexpr (F8) av *= (col < p.N ? p.a_scale * p.w_scale[col] : 0.f);
            float v = actf(av + bvv) * p.alpha;
            if (p.res) v += p.res[(size_t)row * p.ldres + col];
            if constexpr (sizeof(OutT) == 1) C[(size_t)row * p.ldc + col] = (OutT)(pack4_fp8(v * p.out_inv_scale, 0.f, 0.f, 0.f) & 0xffu);
            else C[(size_t)row * p.ldc + col] = Cvt<OutT>::from_f32(v);
          }
        }
    };
    if (p.act == ACT_SILU) finish32([](float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * v)); });
    else if (p.act == ACT_RELU) finish32([](float v) { return fmaxf(v, 0.0f); });
    else finish32([](float v) { return v; });
    return;
  }
  if constexpr (!M32) {
  float bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + wc * 64 + j * 16 + ccol;
    bv[j] = (p.bias && col < p.N) ? p.bias[col] : 0.0f;
  }
  auto finish = [&](auto actf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 128 + i * 16 + crow + r;
        if (!full && row >= p.M) continue;
        const float* rrow = (p.res && !res_acc) ? p.res + (size_t)row * p.ldres : nullptr;
        OutT* crow_p = C + (size_t)row * p.ldc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = n0 + wc * 64 + j * 16 + ccol;
          if (!full && col >= p.N) continue;
          float v = actf(acc[i][j][r] + bv[j]) * p.alpha;
          if (rrow) v += rrow[col];
          crow_p[col] = Cvt<OutT>::from_f32(v);
        }
      }
    }
  };
  if (p.act == ACT_SILU) {
    if constexpr (sizeof(T) == 2) finish([](float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * v)); });
    else finish([](float v) { return v / (1.0f + expf(-v)); });
  } else if (p.act == ACT_RELU) {
    finish([](float v) { return fmaxf(v, 0.0f); });
  } else {
    finish([](float v) { return v; });
  }
  }
}

// ================================================================================================
// bf16, phase-interleaved main loop (round 3).  Same tile and fragment geometry as gemm2_kernel; what changes is
// how the K loop is fed (derivation and side-by-side measurements: scripts/micro/gemm_lab.hip `phase`,
// profiles/r03_gemm_lab_phase.txt -- +7..22 % over the register-pipelined loop above on the engine's shapes):
//
//   * a K step of 64 elements is four PHASES, one per 64x32 quadrant of the wave's 128x64 tile (16 MFMAs each), and
//     its operands are four 16-KiB HALF-TILES -- A-h0 / A-h1 = the first / second 64 rows of each wave row's 128,
//     B-h0 / B-h1 = the first / second 32 columns of each wave column's 64 -- one per phase, in consumption order:
//
//         phase   MFMAs      fragment reads issued (into registers that just died)
//           0     A0 x B0    B0 <- B-h0(t), k-half 1 of A0 <- A-h0(t)      4 + 4 ds_read_b128
//           1     A0 x B1    B1 <- B-h1(t)                                 4
//           2     A1 x B1    A  <- A-h1(t)  (both k-halves)                8
//           3     A1 x B0    k-half 0 of the next A0 <- A-h0(t+1)          4      (B0 stays in registers)
//
//     One 32-register A set (whose k-half 1 serves A0 in phases 0-1 and A1 in phases 2-3) + a 16-register set for the
//     prefetched k-half 0 of A0 + two 16-register B sets = 80 fragment registers next to the 128 accumulators; reads are
//     8 / 4 / 8 / 4 per phase -- never more than the other wave row's 16 MFMAs cover;
//   * the half-tiles stream through a ring of P_NSLOT LDS slots: phase p reads half-tile p+1 (and, for A-h0, finishes
//     half-tile p), waits -- counted `vmcnt`, never 0 in steady state -- until half-tile p+2 has landed, and requests
//     half-tile p+P_LEAD into the slot whose last reader finished two phases earlier: 3 half-tiles stay in flight ACROSS
//     the barriers instead of one 64-KiB stage drained to zero at every K step;
//   * a phase is {DMA issue, reads, vmcnt | s_barrier | 16 MFMAs, each behind a counted lgkmcnt wait for just the
//     fragments it consumes | s_barrier}, and the second wave row runs ONE BARRIER BEHIND the first: on every SIMD one
//     wave multiplies while the other reads and issues.  Without that stagger the same loop is slower than the old one
//     (846 vs 919 TFLOP/s in the lab).  (Until late in round 3 a blanket `s_waitcnt lgkmcnt(0)` sat behind the first
//     barrier: every wave then waited for all 8 of its reads before its first MFMA and the compiler's own counted waits
//     were dead code; without it the GEMMs of the bench hour take 103.3 instead of ~105.7 ms on the same box class.)
//   RAW: a wave waits for its own pieces of half-tile h before the first barrier of phase h-2; every reader is past
//   that barrier (or the one after it, for the staggered row) when it reads in phase h-1.  WAR: the reads of a phase have
//   all been consumed by that phase's MFMAs, i.e. have retired before its closing barrier G (one barrier later for the
//   staggered row); the slot is requested again in the read section of the phase two phases on, which no wave enters
//   before barrier G + 2.
// The ring takes 112 KiB; the 34 KiB above it are the epilogue's transposition slabs (and the residual prologue's), so
// neither ever shares a buffer with the operand stream.
extern int g_gemm2_flags;
static constexpr int P_NSLOT = 7, P_HT = 128 * ROW2, P_LEAD = P_NSLOT - 2, P_DEPTH = P_LEAD - 2;
static constexpr int P_SROW = 64 * 4 + 16;                        // padded fp32 slab row (bytes)
static constexpr int GEMM2P_LDS = P_NSLOT * P_HT + 8 * 16 * P_SROW;

__device__ inline void dma2(unsigned off0, unsigned off1, const void* sbase, unsigned lds0) {
  // two 1-KiB LDS-DMA pieces (8 rows x 128 B each, consecutive in LDS) from scalar base + per-lane 32-bit offsets.
  // Inline asm on purpose (see issue() above); M0 belongs to the compiler: saved and restored.
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %3\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %3\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(off0), "v"(off1), "s"(sbase), "s"(lds0)
      : "memory", "scc");
}
template <int N> __device__ inline void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// a wave-uniform pointer, provably so for the "s" operand of dma2 (uniform values that went through a VALU division live
// in VGPRs otherwise); folds away when the value already sits in SGPRs
__device__ inline const char* uniform_ptr(const char* q) {
  const unsigned long long v = (unsigned long long)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}

// T = bf16_t: 16x16x32 MFMAs on 16x16 accumulator fragments.  T = fp8_t (OCP e4m3, RVB_FP8 mode): the same ring and phases
// with twice the K per 128-byte row and v_mfma_scale_f32_32x32x64_f8f6f4 at unit block scales on 32x32 accumulator blocks
// -- a phase is then four 64-cycle MFMAs (a 64x32 quadrant = 2 x 1 blocks x two k64 slices), the same 256 matrix-pipe
// cycles and the same 8 / 4 fragment reads; scaling as in gemm2_kernel (a_scale per tensor, w_scale per output channel).
// PERSIST (full tiles, no convolution gather): one workgroup per CU walks the tiles of its XCD's contiguous run, and the
// operand stream does not stop at a tile's edge -- the last five phases of a tile request the first five half-tiles of the
// NEXT tile (same per-lane offsets, the next tile's scalar bases), so they land underneath the epilogue's stores and the
// next tile starts multiplying after one barrier: the 2 us first-stage wait and the 0.5-2 us dispatch gap of every tile
// (scripts/gemm_timeline.py) are gone.  The epilogue's slabs have their own LDS, so nothing waits for anything else.
// PMODE 2 (round 4): the persistent LOOP without the cross-tile prefetch -- a workgroup per CU walks its tiles, every tile with
// its own first-stage requests and waits; only the dispatch gap between two workgroups of a CU is gone.
// PH2 (bf16, one tile per workgroup; round 6): the K step as TWO phases instead of four --
//         X: (A0 x B0), (A0 x B1)    32 MFMAs    reads before it: A-h0 (both k-halves), B-h0, B-h1     16 ds_read_b128
//         Y: (A1 x B1), (A1 x B0)    32 MFMAs    reads before it: A-h1                                  8
//     i.e. four barriers per K step instead of eight: the 16-MFMA sections of the four-phase loop are 272 matrix-pipe cycles between
//     two barriers that cost 60-120 cycles together (scripts/gemm_timeline.py: 1.31-1.48 us per K step against 0.86 of matrix pipe).
//     64 fragment registers (one A set, two B sets; no prefetched k-half), the second wave row still one barrier behind the first.
//     The ring becomes EIGHT slots with static addresses: half-tile ty of K step t lives in slot 4 (t & 1) + ty; X(t) requests the
//     three half-tiles of X(t + 1) into the slots X(t - 1) left two phases ago, Y(t) the one of Y(t + 1); waits are vmcnt(6) / vmcnt(2)
//     (everything but what the phase itself just requested).  128 KiB of ring: the epilogue's slabs reuse slot memory (nothing is in
//     flight after the last K step, and every wave is past the last barrier).
template <typename T, typename OutT, bool CONV, int PMODE = 0, bool M32 = false, bool PH2 = false>
__global__ __launch_bounds__(512) void gemm2p_kernel(GemmArgs p) {
  static_assert(!PH2 || (sizeof(T) == 2 && !M32 && PMODE == 0), "PH2: bf16, 16x16x32 fragments, one tile per workgroup");
  // M32 (bf16, round 6): v_mfma_f32_32x32x16_bf16 on 32x32 accumulator blocks instead of v_mfma_f32_16x16x32_bf16 on 16x16
  // fragments -- the same ring, phases, fragment bytes and read counts (a lane's 16-byte vector is 8 k-values of one of 32
  // rows); a phase is 8 MFMAs of 32 cycles instead of 16 of ~17 (MI355X_MICROARCH.md: the 32x32x16 form is the one that
  // reaches the matrix pipe's peak), i.e. half the issue slots next to the SIMD's other wave.
  static_assert(!M32 || sizeof(T) == 2, "M32 is the bf16 form");
  constexpr bool PERSIST = PMODE != 0;        // one workgroup per CU walks tiles
  constexpr bool XPF = PMODE == 1;            // ... and the operand stream runs across tile edges
  static_assert(!(PERSIST && CONV), "the persistent form reuses the per-lane DMA offsets from tile to tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool F8 = std::is_same<T, fp8_t>::value;
  constexpr bool B32 = F8 || M32;             // 32x32 accumulator blocks
  constexpr int BKE = ROW2 / (int)sizeof(T);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = (p.N + B2N - 1) / B2N, tiles_m = (p.M + B2M - 1) / B2M;
  // tile order: each XCD (workgroup id mod 8) owns a contiguous run of the linear tile ids, so that its L2 serves tiles that
  // share operand panels.  One tile per workgroup: id -> the idx-th tile of the run.  PERSIST: workgroup idx of the XCD
  // takes tiles idx, idx + G/8, ... of the run (G = gridDim.x, a multiple of 8).
  int lin, lin_end, lin_step;
  {
    const int ntile = PERSIST ? tiles_m * tiles_n : (int)gridDim.x;
    const int q = ntile >> 3, r = ntile & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    lin = start + idx;
    lin_end = PERSIST ? start + q + (xcd < r ? 1 : 0) : lin + 1;
    lin_step = PERSIST ? (int)(gridDim.x >> 3) : 1;
  }
  auto tile_of = [&](int bid, int& tm, int& tn) __attribute__((always_inline)) {
    if (p.group_m > 1) {
      const int per_group = p.group_m * tiles_n;
      const int g = bid / per_group;
      const int first_m = g * p.group_m;
      const int gsz = min(tiles_m - first_m, p.group_m);
      const int in_g = bid - g * per_group;
      tm = first_m + in_g % gsz;
      tn = in_g / gsz;
    } else {
      tm = bid / tiles_n; tn = bid - tm * tiles_n;
    }
  };
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  // ring state: lives across the tiles of a persistent workgroup
  int s_cur = 0;                    // slot of half-tile ph
  int s_rd = 1;                     // slot of half-tile ph+1
  int s_st = P_LEAD;                // slot of half-tile ph+P_LEAD
  bool first_tile = true;
  // Start stagger (p.stagger_ticks > 0: GEMMs whose epilogue adds an fp32 residual).  With the store drains gone, that epilogue
  // runs AT the HBM roofline -- all 256 CUs reach it together and move 512 KB each (134 MB in 16 us = 8.4 TB/s), while the
  // K loops between two epilogues leave the memory idle.  Every second workgroup of the FIRST round (one per CU) therefore
  // starts half a tile late and stays half a tile behind for the whole launch: at any time half the CUs are in their K loop.
  // The engine's residual shapes have 5.5 rounds of tiles (1408 tiles), so the delayed CUs are the ones that would have idled
  // through the last half round anyway: the delay costs nothing there.
  if constexpr (!XPF) {
    // (groups of four XCD-local ids: a row tile's four column tiles -- or four rows of a patch column -- keep their common panel)
    if (p.stagger_ticks > 0 && blockIdx.x < (unsigned)p.stagger_first && ((blockIdx.x >> 5) & 1)) {
      const long long t0 = wall_clock64();
      while (wall_clock64() - t0 < p.stagger_ticks) __builtin_amdgcn_s_sleep(32);
    }
  }
  for (;;) {
  const int dbg_i = PERSIST ? lin : (int)blockIdx.x;
  if (p.dbg && tid == 0) {
    p.dbg[dbg_i * 6 + 0] = wall_clock64();
    p.dbg[dbg_i * 6 + 4] = __builtin_amdgcn_s_getreg(63492);      // HW_ID
    p.dbg[dbg_i * 6 + 5] = __builtin_amdgcn_s_getreg(63508);      // XCC_ID
  }
  int tm, tn;
  tile_of(lin, tm, tn);
  const int m0 = tm * B2M, n0 = tn * B2N;
  const bool has_next = PERSIST && lin + lin_step < lin_end;

  // ---- DMA sources: every wave stages rows [16 wave, +16) of each half-tile = two 1-KiB pieces; 32-bit byte offsets
  // from the tile's first row (rows past M / N are clamped, never stored), the K position goes into the scalar base
  auto a_elem = [&](int m) -> size_t {                // element index of row m's first K element
    if (CONV) {
      const int tf = p.cT2 * p.cF2;
      const int b = m / tf;
      const int rem = m - b * tf;
      const int t2 = rem / p.cF2, f2 = rem - t2 * p.cF2;
      return (((size_t)b * p.cT1 + 2 * t2) * p.cF1 + 2 * f2) * (size_t)p.cC;
    }
    return (size_t)m * p.lda;
  };
  const size_t a_row0 = a_elem(m0);
  const int lr = lane >> 3, lc = lane & 7;
  unsigned offA[2][2], offW[2][2];                    // [half][piece]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 16 * wave + 8 * i + lr;             // row of the half-tile
    const unsigned col = (unsigned)((lc ^ ((r >> 1) & 7)) * 16);      // swizzled 16-byte source column
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
      const int m = min(m0 + (r >> 6) * 128 + hlf * 64 + (r & 63), p.M - 1);
      const int n = min(n0 + (r >> 5) * 64 + hlf * 32 + (r & 31), p.N - 1);
      offA[hlf][i] = (unsigned)((a_elem(m) - a_row0) * sizeof(T)) + col;
      offW[hlf][i] = (unsigned)((size_t)(n - n0) * p.ldw * sizeof(T)) + col;
    }
  }
  const char* a_base = uniform_ptr((const char*)((const T*)p.A + a_row0));
  const char* w_base = uniform_ptr((const char*)((const T*)p.W + (size_t)n0 * p.ldw));
  const char* a_next = a_base;      // PERSIST: the scalar bases of this workgroup's next tile
  const char* w_next = w_base;
  if (XPF && has_next) {
    int tm2, tn2;
    tile_of(lin + lin_step, tm2, tn2);
    a_next = uniform_ptr((const char*)((const T*)p.A + (size_t)tm2 * B2M * p.lda));
    w_next = uniform_ptr((const char*)((const T*)p.W + (size_t)tn2 * B2N * p.ldw));
  }
  const int nk = p.K / BKE, nh = 4 * nk;
  // K serpentine (RVB_GEMM2_FLAGS bit 7): the 32 CUs of an XCD run 32 consecutive tiles of its run in lockstep -- one "wave" of
  // tiles = an 8 x 4 block of the tile grid whose 12 operand panels stream through the XCD's 4-MiB L2 once, front to back --
  // and the next wave shares 8 (A) or 4 (W) of those panels but starts again at K = 0, whose lines the L2 dropped first.
  // Odd waves walk K from the far end instead, so they begin with the lines the wave before them touched last.  The fp32
  // summation order of those tiles is reversed (bf16 engine only; the f32 parity engine runs gemm.hip).
  const bool k_rev = !CONV && !PERSIST && p.k_serp && ((blockIdx.x >> 8) & 1);
  const unsigned lds_wave = lds_base + wave * 2048;
  // byte offset of K step u in an A row, kept incrementally (no division in the loop): +128 per step; the implicit
  // convolution walks the 3x3 taps, whose (kh, kw .. kw+2) channels are contiguous in NHWC -- only a new kernel row kh
  // jumps, by (F1 - 3) pixels (the whole 128-byte K step lies inside one tap: cC % BKE == 0)
  const int row_steps = CONV ? 3 * p.cC / BKE : 0x7fffffff;
  const size_t row_jump = CONV ? (size_t)(p.cF1 - 3) * p.cC * sizeof(T) : 0;
  auto advance = [&](size_t& off, int& cnt) __attribute__((always_inline)) {
    off += ROW2;
    if (++cnt == row_steps) { cnt = 0; off += row_jump; }
  };
  size_t ak1 = 0, ak2 = 0;          // offsets of K steps t+1 and t+2 (t = the K step being multiplied)
  int ac1 = 0, ac2 = 0;
  advance(ak1, ac1);
  ak2 = ak1; ac2 = ac1;
  advance(ak2, ac2);
  // half-tile ty = {0: A-h0, 1: B-h0, 2: B-h1, 3: A-h1} of K step t (whose A offset is ak) into ring slot `slot`
  auto stage_ty = [&](auto tyc, int t, size_t ak, int slot) __attribute__((always_inline)) {
    constexpr int ty = decltype(tyc)::value;
    if constexpr (!CONV && !PERSIST) {
      if (k_rev) { t = nk - 1 - t; ak = (size_t)t * ROW2; }      // this tile walks K downwards (see k_rev)
    }
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_wave + slot * P_HT);
    if constexpr (ty == 0) dma2(offA[0][0], offA[0][1], uniform_ptr(a_base + ak), dst);
    else if constexpr (ty == 1) dma2(offW[0][0], offW[0][1], uniform_ptr(w_base + (size_t)t * ROW2), dst);
    else if constexpr (ty == 2) dma2(offW[1][0], offW[1][1], uniform_ptr(w_base + (size_t)t * ROW2), dst);
    else dma2(offA[1][0], offA[1][1], uniform_ptr(a_base + ak), dst);
  };
  // the same for K step t2 (0 or 1) of the NEXT tile (PERSIST; no convolution: K step t2 sits at byte t2 * 128 of a row)
  auto stage_next = [&](auto tyc, int t2, int slot) __attribute__((always_inline)) {
    constexpr int ty = decltype(tyc)::value;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_wave + slot * P_HT);
    if constexpr (ty == 0) dma2(offA[0][0], offA[0][1], uniform_ptr(a_next + (size_t)t2 * ROW2), dst);
    else if constexpr (ty == 1) dma2(offW[0][0], offW[0][1], uniform_ptr(w_next + (size_t)t2 * ROW2), dst);
    else if constexpr (ty == 2) dma2(offW[1][0], offW[1][1], uniform_ptr(w_next + (size_t)t2 * ROW2), dst);
    else dma2(offA[1][0], offA[1][1], uniform_ptr(a_next + (size_t)t2 * ROW2), dst);
  };
  auto stage = [&](int h, int slot) __attribute__((always_inline)) {       // prologue only: K steps 0 and 1
    const int t = h >> 2, ty = h & 3;
    const size_t ak = t == 0 ? 0 : ak1;
    if (ty == 0) stage_ty(std::integral_constant<int, 0>(), t, ak, slot);
    else if (ty == 1) stage_ty(std::integral_constant<int, 1>(), t, ak, slot);
    else if (ty == 2) stage_ty(std::integral_constant<int, 2>(), t, ak, slot);
    else stage_ty(std::integral_constant<int, 3>(), t, ak, slot);
  };
  static_assert(P_LEAD - 1 < 8, "the prologue requests half-tiles of K steps 0 and 1 only");
  static_assert(P_DEPTH == 3, "the tail's counted waits enumerate the counts below P_DEPTH");

  // 16x16 fragments: acc[i][j] = rows 16i.., cols 16j.. of the wave's 128x64 tile (C/D: col = lane&15, row = 4*(lane>>4)+r).
  // 32x32 blocks (fp8): acc32[bi][bj] = rows 32bi.., cols 32bj.. (C/D: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
  f32x4_t acc[B32 ? 1 : 8][B32 ? 1 : 4];
  f32x16_t acc32[B32 ? 4 : 1][B32 ? 2 : 1];
  const bool res_acc = !PH2 && !p.res_epilogue && p.res != nullptr && p.act == ACT_NONE && p.alpha != 0.f && (p.N & 3) == 0 && (p.ldres & 3) == 0 &&
                       ((size_t)p.res & 15) == 0;      // 16-byte residual vectors, whole inside N (PH2: its slabs share the ring's memory)
  // ---- prologue: half-tiles 0 .. P_LEAD-1 requested; then (optionally) the fp32 residual tile becomes the initial
  // accumulator: C = res + alpha * (A.W^T + bias) = alpha * (res / alpha + A.W^T + bias), see gemm2_kernel
  if constexpr (PMODE == 2) { s_cur = 0; s_rd = 1; s_st = P_LEAD; }      // every tile restarts the ring (all readers are past the K loop's last barrier)
  if constexpr (PH2) {
    for (int h = 0; h < 4; ++h) stage(h, h);      // X(0) and Y(0): slots 0..3
  } else if (first_tile || !XPF) {       // a later tile of a cross-prefetching workgroup found its first half-tiles requested by the tile before it
    for (int h = 0; h < P_LEAD && h < nh; ++h) stage(h, h);
  }
  if (res_acc) {
    const float inv_alpha = 1.0f / p.alpha;
    const int lr4 = lane >> 4, lc4 = (lane & 15) * 4;
    const int colr = min(n0 + wc * 64 + lc4, p.N - 4);
    char* tb = smem + P_NSLOT * P_HT + wave * (16 * P_SROW);
    float csc[2] = {inv_alpha, inv_alpha};          // fp8: the accumulator is multiplied by a_scale * w_scale[n] in the epilogue
    if constexpr (F8) {
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) {
        const int col = n0 + wc * 64 + bj * 32 + (lane & 31);
        const float sc = col < p.N ? p.a_scale * p.w_scale[col] : 1.0f;
        csc[bj] = sc != 0.f ? inv_alpha / sc : 0.f;
      }
    }
    // the wave's 128 x 64 residual sub-tile, 4 rows x 256 B per instruction; bf16: all 32 vectors in flight together (they
    // land in the accumulator registers themselves); fp8: two halves of 16 (the 16-register accumulator blocks cannot
    // alias the staging vectors, and 128 + 128 registers do not exist)
    constexpr int NPART = B32 ? 2 : 1, QP = 32 / NPART;
#pragma unroll
    for (int part = 0; part < NPART; ++part) {
      f32x4_t rv[QP];
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const int qq = part * QP + q;
        const int row = min(m0 + wr * 128 + (qq >> 2) * 16 + (qq & 3) * 4 + lr4, p.M - 1);
        rv[q] = *(const f32x4_t*)(p.res + (size_t)row * p.ldres + colr);
      }
#pragma unroll
      for (int ii = 0; ii < 8 / NPART; ++ii) {     // 16-row slab i: coalesced rows in, accumulator layout out
        const int i = part * (8 / NPART) + ii;
#pragma unroll
        for (int q = 0; q < 4; ++q) *(f32x4_t*)(tb + (q * 4 + lr4) * P_SROW + lc4 * 4) = rv[ii * 4 + q];
        __builtin_amdgcn_wave_barrier();
        if constexpr (B32) {
          const int bi = i >> 1, hh = i & 1;
          const int r32 = 4 * (lane >> 5), c32 = lane & 31;
#pragma unroll
          for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                acc32[bi][bj][8 * hh + 4 * qq + r] = *(const float*)(tb + (8 * qq + r32 + r) * P_SROW + (bj * 32 + c32) * 4) * csc[bj];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = *(const float*)(tb + (lr4 * 4 + r) * P_SROW + (j * 16 + (lane & 15)) * 4) * inv_alpha;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  } else {
    if constexpr (B32) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
  }
  if constexpr (PH2) {
    wait_vm<2>();                   // X(0)'s three half-tiles have landed (Y(0)'s two pieces may still be in flight)
  } else if (first_tile || !XPF) {       // half-tiles 0 and 1 have landed (later tiles: the previous tile's last two phases waited for them)
    const int infl = (nh < P_LEAD ? nh : P_LEAD) - 2;
    if (infl >= P_DEPTH) wait_vm<2 * P_DEPTH>(); else wait_vm<0>();
  }
  __builtin_amdgcn_s_barrier();
  if (p.dbg && tid == 0) p.dbg[dbg_i * 6 + 1] = wall_clock64();

  // fragment addressing (as gemm2_kernel).  bf16: lane = (row 0..15, 16-byte column group 0..3 of the 64-byte k32 slice), a
  // fragment set is [16-row fragment][k32 slice].  fp8: lane = (row 0..31, half g); the lane's 32 bytes of a k64 slice are
  // the 16-byte columns 2g, 2g+1; a set is [32-row block x k64 slice][column].
  // bf16 M32: lane = (row 0..31, half g); MFMA u = 0..3 of the K step takes the 16-byte column 2u + g of the 128-byte row;
  // a set is [32-row block x k-half][MFMA of the half] with the same four offsets (ro0, rh0 | ro1, rh1) as fp8's.
  const int frow = B32 ? (lane & 31) : (lane & 15), lgrp = B32 ? (lane >> 5) : (lane >> 4);
  const int swz = (frow >> 1) & 7;
  const int ro0 = F8 ? ((0 * 4 + 2 * lgrp) ^ swz) << 4 : ((0 * 4 + lgrp) ^ swz) << 4;
  const int ro1 = F8 ? ((1 * 4 + 2 * lgrp) ^ swz) << 4 : ((1 * 4 + lgrp) ^ swz) << 4;
  const int rh0 = F8 ? ((0 * 4 + 2 * lgrp + 1) ^ swz) << 4 : ((0 * 4 + 2 + lgrp) ^ swz) << 4;      // fp8: second column; M32: second MFMA
  const int rh1 = F8 ? ((1 * 4 + 2 * lgrp + 1) ^ swz) << 4 : ((1 * 4 + 2 + lgrp) ^ swz) << 4;
  const int a_off = (wr * 64 + frow) * ROW2;       // in an A half-tile: rows wr*64 + 16 i (32 bi) + frow
  const int b_off = (wc * 32 + frow) * ROW2;       // in a B half-tile: rows wc*32 + 16 j + frow
  // A set: k-half 0 / 1 = the first / second 64 bytes of the 128-byte row.  bf16: [16-row fragment i]; fp8: [32-row block
  // bi][16-byte column v] (a lane's 32 bytes of a k64 slice).  B sets: bf16 [fragment j][k32 slice], fp8 [k64 slice][v].
  uint4 fa_lo[4], fa_hi[4], fh[4], fb0[2][2], fb1[2][2];
  auto read_a_half = [&](auto halfc, uint4 (&f)[4], int slot) __attribute__((always_inline)) {
    constexpr int hf = decltype(halfc)::value;
    const char* sp = smem + slot * P_HT + a_off;
    if constexpr (B32) {
#pragma unroll
      for (int bi = 0; bi < 2; ++bi) {
        f[bi * 2 + 0] = *(const uint4*)(sp + bi * 4096 + (hf ? ro1 : ro0));
        f[bi * 2 + 1] = *(const uint4*)(sp + bi * 4096 + (hf ? rh1 : rh0));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) f[i] = *(const uint4*)(sp + i * 2048 + (hf ? ro1 : ro0));
    }
  };
  // bf16: the two k-halves of a B set are read separately, k-half 0 of both fragments first: the first eight MFMAs of a
  // phase (k-half 0) then wait for two reads, not for the whole set
  auto read_b_khalf = [&](auto sc, uint4 (&f)[2][2], int slot) __attribute__((always_inline)) {
    constexpr int sk = decltype(sc)::value;
    const char* sp = smem + slot * P_HT + b_off;
    if constexpr (B32) {      // [k-half][MFMA of the half]
      f[sk][0] = *(const uint4*)(sp + (sk ? ro1 : ro0)); f[sk][1] = *(const uint4*)(sp + (sk ? rh1 : rh0));
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) f[j][sk] = *(const uint4*)(sp + j * 2048 + (sk ? ro1 : ro0));
    }
  };
  auto read_b = [&](uint4 (&f)[2][2], int slot) __attribute__((always_inline)) {
    const char* sp = smem + slot * P_HT + b_off;
    if constexpr (F8) {
      f[0][0] = *(const uint4*)(sp + ro0); f[0][1] = *(const uint4*)(sp + rh0);
      f[1][0] = *(const uint4*)(sp + ro1); f[1][1] = *(const uint4*)(sp + rh1);
    } else {
      read_b_khalf(std::integral_constant<int, 0>(), f, slot);
      read_b_khalf(std::integral_constant<int, 1>(), f, slot);
    }
  };
  // quadrant (ai, bj) of the wave tile += A (k-halves alo, ahi) x B
  auto mma_q = [&](auto aic, auto bjc, const uint4 (&alo)[4], const uint4 (&ahi)[4], const uint4 (&fb)[2][2]) __attribute__((always_inline)) {
    constexpr int ai = decltype(aic)::value, bj = decltype(bjc)::value;
    if constexpr (F8) {
      struct Pair { uint4 lo, hi; };      // the two 16-byte vectors of a lane = one 8-register operand
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
          const i32x8_t a8 = __builtin_bit_cast(i32x8_t, (Pair{sl ? ahi[bi * 2] : alo[bi * 2], sl ? ahi[bi * 2 + 1] : alo[bi * 2 + 1]}));
          const i32x8_t b8 = __builtin_bit_cast(i32x8_t, (Pair{fb[sl][0], fb[sl][1]}));
          acc32[ai * 2 + bi][bj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc32[ai * 2 + bi][bj], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    } else if constexpr (M32) {
      // 8 MFMAs: k-half sl, MFMA uu of the half, the two 32-row blocks alternating (an accumulator is touched every second MFMA)
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int uu = 0; uu < 2; ++uu)
#pragma unroll
          for (int bi = 0; bi < 2; ++bi) {
            const bf16x8_t a8 = __builtin_bit_cast(bf16x8_t, sl ? ahi[bi * 2 + uu] : alo[bi * 2 + uu]);
            const bf16x8_t b8 = __builtin_bit_cast(bf16x8_t, fb[sl][uu]);
            acc32[ai * 2 + bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc32[ai * 2 + bi][bj], 0, 0, 0);
          }
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) Mma16<T>::run(s ? ahi[i] : alo[i], fb[j][s], acc[ai * 4 + i][bj * 2 + j]);
    }
  };
  std::integral_constant<int, 0> c0;
  std::integral_constant<int, 1> c1;
  auto mid = [&]() __attribute__((always_inline)) {          // read section -> MFMA section
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto end = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  std::integral_constant<int, 0> h0;
  std::integral_constant<int, 1> h1;
  if constexpr (PH2) {
    if (wr == 1) __builtin_amdgcn_s_barrier();     // the second wave row runs one barrier behind the first from here on
    std::integral_constant<int, 2> c2;
    std::integral_constant<int, 3> c3;
    // (one loop body with the slot group as a scalar: two parity copies of the body made hipcc rename the accumulators between them --
    // 180 spilled VGPRs, and every scratch reload is a vector-memory operation whose wait drains the DMA queue)
    for (int t = 0; t < nk; ++t) {
      const int base = (t & 1) * 4, nb = 4 - base;
      const bool nx = t + 1 < nk;
      // ---- phase X: requests for X(t + 1), all of this phase's fragments, then "Y(t)'s half-tile has landed"
      if (nx) { stage_ty(c0, t + 1, ak1, nb + 0); stage_ty(c1, t + 1, ak1, nb + 1); stage_ty(c2, t + 1, ak1, nb + 2); }
      read_b_khalf(h0, fb0, base + 1); read_a_half(h0, fa_lo, base + 0); read_a_half(h1, fa_hi, base + 0); read_b_khalf(h1, fb0, base + 1);
      read_b(fb1, base + 2);
      if (nx) wait_vm<6>(); else wait_vm<0>();
      mid(); mma_q(c0, c0, fa_lo, fa_hi, fb0); mma_q(c0, c1, fa_lo, fa_hi, fb1); end();
      // ---- phase Y: the request for Y(t + 1), A1, then "X(t + 1)'s half-tiles have landed"
      if (nx) stage_ty(c3, t + 1, ak1, nb + 3);
      read_a_half(h0, fa_lo, base + 3); read_a_half(h1, fa_hi, base + 3);
      if (nx) wait_vm<2>(); else wait_vm<0>();
      mid(); mma_q(c1, c1, fa_lo, fa_hi, fb1); mma_q(c1, c0, fa_lo, fa_hi, fb0); end();
      ak1 = ak2; ac1 = ac2;
      advance(ak2, ac2);
    }
  } else {
  read_a_half(h0, fh, s_cur);
  if (wr == 1) __builtin_amdgcn_s_barrier();       // the second wave row runs one barrier behind the first from here on
  int ph = 0;                       // phase counter (of this tile)
  auto adv = [&]() __attribute__((always_inline)) {
    ++ph;
    s_cur = s_rd;
    s_rd = s_rd + 1 == P_NSLOT ? 0 : s_rd + 1;
    s_st = s_st + 1 == P_NSLOT ? 0 : s_st + 1;
  };
  // one K step = four phases.  MODE 0: steady state.  MODE 1: the last K steps of a workgroup's last tile, where the ring
  // runs dry: requests and waits become conditional.  MODE 2 (PERSIST): the last K steps of a tile that has a successor --
  // requests past this tile's K go to the next tile, the counted wait stays the steady one.
  // R (tail form only) = K steps left including this one, a compile-time constant: with it every "is there anything left to
  // request / to wait for" decision of the tail folds away (round 3 chose them by chains of scalar branches on the phase
  // counter -- 3 of the 16 K steps of a K = 1024 tile ran that way).
  auto kstep = [&](auto modec, auto leftc, int t) __attribute__((always_inline)) {
    constexpr int MODE = decltype(modec)::value;
    constexpr int R = decltype(leftc)::value;
    constexpr bool TAIL = MODE == 1;
    // MODE 4 (PERSIST): the first K step of a tile that has a predecessor.  Its first three phases wait for half-tiles 2, 3, 4,
    // which the tile before requested BEFORE its epilogue: behind them in the in-order queue sit that epilogue's NST stores, so
    // the steady count (everything but the newest 2 * P_DEPTH requests) would also wait for the stores -- the stall that made
    // the persistent form lose in round 3 (bit 4 in the flag list).  The count is relaxed by exactly those stores; phase 3
    // waits for half-tile 5, requested behind them, with the steady count again.
    constexpr int NST = 16 / (64 / (64 / (sizeof(OutT) == 4 ? 4 : (sizeof(OutT) == 2 ? 8 : 16)))) * 8;
    auto ph_head = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      constexpr int dt = (j + P_LEAD) / 4;           // 1 or 2 K steps ahead
      static_assert(dt == 1 || dt == 2, "ring depth");
      constexpr int ty = (j + P_LEAD) & 3;
      if constexpr (MODE == 2) {          // K step nk-2 of a persistent tile: its last phase requests the next tile's first half-tile
        if constexpr (dt == 1) stage_ty(std::integral_constant<int, ty>(), t + 1, (size_t)(t + 1) * ROW2, s_st);
        else stage_next(std::integral_constant<int, ty>(), 0, s_st);
      } else if constexpr (MODE == 3) {   // K step nk-1: everything requested belongs to the next tile
        stage_next(std::integral_constant<int, ty>(), dt - 1, s_st);
      } else {
        // tail: half-tile ph + P_LEAD exists iff 4 t + j + P_LEAD < 4 nk, i.e. j + P_LEAD < 4 R
        if constexpr (!TAIL || j + P_LEAD < 4 * R) stage_ty(std::integral_constant<int, ty>(), t + dt, dt == 1 ? ak1 : ak2, s_st);
      }
    };
    // half-tile ph + 2 (read in the next phase) has landed: everything but the newest min(P_DEPTH, nh - 3 - ph) half-tiles
    auto ph_wait = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      if constexpr (TAIL) {
        constexpr int infl = 4 * R - 3 - j;          // = nh - 3 - ph
        if constexpr (infl >= P_DEPTH) wait_vm<2 * P_DEPTH>();
        else if constexpr (infl >= 0) wait_vm<2 * (infl < 0 ? 0 : infl)>();
      } else if constexpr (MODE == 4 && j < 3) {
        wait_vm<2 * P_DEPTH + NST>();
      } else {
        wait_vm<2 * P_DEPTH>();
      }
    };
    ph_head(std::integral_constant<int, 0>());
    if constexpr (F8) {
      read_a_half(h1, fa_hi, s_cur); read_b(fb0, s_rd);
    } else {      // in consumption order: B0's k-half 0 (with the prefetched A0 k-half 0), then the k-half-1 operands
      read_b_khalf(h0, fb0, s_rd); read_a_half(h1, fa_hi, s_cur); read_b_khalf(h1, fb0, s_rd);
    }
    ph_wait(std::integral_constant<int, 0>());
    mid(); mma_q(c0, c0, fh, fa_hi, fb0); end(); adv();
    ph_head(std::integral_constant<int, 1>()); read_b(fb1, s_rd); ph_wait(std::integral_constant<int, 1>());
    mid(); mma_q(c0, c1, fh, fa_hi, fb1); end(); adv();
    ph_head(std::integral_constant<int, 2>()); read_a_half(h0, fa_lo, s_rd); read_a_half(h1, fa_hi, s_rd); ph_wait(std::integral_constant<int, 2>());
    mid(); mma_q(c1, c1, fa_lo, fa_hi, fb1); end(); adv();
    ph_head(std::integral_constant<int, 3>()); if constexpr (MODE == 0 || MODE == 2 || MODE == 4 || (MODE == 1 && R > 1)) read_a_half(h0, fh, s_rd); ph_wait(std::integral_constant<int, 3>());
    mid(); mma_q(c1, c0, fa_lo, fa_hi, fb0); end(); adv();
    ak1 = ak2; ac1 = ac2;
    advance(ak2, ac2);
  };
  constexpr int NTAIL = (P_LEAD + 3) / 4 + 1;        // K steps whose phases may find nothing left to request / wait for
  int t = 0;
  if constexpr (XPF) {
    // one loop shape for every tile (nk >= 4, host-checked): the workgroup's last tile "prefetches" its own first half-tiles
    // again (a_next = a_base: 80 KiB of harmless reads, drained before the workgroup ends) instead of a third form of the loop
    if (!first_tile) { kstep(std::integral_constant<int, 4>(), c0, 0); t = 1; }
    for (; t < nk - 2; ++t) kstep(std::integral_constant<int, 0>(), c0, t);
    kstep(std::integral_constant<int, 2>(), c0, nk - 2);
    kstep(std::integral_constant<int, 3>(), c0, nk - 1);
  } else {
    static_assert(NTAIL == 3, "the tail is unrolled by hand: K steps with 3, 2 and 1 steps left");
    for (; t < nk - NTAIL; ++t) kstep(std::integral_constant<int, 0>(), c0, t);
    if (nk >= 3) kstep(c1, std::integral_constant<int, 3>(), nk - 3);
    if (nk >= 2) kstep(c1, std::integral_constant<int, 2>(), nk - 2);
    kstep(c1, c1, nk - 1);
  }
  }      // !PH2
  if (wr == 0) __builtin_amdgcn_s_barrier();       // balances the stagger: every wave has executed the same number of barriers
  if (p.dbg && tid == 0) p.dbg[dbg_i * 6 + 2] = wall_clock64();
  auto epilogue = [&]() __attribute__((always_inline)) {

  // ---- epilogue (as gemm2_kernel's 16x16 path; the slabs have their own LDS above the ring)
  OutT* __restrict__ C = (OutT*)p.C;
  const int crow = (lane >> 4) * 4;
  const int ccol = lane & 15;
  const bool full = (m0 + B2M <= p.M) && (n0 + B2N <= p.N);
  const bool vec_ok = ((p.ldc * (int)sizeof(OutT)) % 16 == 0) && (((size_t)p.C & 15) == 0) &&
                      (p.res == nullptr || ((p.ldres % 4) == 0 && ((size_t)p.res & 15) == 0));
  if (vec_ok) {
    char* slab = smem + (PH2 ? 0 : P_NSLOT * P_HT) + wave * (16 * P_SROW);      // PH2: slot memory, idle by now
    constexpr int CPL = sizeof(OutT) == 4 ? 4 : (sizeof(OutT) == 2 ? 8 : 16);   // columns per lane
    constexpr int LPR = 64 / CPL;                                                // lanes per slab row
    constexpr int RPP = 64 / LPR;                                                // rows per pass
    constexpr int NQ = 16 / RPP;                                                 // passes per slab
    const int orow = lane / LPR;
    const int ocol = (lane % LPR) * CPL;
    const int col0 = n0 + wc * 64 + ocol;
    float biasv[CPL], scv[F8 ? CPL : 1];
#pragma unroll
    for (int e = 0; e < CPL; ++e) biasv[e] = (p.bias && col0 + e < p.N) ? p.bias[col0 + e] : 0.0f;
    if constexpr (F8) {
#pragma unroll
      for (int e = 0; e < CPL; ++e) scv[e] = col0 + e < p.N ? p.a_scale * p.w_scale[col0 + e] : 0.0f;
    }
    const bool seg_full = col0 + CPL <= p.N;
    // Row-periodic bf16 addend (GemmArgs::rowadd; bf16 output only): `ra_any` = this tile touches its column range, `ra_vec` =
    // this lane's whole 8-column segment lies inside it and is one aligned 16-byte vector of the addend's row
    constexpr bool RA_OK = sizeof(OutT) == 2 && !F8;
    const bool glu = sizeof(OutT) == 2 && !F8 && p.act == ACT_GLU;      // gemm2() admits it on this kernel only (full column segments)
    const bool ra_any = RA_OK && p.rowadd != nullptr && n0 < p.rowadd_col0 + p.rowadd_cols && n0 + B2N > p.rowadd_col0;
    const bool ra_tile = ra_any && n0 >= p.rowadd_col0 && n0 + B2N <= p.rowadd_col0 + p.rowadd_cols && (p.rowadd_ld & 7) == 0 &&
                         ((p.rowadd_col0 - n0) & 7) == 0 && ((size_t)p.rowadd & 15) == 0;      // every lane's segment is such a vector
    const int ra_t0 = ra_any ? (m0 + wr * 128) % p.rowadd_rows : 0;       // addend row of the wave's first output row
    auto ra_row = [&](int local) __attribute__((always_inline)) -> int {      // addend row of output row m0 + wr*128 + local
      int t = ra_t0 + local;
      if (t >= p.rowadd_rows) { t -= p.rowadd_rows; if (t >= p.rowadd_rows) t %= p.rowadd_rows; }
      return t;
    };
    const bf16_t* ra_base = (const bf16_t*)p.rowadd + (col0 - p.rowadd_col0);
    // The residual of a tile whose residual is added HERE (p.res_epilogue, or whenever it cannot start out in the
    // accumulators): a ring of RING slabs' worth of 16-byte vectors per lane, requested that far ahead of the slab that
    // consumes them -- the K loop's fragment registers are dead by now, so 16 vectors (64 registers) fit next to the
    // accumulators, and a slab's residual rows are in flight while the slabs before it are transposed and stored.
    constexpr int RING = sizeof(OutT) == 4 ? 3 : 2, RV = NQ * (CPL / 4);       // slabs ahead, vectors per slab and lane
    float4 rring[RING][RV];
    const bool res_pre = p.res != nullptr && !res_acc && seg_full;
    const int colc = min(col0, p.N - CPL);
    auto res_issue = [&](int i, int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int row = min(m0 + wr * 128 + i * 16 + q * RPP + orow, p.M - 1);
        const float4* rp = (const float4*)(p.res + (size_t)row * p.ldres + colc);
#pragma unroll
        for (int c4 = 0; c4 < CPL / 4; ++c4) rring[slot][q * (CPL / 4) + c4] = rp[c4];
      }
    };
    auto finish_v = [&](auto actf) {
      if (res_pre) {
#pragma unroll
        for (int i = 0; i < RING; ++i) res_issue(i, i);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_wave_barrier();
        if constexpr (B32) {
          // rows [16i, 16i+16) of the wave tile = half hh = i&1 of block row bi = i>>1: registers 8hh .. 8hh+7
          const int bi = i >> 1, hh = i & 1;
          const int r32 = 4 * (lane >> 5), c32 = lane & 31;
#pragma unroll
          for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                *(float*)(slab + (8 * qq + r32 + r) * P_SROW + (bj * 32 + c32) * 4) = acc32[bi][bj][8 * hh + 4 * qq + r];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              *(float*)(slab + (crow + r) * P_SROW + (j * 16 + ccol) * 4) = acc[i][j][r];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int srow = q * RPP + orow;
          float v[CPL];
#pragma unroll
          for (int c4 = 0; c4 < CPL / 4; ++c4) {
            const float4 tt = *(const float4*)(slab + srow * P_SROW + (ocol + c4 * 4) * 4);
            v[c4 * 4 + 0] = tt.x; v[c4 * 4 + 1] = tt.y; v[c4 * 4 + 2] = tt.z; v[c4 * 4 + 3] = tt.w;
          }
          const int row = m0 + wr * 128 + i * 16 + srow;
          if (row >= p.M || col0 >= p.N) continue;
          if constexpr (F8) {
#pragma unroll
            for (int e = 0; e < CPL; ++e) v[e] *= scv[e];
          }
#pragma unroll
          for (int e = 0; e < CPL; ++e) v[e] = actf(v[e] + biasv[e]) * p.alpha;
          OutT* cp = C + (size_t)row * p.ldc + col0;
          if (seg_full) {
            if (res_pre) {
#pragma unroll
              for (int c4 = 0; c4 < CPL / 4; ++c4) {
                const float4 tt = rring[i % RING][q * (CPL / 4) + c4];
                v[c4 * 4 + 0] += tt.x; v[c4 * 4 + 1] += tt.y; v[c4 * 4 + 2] += tt.z; v[c4 * 4 + 3] += tt.w;
              }
            }
            if constexpr (RA_OK) {
              if (ra_any) {
                const bf16_t* ar = ra_base + (size_t)ra_row(i * 16 + srow) * p.rowadd_ld;
#pragma unroll
                for (int e = 0; e < CPL; ++e)
                  if (col0 + e >= p.rowadd_col0 && col0 + e < p.rowadd_col0 + p.rowadd_cols) v[e] += bf16_to_f32(ar[e]);
              }
            }
            if constexpr (sizeof(OutT) == 1) {
              const float qs = p.out_inv_scale;
              if (p.sat) {      // saturation counter of this fp8 tensor (rvb_get_fp8_saturation)
                // common path: one maximum over the lane's 16 magnitudes (v_max3 with |.| modifiers) and one compare; the exact
                // count and the atomic only when something clips (per-element compares here cost 1.5 ms per hour of audio)
                float mx = fmaxf(fabsf(v[0]), fabsf(v[1]));
#pragma unroll
                for (int e2 = 2; e2 < 16; ++e2) mx = fmaxf(mx, fabsf(v[e2]));
                if (mx * qs > 448.f) {
                  const unsigned ns = fp8_clipped(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs) + fp8_clipped(v[4] * qs, v[5] * qs, v[6] * qs, v[7] * qs) +
                                      fp8_clipped(v[8] * qs, v[9] * qs, v[10] * qs, v[11] * qs) + fp8_clipped(v[12] * qs, v[13] * qs, v[14] * qs, v[15] * qs);
                  atomicAdd(p.sat, ns);
                }
              }
              *(uint4*)cp = make_uint4(pack4_fp8(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs), pack4_fp8(v[4] * qs, v[5] * qs, v[6] * qs, v[7] * qs),
                                       pack4_fp8(v[8] * qs, v[9] * qs, v[10] * qs, v[11] * qs), pack4_fp8(v[12] * qs, v[13] * qs, v[14] * qs, v[15] * qs));
            } else if constexpr (sizeof(OutT) == 2) {
              if (glu) {        // (a, b) column pairs -> a * sigmoid(b), half as many columns
                uint2 o0;
                o0.x = pack2_bf16(glu_gate(v[0], v[1]), glu_gate(v[2], v[3]));
                o0.y = pack2_bf16(glu_gate(v[4], v[5]), glu_gate(v[6], v[7]));
                *(uint2*)(C + (size_t)row * p.ldc + (col0 >> 1)) = o0;
              } else {
                uint4 o0;
                o0.x = pack2_bf16(v[0], v[1]);
                o0.y = pack2_bf16(v[2], v[3]);
                o0.z = pack2_bf16(v[4], v[5]);
                o0.w = pack2_bf16(v[6], v[7]);
                *(uint4*)cp = o0;
              }
            } else {
              *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
            }
          } else {            // ragged last column segment of the matrix
            for (int e = 0; e < CPL && col0 + e < p.N; ++e) {
              float o = v[e];
              if (p.res && !res_acc) o += p.res[(size_t)row * p.ldres + col0 + e];
              if constexpr (RA_OK) {
                if (ra_any && col0 + e >= p.rowadd_col0 && col0 + e < p.rowadd_col0 + p.rowadd_cols)
                  o += bf16_to_f32(ra_base[(size_t)ra_row(i * 16 + srow) * p.rowadd_ld + e]);
              }
              if constexpr (sizeof(OutT) == 1) cp[e] = (OutT)(pack4_fp8(o * p.out_inv_scale, 0.f, 0.f, 0.f) & 0xffu);
              else cp[e] = Cvt<OutT>::from_f32(o);
            }
          }
        }
        if (res_pre && i + RING < 8) res_issue(i + RING, i % RING);
      }
    };
    // Full tiles (all but the last row / column of tiles).  The generic loop above is correct for them too, but slow for a
    // reason that has nothing to do with its arithmetic: vmcnt counts loads and stores in one in-order queue, so every wait
    // hipcc places for a residual vector -- and, because its bookkeeping merges the ragged branch's element loads into the
    // straight path, it places one in front of the slab reads of EVERY pass, residual or not -- also waits for all the
    // stores issued before it: each of the 16 (bf16) or 32 (fp32) stores of a wave's tile was drained to memory before the
    // next pass began (s_waitcnt vmcnt(0) behind every global_store in the round-3 code object; epilogue 4.9 us for a bf16
    // tile, 7.6 us with SiLU, 16-18 us with an fp32 residual -- scripts/gemm_timeline.py).  Here the stores are inline asm,
    // invisible to the waitcnt pass: without a residual the loop contains no vector-memory wait at all; with one the
    // compiler sees loads only and counts them in order (two slabs' worth stay in flight), which is conservative by
    // exactly the interleaved stores -- a slab's worth of stores drains underneath the next slab instead of in front of it.
    auto finish_fast = [&](auto actf, auto resc, auto addc) {
      constexpr bool HR = decltype(resc)::value;
      constexpr bool RA = decltype(addc)::value && RA_OK;       // the row-periodic addend, one 16-byte vector per lane and pass
      // requested RING slabs ahead like the residual: a load waited for right where it is issued would also wait for every store
      // issued before it (one in-order queue)
      uint4 aring[RA ? RING : 1][RA ? NQ : 1];
      auto add_issue = [&](int i, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          aring[RA ? slot : 0][RA ? q : 0] = *(const uint4*)(ra_base + (size_t)ra_row(i * 16 + q * RPP + orow) * p.rowadd_ld);
      };
      if constexpr (RA) {
#pragma unroll
        for (int i = 0; i < RING; ++i) add_issue(i, i);
      }
      auto res_issue_f = [&](int i, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const float4* rp = (const float4*)(p.res + (size_t)(m0 + wr * 128 + i * 16 + q * RPP + orow) * p.ldres + col0);
#pragma unroll
          for (int c4 = 0; c4 < CPL / 4; ++c4) rring[slot][q * (CPL / 4) + c4] = rp[c4];
        }
      };
      if constexpr (HR) {
#pragma unroll
        for (int i = 0; i < RING; ++i) res_issue_f(i, i);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_wave_barrier();
        if constexpr (B32) {
          const int bi = i >> 1, hh = i & 1;
          const int r32 = 4 * (lane >> 5), c32 = lane & 31;
#pragma unroll
          for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                *(float*)(slab + (8 * qq + r32 + r) * P_SROW + (bj * 32 + c32) * 4) = acc32[bi][bj][8 * hh + 4 * qq + r];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              *(float*)(slab + (crow + r) * P_SROW + (j * 16 + ccol) * 4) = acc[i][j][r];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int srow = q * RPP + orow;
          float v[CPL];
#pragma unroll
          for (int c4 = 0; c4 < CPL / 4; ++c4) {
            const float4 tt = *(const float4*)(slab + srow * P_SROW + (ocol + c4 * 4) * 4);
            v[c4 * 4 + 0] = tt.x; v[c4 * 4 + 1] = tt.y; v[c4 * 4 + 2] = tt.z; v[c4 * 4 + 3] = tt.w;
          }
          if constexpr (F8) {
#pragma unroll
            for (int e = 0; e < CPL; ++e) v[e] *= scv[e];
          }
#pragma unroll
          for (int e = 0; e < CPL; ++e) v[e] = actf(v[e] + biasv[e]) * p.alpha;
          if constexpr (HR) {
#pragma unroll
            for (int c4 = 0; c4 < CPL / 4; ++c4) {
              const float4 tt = rring[i % RING][q * (CPL / 4) + c4];
              v[c4 * 4 + 0] += tt.x; v[c4 * 4 + 1] += tt.y; v[c4 * 4 + 2] += tt.z; v[c4 * 4 + 3] += tt.w;
            }
          }
          if constexpr (RA) {
            const uint4 av = aring[i % RING][q];
            v[0] += __uint_as_float(av.x << 16); v[1] += __uint_as_float(av.x & 0xffff0000u);
            v[2] += __uint_as_float(av.y << 16); v[3] += __uint_as_float(av.y & 0xffff0000u);
            v[4] += __uint_as_float(av.z << 16); v[5] += __uint_as_float(av.z & 0xffff0000u);
            v[6] += __uint_as_float(av.w << 16); v[7] += __uint_as_float(av.w & 0xffff0000u);
          }
          OutT* cp = C + (size_t)(m0 + wr * 128 + i * 16 + srow) * p.ldc + col0;
          if constexpr (sizeof(OutT) == 1) {
            const float qs = p.out_inv_scale;
            if (p.sat) {
              float mx = fmaxf(fabsf(v[0]), fabsf(v[1]));
#pragma unroll
              for (int e2 = 2; e2 < 16; ++e2) mx = fmaxf(mx, fabsf(v[e2]));
              if (mx * qs > 448.f) {
                const unsigned ns = fp8_clipped(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs) + fp8_clipped(v[4] * qs, v[5] * qs, v[6] * qs, v[7] * qs) +
                                    fp8_clipped(v[8] * qs, v[9] * qs, v[10] * qs, v[11] * qs) + fp8_clipped(v[12] * qs, v[13] * qs, v[14] * qs, v[15] * qs);
                atomicAdd(p.sat, ns);
              }
            }
            store16_hidden(cp, pack4_fp8(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs), pack4_fp8(v[4] * qs, v[5] * qs, v[6] * qs, v[7] * qs),
                           pack4_fp8(v[8] * qs, v[9] * qs, v[10] * qs, v[11] * qs), pack4_fp8(v[12] * qs, v[13] * qs, v[14] * qs, v[15] * qs));
          } else if constexpr (sizeof(OutT) == 2) {
            if (glu) store8_hidden(C + (size_t)(m0 + wr * 128 + i * 16 + srow) * p.ldc + (col0 >> 1),
                                   pack2_bf16(glu_gate(v[0], v[1]), glu_gate(v[2], v[3])), pack2_bf16(glu_gate(v[4], v[5]), glu_gate(v[6], v[7])));
            else store16_hidden(cp, pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
          } else {
            store16_hidden(cp, __float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
          }
        }
        if constexpr (HR) {
          if (i + RING < 8) res_issue_f(i + RING, i % RING);
        }
        if constexpr (RA) {
          if (i + RING < 8) add_issue(i + RING, i % RING);
        }
      }
    };
    const bool hr = p.res != nullptr && !res_acc;
    // a tile that the addend's column range cuts through takes the generic loop (per-element range checks)
    const bool fast = full && (p.fast_epilogue & (hr ? 2 : 1)) && (!ra_any || (ra_tile && !hr));
    auto run = [&](auto actf) __attribute__((always_inline)) {
      if (fast) {
        if (hr) finish_fast(actf, std::true_type(), std::false_type());
        else if (ra_any) finish_fast(actf, std::false_type(), std::true_type());
        else finish_fast(actf, std::false_type(), std::false_type());
      } else {
        finish_v(actf);
      }
    };
    if (p.act == ACT_SILU) run([](float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x)); });
    else if (p.act == ACT_RELU) run([](float x) { return fmaxf(x, 0.0f); });
    else run([](float x) { return x; });
    if (p.dbg && tid == 0) { if (!has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); p.dbg[dbg_i * 6 + 3] = wall_clock64(); }
    return;
  }
  // unaligned output / residual rows: element-wise stores
  if constexpr (B32) {
    auto finish32 = [&](auto actf) {
#pragma unroll
      for (int bi = 0; bi < 4; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) {
          const int col = n0 + wc * 64 + bj * 32 + (lane & 31);
          const float bvv = (p.bias && col < p.N) ? p.bias[col] : 0.0f;
          float sc = 1.f;
          if constexpr (F8) sc = col < p.N ? p.a_scale * p.w_scale[col] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wr * 128 + bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row >= p.M || col >= p.N) continue;
            float v = actf(acc32[bi][bj][r] * sc + bvv) * p.alpha;
            if (p.res && !res_acc) v += p.res[(size_t)row * p.ldres + col];
            if constexpr (sizeof(OutT) == 2 && !F8) {
              if (p.rowadd && col >= p.rowadd_col0 && col < p.rowadd_col0 + p.rowadd_cols)
                v += bf16_to_f32(((const bf16_t*)p.rowadd)[(size_t)(row % p.rowadd_rows) * p.rowadd_ld + (col - p.rowadd_col0)]);
            }
            if constexpr (sizeof(OutT) == 1) C[(size_t)row * p.ldc + col] = (OutT)(pack4_fp8(v * p.out_inv_scale, 0.f, 0.f, 0.f) & 0xffu);
            else C[(size_t)row * p.ldc + col] = Cvt<OutT>::from_f32(v);
          }
        }
    };
    if (p.act == ACT_SILU) finish32([](float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * v)); });
    else if (p.act == ACT_RELU) finish32([](float v) { return fmaxf(v, 0.0f); });
    else finish32([](float v) { return v; });
    return;
  } else {
  float bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + wc * 64 + j * 16 + ccol;
    bv[j] = (p.bias && col < p.N) ? p.bias[col] : 0.0f;
  }
  auto finish = [&](auto actf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 128 + i * 16 + crow + r;
        if (!full && row >= p.M) continue;
        const float* rrow = (p.res && !res_acc) ? p.res + (size_t)row * p.ldres : nullptr;
        OutT* crow_p = C + (size_t)row * p.ldc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = n0 + wc * 64 + j * 16 + ccol;
          if (!full && col >= p.N) continue;
          float v = actf(acc[i][j][r] + bv[j]) * p.alpha;
          if (rrow) v += rrow[col];
          if constexpr (sizeof(OutT) == 2) {
            if (p.rowadd && col >= p.rowadd_col0 && col < p.rowadd_col0 + p.rowadd_cols)
              v += bf16_to_f32(((const bf16_t*)p.rowadd)[(size_t)(row % p.rowadd_rows) * p.rowadd_ld + (col - p.rowadd_col0)]);
          }
          crow_p[col] = Cvt<OutT>::from_f32(v);
        }
      }
    }
  };
  if (p.act == ACT_SILU) finish([](float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * v)); });
  else if (p.act == ACT_RELU) finish([](float v) { return fmaxf(v, 0.0f); });
  else finish([](float v) { return v; });
  }
  };     // epilogue
  epilogue();
  if (!has_next) {
    if (XPF) wait_vm<0>();        // the ring's last (unused) requests must not outlive the workgroup's LDS
    break;
  }
  lin += lin_step;
  first_tile = false;
  }      // tiles of this workgroup
}

template <typename T, typename OutT, bool CONV, bool M32 = false, bool PH2 = false>
static int launch2p(hipStream_t s, const GemmArgs& p) {
  static bool attr_set = false;
  static int ncu = 0;
  auto kern = gemm2p_kernel<T, OutT, CONV, 0, M32, PH2>;
  if (!attr_set) {
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM2P_LDS));
    if constexpr (!CONV && !M32 && !PH2 && std::is_same<T, bf16_t>::value) {
      RVB_HIP_CHECK(hipFuncSetAttribute((const void*)gemm2p_kernel<T, OutT, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM2P_LDS));
      RVB_HIP_CHECK(hipFuncSetAttribute((const void*)gemm2p_kernel<T, OutT, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM2P_LDS));
    }
    int dev = 0;
    hipDeviceProp_t pr;
    RVB_HIP_CHECK(hipGetDevice(&dev));
    RVB_HIP_CHECK(hipGetDeviceProperties(&pr, dev));
    ncu = pr.multiProcessorCount & ~7;
    attr_set = true;
  }
  const int tiles = cdiv(p.M, B2M) * cdiv(p.N, B2N);
  if constexpr (!CONV && !M32 && !PH2 && std::is_same<T, bf16_t>::value) {
    // persistent forms: full tiles only, and at least two tiles per CU; bit 4 = with the cross-tile prefetch, bit 12 = loop only
    if ((g_gemm2_flags & (16 | 4096)) && ncu >= 8 && p.M % B2M == 0 && p.N % B2N == 0 && tiles >= 2 * ncu && p.K >= 4 * (ROW2 / 2)) {
      if (g_gemm2_flags & 16) hipLaunchKernelGGL((gemm2p_kernel<T, OutT, false, 1>), dim3(ncu), dim3(512), GEMM2P_LDS, s, p);
      else hipLaunchKernelGGL((gemm2p_kernel<T, OutT, false, 2>), dim3(ncu), dim3(512), GEMM2P_LDS, s, p);
      RVB_HIP_CHECK(hipGetLastError());
      return OK;
    }
  }
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), GEMM2P_LDS, s, p);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

template <typename T, typename OutT, bool CONV, bool MMA32 = false>
static int launch2(hipStream_t s, const GemmArgs& p) {
  static bool attr_set = false;
  auto kern = gemm2_kernel<T, OutT, CONV, MMA32>;
  if (!attr_set) {
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM2_LDS));
    attr_set = true;
  }
  const int tiles = cdiv(p.M, B2M) * cdiv(p.N, B2N);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), GEMM2_LDS, s, p);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

bool gemm_glu_supported(int dtype, const GemmArgs& p) {
  return dtype == DT_BF16 && p.act == ACT_GLU && gemm2_applicable(dtype, p) && !p.in_fp8 && !p.out_fp8 && !p.out_f32 && !p.conv && p.res == nullptr &&
         p.rowadd == nullptr && (p.N % 16) == 0 && p.ldc >= p.N / 2 && (p.ldc % 8) == 0 && ((size_t)p.C & 15) == 0 && p.alpha == 1.f;
}

bool gemm2_applicable(int dtype, const GemmArgs& p) {
  // fp8 with the convolution gather (round 4: conv2 of the subsampling, K = 9 d): the phase-interleaved loop, bf16 output
  if (p.in_fp8 && p.conv) return dtype == DT_BF16 && p.K % 128 == 0 && p.cC % 128 == 0 && p.ldw % 16 == 0 && p.M >= 1 && p.N >= 64 &&
                                 p.act != ACT_LRELU && p.w_scale != nullptr && !p.out_f32 && !p.out_fp8 && p.res == nullptr;
  if (p.in_fp8) return dtype == DT_BF16 && p.K % 128 == 0 && p.lda % 16 == 0 && p.ldw % 16 == 0 && p.M >= 1 && p.N >= 64 &&
                       p.act != ACT_LRELU && p.w_scale != nullptr;
  const int bke = dtype == DT_BF16 ? 64 : 32;
  if (p.K % bke || p.lda % (bke / 8) || p.ldw % (bke / 8)) return false;
  if (p.conv && (p.cC % bke)) return false;
  if (p.M < 128 || p.N < 64 || p.act == ACT_LRELU) return false;      // tiny problems: the 128x128 kernel wastes less
  return true;
}

// tuning switches (tests / scripts/gemm_bench.py / environment; none changes results beyond fp32 summation order):
//   bit 0  32x32x16 MFMAs        bit 1  s_setprio 1 for waves 4-7 in the K loop        group_m: tile order (0/1 = row-major)
//   bit 2  bf16: the round-2 register-pipelined loop (one 64-KiB stage per K step) instead of the phase-interleaved one
//   bit 4  bf16: the persistent form of the phase-interleaved loop (cross-tile prefetch) where it applies.  Opt-in: exact
//          (bit-identical), removes the dispatch gap (2.1 -> 0.1 us per tile) and the first-stage wait, but the epilogue of a
//          tile then sits on the next tile's critical path -- its stores share the in-order vmcnt queue with the operand
//          stream, so the next tile's first counted wait also waits for them (epilogue 7.0 -> 10.0 us for ffn1, 7.5 -> 13.5 us
//          for out-proj; one tile per workgroup just ends the wave with its stores in flight): GEMM time 108.2 -> 113.1 ms
//          per hour (gpurun_out/s6, profiles/r03_gemm_timeline_persistent.txt)
//   bit 5  bf16: start the accumulators from the fp32 residual (round 2's prologue) instead of adding it in the epilogue from a
//          ring of prefetched vectors (3 slabs ahead for fp32 output).  The ring is the default since round 3: the K loop's
//          fragment registers are dead in the epilogue, so the prefetch fits where round 2's spilled; ffn2 941 -> 1040 TFLOP/s,
//          GEMM time 103.1 -> 101.5 ms per hour (gpurun_out/s10), and the sum is formed as the reference forms it, x + alpha * (.)
//   bit 3  fp8: the phase-interleaved loop (gemm2p_kernel<fp8_t>) instead of gemm2_kernel's plain loop.  Exact, but slower on
//          the engine's shapes (1 h r640: fp8 GEMMs 58.6 vs 49.7 ms, step 141.4 vs 132.4 ms, gpurun_out/s5): K = 1024 is only 8
//          fp8 K steps, so 3 of them run the tail form of the loop, and next to 16-register accumulator blocks the allocator
//          spills (4-8 scratch accesses per K step in two of the three output variants)
//   bit 6  (round 4, removed) half-tile start stagger of every second CU for the fp32 + residual shapes: measured +3.4 % GEMM time
//          (profiles/r04_candidates.txt) -- the epilogue is not burst-bound, the delayed CUs just finish half a tile later
//   bit 7  K serpentine on for every shape, bit 8 on for N <= 2048 (default: off -- measured neutral in the engine): odd waves
//          of tiles walk K downwards (see k_rev in gemm2p_kernel)
//   bit 9  (round 4, measured and removed) transposed accumulators -- the MFMA operands swapped, so that a lane owns four
//          consecutive columns of a row -- and the epilogue storing straight from registers (8-byte bf16 vectors, 32-byte runs
//          per row) instead of through the LDS transposition: exact, 234 VGPRs, but the epilogue of a K = 1024 tile goes
//          7.75 -> 10.4 us (ffn1), 5.2 -> 9.8 us (qkv), 4.9 -> 7.5 us (plain) and the GEMMs of the bench hour 105.4 -> 107.4 ms
//          (profiles/r04_call5_tr_epilogue_linkage_eager.txt): partial-line stores cost more than the LDS round trip saves
//   bit 10 (round 4) OFF switch of the full-tile epilogue whose stores the waitcnt pass does not see (see finish_fast in
//          gemm2p_kernel): with the bit set every tile runs the generic epilogue, as until round 4
//   bit 11 residual tiles take the fast epilogue only where K is short (<= 2048 bf16 / 4096 fp8 elements)
//   bit 12 the persistent LOOP without cross-tile prefetch (gemm2p_kernel PMODE 2): removes the dispatch gap only
//   bit 13 (round 6) the phase-interleaved loop on v_mfma_f32_32x32x16_bf16 (gemm2p_kernel M32): 8 MFMAs of 32 cycles per phase
//   bit 14 (round 6) two phases of 32 MFMAs per K step instead of four of 16 (gemm2p_kernel PH2): half the barriers; exact, 211 VGPRs, and
//          SLOWER (kernel benchmark -2.4 %, the GEMMs of the bench hour 100.2 -> 105.7 ms): compiled with -DRVB_GEMM2_PH2 only
#ifndef GEMM2_STAGGER_DEFAULT
#define GEMM2_STAGGER_DEFAULT 1
#endif
int g_gemm2_flags = -1, g_gemm2_group_m = -1;
static void gemm2_opts_from_env() {
  if (g_gemm2_flags < 0) { const char* e = lab_env("RVB_GEMM2_FLAGS"); g_gemm2_flags = e ? atoi(e) : GEMM2_DEFAULT_FLAGS; }
  if (g_gemm2_group_m == -1) { const char* e = lab_env("RVB_GEMM2_GROUP_M"); g_gemm2_group_m = e ? atoi(e) : GEMM2_DEFAULT_GROUP_M; }
}

int gemm2(hipStream_t s, int dtype, const GemmArgs& p0) {
  gemm2_opts_from_env();
  GemmArgs p = p0;
  if (p.rowadd && (dtype != DT_BF16 || p.in_fp8 || p.out_f32 || p.out_fp8 || (g_gemm2_flags & (1 | 4)))) {
    set_error("gemm2: the row-periodic addend exists on the phase-interleaved bf16 kernel with bf16 output only");
    return E_UNSUPPORTED;
  }
  if (p.act == ACT_GLU && (!gemm_glu_supported(dtype, p) || (g_gemm2_flags & (1 | 4 | 16 | 4096 | 8192 | 16384)))) {
    set_error("gemm2: ACT_GLU runs on the default phase-interleaved bf16 kernel only (bf16 output, no residual / addend / fp8, N % 16 == 0)");
    return E_UNSUPPORTED;
  }
  p.group_m = g_gemm2_group_m == GROUP_M_AUTO ? (p.K * (p.in_fp8 ? 1 : 2) >= 4096 ? 0 : 8) : g_gemm2_group_m;
  p.prio = (g_gemm2_flags >> 1) & 1;
  p.res_epilogue = ((g_gemm2_flags >> 5) & 1) ^ 1;
  // bit 0: tiles without a residual in the epilogue; bit 1: tiles with one.  Both on by default: in the engine the GEMMs of the
  // bench hour take 104.6-105.2 ms with the round-3 epilogue (flag bit 10), 101.2 ms with the fast form for tiles without a
  // residual and for residual tiles of short-K GEMMs only (flag bit 11), 99.7-100.5 ms with it everywhere
  // (profiles/r04_call16_fast_epilogue.txt; the isolated kernel benchmark is too noisy to rank the last two)
  p.fast_epilogue = (g_gemm2_flags & 1024) ? 0 : (1 | ((!(g_gemm2_flags & 2048) || p.K * (p.in_fp8 ? 1 : 2) <= 4096) ? 2 : 0));
  // K serpentine (round 4): in the kernel benchmark, where one launch is repeated and its operands sit in the Infinity Cache, it
  // is worth +4.6 % on ffn2, +4.3 % on pw1, +5.5 % on embed, -3.7 % on ffn1 (profiles/r04_call2_ab.txt); in the ENGINE, where a
  // GEMM's operands were written by the kernel before it, neither "all shapes" nor "N <= 2048 only" moves the GEMM time of the
  // bench hour (104.5 vs 104.6 ms; 102.7 vs 102.7 ms, profiles/r04_call3_linkage_serp.txt).  Off by default; bit 7 = every shape,
  // bit 8 = the shapes with N <= 2048.
  p.k_serp = (g_gemm2_flags & 128) ? 1 : ((g_gemm2_flags & 256) ? (p.N <= 2048 ? 1 : 0) : 0);
  // start stagger of the residual shapes (see gemm2p_kernel): half a tile, estimated from the K loop (1.4 us per 128-byte K
  // step) + 18 us of prologue / epilogue; only where the last round of tiles leaves at least half the CUs idle anyway.
  // RVB_GEMM2_STAGGER: 0 = off, 1 = on (default: GEMMs of the bench hour 100.2 -> 99.0-99.4 ms, out / pw2 launch 206.9 -> 190.9 us,
  // ffn2 595.6 -> 554.9 us, profiles/r04_call18_stagger.txt), n > 1 = that many microseconds instead of the estimate
  {
    static int mode = -1, ncu = 0;
    if (mode < 0) {
      const char* e = lab_env("RVB_GEMM2_STAGGER"); mode = e ? atoi(e) : GEMM2_STAGGER_DEFAULT;
      int dev = 0; hipDeviceProp_t pr;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
    }
    p.stagger_ticks = 0; p.stagger_first = 0;
    if (mode > 0 && ncu > 0 && p.res != nullptr && p.out_f32 && !p.in_fp8 && dtype == DT_BF16 && !p.conv && (p.fast_epilogue & 2)) {
      const long long tiles = (long long)cdiv(p.M, B2M) * cdiv(p.N, B2N);
      const long long last = tiles % ncu;
      if (tiles > ncu && last > 0 && last <= ncu / 2) {
        const double us = mode > 1 ? (double)mode : 0.5 * (1.4 * (p.K * 2 / ROW2) + 18.0);
        p.stagger_ticks = (int)(us * 100.0);          // wall_clock64 ticks of 10 ns
        p.stagger_first = ncu;
      }
    }
  }
  if (p.in_fp8 && p.conv) return launch2p<fp8_t, bf16_t, true>(s, p);      // K = 9 d: 72 fp8 K steps, where the phase loop pays
  if (p.in_fp8 && (g_gemm2_flags & 8)) {      // fp8 on the phase-interleaved loop: opt-in, measured slower (see the flag list)
    if (p.out_fp8) return launch2p<fp8_t, fp8_t, false>(s, p);
    if (p.out_f32) return launch2p<fp8_t, float, false>(s, p);
    return launch2p<fp8_t, bf16_t, false>(s, p);
  }
  if (p.in_fp8) {
    if (p.out_fp8) return launch2<fp8_t, fp8_t, false, true>(s, p);
    if (p.out_f32) return launch2<fp8_t, float, false, true>(s, p);
    return launch2<fp8_t, bf16_t, false, true>(s, p);
  }
  if (dtype == DT_BF16) {
    if (g_gemm2_flags & 1) {
      if (p.out_f32) return p.conv ? launch2<bf16_t, float, true, true>(s, p) : launch2<bf16_t, float, false, true>(s, p);
      return p.conv ? launch2<bf16_t, bf16_t, true, true>(s, p) : launch2<bf16_t, bf16_t, false, true>(s, p);
    }
#ifdef RVB_GEMM2_PH2      // measured slower (profiles/r06_call7_gemm_two_phase_not_kept.txt): compiled only on request -- four instantiations, 3 min
    if (!(g_gemm2_flags & 4) && (g_gemm2_flags & 16384) && p.res_epilogue) {     // two phases per K step (round 6, gemm2p_kernel PH2)
      if (p.out_f32) return p.conv ? launch2p<bf16_t, float, true, false, true>(s, p) : launch2p<bf16_t, float, false, false, true>(s, p);
      return p.conv ? launch2p<bf16_t, bf16_t, true, false, true>(s, p) : launch2p<bf16_t, bf16_t, false, false, true>(s, p);
    }
#endif
    if (!(g_gemm2_flags & 4) && (g_gemm2_flags & 8192) && !p.conv) {     // the phase-interleaved loop on 32x32x16 MFMAs (round 6: measured
      // slower, kept as the lab's comparison partner for the plain shapes only -- every instantiation costs 40 s of build time)
      return p.out_f32 ? launch2p<bf16_t, float, false, true>(s, p) : launch2p<bf16_t, bf16_t, false, true>(s, p);
    }
    if (!(g_gemm2_flags & 4)) {     // the phase-interleaved loop (default)
      if (p.out_f32) return p.conv ? launch2p<bf16_t, float, true>(s, p) : launch2p<bf16_t, float, false>(s, p);
      return p.conv ? launch2p<bf16_t, bf16_t, true>(s, p) : launch2p<bf16_t, bf16_t, false>(s, p);
    }
    if (p.out_f32) return p.conv ? launch2<bf16_t, float, true>(s, p) : launch2<bf16_t, float, false>(s, p);
    return p.conv ? launch2<bf16_t, bf16_t, true>(s, p) : launch2<bf16_t, bf16_t, false>(s, p);
  }
  return p.conv ? launch2<float, float, true>(s, p) : launch2<float, float, false>(s, p);
}

}  // namespace rvb
