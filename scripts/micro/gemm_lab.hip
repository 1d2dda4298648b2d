// GEMM main-loop laboratory (gfx950, bf16): C[M][N] (fp32) = A[M][K] * W[N][K]^T with the 256x256 tile and the
// LDS-DMA fill of reverb_amd/csrc/gemm2.hip, parametrised over the things DESIGN.md section 8 item 1 names as
// the next steps, so that one run on an MI355X measures them side by side:
//
//   NWM x NWN   waves per workgroup and their layout: 2x4 = 8 waves of 128x64 (gemm2 today; 192 KiB of fragment
//               reads per 64-element K step) or 2x2 = 4 waves of 128x128 (128 KiB: LDS reads + DMA writes drop
//               from 100 % to 75 % of the MFMA time at peak rate; accumulators take 256 registers, one wave/SIMD)
//   KSUBS       K step of 64 (128-byte rows, as gemm2) or 32 elements (64-byte rows: half-size stages)
//   NST         LDS stages: 2 x 64 KiB, or 4 x 32 KiB = three K steps of fill in flight
//   PIPE        0: wait - barrier - refill - read - MFMA (gemm2);  1: fragments double-buffered in registers, the
//               barrier and the refill sit between two MFMA blocks, reads of the next k32 slice land under MFMAs
//
// Inputs are small integers stored as bf16, so every variant must reproduce the naive kernel EXACTLY.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/micro/gemm_lab scripts/micro/gemm_lab.hip && scripts/micro/gemm_lab
//   scripts/micro/gemm_lab 5 short   main-loop variants without the 4-wave / 64-byte-row ones
//   scripts/micro/gemm_lab 5 real    pipelined loop + the engine's epilogue, one tile per workgroup vs persistent
//                                    (profiles/r01_gemm_lab_real.txt: no gain)
//   scripts/micro/gemm_lab 5 conv    3x3 convolution of the ResNet34 128/256-channel stages as an implicit GEMM on the same loop
//                                    (profiles/r01_gemm_lab_conv.txt: 855 / 1193 TFLOP/s vs 555-598 for the direct kernel)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ inline void mma(const uint4& a, const uint4& b, f32x4& c) {
  union U { uint4 u; bf16x8 v; };
  U ua, ub;
  ua.u = a; ub.u = b;
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, c, 0, 0, 0);
}

// The same instruction with the accumulator pinned to AGPRs.  A 128x128 wave tile owns 64 accumulator tuples = all 256
// AGPRs; left to itself the register allocator parks some of them in VGPRs and copies them in and out around every
// MFMA.  Opaque to the scheduler and to the hazard recognizer: the caller orders instructions by hand, never touches
// one accumulator twice within 16 MFMAs, and idles before reading the accumulators with VALU instructions.
__device__ inline void mma_a(const uint4& a, const uint4& b, f32x4& c) {
  union U { uint4 u; bf16x8 v; };
  U ua, ub;
  ua.u = a; ub.u = b;
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(ua.v), "v"(ub.v));
}

// one 1-KiB LDS-DMA piece: 64 lanes x 16 bytes land at lds .. lds+1024 in lane order.  Inline asm on purpose: the
// compiler's waitcnt pass must not see these loads (it would drain vmcnt(0) before every ds_read), the explicit
// s_waitcnt below orders them.  M0 belongs to the compiler: saved and restored.
__device__ inline void dma1(const void* g, unsigned lds) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g), "s"(lds)
      : "memory");
}

template <int N> __device__ inline void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// wait until at most `groups` of the most recent DMA groups (PPW pieces each) are still in flight
template <int PPW> __device__ inline void wait_groups(int groups) {
  if (groups <= 0) wait_vm<0>();
  else if (groups == 1) wait_vm<PPW>();
  else if (groups == 2) wait_vm<2 * PPW>();
  else wait_vm<3 * PPW>();
}

template <int NWM, int NWN, int KSUBS, int NST, int PIPE, int AMMA, int STORE, int BN = 256, int OCC = 1>
__global__ __launch_bounds__(NWM* NWN * 64, OCC) void gemm_lab(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                           float* __restrict__ C, int M, int N, int K, int phases) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = NWM * NWN;
  constexpr int BKB = 64 * KSUBS;        // bytes of K per tile row and stage
  constexpr int BKE = BKB / 2;           // bf16 elements
  constexpr int CPR = BKB / 16;          // 16-byte columns per row: 8 or 4
  constexpr int RPI = 64 / CPR;          // rows one DMA instruction covers: 8 or 16
  constexpr int STAGE = (256 + BN) * BKB;       // 256 rows of A then BN rows of W
  constexpr int PPW = (256 + BN) / RPI / NW;    // DMA pieces per wave per stage
  static_assert(PPW * RPI * NW == 256 + BN, "pieces must divide among the waves");
  constexpr int TM = 256 / NWM, TN = BN / NWN, FI = TM / 16, FJ = TN / 16;
  static_assert(3 * PPW <= 63, "vmcnt is a 6-bit counter");
  static_assert(NST >= 2 && NST <= 4, "stages");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / NWN, wc = wave % NWN;

  // phases > 1: the workgroups of the first wave (one per CU) start `phase/phases` of a tile time apart.  Every CU runs
  // equal tiles back to back, so without this all 256 CUs reach their epilogue at the same moment, the stores of a whole
  // wave of tiles hit HBM as one burst, and nothing computes while it drains.
  if (phases > 1 && blockIdx.x < 256) {
    const int phase = (blockIdx.x >> 3) % phases;                 // bits above the XCD index: every XCD gets every phase
    const int kilocycles = phase * (K / 64) * 1260 / phases / 1024;
    for (int i = 0; i < kilocycles; ++i) __builtin_amdgcn_s_sleep(16);      // 16 x 64 clocks
  }
  const int tiles_n = N / BN;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {   // XCD-aware bijective tile order (as gemm2): consecutive tiles of one XCD share the A panel
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * 256, n0 = tn * BN;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  // swizzle: LDS[row][c] = G[row][c ^ swz(row)] (16-byte columns): the 16 lanes of a fragment read (16 consecutive
  // rows, one logical column) then touch 16 distinct 16-byte slots of a 256-byte bank row
  auto swz = [](int row) { return CPR == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

  // ---- DMA sources: the stage image is 256 rows of A then BN rows of W; wave w stages its PPW consecutive pieces of RPI rows
  const int lr = lane / CPR, lc = lane % CPR;
  const uint16_t* src[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int irow = (wave * PPW + i) * RPI + lr;          // row of the image
    const bool isA = irow < 256;
    const int row = isA ? irow : irow - 256;
    const uint16_t* base = isA ? A + (size_t)(m0 + row) * K : W + (size_t)(n0 + row) * K;
    src[i] = base + (lc ^ swz(row)) * 8;
  }
  const unsigned lds_wave = lds_base + wave * PPW * RPI * BKB;
  auto issue = [&](int kt) __attribute__((always_inline)) {
    const unsigned dst = lds_wave + (kt % NST) * STAGE;
#pragma unroll
    for (int i = 0; i < PPW; ++i) dma1(src[i] + (size_t)kt * BKE, dst + i * 1024);
  };

  f32x4 acc[FI][FJ];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, lgrp = lane >> 4;
  int roff[KSUBS];
#pragma unroll
  for (int s = 0; s < KSUBS; ++s) roff[s] = ((s * 4 + lgrp) ^ swz(frow)) << 4;
  const int a_off = (wr * TM + frow) * BKB;
  const int b_off = 256 * BKB + (wc * TN + frow) * BKB;
  const int nk = K / BKE;

  // Both loop shapes END with the wait-barrier-refill step, never with MFMAs: the register copies the allocator places
  // on loop edges then never read an accumulator that an (opaque, inline-asm) MFMA is still writing.  Straight-line
  // MFMA blocks after the loops carry trailing s_nops for the same reason.
  uint4 fa[2][FI], fb[2][FJ];
  auto st_of = [&](int q) __attribute__((always_inline)) { return smem + ((q / KSUBS) % NST) * STAGE; };
  auto read_slice = [&](auto bufc, int q) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
    const char* st = st_of(q);
    const int s = q % KSUBS;
#pragma unroll
    for (int i = 0; i < FI; ++i) fa[buf][i] = *(const uint4*)(st + a_off + i * 16 * BKB + roff[s]);
#pragma unroll
    for (int j = 0; j < FJ; ++j) fb[buf][j] = *(const uint4*)(st + b_off + j * 16 * BKB + roff[s]);
  };
  auto settle = [&]() __attribute__((always_inline)) { if (AMMA) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15"); };   // in-flight MFMAs retire
  auto mma_slice = [&](auto bufc) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) {
        if (AMMA) mma_a(fa[buf][i], fb[buf][j], acc[i][j]);
        else mma(fa[buf][i], fb[buf][j], acc[i][j]);
      }
  };
  // multiply slice `mbuf` while slice q is read into the other buffer: two MFMAs, one fragment read, ..., then the
  // remaining MFMAs.  AMMA: written in exactly that order (asm volatile MFMAs and loads keep their program order);
  // otherwise sched_group_barriers ask the scheduler for it (left alone it sinks the reads to the end of the block to
  // shorten live ranges, and the next block waits for them).
  auto block = [&](auto mbufc, int q) __attribute__((always_inline)) {
    constexpr int mb = decltype(mbufc)::value, rb = 1 - mb;
    if (AMMA) {
      const char* st = st_of(q);
      const int s = q % KSUBS;
#pragma unroll
      for (int m = 0; m < FI * FJ; ++m) {
        mma_a(fa[mb][m / FJ], fb[mb][m % FJ], acc[m / FJ][m % FJ]);
        if ((m & 1) && (m >> 1) < FI + FJ) {
          const int r = m >> 1;
          if (r < FJ) fb[rb][r] = *(const uint4*)(st + b_off + r * 16 * BKB + roff[s]);
          else fa[rb][r - FJ] = *(const uint4*)(st + a_off + (r - FJ) * 16 * BKB + roff[s]);
        }
      }
    } else {
      read_slice(std::integral_constant<int, rb>(), q);
      mma_slice(mbufc);
#pragma unroll
      for (int r = 0; r < FI + FJ; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);     // MFMA (first: they wait for the previous block's reads only)
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, FI * FJ - 2 * (FI + FJ), 0);
    }
  };
  std::integral_constant<int, 0> b0;
  std::integral_constant<int, 1> b1;

  if (PIPE == 0) {
    // stage kt: wait - barrier - refill the buffer of stage kt-1 - read all fragments - MFMAs          (gemm2's loop)
    auto enter = [&](int kt) __attribute__((always_inline)) {
      const int issued = kt + NST - 1 < nk ? kt + NST - 1 : nk;
      wait_groups<PPW>(issued - kt - 1);           // this wave's pieces of stage kt have landed
      __syncthreads();                             // ... and everybody else's; stage kt-1 is no longer read by anyone
      if (kt + NST - 1 < nk) issue(kt + NST - 1);
    };
    auto compute = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < KSUBS; ++s) {
        read_slice(b0, kt * KSUBS + s);
        mma_slice(b0);
      }
    };
    for (int s = 0; s < NST - 1 && s < nk; ++s) issue(s);
    enter(0);
    for (int kt = 0; kt + 1 < nk; ++kt) {
      compute(kt);
      enter(kt + 1);
    }
    compute(nk - 1);
    settle();
  } else {
    // k32 slices q = kt*KSUBS + s stream through the two register buffers; block u multiplies slice u and reads slice
    // u+1.  A stage is "entered" (its DMA awaited, barrier, the buffer of the stage before it refilled with stage
    // kt-1+NST) before its first slice is read.  Host guarantees nk >= 2 and an even number of slices.
    auto enter_stage = [&](int kt) __attribute__((always_inline)) {               // kt >= 1
      const int issued = kt - 1 + NST < nk ? kt - 1 + NST : nk;
      wait_groups<PPW>(issued - kt - 1);
      __syncthreads();
      if (kt - 1 + NST < nk) issue(kt - 1 + NST);
    };
    const int nq = nk * KSUBS;
    for (int s = 0; s < NST && s < nk; ++s) issue(s);
    {
      const int issued = NST < nk ? NST : nk;
      wait_groups<PPW>(issued - 1);
      __syncthreads();
    }
    read_slice(b0, 0);
    if (KSUBS == 2) {
      block(b0, 1);
      enter_stage(1);
      int u = 1;
      for (; u <= nq - 5; u += 2) {                // slices u (sub 1 of a stage), u+1 (sub 0 of the next), then enter the one after
        block(b1, u + 1);
        block(b0, u + 2);
        enter_stage((u + 3) / 2);
      }
      block(b1, nq - 2); settle();
      block(b0, nq - 1); settle();
      mma_slice(b1); settle();
    } else {
      enter_stage(1);
      int u = 0;
      for (; u <= nq - 4; u += 2) {
        block(b0, u + 1);
        enter_stage(u + 2);
        block(b1, u + 2);
        enter_stage(u + 3);
      }
      block(b0, nq - 1); settle();
      mma_slice(b1); settle();
    }
  }

  // ---- plain epilogue (the lab measures main loops; gemm2's LDS-transposed epilogue is not repeated here)
  const int crow = (lane >> 4) * 4, ccol = lane & 15;
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* crow_p = C + (size_t)(m0 + wr * TM + i * 16 + crow + r) * N + n0 + wc * TN + ccol;
      if (STORE) {
#pragma unroll
        for (int j = 0; j < FJ; ++j) crow_p[j * 16] = acc[i][j][r];
      } else {                                                   // main loop only: keep the accumulators alive, store nothing
#pragma unroll
        for (int j = 0; j < FJ; ++j)
          if (acc[i][j][r] == 12345.678f) crow_p[j * 16] = acc[i][j][r];
      }
    }
}

// ---- phase-interleaved main loop ("8-phase" schedule of cdna_hip_programming.md section 5, re-derived for this tile):
// the K step of 64 elements is cut into four PHASES, one per 64x32 quadrant of the wave's 128x64 tile (16 MFMAs), and the
// operands into four HALF-TILES of 16 KiB -- A-h0 / A-h1 = the first / second 64 rows of every wave row's 128, B-h0 / B-h1 =
// the first / second 32 columns of every wave column's 64 -- which is exactly what one phase newly needs:
//
//     phase j of K step t     MFMAs        fragment reads issued in this phase (into the register set that just died)
//       0                     A0 x B0      B0 <- B-h0(t)      4 ds_read_b128
//       1                     A0 x B1      B1 <- B-h1(t)      4
//       2                     A1 x B1      A1 <- A-h1(t)      8
//       3                     A1 x B0      A0 <- A-h0(t+1)    8            (B0 stays in registers from phase 0)
//
// so the operand stream is ONE half-tile per phase, in consumption order, through a ring of NSLOT 16-KiB LDS slots:
// phase p reads half-tile p+1, waits (counted vmcnt, never 0 in steady state) until half-tile p+2 has landed, and requests
// half-tile p+NSLOT-1 into the slot whose last reader finished two phases ago -- NSLOT-3 half-tiles (80 KiB of 128) stay in
// flight ACROSS the barriers instead of one 64-KiB stage that is drained to zero every K step.  A phase is
// {reads + DMA issue + vmcnt | barrier | lgkmcnt(0), 16 MFMAs | barrier}; with STAGGER the second wave row runs one barrier
// behind the first, so on every SIMD one wave multiplies while the other reads and issues (s_setprio around the MFMAs, PRIO).
template <int HT> __device__ inline void dma2(unsigned off0, unsigned off1, const void* sbase, unsigned lds0) {
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

template <int NSLOT, int STAGGER, int PRIO, int STORE>
__global__ __launch_bounds__(512) void gemm_ph(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                               float* __restrict__ C, int M, int N, int K, int /*unused*/) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HT = 128 * 128;                    // bytes of a half-tile: 128 rows x 128 bytes of K
  constexpr int DEPTH = NSLOT - 3;                 // half-tiles in flight behind the one being waited for
  static_assert(2 * DEPTH <= 63, "vmcnt is a 6-bit counter");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = N / 256;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  // ---- DMA sources: every wave stages rows [16 wave, +16) of each half-tile = two 1-KiB pieces (8 rows x 128 B);
  // 32-bit byte offsets from the tile's first row, the K position is added to the scalar base
  const int lr = lane >> 3, lc = lane & 7;
  unsigned offA[2][2], offW[2][2];                 // [half][piece]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 16 * wave + 8 * i + lr;          // row of the half-tile
    const unsigned col = (unsigned)((lc ^ ((r >> 1) & 7)) * 16);
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
      offA[hlf][i] = (unsigned)((r >> 6) * 128 + hlf * 64 + (r & 63)) * (unsigned)(K * 2) + col;
      offW[hlf][i] = (unsigned)((r >> 5) * 64 + hlf * 32 + (r & 31)) * (unsigned)(K * 2) + col;
    }
  }
  const char* a_base = (const char*)(A + (size_t)m0 * K);
  const char* w_base = (const char*)(W + (size_t)n0 * K);
  const int nk = K / 64, nh = 4 * nk;
  const unsigned lds_wave = lds_base + wave * 2048;
  // half-tile h = 4t + ty, ty = {0: A-h0, 1: B-h0, 2: B-h1, 3: A-h1} of K step t, into ring slot `slot`
  auto stage_ty = [&](auto tyc, int t, int slot) __attribute__((always_inline)) {
    constexpr int ty = decltype(tyc)::value;
    const unsigned dst = lds_wave + slot * HT;
    if constexpr (ty == 0) dma2<HT>(offA[0][0], offA[0][1], a_base + (size_t)t * 128, dst);
    else if constexpr (ty == 1) dma2<HT>(offW[0][0], offW[0][1], w_base + (size_t)t * 128, dst);
    else if constexpr (ty == 2) dma2<HT>(offW[1][0], offW[1][1], w_base + (size_t)t * 128, dst);
    else dma2<HT>(offA[1][0], offA[1][1], a_base + (size_t)t * 128, dst);
  };
  auto stage = [&](int h, int slot) __attribute__((always_inline)) {       // prologue only
    const int t = h >> 2, ty = h & 3;
    if (ty == 0) stage_ty(std::integral_constant<int, 0>(), t, slot);
    else if (ty == 1) stage_ty(std::integral_constant<int, 1>(), t, slot);
    else if (ty == 2) stage_ty(std::integral_constant<int, 2>(), t, slot);
    else stage_ty(std::integral_constant<int, 3>(), t, slot);
  };
  // wait until half-tile p+2 (read in the next phase) has landed: everything but the newest min(DEPTH, nh-3-p) half-tiles
  auto wait_tail = [&](int p) __attribute__((always_inline)) {
    const int infl = nh - 3 - p;
    if (infl >= DEPTH) wait_vm<2 * DEPTH>();
    else if (infl < 0) {}
    else if (infl == 0) wait_vm<0>();
    else if (infl == 1) wait_vm<2>();
    else if (infl == 2) wait_vm<4>();
    else if (infl == 3) wait_vm<6>();
    else if (infl == 4) wait_vm<8>();
    else if (infl == 5) wait_vm<10>();
    else wait_vm<12>();
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, lgrp = lane >> 4;
  const int swz = (frow >> 1) & 7;
  const int ro0 = ((0 * 4 + lgrp) ^ swz) << 4, ro1 = ((1 * 4 + lgrp) ^ swz) << 4;
  const int a_off = (wr * 64 + frow) * 128;        // in an A half-tile: rows wr*64 + 16 i + frow
  const int b_off = (wc * 32 + frow) * 128;        // in a B half-tile: rows wc*32 + 16 j + frow
  uint4 fa0[4][2], fa1[4][2], fb0[2][2], fb1[2][2];          // [fragment][k32 slice]
  auto read_a = [&](uint4 (&f)[4][2], int slot) __attribute__((always_inline)) {
    const char* sp = smem + slot * HT + a_off;
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i][0] = *(const uint4*)(sp + i * 2048 + ro0); f[i][1] = *(const uint4*)(sp + i * 2048 + ro1); }
  };
  auto read_b = [&](uint4 (&f)[2][2], int slot) __attribute__((always_inline)) {
    const char* sp = smem + slot * HT + b_off;
#pragma unroll
    for (int j = 0; j < 2; ++j) { f[j][0] = *(const uint4*)(sp + j * 2048 + ro0); f[j][1] = *(const uint4*)(sp + j * 2048 + ro1); }
  };
  auto mma_q = [&](auto aic, auto bjc, const uint4 (&fa)[4][2], const uint4 (&fb)[2][2]) __attribute__((always_inline)) {
    constexpr int ai = decltype(aic)::value, bj = decltype(bjc)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma(fa[i][s], fb[j][s], acc[ai * 4 + i][bj * 2 + j]);
  };
  std::integral_constant<int, 0> c0;
  std::integral_constant<int, 1> c1;
  auto mid = [&]() __attribute__((always_inline)) {          // end of the read section -> MFMA section
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (PRIO) __builtin_amdgcn_s_setprio(1);
  };
  auto end = [&]() __attribute__((always_inline)) {
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: half-tiles 0 .. NSLOT-2 requested, 0 and 1 awaited, A0 of K step 0 read
  for (int h = 0; h < NSLOT - 1 && h < nh; ++h) stage(h, h);
  {
    const int infl = (nh < NSLOT - 1 ? nh : NSLOT - 1) - 2;     // host guarantees nk >= 2
    if (infl >= DEPTH) wait_vm<2 * DEPTH>(); else wait_vm<0>();
  }
  __builtin_amdgcn_s_barrier();
  read_a(fa0, 0);
  if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();
  int p = 0;                       // phase counter
  int s_rd = 1;                    // slot of half-tile p+1
  int s_st = NSLOT - 1;            // slot of half-tile p+NSLOT-1
  auto adv = [&]() __attribute__((always_inline)) {
    ++p;
    s_rd = s_rd + 1 == NSLOT ? 0 : s_rd + 1;
    s_st = s_st + 1 == NSLOT ? 0 : s_st + 1;
  };
  // one K step = four phases.  TAIL = the last K steps, where the ring runs dry: requests and waits become conditional
  auto kstep = [&](auto tailc, int t) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(tailc)::value;
    constexpr int LEAD = NSLOT - 1;                  // phase p requests half-tile p + LEAD = K step t + (j + LEAD) / 4
    auto ph_head = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      if (!TAIL || p + LEAD < nh) stage_ty(std::integral_constant<int, (j + LEAD) & 3>(), t + (j + LEAD) / 4, s_st);
    };
    auto ph_wait = [&]() __attribute__((always_inline)) {
      if (TAIL) wait_tail(p); else wait_vm<2 * DEPTH>();
    };
    ph_head(std::integral_constant<int, 0>()); read_b(fb0, s_rd); ph_wait();
    mid(); mma_q(c0, c0, fa0, fb0); end(); adv();
    ph_head(std::integral_constant<int, 1>()); read_b(fb1, s_rd); ph_wait();
    mid(); mma_q(c0, c1, fa0, fb1); end(); adv();
    ph_head(std::integral_constant<int, 2>()); read_a(fa1, s_rd); ph_wait();
    mid(); mma_q(c1, c1, fa1, fb1); end(); adv();
    ph_head(std::integral_constant<int, 3>()); if (!TAIL || t + 1 < nk) read_a(fa0, s_rd); ph_wait();
    mid(); mma_q(c1, c0, fa1, fb0); end(); adv();
  };
  constexpr int NTAIL = (NSLOT - 1 + 3) / 4 + 1;     // K steps whose phases may find nothing left to request / wait for
  int t = 0;
  for (; t < nk - NTAIL; ++t) kstep(std::false_type(), t);
  for (; t < nk; ++t) kstep(std::true_type(), t);
  if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();

  const int crow = (lane >> 4) * 4, ccol = lane & 15;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* crow_p = C + (size_t)(m0 + wr * 128 + i * 16 + crow + r) * N + n0 + wc * 64 + ccol;
      if (STORE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) crow_p[j * 16] = acc[i][j][r];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (acc[i][j][r] == 12345.678f) crow_p[j * 16] = acc[i][j][r];
      }
    }
}

// ---- persistent variant of "8w 128x64 BK64 2st pipelined": one workgroup per CU walks the tiles (same XCD-aware order);
// the first two stages of the NEXT tile are requested before the epilogue of the current one (stage 0 into the buffer
// that frees up at the last barrier of the K loop, stage 1 after one more barrier into the buffer of the last K step),
// so the DMA start-up latency of a tile hides under the previous tile's stores.
template <int STORE>
__global__ __launch_bounds__(512) void gemm_persist(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                    float* __restrict__ C, int M, int N, int K, int store) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKB = 128, BKE = 64, STAGE = 65536, PPW = 8, FI = 8, FJ = 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = N / 256;
  const int ntiles = (M / 256) * tiles_n;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const bool isA = wave < 4;
  const int first = (wave & 3) * 64;                     // 8 pieces x 8 rows
  const uint16_t* src[PPW];
  auto tile_origin = [&](int vb, int& m0, int& n0) __attribute__((always_inline)) {
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = vb & 7, idx = vb >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    m0 = tm * 256; n0 = tn * 256;
  };
  // lane id re-derived through volatile asm where it is needed outside the K loop: everything computed from it then
  // stays out of the loop's live set (the K loop has no spare register: 128 acc + 96 fragment + 16 pointer VGPRs)
  auto fresh_lane = [&]() __attribute__((always_inline)) {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  auto set_src = [&](int m0, int n0) __attribute__((always_inline)) {
    const int l = fresh_lane();
    const int lr = l >> 3, lc = l & 7;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int row = first + i * 8 + lr;
      const uint16_t* base = isA ? A + (size_t)(m0 + row) * K : W + (size_t)(n0 + row) * K;
      src[i] = base + (lc ^ ((row >> 1) & 7)) * 8;
    }
  };
  const unsigned lds_wave = lds_base + (isA ? 0 : 256 * BKB) + first * BKB;
  auto issue = [&](int kt, int buf) __attribute__((always_inline)) {      // stage kt of the tile `src` points at
    const unsigned dst = lds_wave + buf * STAGE;
#pragma unroll
    for (int i = 0; i < PPW; ++i) dma1(src[i] + (size_t)kt * BKE, dst + i * 1024);
  };
  const int frow = lane & 15, lgrp = lane >> 4;
  int roff[2];
  roff[0] = ((0 + lgrp) ^ ((frow >> 1) & 7)) << 4;
  roff[1] = ((4 + lgrp) ^ ((frow >> 1) & 7)) << 4;
  const int a_off = (wr * 128 + frow) * BKB;
  const int b_off = 256 * BKB + (wc * 64 + frow) * BKB;
  const int nk = K / BKE, nq = 2 * nk;                   // host guarantees nk >= 2

  int vb = blockIdx.x;
  if (vb >= ntiles) return;
  int m0, n0, par = 0;                                   // stage kt of the current tile lives in buffer (kt + par) & 1
  tile_origin(vb, m0, n0);
  set_src(m0, n0);
  issue(0, 0);
  issue(1, 1);
  std::integral_constant<int, 0> b0;
  std::integral_constant<int, 1> b1;
  for (;;) {
    const int nvb = vb + gridDim.x;
    const bool has_next = nvb < ntiles;
    int nm0 = 0, nn0 = 0;
    if (has_next) tile_origin(nvb, nm0, nn0);
    f32x4 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4 fa[2][FI], fb[2][FJ];
    auto read_slice = [&](auto bufc, int q) __attribute__((always_inline)) {
      constexpr int buf = decltype(bufc)::value;
      const char* st = smem + (((q >> 1) + par) & 1) * STAGE;
      const int ro = roff[q & 1];
#pragma unroll
      for (int i = 0; i < FI; ++i) fa[buf][i] = *(const uint4*)(st + a_off + i * 16 * BKB + ro);
#pragma unroll
      for (int j = 0; j < FJ; ++j) fb[buf][j] = *(const uint4*)(st + b_off + j * 16 * BKB + ro);
    };
    auto mma_slice = [&](auto bufc) __attribute__((always_inline)) {
      constexpr int buf = decltype(bufc)::value;
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) mma(fa[buf][i], fb[buf][j], acc[i][j]);
    };
    auto block = [&](auto mbufc, int q) __attribute__((always_inline)) {
      constexpr int mb = decltype(mbufc)::value;
      read_slice(std::integral_constant<int, 1 - mb>(), q);
      mma_slice(mbufc);
#pragma unroll
      for (int r = 0; r < FI + FJ; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, FI * FJ - 2 * (FI + FJ), 0);
    };
    auto enter_stage = [&](int kt) __attribute__((always_inline)) {        // kt >= 1
      wait_vm<0>();
      __syncthreads();
      if (kt + 1 < nk) {
        issue(kt + 1, (kt + 1 + par) & 1);
      } else if (has_next) {                                               // last barrier of this tile's K loop:
        set_src(nm0, nn0);                                                 // the buffer of stage nk-2 is free ->
        issue(0, (nk + par) & 1);                                          // next tile's stage 0
      }
    };
    wait_vm<0>();                 // stages 0 and 1 of this tile (and the previous tile's stores)
    __syncthreads();
    read_slice(b0, 0);
    block(b0, 1);
    enter_stage(1);
    for (int u = 1; u <= nq - 5; u += 2) {
      block(b1, u + 1);
      block(b0, u + 2);
      enter_stage((u + 3) >> 1);
    }
    block(b1, nq - 2);
    block(b0, nq - 1);
    mma_slice(b1);
    if (has_next) {
      __syncthreads();            // nobody reads the last stage any more
      issue(1, (nk + 1 + par) & 1);
    }
    const int el = fresh_lane();
    const int crow = (el >> 4) * 4, ccol = el & 15;
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* crow_p = C + (size_t)(m0 + wr * 128 + i * 16 + crow + r) * N + n0 + wc * 64 + ccol;
        if (STORE) {
#pragma unroll
          for (int j = 0; j < FJ; ++j) crow_p[j * 16] = acc[i][j][r];
        } else {
#pragma unroll
          for (int j = 0; j < FJ; ++j)
            if (acc[i][j][r] == 12345.678f) crow_p[j * 16] = acc[i][j][r];
        }
      }
    if (!has_next) break;
    par = (par + nk) & 1;
    vb = nvb; m0 = nm0; n0 = nn0;
  }
}

__global__ void fill_ints(uint16_t* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ (unsigned)(i >> 32) ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const int v = (int)(h % 5) - 2;                             // -2..2, exact in bf16; sums stay exact in fp32
    const float f = (float)v;
    p[i] = (uint16_t)(__float_as_uint(f) >> 16);
  }
}

__global__ void naive_rows(const uint16_t* A, const uint16_t* W, float* C, int N, int K, int row0) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = row0 + blockIdx.y;
  if (n >= N) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k)
    s += __uint_as_float((unsigned)A[(size_t)m * K + k] << 16) * __uint_as_float((unsigned)W[(size_t)n * K + k] << 16);
  C[(size_t)blockIdx.y * N + n] = s;
}

// ---- the engine's epilogue (bias, optional fp32 residual, bf16 or fp32 output, full-line vector stores through an LDS
// transposition slab) behind the pipelined main loop, one tile per workgroup (PERSIST = 0, = what gemm2.hip does) or
// persistent with the next tile's two stages requested before the epilogue (PERSIST = 1).  The slab lives in the 32 KiB
// of LDS above the two 64 KiB stages (8-row passes, 17 KiB), so the prefetch and the epilogue do not share a buffer.
__device__ inline uint16_t f32_to_bf16_rne(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

template <int PERSIST, int OUTBF>
__global__ __launch_bounds__(512) void gemm_real(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, void* __restrict__ Cv,
                                                 const float* __restrict__ bias, const float* __restrict__ res, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKB = 128, BKE = 64, STAGE = 65536, PPW = 8, FI = 8, FJ = 4, SROW = 64 * 4 + 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = N / 256;
  const int ntiles = (M / 256) * tiles_n;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const bool isA = wave < 4;
  const int first = (wave & 3) * 64;
  const uint16_t* src[PPW];
  auto tile_origin = [&](int vb, int& m0, int& n0) __attribute__((always_inline)) {
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = vb & 7, idx = vb >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    m0 = tm * 256; n0 = tn * 256;
  };
  auto fresh_lane = [&]() __attribute__((always_inline)) {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  auto set_src = [&](int m0, int n0) __attribute__((always_inline)) {
    const int l = fresh_lane();
    const int lr = l >> 3, lc = l & 7;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int row = first + i * 8 + lr;
      const uint16_t* base = isA ? A + (size_t)(m0 + row) * K : W + (size_t)(n0 + row) * K;
      src[i] = base + (lc ^ ((row >> 1) & 7)) * 8;
    }
  };
  const unsigned lds_wave = lds_base + (isA ? 0 : 256 * BKB) + first * BKB;
  auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
    const unsigned dst = lds_wave + buf * STAGE;
#pragma unroll
    for (int i = 0; i < PPW; ++i) dma1(src[i] + (size_t)kt * BKE, dst + i * 1024);
  };
  const int frow = lane & 15, lgrp = lane >> 4;
  int roff[2];
  roff[0] = ((0 + lgrp) ^ ((frow >> 1) & 7)) << 4;
  roff[1] = ((4 + lgrp) ^ ((frow >> 1) & 7)) << 4;
  const int a_off = (wr * 128 + frow) * BKB;
  const int b_off = 256 * BKB + (wc * 64 + frow) * BKB;
  const int nk = K / BKE, nq = 2 * nk;

  int vb = blockIdx.x;
  if (vb >= ntiles) return;
  int m0, n0, par = 0;
  tile_origin(vb, m0, n0);
  set_src(m0, n0);
  issue(0, 0);
  issue(1, 1);
  std::integral_constant<int, 0> b0;
  std::integral_constant<int, 1> b1;
  for (;;) {
    const int nvb = vb + gridDim.x;
    const bool has_next = PERSIST && nvb < ntiles;
    int nm0 = 0, nn0 = 0;
    if (has_next) tile_origin(nvb, nm0, nn0);
    f32x4 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4 fa[2][FI], fb[2][FJ];
    auto read_slice = [&](auto bufc, int q) __attribute__((always_inline)) {
      constexpr int buf = decltype(bufc)::value;
      const char* st = smem + (((q >> 1) + par) & 1) * STAGE;
      const int ro = roff[q & 1];
#pragma unroll
      for (int i = 0; i < FI; ++i) fa[buf][i] = *(const uint4*)(st + a_off + i * 16 * BKB + ro);
#pragma unroll
      for (int j = 0; j < FJ; ++j) fb[buf][j] = *(const uint4*)(st + b_off + j * 16 * BKB + ro);
    };
    auto mma_slice = [&](auto bufc) __attribute__((always_inline)) {
      constexpr int buf = decltype(bufc)::value;
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) mma(fa[buf][i], fb[buf][j], acc[i][j]);
    };
    auto block = [&](auto mbufc, int q) __attribute__((always_inline)) {
      constexpr int mb = decltype(mbufc)::value;
      read_slice(std::integral_constant<int, 1 - mb>(), q);
      mma_slice(mbufc);
#pragma unroll
      for (int r = 0; r < FI + FJ; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, FI * FJ - 2 * (FI + FJ), 0);
    };
    auto enter_stage = [&](int kt) __attribute__((always_inline)) {
      wait_vm<0>();
      __syncthreads();
      if (kt + 1 < nk) {
        issue(kt + 1, (kt + 1 + par) & 1);
      } else if (has_next) {
        set_src(nm0, nn0);
        issue(0, (nk + par) & 1);
      }
    };
    wait_vm<0>();
    __syncthreads();
    read_slice(b0, 0);
    block(b0, 1);
    enter_stage(1);
    for (int u = 1; u <= nq - 5; u += 2) {
      block(b1, u + 1);
      block(b0, u + 2);
      enter_stage((u + 3) >> 1);
    }
    block(b1, nq - 2);
    block(b0, nq - 1);
    mma_slice(b1);
    if (has_next) {
      __syncthreads();
      issue(1, (nk + 1 + par) & 1);
    }
    // ---- epilogue: 8-row passes through the wave's private slab above the stages
    {
      char* slab = smem + 2 * STAGE + wave * (8 * SROW);
      const int el = fresh_lane();
      const int crow = ((el >> 4) & 1) * 4, ccol = el & 15, half = el >> 5;
      const int orow = el >> 3, ocol = (el & 7) * 8;
      const int col0 = n0 + wc * 64 + ocol;
      float b8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) b8[e] = bias ? bias[col0 + e] : 0.f;
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          __builtin_amdgcn_wave_barrier();
          if (half == h) {
#pragma unroll
            for (int j = 0; j < FJ; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) *(float*)(slab + (crow + r) * SROW + (j * 16 + ccol) * 4) = acc[i][j][r];
          }
          __builtin_amdgcn_wave_barrier();
          const float4 x0 = *(const float4*)(slab + orow * SROW + ocol * 4);
          const float4 x1 = *(const float4*)(slab + orow * SROW + ocol * 4 + 16);
          float v[8] = {x0.x + b8[0], x0.y + b8[1], x0.z + b8[2], x0.w + b8[3], x1.x + b8[4], x1.y + b8[5], x1.z + b8[6], x1.w + b8[7]};
          const size_t off = (size_t)(m0 + wr * 128 + i * 16 + h * 8 + orow) * N + col0;
          if (res) {
            const float4 r0 = *(const float4*)(res + off), r1 = *(const float4*)(res + off + 4);
            v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
          }
          if (OUTBF) {
            uint4 o;
            o.x = (unsigned)f32_to_bf16_rne(v[0]) | ((unsigned)f32_to_bf16_rne(v[1]) << 16);
            o.y = (unsigned)f32_to_bf16_rne(v[2]) | ((unsigned)f32_to_bf16_rne(v[3]) << 16);
            o.z = (unsigned)f32_to_bf16_rne(v[4]) | ((unsigned)f32_to_bf16_rne(v[5]) << 16);
            o.w = (unsigned)f32_to_bf16_rne(v[6]) | ((unsigned)f32_to_bf16_rne(v[7]) << 16);
            *(uint4*)((uint16_t*)Cv + off) = o;
          } else {
            *(float4*)((float*)Cv + off) = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)((float*)Cv + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
          }
        }
    }
    if (!has_next) break;
    par = (par + nk) & 1;
    vb = nvb; m0 = nm0; n0 = nn0;
  }
}

__global__ void fill_small_f32(float* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2246822519u ^ (unsigned)(i >> 32) ^ seed;
    h ^= h >> 13; h *= 2654435761u; h ^= h >> 16;
    p[i] = (float)((int)(h % 9) - 4);
  }
}

// second half of the program: `gemm_lab <reps> real`
static int real_epilogue_run(int reps, int ncu) {
  struct Shape { int M, N, K, outbf, use_res; const char* what; };
  const Shape shapes[] = {{90112, 1024, 1024, 0, 1, "out-proj / pw2: f32 out + residual"}, {90112, 4096, 1024, 1, 0, "ff1: bf16 out"},
                          {90112, 1024, 4096, 0, 1, "ff2: f32 out + residual"}, {90112, 3072, 1024, 1, 0, "qkv: bf16 out"}};
  const int CHK = 512;
  auto k00 = gemm_real<0, 0>; auto k01 = gemm_real<0, 1>; auto k10 = gemm_real<1, 0>; auto k11 = gemm_real<1, 1>;
  const int LDS = 2 * 65536 + 8 * 8 * (64 * 4 + 16);
  CHECK(hipFuncSetAttribute((const void*)k00, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CHECK(hipFuncSetAttribute((const void*)k01, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CHECK(hipFuncSetAttribute((const void*)k10, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CHECK(hipFuncSetAttribute((const void*)k11, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  for (const Shape& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K;
    uint16_t *A, *W; float *R, *bias, *resid = nullptr; void* C;
    const size_t esz = sh.outbf ? 2 : 4;
    CHECK(hipMalloc(&A, (size_t)M * K * 2)); CHECK(hipMalloc(&W, (size_t)N * K * 2));
    CHECK(hipMalloc(&C, (size_t)M * N * esz)); CHECK(hipMalloc(&R, (size_t)2 * CHK * N * 4)); CHECK(hipMalloc(&bias, (size_t)N * 4));
    if (sh.use_res) { CHECK(hipMalloc(&resid, (size_t)M * N * 4)); hipLaunchKernelGGL(fill_small_f32, dim3(4096), dim3(256), 0, 0, resid, (size_t)M * N, 5u); }
    hipLaunchKernelGGL(fill_ints, dim3(4096), dim3(256), 0, 0, A, (size_t)M * K, 17u);
    hipLaunchKernelGGL(fill_ints, dim3(4096), dim3(256), 0, 0, W, (size_t)N * K, 91u);
    hipLaunchKernelGGL(fill_small_f32, dim3(16), dim3(256), 0, 0, bias, (size_t)N, 3u);
    hipLaunchKernelGGL(naive_rows, dim3((N + 255) / 256, CHK), dim3(256), 0, 0, A, W, R, N, K, 0);
    hipLaunchKernelGGL(naive_rows, dim3((N + 255) / 256, CHK), dim3(256), 0, 0, A, W, R + (size_t)CHK * N, N, K, M - CHK);
    CHECK(hipDeviceSynchronize());
    std::vector<float> ref((size_t)2 * CHK * N), hb(N), hres;
    CHECK(hipMemcpy(ref.data(), R, ref.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hb.data(), bias, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (sh.use_res) {
      hres.resize((size_t)2 * CHK * N);
      CHECK(hipMemcpy(hres.data(), resid, (size_t)CHK * N * 4, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(hres.data() + (size_t)CHK * N, resid + (size_t)(M - CHK) * N, (size_t)CHK * N * 4, hipMemcpyDeviceToHost));
    }
    for (size_t i = 0; i < ref.size(); ++i) ref[i] += hb[i % N] + (sh.use_res ? hres[i] : 0.f);      // small integers: exact
    printf("M=%d N=%d K=%d  (%s)\n", M, N, K, sh.what);
    const int tiles = (M / 256) * (N / 256);
    for (int persist = 0; persist < 2; ++persist) {
      auto kern = sh.outbf ? (persist ? k11 : k01) : (persist ? k10 : k00);
      const int grid = persist ? (tiles < ncu ? tiles : ncu) : tiles;
      CHECK(hipMemset(C, 0xff, (size_t)M * N * esz));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, 0, A, W, C, bias, resid, M, N, K);
      CHECK(hipGetLastError());
      CHECK(hipDeviceSynchronize());
      std::vector<char> got((size_t)2 * CHK * N * esz);
      CHECK(hipMemcpy(got.data(), C, (size_t)CHK * N * esz, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(got.data() + (size_t)CHK * N * esz, (char*)C + (size_t)(M - CHK) * N * esz, (size_t)CHK * N * esz, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < ref.size(); ++i) {
        if (sh.outbf) {
          unsigned u; memcpy(&u, &ref[i], 4); u += 0x7fffu + ((u >> 16) & 1u);
          bad += ((uint16_t*)got.data())[i] != (uint16_t)(u >> 16);
        } else {
          bad += !(((float*)got.data())[i] == ref[i]);
        }
      }
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, 0, A, W, C, bias, resid, M, N, K);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      ms /= reps;
      printf("  pipelined loop + engine epilogue, %s  %8.3f ms  %7.1f TFLOP/s  %s\n", persist ? "persistent + next-tile prefetch" : "one tile per workgroup         ",
             ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12, bad ? "WRONG" : "exact");
      if (bad) printf("    (%zu of %zu checked elements differ)\n", bad, ref.size());
      fflush(stdout);
    }
    CHECK(hipFree(A)); CHECK(hipFree(W)); CHECK(hipFree(C)); CHECK(hipFree(R)); CHECK(hipFree(bias));
    if (resid) CHECK(hipFree(resid));
  }
  return 0;
}

// ---- 3x3 stride-1 convolution as an implicit GEMM on the same pipelined LDS-DMA loop (candidate for the ResNet34 stages
// with 128 / 256 channels): rows = output pixels of a bordered NHWC tensor
// [B][F+2][T+2][Cin] (zero border), one K step = 64 channels of one tap gathered from the shifted pixels, weights
// [Cout][9][Cin].  Tile 256 pixels x BN channels, 8 waves: 2x4 of 128x64 (BN = 256) or 4x2 of 64x64 (BN = 128).
template <int BN>
__global__ __launch_bounds__(512) void conv_igemm(const uint16_t* __restrict__ in, const uint16_t* __restrict__ w, float* __restrict__ out,
                                                  int B, int F, int T, int Cin, int Cout) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKB = 128, BKE = 64;
  constexpr int NWN = BN / 64, NWM = 8 / NWN;                 // 4 or 2 waves along N
  constexpr int TM = 256 / NWM, FI = TM / 16, FJ = 4;         // wave tile TM x 64
  constexpr int STAGE = (256 + BN) * BKB;
  constexpr int WP = BN / 64;                                  // W pieces per wave (8 rows each)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / NWN, wc = wave % NWN;
  const int TP = T + 2, FP = F + 2;
  const int tiles_n = Cout / BN;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * 256, n0 = tn * BN;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int lr = lane >> 3, lc = lane & 7;
  const uint16_t* a_src[4];
  const uint16_t* w_src[WP];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + lr;
    const int m = m0 + row;
    const int b = m / (F * T), rem = m - b * (F * T);
    const int fo = rem / T, to = rem - fo * T;
    a_src[i] = in + ((size_t)(b * FP + fo) * TP + to) * Cin + (lc ^ ((row >> 1) & 7)) * 8;      // tap (0,0), channel 0
  }
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int row = wave * (WP * 8) + i * 8 + lr;
    w_src[i] = w + (size_t)(n0 + row) * 9 * Cin + (lc ^ ((row >> 1) & 7)) * 8;
  }
  const int cpt = Cin / BKE;                                   // K steps per tap
  auto issue = [&](int kt) __attribute__((always_inline)) {
    const int tap = kt / cpt, c0 = (kt - tap * cpt) * BKE;
    const int kh = tap / 3, kw = tap - kh * 3;
    const size_t aoff = (size_t)(kh * TP + kw) * Cin + c0;
    const size_t woff = (size_t)tap * Cin + c0;
    const unsigned dst = lds_base + (kt & 1) * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma1(a_src[i] + aoff, dst + (wave * 32 + i * 8) * BKB);
#pragma unroll
    for (int i = 0; i < WP; ++i) dma1(w_src[i] + woff, dst + 256 * BKB + (wave * (WP * 8) + i * 8) * BKB);
  };
  f32x4 acc[FI][FJ];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, lgrp = lane >> 4;
  int roff[2];
  roff[0] = ((0 + lgrp) ^ ((frow >> 1) & 7)) << 4;
  roff[1] = ((4 + lgrp) ^ ((frow >> 1) & 7)) << 4;
  const int a_off = (wr * TM + frow) * BKB;
  const int b_off = 256 * BKB + (wc * 64 + frow) * BKB;
  const int nk = 9 * cpt, nq = 2 * nk;
  uint4 fa[2][FI], fb[2][FJ];
  auto read_slice = [&](auto bufc, int q) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
    const char* st = smem + ((q >> 1) & 1) * STAGE;
    const int ro = roff[q & 1];
#pragma unroll
    for (int i = 0; i < FI; ++i) fa[buf][i] = *(const uint4*)(st + a_off + i * 16 * BKB + ro);
#pragma unroll
    for (int j = 0; j < FJ; ++j) fb[buf][j] = *(const uint4*)(st + b_off + j * 16 * BKB + ro);
  };
  auto mma_slice = [&](auto bufc) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) mma(fa[buf][i], fb[buf][j], acc[i][j]);
  };
  auto block = [&](auto mbufc, int q) __attribute__((always_inline)) {
    constexpr int mb = decltype(mbufc)::value;
    read_slice(std::integral_constant<int, 1 - mb>(), q);
    mma_slice(mbufc);
    constexpr int NR = FI + FJ, NM = FI * FJ, PER = NM / NR >= 2 ? 2 : 1;      // MFMAs between two reads
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    if constexpr (NM - PER * NR > 0) __builtin_amdgcn_sched_group_barrier(0x008, NM - PER * NR, 0);
  };
  auto enter_stage = [&](int kt) __attribute__((always_inline)) {
    wait_vm<0>();
    __syncthreads();
    if (kt + 1 < nk) issue(kt + 1);
  };
  std::integral_constant<int, 0> b0;
  std::integral_constant<int, 1> b1;
  issue(0);
  issue(1);
  wait_vm<4 + WP>();                                           // stage 0 = the older group of 4 + WP pieces
  __syncthreads();
  read_slice(b0, 0);
  block(b0, 1);
  enter_stage(1);
  for (int u = 1; u <= nq - 5; u += 2) {
    block(b1, u + 1);
    block(b0, u + 2);
    enter_stage((u + 3) >> 1);
  }
  block(b1, nq - 2);
  block(b0, nq - 1);
  mma_slice(b1);
  const int crow = (lane >> 4) * 4, ccol = lane & 15;
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* op = out + (size_t)(m0 + wr * TM + i * 16 + crow + r) * Cout + n0 + wc * 64 + ccol;
#pragma unroll
      for (int j = 0; j < FJ; ++j) op[j * 16] = acc[i][j][r];
    }
}

__global__ void naive_conv_rows(const uint16_t* in, const uint16_t* w, float* out, int F, int T, int Cin, int Cout, int row0) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = row0 + blockIdx.y;
  if (n >= Cout) return;
  const int TP = T + 2, FP = F + 2;
  const int b = m / (F * T), rem = m - b * (F * T), fo = rem / T, to = rem - fo * T;
  float s = 0.f;
  for (int tap = 0; tap < 9; ++tap) {
    const int kh = tap / 3, kw = tap - kh * 3;
    const uint16_t* px = in + ((size_t)(b * FP + fo + kh) * TP + to + kw) * Cin;
    const uint16_t* wr = w + ((size_t)n * 9 + tap) * Cin;
    for (int c = 0; c < Cin; ++c) s += __uint_as_float((unsigned)px[c] << 16) * __uint_as_float((unsigned)wr[c] << 16);
  }
  out[(size_t)blockIdx.y * Cout + n] = s;
}

__global__ void zero_border(uint16_t* in, int B, int F, int T, int Cin) {
  const int TP = T + 2, FP = F + 2;
  const size_t npx = (size_t)B * FP * TP;
  for (size_t px = blockIdx.x * (size_t)blockDim.x + threadIdx.x; px < npx; px += (size_t)gridDim.x * blockDim.x) {
    const int t = px % TP, f = (px / TP) % FP;
    if (t == 0 || t == TP - 1 || f == 0 || f == FP - 1)
      for (int c = 0; c < Cin; ++c) in[px * Cin + c] = 0;
  }
}

// `gemm_lab <reps> conv`
static int conv_run(int reps) {
  struct Shape { int B, F, T, C; const char* what; };
  const Shape shapes[] = {{192, 20, 250, 128, "ResNet34 stage 3: 128 -> 128 channels, 192 windows"}, {192, 10, 128, 256, "ResNet34 stage 4: 256 -> 256 channels (T padded 125 -> 128)"}};
  const int CHK = 512;
  auto k128 = conv_igemm<128>; auto k256 = conv_igemm<256>;
  CHECK(hipFuncSetAttribute((const void*)k128, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 128) * 128));
  CHECK(hipFuncSetAttribute((const void*)k256, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 256) * 128));
  for (const Shape& sh : shapes) {
    const int B = sh.B, F = sh.F, T = sh.T, C = sh.C;
    const size_t M = (size_t)B * F * T;
    if (M % 256) { printf("skip %s: pixels not a multiple of 256\n", sh.what); continue; }
    const size_t nin = (size_t)B * (F + 2) * (T + 2) * C;
    uint16_t *in, *w; float *out, *R;
    CHECK(hipMalloc(&in, nin * 2)); CHECK(hipMalloc(&w, (size_t)C * 9 * C * 2));
    CHECK(hipMalloc(&out, M * C * 4)); CHECK(hipMalloc(&R, (size_t)2 * CHK * C * 4));
    hipLaunchKernelGGL(fill_ints, dim3(4096), dim3(256), 0, 0, in, nin, 23u);
    hipLaunchKernelGGL(zero_border, dim3(4096), dim3(256), 0, 0, in, B, F, T, C);
    hipLaunchKernelGGL(fill_ints, dim3(1024), dim3(256), 0, 0, w, (size_t)C * 9 * C, 77u);
    hipLaunchKernelGGL(naive_conv_rows, dim3((C + 255) / 256, CHK), dim3(256), 0, 0, in, w, R, F, T, C, C, 0);
    hipLaunchKernelGGL(naive_conv_rows, dim3((C + 255) / 256, CHK), dim3(256), 0, 0, in, w, R + (size_t)CHK * C, F, T, C, C, (int)(M - CHK));
    CHECK(hipDeviceSynchronize());
    std::vector<float> ref((size_t)2 * CHK * C), got((size_t)2 * CHK * C);
    CHECK(hipMemcpy(ref.data(), R, ref.size() * 4, hipMemcpyDeviceToHost));
    printf("%s: %zu pixels, K = %d\n", sh.what, M, 9 * C);
    for (int bn : {128, 256}) {
      if (C % bn) continue;
      auto kern = bn == 128 ? k128 : k256;
      const int lds = 2 * (256 + bn) * 128;
      const int tiles = (int)(M / 256) * (C / bn);
      CHECK(hipMemset(out, 0xff, M * C * 4));
      hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, 0, in, w, out, B, F, T, C, C);
      CHECK(hipGetLastError());
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(got.data(), out, (size_t)CHK * C * 4, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(got.data() + (size_t)CHK * C, out + (M - CHK) * C, (size_t)CHK * C * 4, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < ref.size(); ++i) bad += !(ref[i] == got[i]);
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, 0, in, w, out, B, F, T, C, C);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      ms /= reps;
      printf("  implicit GEMM, 256 x %3d tile  %8.3f ms  %7.1f TFLOP/s  %s   (direct conv_kernel in the engine: 555-598 TFLOP/s on these stages)\n", bn, ms,
             2.0 * M * C * 9.0 * C / (ms * 1e-3) / 1e12, bad ? "WRONG" : "exact");
      if (bad) printf("    (%zu of %zu checked elements differ)\n", bad, ref.size());
      fflush(stdout);
    }
    CHECK(hipFree(in)); CHECK(hipFree(w)); CHECK(hipFree(out)); CHECK(hipFree(R));
  }
  return 0;
}

struct Variant { const char* name; void (*kern)(const uint16_t*, const uint16_t*, float*, int, int, int, int); int threads, lds, store, persistent, phases; int bn = 256; };

template <int NWM, int NWN, int KSUBS, int NST, int PIPE, int AMMA = 0, int STORE = 1, int BN = 256, int OCC = 1> Variant make(const char* name, int phases = 0) {
  auto k = gemm_lab<NWM, NWN, KSUBS, NST, PIPE, AMMA, STORE, BN, OCC>;
  const int lds = NST * (256 + BN) * 64 * KSUBS;
  CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  return {name, k, NWM * NWN * 64, lds, STORE, 0, phases, BN};
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 5;
  int ncu = 256;
  { hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0)); ncu = pr.multiProcessorCount; }
  if (argc > 2 && std::string(argv[2]) == "real") return real_epilogue_run(reps, ncu);
  if (argc > 2 && std::string(argv[2]) == "conv") return conv_run(reps);
  std::vector<Variant> vs;
  vs.push_back(make<2, 4, 2, 2, 0>("8w 128x64  BK64 2st simple (=gemm2)"));
  vs.push_back(make<2, 4, 2, 2, 1>("8w 128x64  BK64 2st pipelined     "));
  vs.push_back(make<2, 2, 2, 2, 0, 1>("4w 128x128 BK64 2st simple        "));
  vs.push_back(make<2, 2, 2, 2, 1, 1>("4w 128x128 BK64 2st pipelined     "));
  vs.push_back(make<2, 4, 1, 4, 0>("8w 128x64  BK32 4st simple        "));
  vs.push_back(make<2, 4, 1, 4, 1>("8w 128x64  BK32 4st pipelined     "));
  vs.push_back(make<2, 2, 1, 4, 0, 1>("4w 128x128 BK32 4st simple        "));
  vs.push_back(make<2, 2, 1, 4, 1, 1>("4w 128x128 BK32 4st pipelined     "));
  vs.push_back(make<2, 4, 1, 3, 1>("8w 128x64  BK32 3st pipelined     "));
  CHECK(hipFuncSetAttribute((const void*)gemm_persist<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CHECK(hipFuncSetAttribute((const void*)gemm_persist<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  vs.push_back({"8w BK64 2st pipelined PERSISTENT  ", gemm_persist<1>, 512, 131072, 1, 1, 0});
  vs.push_back(make<2, 4, 2, 2, 0, 0, 0>("8w simple     NO STORES          "));
  vs.push_back(make<2, 4, 2, 2, 1, 0, 0>("8w pipelined  NO STORES          "));
  vs.push_back({"8w pipelined PERSISTENT NO STORES ", gemm_persist<0>, 512, 131072, 0, 1, 0});
  vs.push_back(make<2, 4, 2, 2, 1>("8w pipelined  first wave in 2 phases", 2));
  vs.push_back(make<2, 4, 2, 2, 1>("8w pipelined  first wave in 4 phases", 4));
  vs.push_back(make<2, 4, 2, 2, 1>("8w pipelined  first wave in 8 phases", 8));
  vs.push_back(make<2, 4, 2, 2, 0>("8w simple     first wave in 4 phases", 4));
  if (argc > 2) vs.erase(vs.begin() + 2, vs.begin() + 10);      // short run: skip the 4-wave / BK32 variants
  const bool tile_mode = argc > 2 && std::string(argv[2]) == "tile";
  if (tile_mode) vs.erase(vs.begin() + 2, vs.end());            // the two baselines + the phase loops + the 256 x 128 tiles below
  if (argc > 2 && std::string(argv[2]) == "phase") vs.erase(vs.begin() + 2, vs.end());   // the two baselines + the phase-interleaved loops
  {
    auto addph = [&](const char* name, auto kern, int nslot) {
      CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, nslot * 16384));
      vs.push_back({name, kern, 512, nslot * 16384, 1, 0, 0});
    };
    addph("phases  8 slots                    ", gemm_ph<8, 0, 0, 1>, 8);
    addph("phases  8 slots stagger            ", gemm_ph<8, 1, 0, 1>, 8);
    addph("phases  8 slots stagger prio       ", gemm_ph<8, 1, 1, 1>, 8);
    addph("phases  8 slots         prio       ", gemm_ph<8, 0, 1, 1>, 8);
    addph("phases 10 slots stagger prio       ", gemm_ph<10, 1, 1, 1>, 10);
    addph("phases 10 slots stagger            ", gemm_ph<10, 1, 0, 1>, 10);
    addph("phases  6 slots stagger prio       ", gemm_ph<6, 1, 1, 1>, 6);
  }
  if (!(argc > 2 && std::string(argv[2]) == "phase") && !tile_mode) vs.push_back(make<2, 2, 1, 3, 1, 1>("4w 128x128 BK32 3st pipelined     "));
  if (tile_mode) {
    // VERDICT r4, "next" 3 (d): the 256 x 128 tile of FOUR waves (2 x 2 of 128 x 64, the wave tile of the 8-wave kernels), 64-byte
    // rows and three 24-KiB stages = 72 KiB: TWO workgroups per CU, each filling the other's barriers, first-stage wait and epilogue
    vs.push_back(make<2, 2, 1, 3, 1, 0, 1, 128, 2>("4w 256x128 BK32 3st pipelined 2/CU"));
    vs.push_back(make<2, 2, 1, 3, 0, 0, 1, 128, 2>("4w 256x128 BK32 3st simple    2/CU"));
    vs.push_back(make<2, 2, 1, 2, 1, 0, 1, 128, 2>("4w 256x128 BK32 2st pipelined 2/CU"));
    vs.push_back(make<2, 2, 2, 2, 1, 0, 1, 128, 1>("4w 256x128 BK64 2st pipelined 1/CU"));
    vs.push_back(make<2, 4, 2, 2, 1, 0, 1, 128, 1>("8w 256x128 BK64 2st pipelined     "));
  }

  struct Shape { int M, N, K; };
  const Shape shapes[] = {{90112, 1024, 1024}, {90112, 4096, 1024}, {90112, 1024, 4096}, {90112, 1024, 19456}};
  const int CHK = 512;                                           // rows checked at the top and at the bottom of C
  for (const Shape& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K;
    uint16_t *A, *W;
    float *C, *R;
    CHECK(hipMalloc(&A, (size_t)M * K * 2)); CHECK(hipMalloc(&W, (size_t)N * K * 2));
    CHECK(hipMalloc(&C, (size_t)M * N * 4)); CHECK(hipMalloc(&R, (size_t)2 * CHK * N * 4));
    hipLaunchKernelGGL(fill_ints, dim3(4096), dim3(256), 0, 0, A, (size_t)M * K, 17u);
    hipLaunchKernelGGL(fill_ints, dim3(4096), dim3(256), 0, 0, W, (size_t)N * K, 91u);
    hipLaunchKernelGGL(naive_rows, dim3((N + 255) / 256, CHK), dim3(256), 0, 0, A, W, R, N, K, 0);
    hipLaunchKernelGGL(naive_rows, dim3((N + 255) / 256, CHK), dim3(256), 0, 0, A, W, R + (size_t)CHK * N, N, K, M - CHK);
    CHECK(hipDeviceSynchronize());
    std::vector<float> ref((size_t)2 * CHK * N), got((size_t)2 * CHK * N);
    CHECK(hipMemcpy(ref.data(), R, ref.size() * 4, hipMemcpyDeviceToHost));
    printf("M=%d N=%d K=%d\n", M, N, K);
    for (const Variant& v : vs) {
      const int tiles = (M / 256) * (N / v.bn);
      CHECK(hipMemset(C, 0xff, (size_t)M * N * 4));
      const int grid = v.persistent ? (tiles < ncu ? tiles : ncu) : tiles;
      hipLaunchKernelGGL(v.kern, dim3(grid), dim3(v.threads), v.lds, 0, A, W, C, M, N, K, v.persistent ? v.store : v.phases);
      CHECK(hipGetLastError());
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(got.data(), C, (size_t)CHK * N * 4, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(got.data() + (size_t)CHK * N, C + (size_t)(M - CHK) * N, (size_t)CHK * N * 4, hipMemcpyDeviceToHost));
      size_t bad = 0;
      if (v.store) for (size_t i = 0; i < ref.size(); ++i) bad += !(ref[i] == got[i]);
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(v.kern, dim3(grid), dim3(v.threads), v.lds, 0, A, W, C, M, N, K, v.persistent ? v.store : v.phases);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      ms /= reps;
      printf("  %s  %8.3f ms  %7.1f TFLOP/s  %s\n", v.name, ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12,
             !v.store ? "(not checked)" : bad ? "WRONG" : "exact");
      if (bad) printf("    (%zu of %zu checked elements differ)\n", bad, ref.size());
      fflush(stdout);
    }
    CHECK(hipFree(A)); CHECK(hipFree(W)); CHECK(hipFree(C)); CHECK(hipFree(R));
  }
  return 0;
}
