// 3x3 convolution (stride 1 or 2, pad 1) of the ResNet34 trunk as an implicit GEMM on gemm2.hip's pipelined LDS-DMA loop (bf16).
//
//   rows    = output pixels (b, f, t); the input is the bordered NHWC tensor [B][Fi+2][Ti+2][Cin] (zero border = the padding)
//   K steps = 64 input channels of one tap; the A operand of a step is gathered by LDS-DMA from the input pixels
//             (S f + kh, S t + kw) of the bordered tensor (S = stride): one 128-byte run per pixel, no im2col buffer, no
//             register staging.  Round 4: the two stride-2 convolutions that open stages 3 and 4 (64 -> 128, 128 -> 256) run
//             here too -- on the direct kernel they were the 512-VGPR + scratch instantiations (24 ms per hour of audio)
//   columns = output channels, weights [Cout][9][Cin] (BatchNorm folded), tile 256 pixels x 256 channels or (round 5) 512 pixels x
//             128 channels: either way 8 waves of 128 pixels x 64 channels.  The 128-channel stage ran on 256 x 128 tiles until then
//             (wave tile 64 x 64: a third more LDS fragment bytes per MFMA, at the LDS port's limit): 65.5 -> 60.9 ms per hour
//             (profiles/archive/r05_call15_strips_igemm512.txt); the variant with the fused shortcut stays on 256 pixels (it spills at 512)
//
// Main loop and gather measured in scripts/micro/gemm_lab.hip ("conv" mode, exact against a naive convolution):
// 855 TFLOP/s for the 128-channel stage and 1193 for the 256-channel stage, against 555-598 for the direct
// convolution (resnet.hip: conv_kernel), which stages a 6 x 66 pixel patch per block and runs one wave per SIMD.
// The epilogue is conv_kernel's: 16-row slabs transposed through LDS, bias + residual + ReLU, full 128-byte runs of
// channels per pixel into the bordered output.
//
// Measured and not kept (round 5, profiles/archive/r05_call11_igemm_small_tiles.txt): 128 x 128 tiles of four waves, 64 KiB of stages, TWO
// workgroups per CU -- the remedy that paid for the 32- / 64-channel stages: 64.2-64.7 vs 65.0-65.3 ms for the 128-channel stage,
// 24.4-24.8 vs 25.0 ms for the 256-channel one (-1 %): this loop is paced by its stage fill, not by what overlaps what on a CU.
//
// Measured and not kept (round 5, profiles/archive/r05_call18_conv_flat_not_kept.txt): the same convolutions over the FLAT pixel index of the
// bordered tensor (tap (kh, kw) of pixel m is pixel m + (kh - 1)(T + 2) + (kw - 1): a tile's 512 + 2 (T + 2) + 2 pixels resident in
// LDS for all nine taps, persistent workgroups, a third of this loop's LDS-DMA requests per MFMA).  The image fits only for
// T + 2 <= 127 -- the 256-channel stage; the 128-channel stage's 20 x 250 maps would need 2 x 65 KB of pixels beside 48 KB of
// weights -- and there the border positions a flat tile has to compute are 1.22 x the MFMAs: 26.6 ms against 25.4 here.
//
// STATUS (round 2): the default for the 128- and 256-channel stages (diar_engine.hip packs the second weight layout unless
// RVD_CONV_IGEMM=0); tests/test_diar_gpu.py compares it with resnet.hip's direct kernel.  Tried and not kept: the
// row-contiguous read-back + up-front residual prefetch that helped gemm2's fp32 epilogue (no change here: bf16 output and
// residual are a quarter of those bytes, the tile is bound by its K loop).
#include "common.h"
#include "kernels.h"

#include <cstdlib>
#include <type_traits>

namespace rvb {

namespace {

__device__ inline void cg_mma(const uint4& a, const uint4& b, f32x4_t& c) {
  union U { uint4 u; bf16x8_t v; };
  U ua, ub;
  ua.u = a; ub.u = b;
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, c, 0, 0, 0);
}

// one 1-KiB LDS-DMA piece (64 lanes x 16 bytes, lane order); inline asm so that hipcc's waitcnt pass does not see it
// (it would drain vmcnt(0) before every ds_read); ordered by the explicit s_waitcnt in the loop.  M0 is saved/restored.
__device__ inline void cg_dma(const void* g, unsigned lds) {
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

// one 16-byte store the compiler's waitcnt pass does not see (see the epilogue); s_nop 1: the TWO wait states gfx950 wants
// between a store of more than 8 bytes and a VALU write of its data registers (gemm2.hip: store16_hidden)
__device__ inline void cg_store16(void* q, const uint4& v) {
  typedef unsigned cg_u32x4 __attribute__((ext_vector_type(4)));
  const cg_u32x4 d = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(q), "v"(d) : "memory");
}

__device__ inline void cg_pin(uint4& a, uint4& b) {
  typedef unsigned cg_u32x4 __attribute__((ext_vector_type(4)));
  cg_u32x4 x = {a.x, a.y, a.z, a.w}, y = {b.x, b.y, b.z, b.w};
  asm volatile("" : "+v"(x), "+v"(y)::"memory");
  a = make_uint4(x[0], x[1], x[2], x[3]); b = make_uint4(y[0], y[1], y[2], y[3]);
}

// SC: the block's projection shortcut rides in this convolution's K loop (ConvArgs::in2; its own instantiation, so that the plain
// form keeps its registers: the 256-channel tile sits at 254 VGPRs)
template <int BN, bool SC = false, int BM = 256>
__global__ __launch_bounds__(512) void conv_igemm_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char cg_smem[];
  constexpr int BKB = 128, BKE = 64;
  constexpr int NWN = BN / 64, NWM = 8 / NWN;                 // waves along channels / pixels
  constexpr int TM = BM / NWM, FI = TM / 16, FJ = 4;          // wave tile TM pixels x 64 channels
  constexpr int STAGE = (BM + BN) * BKB;
  constexpr int AP = BM / 64, AR = BM / 8;                    // pixel pieces (8 rows each) / pixel rows per wave
  constexpr int WP = BN / 64;                                  // weight pieces (8 rows each) per wave
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / NWN, wc = wave % NWN;
  const int F = p.Fo, T = p.To, Cin = p.Cin, Cout = p.Cout;
  const int TP = T + 2, FP = F + 2;                           // bordered OUTPUT (and residual) geometry
  const int TPi = p.Ti + 2, FPi = p.Fi + 2, S = p.stride;     // bordered INPUT geometry, stride
  const int M = p.B * F * T;
  const int tiles_n = Cout / BN;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {   // XCD-aware bijective tile order (as gemm2)
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const bf16_t* __restrict__ in = (const bf16_t*)p.in;
  const bf16_t* __restrict__ w = (const bf16_t*)p.w_ig;
  const size_t ldw = (size_t)9 * Cin + (SC ? p.Cin2 : 0);      // weight row: 9 taps (+ the fused shortcut's channels)
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)cg_smem;

  // ---- DMA sources: wave w stages pixels [32w, 32w+32) and weight rows [8 WP w, +8 WP); source column swizzled
  const int lr = lane >> 3, lc = lane & 7;
  const bf16_t* a_src[AP];
  const bf16_t* w_src[WP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int row = wave * AR + i * 8 + lr;
    int m = m0 + row;
    if (m >= M) m = M - 1;                                     // clamped rows are computed and never stored
    const int b = m / (F * T), rem = m - b * (F * T);
    const int fo = rem / T, to = rem - fo * T;
    a_src[i] = in + ((size_t)(b * FPi + S * fo) * TPi + S * to) * Cin + (lc ^ ((row >> 1) & 7)) * 8;      // tap (0,0), channel 0
  }
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int row = wave * (WP * 8) + i * 8 + lr;
    w_src[i] = w + (size_t)(n0 + row) * ldw + (lc ^ ((row >> 1) & 7)) * 8;
  }
  const int cpt = Cin / BKE;                                   // K steps per tap
  const int nk9 = 9 * cpt;
  auto issue = [&](int kt) __attribute__((always_inline)) {
    const unsigned dst = lds_base + (kt & 1) * STAGE;
    if (SC && kt >= nk9) {
      // fused projection shortcut (p.in2): K step j of the 1 x 1 / stride-s2 convolution of the block's input -- pixel
      // (s2 fo + 1, s2 to + 1) of its bordered tensor, channels [64 j, +64); the per-lane addresses are rebuilt here (one or
      // two K steps per tile: cheaper than four more pointers alive through the main loop)
      const int j = kt - nk9;
      const bf16_t* in2 = (const bf16_t*)p.in2;
      const int TP2 = p.Ti2 + 2, FP2 = p.Fi2 + 2, s2 = p.stride2, C2 = p.Cin2;
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const int row = wave * AR + i * 8 + lr;
        int m = m0 + row;
        if (m >= M) m = M - 1;
        const int b = m / (F * T), rem = m - b * (F * T);
        const int fo = rem / T, to = rem - fo * T;
        const bf16_t* src = in2 + ((size_t)(b * FP2 + s2 * fo + 1) * TP2 + s2 * to + 1) * C2 + j * BKE + (lc ^ ((row >> 1) & 7)) * 8;
        cg_dma(src, dst + (wave * AR + i * 8) * BKB);
      }
      const size_t woff2 = (size_t)9 * Cin + (size_t)j * BKE;
#pragma unroll
      for (int i = 0; i < WP; ++i) cg_dma(w_src[i] + woff2, dst + BM * BKB + (wave * (WP * 8) + i * 8) * BKB);
      return;
    }
    const int tap = kt / cpt, c0 = (kt - tap * cpt) * BKE;
    const int kh = tap / 3, kw = tap - kh * 3;
    const size_t aoff = (size_t)(kh * TPi + kw) * Cin + c0;
    const size_t woff = (size_t)tap * Cin + c0;
#pragma unroll
    for (int i = 0; i < AP; ++i) cg_dma(a_src[i] + aoff, dst + (wave * AR + i * 8) * BKB);
#pragma unroll
    for (int i = 0; i < WP; ++i) cg_dma(w_src[i] + woff, dst + BM * BKB + (wave * (WP * 8) + i * 8) * BKB);
  };

  f32x4_t acc[FI][FJ];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, lgrp = lane >> 4;
  int roff[2];
  roff[0] = ((0 + lgrp) ^ ((frow >> 1) & 7)) << 4;
  roff[1] = ((4 + lgrp) ^ ((frow >> 1) & 7)) << 4;
  const int a_off = (wr * TM + frow) * BKB;
  const int b_off = BM * BKB + (wc * 64 + frow) * BKB;
  const int nk = nk9 + (SC ? p.Cin2 / BKE : 0), nq = 2 * nk;      // nk >= 9
  uint4 fa[2][FI], fb[2][FJ];
  auto read_slice = [&](auto bufc, int q) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
    const char* st = cg_smem + ((q >> 1) & 1) * STAGE;
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
      for (int j = 0; j < FJ; ++j) cg_mma(fa[buf][i], fb[buf][j], acc[i][j]);
  };
  auto block = [&](auto mbufc, int q) __attribute__((always_inline)) {     // multiply buffer mbuf, read slice q into the other
    constexpr int mb = decltype(mbufc)::value;
    read_slice(std::integral_constant<int, 1 - mb>(), q);
    mma_slice(mbufc);
    constexpr int NR = FI + FJ, NM = FI * FJ, PER = NM / NR >= 2 ? 2 : 1;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    if constexpr (NM - PER * NR > 0) __builtin_amdgcn_sched_group_barrier(0x008, NM - PER * NR, 0);
  };
  auto enter_stage = [&](int kt) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) issue(kt + 1);
  };
  std::integral_constant<int, 0> b0;
  std::integral_constant<int, 1> b1;
  issue(0);
  issue(1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP + WP) : "memory");    // stage 0 = the older group of AP + WP pieces
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
  // (the bias vectors are requested in front of the barrier: behind it they would queue up after the residual vectors, and
  // the first slab, which needs the bias, would wait for the whole ring)
  const int ch0 = n0 + wc * 64 + (lane & 3) * 16;
  float bias_r[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) bias_r[e] = 0.f;
  if (p.bias) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 bq = *(const float4*)(p.bias + ch0 + q * 4);
      bias_r[q * 4 + 0] = bq.x; bias_r[q * 4 + 1] = bq.y; bias_r[q * 4 + 2] = bq.z; bias_r[q * 4 + 3] = bq.w;
    }
  }
  __syncthreads();                                             // every wave is done with the stages: the slabs go there

  // ---- epilogue (as conv_kernel): 16 x 64 slab per wave through LDS so that a lane owns 16 consecutive channels of
  // one pixel; bias + residual + ReLU; 4 lanes write the 128-byte channel run of a pixel
  constexpr int SROW = 64 * 4 + 16;
  char* slab = cg_smem + wave * (16 * SROW);
  const int crow = lgrp * 4, ccol = frow;
  const int orow = lane >> 2, oseg = (lane & 3) * 16;
  // (... and consumed here: where the residual branch joins the straight path hipcc merges the two wait states, and a
  // bias vector that is still "in flight" on the path without a residual costs a wait for the residual vectors on the other)
  asm volatile("" : "+v"(bias_r[0]), "+v"(bias_r[1]), "+v"(bias_r[2]), "+v"(bias_r[3]), "+v"(bias_r[4]), "+v"(bias_r[5]), "+v"(bias_r[6]), "+v"(bias_r[7]),
               "+v"(bias_r[8]), "+v"(bias_r[9]), "+v"(bias_r[10]), "+v"(bias_r[11]), "+v"(bias_r[12]), "+v"(bias_r[13]), "+v"(bias_r[14]), "+v"(bias_r[15]));
  // Round 4: the residual vectors of ALL slabs are requested before the first slab is transposed, and the stores are inline
  // asm (cg_store16).  vmcnt counts loads and stores in one in-order queue: with the residual loaded inside the slab loop,
  // the wait for slab i's residual was also a wait for every store of slab i - 1 (s_waitcnt vmcnt(0) between any two stores
  // of the round-3 code object).  Now the compiler's waitcnt pass sees loads only -- all older than the first store -- and
  // the stores of a tile stream out behind each other.
  // (two copies of the loop, with and without a residual: inside one loop hipcc merges the wait states of the two cases where
  // their branches join, and the counted waits degrade to the smaller count)
  auto epilogue = [&](auto hrc) __attribute__((always_inline)) {
  constexpr bool HR = decltype(hrc)::value;
  // (a ring of RD slabs: the 256-channel tile has 8 slabs per wave and no registers for 16 residual vectors)
  constexpr int RD = FI < 4 ? FI : 4;
  uint4 rpre[RD][2];
  auto pix_of = [&](int i) __attribute__((always_inline)) {
    const int m = min(m0 + wr * TM + i * 16 + orow, M - 1);
    const int b = m / (F * T), rem = m - b * (F * T);
    const int fo = rem / T, to = rem - fo * T;
    return ((size_t)(b * FP + fo + 1) * TP + to + 1) * Cout + ch0;
  };
  auto res_issue = [&](int i) __attribute__((always_inline)) {
    const bf16_t* rp = (const bf16_t*)p.res + pix_of(i);
    rpre[i % RD][0] = *(const uint4*)rp;
    rpre[i % RD][1] = *(const uint4*)(rp + 8);
  };
  if constexpr (HR) {
#pragma unroll
    for (int i = 0; i < RD; ++i) res_issue(i);
  }
#pragma unroll
  for (int i = 0; i < FI; ++i) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < FJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) *(float*)(slab + (crow + r) * SROW + (j * 16 + ccol) * 4) = acc[i][j][r];
    __builtin_amdgcn_wave_barrier();
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 x = *(const float4*)(slab + orow * SROW + (oseg + q * 4) * 4);
      v[q * 4 + 0] = x.x + bias_r[q * 4 + 0]; v[q * 4 + 1] = x.y + bias_r[q * 4 + 1];
      v[q * 4 + 2] = x.z + bias_r[q * 4 + 2]; v[q * 4 + 3] = x.w + bias_r[q * 4 + 3];
    }
    if constexpr (HR) {
      // (pins the use of this slab's vectors here: hipcc otherwise unpacks all of them right behind the loads, i.e. waits
      // for the whole ring before the first slab)
      cg_pin(rpre[i % RD][0], rpre[i % RD][1]);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint4 raw = rpre[i % RD][q];
        const bf16_t* re = (const bf16_t*)&raw;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[q * 8 + e] += bf16_to_f32(re[e]);
      }
    }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (m0 + wr * TM + i * 16 + orow < M) {
      bf16_t* op = (bf16_t*)p.out + pix_of(i);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        bf16_t o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(v[q * 8 + e]);
        cg_store16(op + q * 8, *(const uint4*)o);
      }
    }
    if constexpr (HR) {
      if (i + RD < FI) res_issue(i + RD);
    }
  }
  };
  if (p.res) epilogue(std::true_type()); else epilogue(std::false_type());
}

// ------------------------------------------------------------------------------------------------------------------------
// fp8 form (round 4 CANDIDATE: compiled, not yet run on a GPU).  Same tile (256 pixels x BN channels), same LDS-DMA gather and
// source swizzle; a 128-byte row is now 128 channels of e4m3, so a K step covers a whole tap of the 128-channel stage (half a
// tap of the 256-channel one), and the wave tile is 32 x 32 accumulator blocks fed by v_mfma_scale_f32_32x32x64_f8f6f4 at unit
// block scales -- operand layout, fragment addressing and the slab mapping of the blocks are gemm2_kernel's fp8 path
// (scripts/micro/f8_probe.hip).  Plain loop: one 64/96-KiB stage per K step, the next one in flight under the MFMAs.
// Epilogue: acc * (a_scale * w_scale[n]) + bias (+ bf16 residual) -> ReLU -> bf16 and / or e4m3 output.
typedef __attribute__((ext_vector_type(16))) float cg_f32x16;
typedef __attribute__((ext_vector_type(8))) int cg_i32x8;

template <int BN>
__global__ __launch_bounds__(512) void conv_igemm8_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char cg_smem[];
  constexpr int BKB = 128;                                   // bytes = channels per K step
  constexpr int NWN = BN / 64, NWM = 8 / NWN;
  constexpr int TM = 256 / NWM, MB = TM / 32;                // wave tile TM pixels x 64 channels = MB x 2 blocks
  constexpr int STAGE = (256 + BN) * BKB;
  constexpr int WP = BN / 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / NWN, wc = wave % NWN;
  const int F = p.Fo, T = p.To, Cin = p.Cin, Cout = p.Cout;
  const int TP = T + 2, FP = F + 2;
  const int TPi = p.Ti + 2, FPi = p.Fi + 2, S = p.stride;
  const int M = p.B * F * T;
  const int tiles_n = Cout / BN;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * 256, n0 = tn * BN;
  const char* __restrict__ in = (const char*)p.in8;
  const char* __restrict__ w = (const char*)p.w8;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)cg_smem;

  const int lr = lane >> 3, lc = lane & 7;
  const char* a_src[4];
  const char* w_src[WP];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + lr;
    int m = m0 + row;
    if (m >= M) m = M - 1;
    const int b = m / (F * T), rem = m - b * (F * T);
    const int fo = rem / T, to = rem - fo * T;
    a_src[i] = in + ((size_t)(b * FPi + S * fo) * TPi + S * to) * Cin + (lc ^ ((row >> 1) & 7)) * 16;
  }
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int row = wave * (WP * 8) + i * 8 + lr;
    w_src[i] = w + (size_t)(n0 + row) * 9 * Cin + (lc ^ ((row >> 1) & 7)) * 16;
  }
  const int cpt = Cin / BKB;
  const int nk = 9 * cpt;
  auto issue = [&](int kt) __attribute__((always_inline)) {
    const int tap = kt / cpt, c0 = (kt - tap * cpt) * BKB;
    const int kh = tap / 3, kw = tap - kh * 3;
    const size_t aoff = (size_t)(kh * TPi + kw) * Cin + c0;
    const size_t woff = (size_t)tap * Cin + c0;
    const unsigned dst = lds_base + (kt & 1) * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) cg_dma(a_src[i] + aoff, dst + (wave * 32 + i * 8) * BKB);
#pragma unroll
    for (int i = 0; i < WP; ++i) cg_dma(w_src[i] + woff, dst + 256 * BKB + (wave * (WP * 8) + i * 8) * BKB);
  };

  cg_f32x16 acc[MB][2];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // lane = (row 0..31 of a block, half g); its 32 bytes of a k64 slice are the 16-byte columns 2g, 2g + 1 (swizzled as stored)
  const int frow = lane & 31, g = lane >> 5;
  const int swz = (frow >> 1) & 7;
  int rlo[2], rhi[2];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) { rlo[sl] = ((sl * 4 + 2 * g) ^ swz) << 4; rhi[sl] = ((sl * 4 + 2 * g + 1) ^ swz) << 4; }
  const int a_off = (wr * TM + frow) * BKB;
  const int b_off = 256 * BKB + (wc * 64 + frow) * BKB;

  issue(0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of stage kt
    __syncthreads();                                       // ... everybody's; nobody reads stage kt - 1 any more
    if (kt + 1 < nk) issue(kt + 1);
    const char* st = cg_smem + (kt & 1) * STAGE;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      cg_i32x8 a8[MB], b8[2];
#pragma unroll
      for (int bi = 0; bi < MB; ++bi) {
        const uint4 lo = *(const uint4*)(st + a_off + bi * 32 * BKB + rlo[sl]), hi = *(const uint4*)(st + a_off + bi * 32 * BKB + rhi[sl]);
        a8[bi] = (cg_i32x8){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
      }
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) {
        const uint4 lo = *(const uint4*)(st + b_off + bj * 32 * BKB + rlo[sl]), hi = *(const uint4*)(st + b_off + bj * 32 * BKB + rhi[sl]);
        b8[bj] = (cg_i32x8){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
      }
#pragma unroll
      for (int bi = 0; bi < MB; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
          acc[bi][bj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[bi], b8[bj], acc[bi][bj], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
  }
  const int ch0 = n0 + wc * 64 + (lane & 3) * 16;
  float bias_r[16], sc_r[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 sq = *(const float4*)(p.w8_scale + ch0 + q * 4);
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bq = *(const float4*)(p.bias + ch0 + q * 4);
    sc_r[q * 4 + 0] = sq.x * p.a_scale; sc_r[q * 4 + 1] = sq.y * p.a_scale; sc_r[q * 4 + 2] = sq.z * p.a_scale; sc_r[q * 4 + 3] = sq.w * p.a_scale;
    bias_r[q * 4 + 0] = bq.x; bias_r[q * 4 + 1] = bq.y; bias_r[q * 4 + 2] = bq.z; bias_r[q * 4 + 3] = bq.w;
  }
  __syncthreads();                                             // every wave is done with the stages: the slabs go there

  // ---- epilogue: rows [16 i, 16 i + 16) of the wave tile = half hh = i & 1 of block row bi = i >> 1 (registers 8 hh .. 8 hh + 7:
  // C/D of a 32 x 32 block: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)); through the wave's slab a lane then
  // owns 16 consecutive channels of one pixel
  constexpr int SROW = 64 * 4 + 16;
  char* slab = cg_smem + wave * (16 * SROW);
  const int orow = lane >> 2, oseg = (lane & 3) * 16;
  const int r32 = 4 * (lane >> 5), c32 = lane & 31;
  float vmax = 0.f;
  unsigned nclip = 0;
#pragma unroll
  for (int i = 0; i < TM / 16; ++i) {
    const int bi = i >> 1, hh = i & 1;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int bj = 0; bj < 2; ++bj)
#pragma unroll
      for (int qq = 0; qq < 2; ++qq)
#pragma unroll
        for (int r = 0; r < 4; ++r) *(float*)(slab + (8 * qq + r32 + r) * SROW + (bj * 32 + c32) * 4) = acc[bi][bj][8 * hh + 4 * qq + r];
    __builtin_amdgcn_wave_barrier();
    const int mrow = m0 + wr * TM + i * 16 + orow;
    const int m = min(mrow, M - 1);
    const int b = m / (F * T), rem = m - b * (F * T);
    const int fo = rem / T, to = rem - fo * T;
    const size_t pix = ((size_t)(b * FP + fo + 1) * TP + to + 1) * Cout + ch0;
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 x = *(const float4*)(slab + orow * SROW + (oseg + q * 4) * 4);
      v[q * 4 + 0] = x.x * sc_r[q * 4 + 0] + bias_r[q * 4 + 0]; v[q * 4 + 1] = x.y * sc_r[q * 4 + 1] + bias_r[q * 4 + 1];
      v[q * 4 + 2] = x.z * sc_r[q * 4 + 2] + bias_r[q * 4 + 2]; v[q * 4 + 3] = x.w * sc_r[q * 4 + 3] + bias_r[q * 4 + 3];
    }
    if (p.res) {
      const bf16_t* rp = (const bf16_t*)p.res + pix;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint4 raw = *(const uint4*)(rp + q * 8);
        const bf16_t* re = (const bf16_t*)&raw;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[q * 8 + e] += bf16_to_f32(re[e]);
      }
    }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (mrow >= M) continue;
    if (p.amax8) {
#pragma unroll
      for (int e = 0; e < 16; ++e) vmax = fmaxf(vmax, fabsf(v[e]));
    }
    if (p.out) {
      bf16_t* op = (bf16_t*)p.out + pix;
#pragma unroll
      for (int q = 0; q < 2; ++q)
        *(uint4*)(op + q * 8) = make_uint4(pack2_bf16(v[q * 8 + 0], v[q * 8 + 1]), pack2_bf16(v[q * 8 + 2], v[q * 8 + 3]),
                                           pack2_bf16(v[q * 8 + 4], v[q * 8 + 5]), pack2_bf16(v[q * 8 + 6], v[q * 8 + 7]));
    }
    if (p.out8) {
      const float qs = p.out8_inv_scale;
      if (p.sat8) nclip += fp8_clipped(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs) + fp8_clipped(v[4] * qs, v[5] * qs, v[6] * qs, v[7] * qs) +
                           fp8_clipped(v[8] * qs, v[9] * qs, v[10] * qs, v[11] * qs) + fp8_clipped(v[12] * qs, v[13] * qs, v[14] * qs, v[15] * qs);
      *(uint4*)((char*)p.out8 + pix) = make_uint4(pack4_fp8(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs), pack4_fp8(v[4] * qs, v[5] * qs, v[6] * qs, v[7] * qs),
                                                  pack4_fp8(v[8] * qs, v[9] * qs, v[10] * qs, v[11] * qs), pack4_fp8(v[12] * qs, v[13] * qs, v[14] * qs, v[15] * qs));
    }
  }
  if (p.amax8) {
    vmax = wave_max(vmax);
    if (lane == 0 && vmax > 0.f) atomicMax(p.amax8, __float_as_uint(vmax));
  }
  if (p.sat8 && nclip) atomicAdd(p.sat8, nclip);
}

template <int BN>
int launch_igemm8(hipStream_t st, const ConvArgs& p) {
  const int lds = 2 * (256 + BN) * 128;
  auto kern = conv_igemm8_kernel<BN>;
  static bool attr_set = false;
  if (!attr_set) {
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const int64_t M = (int64_t)p.B * p.Fo * p.To;
  const int64_t tiles = ((M + 255) / 256) * (p.Cout / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds, st, p);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

template <int BN, bool SC, int BM = 256>
int launch_igemm(hipStream_t st, const ConvArgs& p) {
  const int lds = 2 * (BM + BN) * 128;
  auto kern = conv_igemm_kernel<BN, SC, BM>;
  static bool attr_set = false;
  if (!attr_set) {
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const int64_t M = (int64_t)p.B * p.Fo * p.To;
  const int64_t tiles = ((M + BM - 1) / BM) * (p.Cout / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds, st, p);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

// 128-channel stage: 512-pixel tiles (wave tile 128 pixels x 64 channels, the 256-channel tile's) when the launch still fills the
// chip several times over (lab: RVD_IGEMM_BM)
bool igemm_wide_tile(const ConvArgs& p) {
  const char* e = lab_env("RVD_IGEMM_BM");              // lab: 256 = never; 512 = always (tests: small launches too)
  const int bm = e ? atoi(e) : 0;
  if (bm == 256) return false;
  return bm == 512 || (int64_t)p.B * p.Fo * p.To >= (int64_t)512 * 1024;
}

}  // namespace

bool conv_igemm_applicable(int dtype, const ConvArgs& p) {
  if (!(dtype == DT_BF16 && p.w_ig != nullptr && p.taps == 9 && p.Cin % 64 == 0 && p.Cout % 128 == 0 &&
        (int64_t)p.B * p.Fo * p.To < (int64_t)1 << 31)) return false;
  if (p.in2 && (p.Cin2 % 64 || p.res != nullptr || p.Fo != (p.Fi2 - 1) / p.stride2 + 1 || p.To != (p.Ti2 - 1) / p.stride2 + 1)) return false;
  if (p.stride == 1) return p.Fo == p.Fi && p.To == p.Ti;
  return p.stride == 2 && p.Fo == (p.Fi - 1) / 2 + 1 && p.To == (p.Ti - 1) / 2 + 1;      // Conv2d(k 3, stride 2, pad 1)
}

bool conv_igemm_wide(const ConvArgs& p) { return p.Cout % 256 != 0 && !p.in2 && igemm_wide_tile(p); }

bool conv_igemm8_applicable(int dtype, const ConvArgs& p) {
  if (!(dtype == DT_BF16 && p.in8 != nullptr && p.w8 != nullptr && p.w8_scale != nullptr && p.taps == 9 && p.Cin % 128 == 0 &&
        p.Cout % 128 == 0 && p.in2 == nullptr && (p.out != nullptr || p.out8 != nullptr) && (int64_t)p.B * p.Fo * p.To < (int64_t)1 << 31))
    return false;
  if (p.stride == 1) return p.Fo == p.Fi && p.To == p.Ti;
  return p.stride == 2 && p.Fo == (p.Fi - 1) / 2 + 1 && p.To == (p.Ti - 1) / 2 + 1;
}

int conv_igemm8(hipStream_t s, const ConvArgs& p) {
  if (p.B <= 0) return OK;
  return p.Cout % 256 == 0 ? launch_igemm8<256>(s, p) : launch_igemm8<128>(s, p);
}

int conv_igemm(hipStream_t s, const ConvArgs& p) {
  if (p.B <= 0) return OK;
  if (p.Cout % 256 == 0) return p.in2 ? launch_igemm<256, true>(s, p) : launch_igemm<256, false>(s, p);
  if (conv_igemm_wide(p)) return launch_igemm<128, false, 512>(s, p);      // (with the fused shortcut the wide tile spills)
  return p.in2 ? launch_igemm<128, true>(s, p) : launch_igemm<128, false>(s, p);
}

}  // namespace rvb
