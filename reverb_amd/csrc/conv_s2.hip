// The block that opens the 64-channel stage of the ResNet34 trunk (bf16): its stride-2 3x3 convolution (32 -> 64 channels, ReLU) and
// its 1x1 / stride-2 projection shortcut (32 -> 64, no ReLU) read the SAME input tensor -- the largest activation of the network --
// and until round 5 did so in two launches of resnet.hip's direct kernel (10.9 + 5.3 ms per hour of audio, 8 % of the matrix pipe busy,
// 3.5 TB/s).  Here one kernel reads the input once and writes both outputs:
//
//     y  = relu(conv3x3_s2(x) + b)        [B][Fo+2][To+2][64]
//     sc = conv1x1_s2(x) + b_sc           [B][Fo+2][To+2][64]      (the residual of the block's second convolution)
//
// The work is 41 kFLOP per output pixel against 64 B x 4 read + 256 B written: bound by HBM (35 GB per hour of audio, 7 ms at 5 TB/s),
// so the structure is conv_row64.hip's (4 waves, two patch buffers, one barrier per tile, two workgroups per CU, hidden stores) with
// what stride 2 needs:
//   * the patch is stored DE-INTERLEAVED: even and odd input columns in separate planes ([row][parity][32 pixels][64 B]), written that
//     way by the LDS-DMA (a piece's 16 source pixels are every second column).  Output pixel o then finds its taps at E[o], O[o],
//     E[o + 1] -- consecutive pixels of a plane for consecutive lanes, the access pattern (and the chunk swizzle) of the stride-1
//     kernels; read in place, the 128-byte stride between the pixels of neighbouring lanes is an 8-way bank conflict;
//   * a tile is 4 output rows x 31 output frames (E needs one pixel more than there are outputs: 32 per plane = two DMA pieces), its
//     patch 9 input rows; wave w = channel half w & 1 x column strip w >> 1 slides down the 9 rows: even row 2 r is tap row 0 of output
//     row r and tap row 2 of output row r - 1, odd row 2 r + 1 is tap row 1 of output row r and, at its centre column, the ONE tap of
//     the shortcut.  27 fragment reads per tile and wave.
// Results: every output accumulates its taps in the order 0 .. 8 (one 32-channel MFMA each), rounding points as conv_kernel's:
// bit-identical with the two launches (tests/test_diar_gpu.py: test_stride2_opener_equals_the_two_launches).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace rvb {

namespace {

constexpr int CS_OT = 31, CS_PP = 32, CS_PR = 9, CS_OF = 4;
constexpr int CS_PLANE = CS_PP * 64;                // 2 048 B
constexpr int CS_ROW = 2 * CS_PLANE;                // 4 096 B
constexpr int CS_BUF = CS_PR * CS_ROW + 128;
constexpr int CS_LDS = 2 * CS_BUF;

typedef unsigned cs_u32x4 __attribute__((ext_vector_type(4)));

__device__ inline void cs_mma(const uint4& w, const uint4& x, f32x4_t& c) {
  union U { uint4 u; bf16x8_t v; };
  U uw, ux;
  uw.u = w; ux.u = x;
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uw.v, ux.v, c, 0, 0, 0);
}
__device__ inline unsigned cs_swz(unsigned g, unsigned c) { return g * 64u + ((c ^ (((g >> 2) & 1u) << 1)) << 4); }
// (conv_row64.hip: cr_store16)
__device__ inline void cs_store16(void* q, const uint4& v) {
  const cs_u32x4 d = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(q), "v"(d) : "memory");
}
__device__ inline void cs_dma1(unsigned off, const void* sbase, unsigned lds0) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(off), "s"(sbase), "s"(lds0)
      : "memory");
}
__device__ inline const char* cs_uniform(const char* q) {
  const unsigned long long v = (unsigned long long)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}

__global__ __launch_bounds__(256, 2) void conv_s2sc_kernel(ConvS2Args p) {
  extern __shared__ __attribute__((aligned(16))) char cs_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int FPi = p.Fi + 2, TPi = p.Ti + 2, FPo = p.Fo + 2, TPo = p.To + 2;
  const int tiles_f = (p.Fo + CS_OF - 1) / CS_OF, tiles_t = (p.To + CS_OT - 1) / CS_OT;
  int lin;
  {
    const int nblk = (int)gridDim.x, q = nblk >> 3, r = nblk & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tf = lin % tiles_f;
  const int b = lin / tiles_f;
  const int f0 = tf * CS_OF;
  const char* in_b = cs_uniform((const char*)p.in + (size_t)b * FPi * TPi * 64);
  char* out_b = (char*)p.out + (size_t)b * FPo * TPo * 128;
  char* sc_b = (char*)p.sc + (size_t)b * FPo * TPo * 128;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)cs_smem;

  // 36 pieces per tile (9 rows x 2 parities x 2 groups of 16 pixels), 9 per wave: piece q = 9 w + i -> row q >> 2, parity (q >> 1) & 1,
  // group q & 1; bordered input row 2 f0 + row, bordered input column 2 (t0 + 16 group + pixel) + parity
  const int ppx = lane >> 2;
  const unsigned piece_b = (unsigned)((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 16;
  auto issue = [&](int tt) __attribute__((always_inline)) {
    const int t0 = tt * CS_OT;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int q = wave * 9 + i, row = q >> 2, par = (q >> 1) & 1, g = q & 1;
      const unsigned rowterm = (unsigned)(min(2 * f0 + row, FPi - 1) * TPi) * 64u;
      const unsigned colterm = (unsigned)min(2 * (t0 + g * 16 + ppx) + par, TPi - 1) * 64u + piece_b;
      cs_dma1(rowterm + colterm, in_b,
              __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(tt & 1) * CS_BUF + (unsigned)row * CS_ROW + (unsigned)par * CS_PLANE + (unsigned)g * 1024));
    }
  };
  issue(0);
  const int half = wave & 1, col = wave >> 1;
  uint4 w[9][2], wsc[2];
  float bias_r[8], bsc_r[8];
  {
    const char* gw = (const char*)p.w;
    const char* gs = (const char*)p.wsc;
    const int ch0 = half * 32 + (li >> 2) * 8 + (li & 3);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int j = 0; j < 2; ++j) w[tap][j] = *(const uint4*)(gw + (tap * 64 + ch0 + j * 4) * 64 + lg * 16);
#pragma unroll
    for (int j = 0; j < 2; ++j) wsc[j] = *(const uint4*)(gs + (ch0 + j * 4) * 64 + lg * 16);
    const float4 b0 = *(const float4*)(p.bias + half * 32 + lg * 8), b1 = *(const float4*)(p.bias + half * 32 + lg * 8 + 4);
    const float4 s0 = *(const float4*)(p.bsc + half * 32 + lg * 8), s1 = *(const float4*)(p.bsc + half * 32 + lg * 8 + 4);
    bias_r[0] = b0.x; bias_r[1] = b0.y; bias_r[2] = b0.z; bias_r[3] = b0.w; bias_r[4] = b1.x; bias_r[5] = b1.y; bias_r[6] = b1.z; bias_r[7] = b1.w;
    bsc_r[0] = s0.x; bsc_r[1] = s0.y; bsc_r[2] = s0.z; bsc_r[3] = s0.w; bsc_r[4] = s1.x; bsc_r[5] = s1.y; bsc_r[6] = s1.z; bsc_r[7] = s1.w;
  }
  // tap column 0 = E[o], 1 = O[o], 2 = E[o + 1]
  unsigned aoff[3];
  aoff[0] = cs_swz((unsigned)(col * 16 + li), lg);
  aoff[1] = aoff[0] + CS_PLANE;
  aoff[2] = cs_swz((unsigned)(col * 16 + li + 1), lg);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  // (the weights are "used" here, so that the compiler carries no pending load of its own into the loop: its waits for one would
  // also wait for the LDS-DMA pieces and the stores it cannot see)
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    cs_u32x4 x = {w[tap][0].x, w[tap][0].y, w[tap][0].z, w[tap][0].w}, y = {w[tap][1].x, w[tap][1].y, w[tap][1].z, w[tap][1].w};
    asm volatile("" : "+v"(x), "+v"(y));
    w[tap][0] = make_uint4(x[0], x[1], x[2], x[3]); w[tap][1] = make_uint4(y[0], y[1], y[2], y[3]);
  }
  {
    cs_u32x4 x = {wsc[0].x, wsc[0].y, wsc[0].z, wsc[0].w}, y = {wsc[1].x, wsc[1].y, wsc[1].z, wsc[1].w};
    asm volatile("" : "+v"(x), "+v"(y));
    wsc[0] = make_uint4(x[0], x[1], x[2], x[3]); wsc[1] = make_uint4(y[0], y[1], y[2], y[3]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(bias_r[e]), "+v"(bsc_r[e]));
  __syncthreads();
  const int o = col * 16 + li;

  for (int tt = 0; tt < tiles_t; ++tt) {
    const int t0 = tt * CS_OT;
    if (tt + 1 < tiles_t) issue(tt + 1);       // into the other buffer: its readers (tile tt - 1) are behind the last barrier
    const char* img = cs_smem + (tt & 1) * CS_BUF;
    f32x4_t acc[CS_OF][2], accs[CS_OF][2];
    uint4 xf[2][3];
    auto read_row = [&](int i, int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) xf[buf][kw] = *(const uint4*)(img + aoff[kw] + i * CS_ROW);
    };
    read_row(0, 0);
#pragma unroll
    for (int i = 0; i < CS_PR; ++i) {
      const int cur = i & 1;
      if (i + 1 < CS_PR) read_row(i + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      if ((i & 1) == 0) {
        const int r = i >> 1;                  // tap row 2 of output row r - 1, tap row 0 of output row r
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          if (r >= 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j) cs_mma(w[6 + kw][j], xf[cur][kw], acc[r - 1][j]);
          }
          if (r < CS_OF) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              if (kw == 0) acc[r][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
              cs_mma(w[kw][j], xf[cur][kw], acc[r][j]);
            }
          }
        }
      } else {
        const int r = i >> 1;                  // tap row 1 of output row r; its centre column is the shortcut's tap
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int j = 0; j < 2; ++j) cs_mma(w[3 + kw][j], xf[cur][kw], acc[r][j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          accs[r][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
          cs_mma(wsc[j], xf[cur][1], accs[r][j]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // this wave's pieces of the next patch (requested a whole tile of MFMAs ago) have landed, and the stores of the previous tile
    // with them (older in the same queue); the stores below are younger than this wait: nothing ever waits for them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int t = t0 + o;
    const bool t_ok = o < CS_OT && t < p.To;
#pragma unroll
    for (int r = 0; r < CS_OF; ++r) {
      float v[8], s[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = fmaxf(acc[r][e >> 2][e & 3] + bias_r[e], 0.f);
        s[e] = accs[r][e >> 2][e & 3] + bsc_r[e];
      }
      const int fo = f0 + r;
      if (t_ok && fo < p.Fo) {
        const size_t at = ((size_t)((unsigned)(fo + 1) * TPo + t + 1) * 64 + half * 32 + lg * 8) * 2;
        cs_store16(out_b + at, make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])));
        cs_store16(sc_b + at, make_uint4(pack2_bf16(s[0], s[1]), pack2_bf16(s[2], s[3]), pack2_bf16(s[4], s[5]), pack2_bf16(s[6], s[7])));
      }
    }
    __builtin_amdgcn_s_barrier();              // every wave's pieces of the next patch are in; this tile's buffer is free
    asm volatile("" ::: "memory");
  }
}

}  // namespace

bool conv_s2sc_applicable(int dtype, int cin, int cout, int stride, int taps, int sc_cin, int sc_cout, int sc_stride, int sc_taps,
                          int Fi, int Ti, int Fo, int To) {
  const char* e = lab_env("RVD_CONV_S2SC");           // lab: 0 = two launches of resnet.hip's direct kernel (until round 5)
  if (e && atoi(e) == 0) return false;
  return dtype == DT_BF16 && cin == 32 && cout == 64 && stride == 2 && taps == 9 && sc_cin == 32 && sc_cout == 64 && sc_stride == 2 && sc_taps == 1 &&
         Fo == (Fi - 1) / 2 + 1 && To == (Ti - 1) / 2 + 1 && (int64_t)(Fi + 2) * (Ti + 2) * 64 < ((int64_t)1 << 31);
}

int conv_s2sc(hipStream_t s, const ConvS2Args& a) {
  if (a.B <= 0) return OK;
  static bool attr_set = false;
  if (!attr_set) {
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)conv_s2sc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CS_LDS));
    attr_set = true;
  }
  const int64_t blocks = (int64_t)a.B * cdiv(a.Fo, CS_OF);
  if (blocks >= ((int64_t)1 << 31)) { set_error("conv_s2sc: too many workgroups"); return E_ARG; }
  hipLaunchKernelGGL(conv_s2sc_kernel, dim3((unsigned)blocks), dim3(256), CS_LDS, s, a);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace rvb
