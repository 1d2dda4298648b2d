// The stride-1 3x3 convolutions of the ResNet34 trunk's 128- and 256-channel stages (bf16) over the FLAT pixel index of the bordered
// tensor, with the input pixels of a tile resident in LDS for all nine taps.
//
// Why (round 5): the implicit GEMM of conv_gemm.hip gathers the A operand of every K step from the L2 -- the same pixels nine times,
// once per tap, 64 of the 80 KiB of a stage -- and its loop is paced by those LDS-DMA requests (10 one-KiB pieces per wave for 64
// MFMAs; a piece costs 60-180 cycles of issue beside MFMAs, MI355X_MICROARCH.md).  The bordered NHWC layout offers better: in the
// flat pixel index m = (b (F + 2) + f) (T + 2) + t of the bordered tensor, tap (kh, kw) of output pixel m is input pixel
// m + (kh - 1)(T + 2) + (kw - 1), for EVERY m -- a constant shift.  So a tile of 512 consecutive m needs the 512 + 2 (T + 2) + 2
// consecutive input pixels around it ONCE per 32-channel chunk, and the nine taps are nine row offsets into that LDS image.
//   + LDS-DMA per 96 MFMAs of a wave: 3 pieces of weights + 2 of pixels (was 15);
//   - the border positions are computed like any other pixel and not stored: (F + 2)(T + 2) / (F T) = 1.12 x the MFMAs for the
//     20 x 125 maps of the 128-channel stage, 1.24 x for the 10 x 63 maps of the 256-channel stage.
//
// Structure: persistent workgroups of 8 waves (4 along pixels x 2 along channels, wave tile 128 pixels x 64 channels = 8 x 4
// accumulator fragments), tile = 512 pixels x 128 output channels.  A STAGE is one tap row (3 taps) of one 32-channel chunk:
// 24 KiB of weights ([chunk][tap][128 rows][64 B], packed at load in the order the fragments want them) in one of two buffers; the
// chunk's 768 x 64 B of pixels in one of two buffers.  At the top of a stage the NEXT stage's weights are requested -- and, in the
// first stage of a chunk, the next chunk's pixels (of the next tile, if this was the tile's last chunk: the sequence of stages runs
// across tiles, only the epilogue is exposed); one barrier per stage.  The pixel fragments of a stage's first tap are read during
// the previous stage's last tap (pixels have been visible since that chunk's second stage).
// Operand roles as in conv_block.hip: A = weights, B = pixels, and the weight rows ordered so that a lane's accumulators of an
// n-tile pair are 8 consecutive channels of one pixel = one 16-byte store straight from registers; both 64-byte-row images carry the
// chunk swizzle of conv_block.hip (chunk ^= 2 for rows 4-7 mod 8, applied on the LDS-DMA's source side).
//
// Accumulation order: chunks outer, taps inner (conv_gemm.hip: taps outer) -- not bit-identical with it; tests/test_diar_gpu.py
// compares the two and the fp32 oracle (test_flat_convolutions_against_the_implicit_gemm).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace rvb {

namespace {

constexpr int FL_BM = 512, FL_BN = 128;
constexpr int FL_PPX = 768;                         // pixels of a chunk image: 512 + 2 (T + 2) + 2 <= 768
constexpr int FL_PBUF = FL_PPX * 64;                // 49 152 B
constexpr int FL_WSTAGE = 3 * FL_BN * 64;           // 24 576 B
constexpr int FL_OFF_W = 2 * FL_PBUF;
constexpr int FL_LDS = FL_OFF_W + 2 * FL_WSTAGE;    // 147 456 B

typedef unsigned fl_u32x4 __attribute__((ext_vector_type(4)));

__device__ inline void fl_mma(const uint4& w, const uint4& x, f32x4_t& c) {
  union U { uint4 u; bf16x8_t v; };
  U uw, ux;
  uw.u = w; ux.u = x;
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uw.v, ux.v, c, 0, 0, 0);
}
__device__ inline unsigned fl_swz(unsigned g, unsigned c) { return g * 64u + ((c ^ (((g >> 2) & 1u) << 1)) << 4); }
__device__ inline void fl_store16(void* q, const uint4& v) {
  const fl_u32x4 d = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(q), "v"(d) : "memory");
}
__device__ inline void fl_dma1(unsigned off, const void* sbase, unsigned lds0) {
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
// three consecutive 1-KiB pieces from a scalar base: lane offset `off`, LDS and source advance by 1 KiB per piece
__device__ inline void fl_dma3(unsigned off, const void* sbase, unsigned lds0) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(off), "s"(sbase), "s"(lds0)
      : "memory", "scc");
}
__device__ inline const char* fl_uniform(const char* q) {
  const unsigned long long v = (unsigned long long)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}
template <int N> __device__ inline void fl_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool RES, bool RELU>
__global__ __launch_bounds__(512) void conv_flat_kernel(ConvArgs p, int tiles_total, int tiles_per_xcd) {
  extern __shared__ __attribute__((aligned(16))) char fl_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int F = p.Fo, T = p.To, TP = T + 2, FP = F + 2, Cin = p.Cin, Cout = p.Cout;
  const int Mtot = p.B * FP * TP;
  const int nch = Cin >> 5, tiles_n = Cout / FL_BN;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)fl_smem;
  const char* in_u = fl_uniform((const char*)p.in);
  const char* w_u = fl_uniform((const char*)p.w_fl);

  // this workgroup's tiles: XCD x owns the tiles [x per, (x + 1) per); its workgroups take them round robin, so that the
  // workgroups of an XCD work on neighbouring tiles (which share a third of their pixels) at any time
  const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, wper = (int)gridDim.x >> 3;
  const int t_end = min((xcd + 1) * tiles_per_xcd, tiles_total);
  int tile = xcd * tiles_per_xcd + widx;
  if (tile >= t_end) return;

  // ---- LDS-DMA sources.  Weights: a stage is 24 contiguous KiB, wave w brings pieces 3 w .. 3 w + 2 (16 rows each); the chunk
  // swizzle of row r depends on bit 2 of r = bit 4 of the lane.  Pixels: 48 pieces per chunk, 6 per wave.
  const unsigned piece_b = (unsigned)((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 16;
  const unsigned w_lane = (unsigned)(wave * 3 * 1024) + (unsigned)(lane >> 2) * 64u + piece_b;
  auto issue_w = [&](int tl, int q, int par) __attribute__((always_inline)) {
    const int tn = tl % tiles_n;
    const char* src = w_u + ((size_t)(tn * nch * 3 + q)) * FL_WSTAGE;
    fl_dma3(w_lane, src, __builtin_amdgcn_readfirstlane(lds_base + FL_OFF_W + (unsigned)par * FL_WSTAGE + (unsigned)(wave * 3 * 1024)));
  };
  auto issue_px = [&](int tl, int c, int par) __attribute__((always_inline)) {
    const int m0 = (tl / tiles_n) * FL_BM;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int row = (wave * 6 + k) * 16 + (lane >> 2);
      const int px = min(max(m0 - TP - 1 + row, 0), Mtot - 1);
      const unsigned off = (unsigned)px * (unsigned)(Cin * 2) + (unsigned)c * 64u + piece_b;
      fl_dma1(off, in_u, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)par * FL_PBUF + (unsigned)((wave * 6 + k) * 1024)));
    }
  };

  // ---- fragment addresses
  const unsigned wfrag = FL_OFF_W + fl_swz((unsigned)(wc * 64 + li), lg);      // + buffer, + tap * 8192, + n-tile * 1024
  int prow = wr * 128 + li;                                                    // + kh TP + kw: the pixel row of m-tile 0 at a tap
  // Fragments: the 4 weight fragments of a tap (double-buffered per tap) and the pixel fragments of HALF a tap (4 m-tiles,
  // double-buffered per half): 64 registers beside the 128 of the accumulators (whole taps double-buffered: 96, and hipcc spilled)
  uint4 fa[2][4], fb[2][4];
  f32x4_t acc[8][4];
  auto read_px = [&](int kh, int kw, int h, int ppar, int buf) __attribute__((always_inline)) {
    const unsigned g = (unsigned)(prow + kh * TP + kw);
    const char* q = fl_smem + ((unsigned)ppar * FL_PBUF + fl_swz(g, lg));
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[buf][i] = *(const uint4*)(q + (h * 4 + i) * 1024);
  };
  auto read_w = [&](int tapi, int wpar, int buf) __attribute__((always_inline)) {
    const char* q = fl_smem + (wfrag + (unsigned)wpar * FL_WSTAGE);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[buf][j] = *(const uint4*)(q + tapi * 8192 + j * 1024);
  };

  // ---- prologue: the first tile's first chunk and first stage
  issue_px(tile, 0, 0);
  issue_w(tile, 0, 0);
  fl_wait_vm<0>();
  __syncthreads();
  read_px(0, 0, 0, 0, 0);

  for (;;) {
    const int next_tile = tile + wper;
    const bool has_next = next_tile < t_end;
    const int nt = has_next ? next_tile : tile;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // Two chunks = six stages per trip: a stage has three taps, so the weight fragment registers a stage starts in alternate from
    // stage to stage -- with six per trip every buffer index below (fragments, weight buffer, pixel buffer) is a constant.
    for (int cp = 0; cp < nch; cp += 2) {
      const bool last_pair = cp + 2 == nch;
      asm volatile("" : "+v"(prow));            // (keeps the tap addresses from being hoisted out of the loop: 18 live registers)
#pragma unroll
      for (int sidx = 0; sidx < 6; ++sidx) {
        // top of stage (c, s): everybody is past the barrier that ended the previous stage.  No branches in here: the requests
        // behind a launch's very last stage re-fetch something nobody reads (with branches around the requests hipcc spreads the
        // stage over basic blocks, sinks the MFMAs below the barriers and hoists every fragment read of a stage to its top: 500
        // spilled registers)
        const int cc = sidx / 3, s = sidx - cc * 3, c = cp + cc;
        const int wpar = sidx & 1, ppar = cc, f0 = sidx & 1;        // this stage's weight / pixel buffer, its first tap's weight registers
        const bool tile_ends = last_pair && cc == 1;                // this chunk is the tile's last
        {
          const bool wrap = tile_ends && s == 2;
          issue_w(wrap ? nt : tile, wrap ? 0 : c * 3 + s + 1, wpar ^ 1);
        }
        if (s == 0) issue_px(tile_ends ? nt : tile, tile_ends ? 0 : c + 1, ppar ^ 1);
        read_w(0, wpar, f0);
#pragma unroll
        for (int hh = 0; hh < 6; ++hh) {
          const int kw = hh >> 1, h = hh & 1, curp = hh & 1, curw = f0 ^ (kw & 1);
          // the pixel fragments of the next half tap; behind the stage's last one: those of the next stage's first (its weights
          // come after the barrier) -- the next tap row of this chunk, or tap row 0 of the next chunk / tile, whose pixels have
          // been visible since this chunk's second stage
          if (hh < 5) read_px(s, h ? kw + 1 : kw, h ^ 1, ppar, curp ^ 1);
          else if (s < 2) read_px(s + 1, 0, 0, ppar, curp ^ 1);
          else read_px(0, 0, 0, ppar ^ 1, curp ^ 1);
          if (h == 0 && kw < 2) read_w(kw + 1, wpar, curw ^ 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) fl_mma(fb[curw][j], fa[curp][i], acc[h * 4 + i][j]);
          __builtin_amdgcn_sched_barrier(0);
        }
        // (pins the stage's MFMAs in front of its barrier)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));
        // this wave's pieces of the next stage's weights have landed (first stage of a chunk: the 6 pixel pieces requested AFTER
        // them may still be under way; they are waited for at the end of the chunk's second stage)
        if (s == 0) fl_wait_vm<6>(); else fl_wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }

    // ---- epilogue: a lane holds channels [n0 + 64 wc + 32 P + 8 lg, + 8) of pixel m0 + 128 wr + 16 i + li for P = 0, 1
    {
      const int tn = tile % tiles_n, m0 = (tile / tiles_n) * FL_BM;
      const int chb = tn * FL_BN + wc * 64 + lg * 8;
      float bias_r[2][8];
#pragma unroll
      for (int P = 0; P < 2; ++P) {
        const float4 b0 = *(const float4*)(p.bias + chb + P * 32), b1 = *(const float4*)(p.bias + chb + P * 32 + 4);
        bias_r[P][0] = b0.x; bias_r[P][1] = b0.y; bias_r[P][2] = b0.z; bias_r[P][3] = b0.w;
        bias_r[P][4] = b1.x; bias_r[P][5] = b1.y; bias_r[P][6] = b1.z; bias_r[P][7] = b1.w;
      }
      const unsigned plane = (unsigned)(FP * TP);
      // (all residual vectors are requested before the first store: the compiler's counted waits for them see loads only, and the
      // stores -- inline asm, younger than every load -- stream out behind each other)
      fl_u32x4 rp[8][2];
      if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int m = min(m0 + wr * 128 + i * 16 + li, Mtot - 1);
          const size_t atc = ((size_t)m * Cout + chb) * 2;
#pragma unroll
          for (int P = 0; P < 2; ++P) rp[i][P] = *(const fl_u32x4*)((const char*)p.res + atc + P * 64);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = m0 + wr * 128 + i * 16 + li;
        const unsigned rem = (unsigned)m % plane;
        const unsigned fb_ = rem / (unsigned)TP, tb_ = rem - fb_ * (unsigned)TP;
        const bool ok = m < Mtot && fb_ >= 1u && fb_ <= (unsigned)F && tb_ >= 1u && tb_ <= (unsigned)T;
        const size_t at = ((size_t)m * Cout + chb) * 2;
#pragma unroll
        for (int P = 0; P < 2; ++P) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = acc[i][2 * P + (e >> 2)][e & 3] + bias_r[P][e];
          if constexpr (RES) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * e] += __uint_as_float(rp[i][P][e] << 16);
              v[2 * e + 1] += __uint_as_float(rp[i][P][e] & 0xffff0000u);
            }
          }
          if constexpr (RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (ok)
            fl_store16((char*)p.out + at + P * 64,
                       make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])));
        }
      }
    }
    if (!has_next) break;
    tile = next_tile;
  }
}

}  // namespace

bool conv_flat_applicable(int dtype, const ConvArgs& p) {
  const char* e = lab_env("RVD_CONV_FLAT");           // lab: 0 = conv_gemm.hip's implicit GEMM for these too (until round 5); 2 = small launches too (tests)
  const int mode = e ? atoi(e) : 1;
  if (mode == 0) return false;
  return dtype == DT_BF16 && p.w_fl != nullptr && p.taps == 9 && p.stride == 1 && p.Cin % 64 == 0 && p.Cout % FL_BN == 0 &&
         p.Fo == p.Fi && p.To == p.Ti && !p.in2 && !p.in8 && 2 * (p.To + 2) + 2 + FL_BM <= FL_PPX &&
         (int64_t)p.B * (p.Fo + 2) * (p.To + 2) * std::max(p.Cin, p.Cout) * 2 < ((int64_t)1 << 32) &&
         (mode == 2 || (int64_t)p.B * (p.Fo + 2) * (p.To + 2) >= (int64_t)256 * 1024);
}

template <bool RES, bool RELU>
static int launch_flat(hipStream_t s, const ConvArgs& p) {
  static bool attr_set = false;
  if (!attr_set) {
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)conv_flat_kernel<RES, RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, FL_LDS));
    attr_set = true;
  }
  const int64_t M = (int64_t)p.B * (p.Fo + 2) * (p.To + 2);
  const int tiles = (int)(cdiv(M, (int64_t)FL_BM) * (p.Cout / FL_BN));
  const int per_xcd = cdiv(tiles, 8);
  const int wgs = 8 * std::min(32, per_xcd);           // one workgroup per CU (147 KB of LDS), 32 CUs per XCD
  hipLaunchKernelGGL((conv_flat_kernel<RES, RELU>), dim3(wgs), dim3(512), FL_LDS, s, p, tiles, per_xcd);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

int conv_flat(hipStream_t s, const ConvArgs& p) {
  if (p.B <= 0) return OK;
  if (p.res) return p.relu ? launch_flat<true, true>(s, p) : launch_flat<true, false>(s, p);
  return p.relu ? launch_flat<false, true>(s, p) : launch_flat<false, false>(s, p);
}

}  // namespace rvb
