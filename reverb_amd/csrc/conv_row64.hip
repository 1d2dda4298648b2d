// The stride-1 3x3 convolutions of the ResNet34 trunk's 64-channel stage (bf16), built from what conv_block.hip's history showed
// for the 32-channel stage (round 5): these kernels are bound by their LDS operations and by what overlaps what on a CU, not by
// the matrix pipe.  resnet.hip's conv_kernel<64> (the default until now) spent 36 % of its LDS-active cycles in bank conflicts and
// ran at 3.2 TB/s of HBM traffic with the waves 30 % parked / 37 % stalled at issue (profiles/archive/r05_call5_*.txt, call 6); the
// streamed form of round 4 (conv_stream.hip, 64 channels: one workgroup per CU) measured equal to it.
//
//   out = relu?(conv3x3(x) + b [+ res])        x, res, out: bordered NHWC [B][F+2][T+2][64], weights [tap][chunk][64][64 B]
//
// Ingredients (see conv_block.hip for the measurements behind each):
//   * A = weights (M = channels), B = pixels (N = pixels), and the weight rows a lane supplies chosen so that its accumulators are
//     8 consecutive channels of one pixel: the 16 bytes it stores -- no transposition through LDS;
//   * a wave owns ONE HALF of the output channels (2 n-tiles) and keeps that half's weights in registers for the whole walk:
//     9 taps x 2 chunks x 2 n-tiles x 16 B = 144 VGPRs; LDS holds pixels only;
//   * the 16-byte chunks of a pixel's 64-byte channel chunk are XOR-swizzled with bit 2 of the pixel index (source side of the
//     LDS-DMA), so a ds_read_b128 of 16 consecutive pixels covers all 64 banks once;
//   * 4 waves per workgroup (49 KB of LDS, 2 waves per SIMD): TWO workgroups per CU fill each other's barriers and waits.
//
// Geometry: a workgroup owns 4 output rows of one window and walks tiles of 30 frames (patch 6 rows x 32 pixels, stored as two
// chunk planes [chunk][row][pixel][64 B] of 12 KiB; TWO patch buffers = 49 KB of LDS).  Wave w = channel half w & 1 x the 16-pixel
// column strip w >> 1, all FOUR output rows; it slides down the six patch rows once per channel chunk.  The fragment of input row i
// at column shift kw is the operand of tap (kh, kw) of output row i - kh for kh = 0, 1, 2: read once, used for up to three rows --
// 2 x 18 fragment reads per tile and wave for 144 MFMAs.  Every pixel fragment is read by two waves (the two channel halves).
// Iteration: LDS-DMA of the NEXT tile's patch into the other buffer; residual loads; the two slides; when the first output row has
// its last tap: `s_waitcnt vmcnt(0)` (this wave's pieces of the next patch and the residual vectors, both requested most of a tile of
// MFMAs ago); the epilogue of row r between the MFMAs of the rows still open, with stores that are younger than that wait and
// hidden from the compiler's, so nothing ever waits for them; ONE barrier.
//
// Measured (same box A/B, profiles/archive/r05_call10_conv_row64.txt, r05_call14_*, r05_call15_*): the 64-channel stage 51.6-52.5 ms per hour
// of audio on the direct kernel -> 43.3-43.8 ms in this kernel's first form (62-frame tiles, ONE patch buffer: barrier, DMA, wait,
// barrier at the end of every tile, covered only by the CU's other workgroup) -> 39.1 ms with 30-frame tiles and two buffers (+3 %
// MFMA work for the narrower tiles, the DMA latency under the tile's own MFMAs; wave = channel half x two rows, a fragment read per
// tap: 72 reads per tile and wave) -> **37.0-38.0 ms** with the column strips (half the LDS reads: a small gain -- not what bounds it).
//
// Results: accumulation order (chunks outer, taps inner: rows arrive in order, so every output still sees taps 0 .. 8 of chunk 0, then of chunk 1), operand values and rounding points are conv_kernel's: bit-identical
// (tests/test_diar_gpu.py: test_row64_convolutions_equal_the_direct_kernel).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace rvb {

namespace {

constexpr int CR_PF = 6, CR_OF = 4, CR_NT = 64;

typedef unsigned cr_u32x4 __attribute__((ext_vector_type(4)));

__device__ inline void cr_mma(const uint4& w, const uint4& x, f32x4_t& c) {
  union U { uint4 u; bf16x8_t v; };
  U uw, ux;
  uw.u = w; ux.u = x;
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uw.v, ux.v, c, 0, 0, 0);
}
__device__ inline unsigned cr_swz(unsigned g, unsigned c) { return g * 64u + ((c ^ (((g >> 2) & 1u) << 1)) << 4); }

// one 16-byte store the compiler's waitcnt pass does not see: it waits for a store (vmcnt is one queue for loads and stores)
// before the next m-tile's epilogue touches the registers the store read, i.e. it would drain every store one by one.  s_nop 1:
// the TWO wait states gfx950 wants between a store of more than 8 bytes and a VALU write of its data registers
// (scripts/micro/store_hazard.hip).  Sound with the compiler's counting of the residual loads: hidden stores are OLDER than the
// loads it waits for, so they only make such a wait cover more than it thinks.
__device__ inline void cr_store16(void* q, const uint4& v) {
  const cr_u32x4 d = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(q), "v"(d) : "memory");
}
__device__ inline void cr_wait_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
__device__ inline void cr_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ inline const char* cr_uniform(const char* q) {
  const unsigned long long v = (unsigned long long)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}

// RES / RELU are template parameters: with `if (p.res)` blocks in the body the compiler re-waits for the residual loads inside each
// of them, and each such wait (counted against ITS loads only) also drains the hidden stores issued in between
constexpr int CN_OT = 30, CN_PT = 32;
constexpr int CN_ROW = CN_PT * 64;                  // 2 048 B
constexpr int CN_PLANE = CR_PF * CN_ROW;            // 12 288 B
constexpr int CN_BUF = 2 * CN_PLANE + 128;          // both chunk planes of a tile + the spill pad
constexpr int CN_LDS = 2 * CN_BUF;

// two 1-KiB LDS-DMA pieces: one patch row of one chunk plane (2 KiB), 16 pixels per piece
__device__ inline void cr_dma2(const unsigned (&off)[2], const void* sbase, unsigned lds0) {
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
      : "v"(off[0]), "v"(off[1]), "s"(sbase), "s"(lds0)
      : "memory", "scc");
}

template <bool RES, bool RELU>
__global__ __launch_bounds__(256, 2) void conv_row64_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char cr_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int FP = p.Fi + 2, TP = p.Ti + 2;
  const int tiles_f = (p.Fo + CR_OF - 1) / CR_OF, tiles_t = (p.To + CN_OT - 1) / CN_OT;
  int lin;
  {
    const int nblk = (int)gridDim.x, q = nblk >> 3, r = nblk & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tf = lin % tiles_f;
  const int b = lin / tiles_f;
  const int f0 = tf * CR_OF;
  const char* in_b = cr_uniform((const char*)p.in + (size_t)b * FP * TP * CR_NT * 2);
  const char* res_b = (const char*)p.res + (size_t)b * FP * TP * CR_NT * 2;
  char* out_b = (char*)p.out + (size_t)b * FP * TP * CR_NT * 2;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)cr_smem;

  unsigned lineoff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int q = wave * 3 + i;
    lineoff[i] = (unsigned)(min(f0 + (q >> 1), FP - 1) * TP) * (CR_NT * 2) + (unsigned)(q & 1) * 64;
  }
  const int ppx = lane >> 2;
  const unsigned piece_b = (unsigned)((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 16;
  auto issue = [&](int tt) __attribute__((always_inline)) {          // 6 pieces per wave
    const int t0 = tt * CN_OT;
    unsigned px[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) px[g] = (unsigned)min(t0 + g * 16 + ppx, TP - 1) * (CR_NT * 2) + piece_b;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int q = wave * 3 + i;
      unsigned off[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) off[g] = lineoff[i] + px[g];
      cr_dma2(off, in_b, __builtin_amdgcn_readfirstlane(lds_base + (tt & 1) * CN_BUF + (q & 1) * CN_PLANE + (q >> 1) * CN_ROW));
    }
  };
  issue(0);
  const int half = wave & 1, col = wave >> 1;
  uint4 w[9][2][2];
  float bias_r[8];
  {
    const char* gw = (const char*)p.w;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          w[tap][c][j] = *(const uint4*)(gw + ((tap * 2 + c) * CR_NT + half * 32 + (li >> 2) * 8 + j * 4 + (li & 3)) * 64 + lg * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) bias_r[e] = 0.f;
    if (p.bias) {
      const float4 b0 = *(const float4*)(p.bias + half * 32 + lg * 8), b1 = *(const float4*)(p.bias + half * 32 + lg * 8 + 4);
      bias_r[0] = b0.x; bias_r[1] = b0.y; bias_r[2] = b0.z; bias_r[3] = b0.w; bias_r[4] = b1.x; bias_r[5] = b1.y; bias_r[6] = b1.z; bias_r[7] = b1.w;
    }
  }
  unsigned aoff[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) aoff[kw] = cr_swz((unsigned)(col * 16 + li + kw), lg);
  cr_wait_all();
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      cr_u32x4 x = {w[tap][c][0].x, w[tap][c][0].y, w[tap][c][0].z, w[tap][c][0].w}, y = {w[tap][c][1].x, w[tap][c][1].y, w[tap][c][1].z, w[tap][c][1].w};
      asm volatile("" : "+v"(x), "+v"(y));
      w[tap][c][0] = make_uint4(x[0], x[1], x[2], x[3]); w[tap][c][1] = make_uint4(y[0], y[1], y[2], y[3]);
    }
  __syncthreads();
  const int o = col * 16 + li;

  for (int tt = 0; tt < tiles_t; ++tt) {
    const int t0 = tt * CN_OT;
    if (tt + 1 < tiles_t) issue(tt + 1);       // into the other buffer: its readers (tile tt - 1) are behind the last barrier
    const int t = t0 + o;
    const unsigned tcl = (unsigned)min(t, p.To - 1) + 1;
    cr_u32x4 rp[4];
    if constexpr (RES) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        rp[q] = *(const cr_u32x4*)(res_b + ((size_t)((unsigned)(min(f0 + q, p.Fo - 1) + 1) * TP + tcl) * CR_NT + half * 32 + lg * 8) * 2);
    }
    const bool t_ok = o < CN_OT && t < p.To;
    f32x4_t acc[4][2];
    const char* img = cr_smem + (tt & 1) * CN_BUF;
    uint4 xf[2][3];
    auto read_row = [&](int step, int buf) __attribute__((always_inline)) {      // step = chunk * 6 + input row
      const int c = step / CR_PF, i = step - c * CR_PF;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) xf[buf][kw] = *(const uint4*)(img + aoff[kw] + i * CN_ROW + c * CN_PLANE);
    };
    read_row(0, 0);
#pragma unroll
    for (int step = 0; step < 2 * CR_PF; ++step) {
      const int cur = step & 1, c = step / CR_PF, i = step - c * CR_PF;
      if (step + 1 < 2 * CR_PF) read_row(step + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int kh = 2; kh >= 0; --kh) {
          const int r = i - kh;
          if (r < 0 || r >= CR_OF) continue;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (c == 0 && kh == 0 && kw == 0) acc[r][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            cr_mma(w[kh * 3 + kw][c][j], xf[cur][kw], acc[r][j]);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
      if (c == 1 && i >= 2) {
        const int q = i - 2;
        if (q == 0) {
          // this wave's pieces of the next patch (requested most of a tile of MFMAs ago) and the residual vectors have landed; the
          // stores below are younger than this wait, so nothing ever waits for them
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if constexpr (RES) {
#pragma unroll
            for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(rp[m]));
          }
        }
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[q][e >> 2][e & 3] + bias_r[e];
        if constexpr (RES) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] += __uint_as_float(rp[q][e] << 16);
            v[2 * e + 1] += __uint_as_float(rp[q][e] & 0xffff0000u);
          }
        }
        if constexpr (RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        const int fo = f0 + q;
        if (t_ok && fo < p.Fo)
          cr_store16(out_b + ((size_t)((unsigned)(fo + 1) * TP + t + 1) * CR_NT + half * 32 + lg * 8) * 2,
                     make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])));
      }
    }
    cr_wait_lds();
    __builtin_amdgcn_s_barrier();              // every wave's pieces of the next patch are in; this tile's buffer is free
    asm volatile("" ::: "memory");
  }
}

}  // namespace

bool conv_row64_applicable(int dtype, const ConvArgs& p) {
  const char* e = lab_env("RVD_CONV_ROW64");          // lab: 0 = resnet.hip's direct kernel (until round 5)
  if (e && atoi(e) == 0) return false;
  return dtype == DT_BF16 && p.taps == 9 && p.stride == 1 && p.Cin == 64 && p.Cout == 64 && p.Fo == p.Fi && p.To == p.Ti && !p.in2 && !p.in8 &&
         (int64_t)(p.Fi + 2) * (p.Ti + 2) * 128 < ((int64_t)1 << 31) && (int64_t)p.B * cdiv(p.Fo, CR_OF) < ((int64_t)1 << 31);
}

template <bool RES, bool RELU>
static int launch_row64(hipStream_t s, const ConvArgs& p) {
  static bool attr_set = false;
  if (!attr_set) {
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)conv_row64_kernel<RES, RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, CN_LDS));
    attr_set = true;
  }
  const int64_t blocks = (int64_t)p.B * cdiv(p.Fo, CR_OF);
  hipLaunchKernelGGL((conv_row64_kernel<RES, RELU>), dim3((unsigned)blocks), dim3(256), CN_LDS, s, p);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

int conv_row64(hipStream_t s, const ConvArgs& p) {
  if (p.B <= 0) return OK;
  if (p.res) return p.relu ? launch_row64<true, true>(s, p) : launch_row64<true, false>(s, p);
  return p.relu ? launch_row64<false, true>(s, p) : launch_row64<false, false>(s, p);
}

}  // namespace rvb
