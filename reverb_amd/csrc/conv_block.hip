// A whole stride-1 BasicBlock of the ResNet34 trunk's 32-channel stage in ONE kernel (bf16):
//
//     mid = relu(conv3x3_a(x) + b_a)          kept in LDS, never written to HBM
//     out = relu(conv3x3_b(mid) + b_b + x)    the residual x comes out of the input patch that is in LDS anyway
//
// Why (round 5; VERDICT r4 "next" #2): as two conv_stream launches a block moves five tensor passes through HBM (x read, mid
// written, mid read, x read again as the residual, out written; 1.55 KB per output pixel with the halo rows) for 37 kFLOP per
// pixel.  Here x is read once (8 patch rows for 4 output rows) and out written once: 0.79 KB per output pixel.
//
// What the first form of this kernel taught (profiles/r05_call3_*.txt: correct, bit-identical, and 61 ms per hour against 54-56
// for two launches): these stages are not bound by HBM alone but by their LDS operations -- (i) the [pixel][64 B] patch layout
// serves a wave's ds_read_b128 with a 2-way bank conflict in every lane group (lanes {0-3, 12-15} of a group sit 12 pixels =
// 768 B = 0 mod 256 apart), so every fragment read took 8 cycles instead of 4; (ii) the accumulators (pixels x channels) went
// through a per-wave LDS slab to reach a store-friendly layout: 61 LDS cycles per 16-pixel m-tile, more than its 18 MFMAs;
// (iii) the weights' fragments were re-read from LDS for every tile.  This form removes all three:
//   * operand roles swapped: A = weights (M = channels), B = pixels (N = pixels), and the weight ROWS a lane supplies are chosen
//     so that accumulator row (lane >> 4) * 4 + r of n-tile j is channel (lane >> 4) * 8 + j * 4 + r: a lane ends up holding 8
//     CONSECUTIVE channels of ONE pixel -- exactly the 16 bytes it writes (to mid in LDS, or to the output in HBM: a wave stores
//     1 KiB contiguous).  No transposition, no slab, no wave barriers;
//   * both weight sets live in REGISTERS (2 x 9 taps x 2 n-tiles x 16 B per lane = 144 VGPRs of the 256 a wave has at two
//     waves per SIMD), loaded once per workgroup; LDS holds only pixels;
//   * the 16-byte chunks of a pixel are XOR-swizzled with bit 2 of the pixel index (chunk ^= 2 for pixels 4-7 mod 8), on the
//     LDS-DMA's source side for the patches and on the write side for mid: every lane group of a fragment read then covers all
//     64 banks once.
//
// Geometry (unbordered coordinates; the tensors carry a one-pixel zero border, element (f, t) sits at bordered (f + 1, t + 1)):
//   workgroup = 512 threads = 8 waves, owns output rows f0 .. f0+3 of one window and walks tiles of 60 frames, t0 = 60 tt
//   patch  8 rows x 64 pixels x 64 B: bordered rows f0-1 .. f0+6, bordered columns t0-1 .. t0+62 (clamped into the plane: what the
//          clamp changes only feeds mid positions outside the image, and those are set to zero)
//   mid    6 rows x 64 pixels: rows f0-1 .. f0+4, columns t0-1 .. t0+62 (columns 62, 63 are never used); ZERO outside the image --
//          the zero border the second convolution sees in the unfused path
//   conv_a 24 m-tiles of 16 pixels, 3 per wave;  conv_b 16 m-tiles, 2 per wave (row w >> 1, pixels 32 (w & 1) .. +31)
//   LDS    patches 2 x 32 768 + mid 24 576 (+ pads) = 90 368 B: one workgroup per CU, two waves per SIMD
//
// Iteration k (tile k of the walk):
//   a. LDS-DMA of patch k+1 into the other patch buffer (its readers -- conv_a and the residual reads of tile k-1 -- finished
//      before barrier B of iteration k-1)
//   b. conv_a on patch k; the wave's residual vectors are read from the patch; bias + ReLU + zeroing; mid written
//      (mid's readers -- conv_b of tile k-1 -- finished before barrier A of iteration k-1)
//   c. barrier B: mid visible
//   d. conv_b on mid
//   e. s_waitcnt vmcnt(0): this wave's pieces of patch k+1 have landed (and the stores of tile k-1 have drained: they were issued
//      a whole tile ago); barrier A: patch k+1 visible, mid free
//   f. epilogue in registers: bias + residual + ReLU, one 16-byte store per lane and m-tile, draining under iteration k+1
//
// Results: operand values, accumulation order (taps 0..8, one 32-channel K step each) and rounding points (mid and out rounded to
// bf16 after bias / residual / ReLU in fp32) are those of two conv_stream / conv_kernel launches
// (tests/test_diar_gpu.py: test_fused_basic_block_equals_two_convolutions).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace rvb {

namespace {

constexpr int CB_OT = 60, CB_PT = 64, CB_PF = 8, CB_MF = 6, CB_OF = 4, CB_NT = 32;
constexpr int CB_ROW = CB_PT * 64;                   // one patch / mid row: 4 096 B
constexpr int CB_PATCH = CB_PF * CB_ROW;             // 32 768 B
constexpr int CB_MID = CB_MF * CB_ROW;               // 24 576 B
constexpr int CB_OFF_P0 = 0;
constexpr int CB_OFF_MID = CB_OFF_P0 + 2 * CB_PATCH + 128;        // 128 B: the two pixels garbage m-tile positions read past a buffer
constexpr int CB_LDS = CB_OFF_MID + CB_MID + 128;

typedef unsigned cb_u32x4 __attribute__((ext_vector_type(4)));

// D[channel][pixel] += W[channel][k] . X[pixel][k]: the WEIGHTS are the A operand (see the file header)
__device__ inline void cb_mma(const uint4& w, const uint4& x, f32x4_t& c) {
  union U { uint4 u; bf16x8_t v; };
  U uw, ux;
  uw.u = w; ux.u = x;
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uw.v, ux.v, c, 0, 0, 0);
}

// byte offset of (pixel g, 16-byte chunk c) in a [pixel][64 B] image whose chunks are swizzled with bit 2 of the pixel index
__device__ inline unsigned cb_swz(unsigned g, unsigned c) { return g * 64u + ((c ^ (((g >> 2) & 1u) << 1)) << 4); }

// four 1-KiB LDS-DMA pieces: one patch row (4 KiB), 16 pixels per piece; per-lane 32-bit byte offsets from a scalar base
__device__ inline void cb_dma4(const unsigned (&off)[4], const void* sbase, unsigned lds0) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %6\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %3, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %4, %5\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(sbase), "s"(lds0)
      : "memory", "scc");
}
__device__ inline void cb_wait_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
__device__ inline void cb_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ inline const char* cb_uniform(const char* q) {
  const unsigned long long v = (unsigned long long)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}

// 9 taps of one convolution for NM m-tiles of this wave.  w: the weight fragments (registers); img: patch or mid; a[m][kw]: this
// lane's swizzled byte offset of m-tile m's pixel at column shift kw, row shift 0 (a row further down is + CB_ROW: the swizzle
// does not depend on the row); the pixel fragments of tap + 1 are read before the MFMAs of tap are issued.
template <int NM>
__device__ inline void cb_conv9(const uint4 (&w)[9][2], const char* img, const unsigned (&a)[NM][3], f32x4_t (&acc)[NM][2]) {
  uint4 xf[2][NM];
  auto read_frags = [&](int tap, int buf) __attribute__((always_inline)) {
    const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
    for (int m = 0; m < NM; ++m) xf[buf][m] = *(const uint4*)(img + a[m][kw] + kh * CB_ROW);
  };
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[m][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  read_frags(0, 0);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int cur = tap & 1;
    if (tap + 1 < 9) read_frags(tap + 1, cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
      for (int j = 0; j < 2; ++j) cb_mma(w[tap][j], xf[cur][m], acc[m][j]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ __launch_bounds__(512, 2) void conv_block32_kernel(ConvBlockArgs p, int tsplit) {
  extern __shared__ __attribute__((aligned(16))) char cb_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int FP = p.F + 2, TP = p.T + 2;
  const int tiles_f = (p.F + CB_OF - 1) / CB_OF, tiles_t = (p.T + CB_OT - 1) / CB_OT;
  const int per = (tiles_t + tsplit - 1) / tsplit;

  // workgroup -> (window, mel-row tile, part of the time axis); each XCD (workgroup id mod 8) takes a contiguous run of the linear
  // order, so that the workgroups that share halo rows run on the same L2 at about the same time (as conv_stream.hip)
  int lin;
  {
    const int nblk = (int)gridDim.x, q = nblk >> 3, r = nblk & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int sp = lin % tsplit;
  const int tf = (lin / tsplit) % tiles_f;
  const int b = lin / (tsplit * tiles_f);
  const int f0 = tf * CB_OF;
  const int tt0 = sp * per, tt1 = min(tiles_t, tt0 + per);
  const int n_tiles = tt1 - tt0;
  if (n_tiles <= 0) return;

  const char* in_b = cb_uniform((const char*)p.in + (size_t)b * FP * TP * CB_NT * 2);
  char* out_b = (char*)p.out + (size_t)b * FP * TP * CB_NT * 2;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)cb_smem;

  // ---- DMA coordinates: wave w brings patch row w (bordered row f0 - 1 + w, clamped), four pieces of 16 pixels.  The LDS image is
  // lane-linear (lane i's 16 bytes land at M0 + 16 i), so the chunk swizzle is applied to the SOURCE: LDS position (pixel, chunk')
  // receives global chunk chunk' ^ 2 [pixel bit 2]; within a piece pixel = lane >> 2, so that bit is lane bit 4
  const unsigned rowoff = (unsigned)(min(max(f0 - 1 + wave, 0), FP - 1) * TP) * (CB_NT * 2);
  const int ppx = lane >> 2;
  const unsigned piece_b = (unsigned)((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 16;
  auto issue = [&](int k) __attribute__((always_inline)) {
    const int t0 = (tt0 + k) * CB_OT;
    unsigned off[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) off[g] = rowoff + (unsigned)min(max(t0 - 1 + g * 16 + ppx, 0), TP - 1) * (CB_NT * 2) + piece_b;
    cb_dma4(off, in_b, __builtin_amdgcn_readfirstlane(lds_base + CB_OFF_P0 + (k & 1) * CB_PATCH + wave * CB_ROW));
  };
  issue(0);

  // ---- both weight sets into registers: lane (li, lg) supplies, for n-tile j, the row of channel (li >> 2) * 8 + j * 4 + (li & 3),
  // k chunk lg (global layout [tap][channel][64 B])
  uint4 wa[9][2], wb[9][2];
  {
    const char* ga = (const char*)p.wa;
    const char* gb = (const char*)p.wb;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ch = (li >> 2) * 8 + j * 4 + (li & 3);
        wa[tap][j] = *(const uint4*)(ga + (tap * CB_NT + ch) * 64 + lg * 16);
        wb[tap][j] = *(const uint4*)(gb + (tap * CB_NT + ch) * 64 + lg * 16);
      }
  }
  // this lane's 8 channels are lg * 8 .. + 7: accumulator (j, r) is channel lg * 8 + j * 4 + r
  float ba_r[8], bb_r[8];
  {
    const float4 a0 = *(const float4*)(p.ba + lg * 8), a1 = *(const float4*)(p.ba + lg * 8 + 4);
    const float4 b0 = *(const float4*)(p.bb + lg * 8), b1 = *(const float4*)(p.bb + lg * 8 + 4);
    ba_r[0] = a0.x; ba_r[1] = a0.y; ba_r[2] = a0.z; ba_r[3] = a0.w; ba_r[4] = a1.x; ba_r[5] = a1.y; ba_r[6] = a1.z; ba_r[7] = a1.w;
    bb_r[0] = b0.x; bb_r[1] = b0.y; bb_r[2] = b0.z; bb_r[3] = b0.w; bb_r[4] = b1.x; bb_r[5] = b1.y; bb_r[6] = b1.z; bb_r[7] = b1.w;
  }

  // ---- this wave's m-tiles.  conv_a: indices 3 w .. 3 w + 2 of (mid row r, 16-pixel group mi) = divmod(idx, 4);
  //      conv_b: output row w >> 1, pixel groups 2 (w & 1) and 2 (w & 1) + 1.  Lane (li, lg) = (pixel li of the m-tile, chunk lg).
  int ar[3], ami[3];
  unsigned aoff[3][3], moff[3];          // conv_a: fragment offsets per column shift; where this lane's mid vector goes
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    ar[m] = (wave * 3 + m) >> 2; ami[m] = (wave * 3 + m) & 3;
    const unsigned g = (unsigned)(ar[m] * CB_PT + ami[m] * 16 + li);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) aoff[m][kw] = cb_swz(g + kw, lg);
    moff[m] = cb_swz(g, lg);
  }
  const int brow = wave >> 1, bmi0 = (wave & 1) * 2;
  unsigned boff[2][3], roff[2];          // conv_b: fragment offsets in mid; the residual vector in the patch
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const unsigned g = (unsigned)(brow * CB_PT + (bmi0 + m) * 16 + li);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) boff[m][kw] = cb_swz(g + kw, lg);
    roff[m] = cb_swz(g + 2 * CB_PT + 2, lg);                   // x(f, t) = patch(row + 2, column + 2)
  }
  char* mid = cb_smem + CB_OFF_MID;
  const int fo = f0 + brow;                                    // this wave's output row
  const unsigned frow_off = (unsigned)(min(fo, p.F - 1) + 1) * TP;

  // ---- prologue: patch 0 landed, visible
  cb_wait_all();
  __syncthreads();

  for (int k = 0; k < n_tiles; ++k) {
    const int t0 = (tt0 + k) * CB_OT;
    const char* patch = cb_smem + CB_OFF_P0 + (k & 1) * CB_PATCH;
    if (k + 1 < n_tiles) issue(k + 1);                                         // (a)

    // (b) first convolution: mid(r, px) = sum_taps wa[tap] . patch(r + kh, px + kw)
    {
      f32x4_t acc[3][2];
      cb_conv9<3>(wa, patch, aoff, acc);
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int f = f0 - 1 + ar[m], t = t0 - 1 + ami[m] * 16 + li;
        const bool inside = f >= 0 && f < p.F && t >= 0 && t < p.T;            // outside: the zero border conv_b must see
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = inside ? fmaxf(acc[m][e >> 2][e & 3] + ba_r[e], 0.f) : 0.f;
        *(uint4*)(mid + moff[m]) = make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
      }
    }
    // the residual of this wave's outputs, read before barrier B (see the file header)
    cb_u32x4 rp[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) rp[m] = *(const cb_u32x4*)(patch + roff[m]);
    cb_wait_lds();
    __builtin_amdgcn_s_barrier();                                              // (c) barrier B
    asm volatile("" ::: "memory");

    // (d) second convolution: out(row, o) = sum_taps wb[tap] . mid(row + kh, o + kw)
    f32x4_t acc2[2][2];
    cb_conv9<2>(wb, mid, boff, acc2);
    cb_wait_all();                                                             // (e)
    __builtin_amdgcn_s_barrier();                                              //     barrier A
    asm volatile("" ::: "memory");

    // (f) epilogue: bias + residual + ReLU in fp32, one 16-byte store per lane and m-tile (a wave stores 1 KiB contiguous)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = acc2[m][e >> 2][e & 3] + bb_r[e];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] += __uint_as_float(rp[m][e] << 16);
        v[2 * e + 1] += __uint_as_float(rp[m][e] & 0xffff0000u);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      const int o = (bmi0 + m) * 16 + li, t = t0 + o;
      if (o < CB_OT && t < p.T && fo < p.F)
        *(uint4*)(out_b + ((size_t)(frow_off + t + 1) * CB_NT + lg * 8) * 2) =
            make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
    }
  }
}

}  // namespace

bool conv_block32_applicable(int dtype, int cin, int cmid, int cout, int stride_a, int stride_b, int taps_a, int taps_b, int F, int T) {
  const char* e = lab_env("RVD_CONV_BLOCK");          // lab: 0 = two convolution launches per block (until round 5)
  if (e && atoi(e) == 0) return false;
  return dtype == DT_BF16 && cin == 32 && cmid == 32 && cout == 32 && stride_a == 1 && stride_b == 1 && taps_a == 9 && taps_b == 9 &&
         (int64_t)(F + 2) * (T + 2) * 64 < ((int64_t)1 << 31);
}

int conv_block32(hipStream_t s, const ConvBlockArgs& a) {
  if (a.B <= 0) return OK;
  static bool attr_set = false;
  if (!attr_set) {
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)conv_block32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CB_LDS));
    attr_set = true;
  }
  const int tsplit = 1;
  const int64_t blocks = (int64_t)a.B * cdiv(a.F, CB_OF) * tsplit;
  if (blocks >= ((int64_t)1 << 31)) { set_error("conv_block32: too many workgroups"); return E_ARG; }
  hipLaunchKernelGGL(conv_block32_kernel, dim3((unsigned)blocks), dim3(512), CB_LDS, s, a, tsplit);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace rvb
