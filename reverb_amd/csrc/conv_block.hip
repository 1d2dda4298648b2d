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
// Third form (the one below) -- the two convolutions are ROLES of different waves, one tile apart:
//   waves 0-3 (A)  first convolution of tile k+1: 24 m-tiles of 16 pixels, 6 per wave, weights of conv_a in registers; bias + ReLU +
//                  zeroing; mid(k+1) written to LDS
//   waves 4-7 (B)  epilogue of tile k-1 (bias + residual + ReLU, stores), then the second convolution of tile k from mid(k): output
//                  row w - 4, 4 m-tiles, weights of conv_b in registers; the residual vectors of tile k read out of patch k
// Every SIMD holds one wave of each role and there is ONE barrier per tile; 149 VGPRs.  Measured EQUAL to the second form (all 8
// waves in the same phase, two barriers per tile: 42.5-43.0 vs 42.8 ms per hour, profiles/r05_call4_*.txt, r05_call5_*.txt), so
// phase serialisation was not what was left.  What the counters say (profiles/r05_call7_conv_block_counters.txt, per hour):
//   HBM       58 GB read (the 4 halo rows of a patch come out of the L2: 1.03 x the plane) + 55 GB written = 113 GB in 42.8 ms =
//             2.6 TB/s; the two-launch form moves 170 + 111 = 281 GB at 5.2 TB/s -- THAT form is bound by HBM, this one is not
//   LDS       bank-conflict cycles 10 % of the LDS-active cycles (two-launch form: 44 %), the LDS 27 % busy
//   MFMA      SQ_VALU_MFMA_BUSY_CYCLES = 41 % of the launch (1.5 x the useful FLOPs with the halo rows and the 64-wide m-tiles: 0.99 PF/s)
//   waves     33 % issuing, 43 % stalled at issue (matrix pipe / dependency), 24 % parked at the barrier or a counter; 3.25 VALU
//             instructions per MFMA
// i.e. 3.0 us per tile of which 1.2 us are MFMA time; the remainder is VALU work sharing the issue slots (bias / ReLU / zeroing /
// packing / residual: next step packed fp32 math), four LDS-DMA issues per wave and tile, and a workgroup's start-up (weights,
// first patch, the one-tile offset of role B) paid once per 17 tiles.
// Measured and rejected (profiles/r05_call8_conv_block_fourth_form.txt): patches requested TWO tiles ahead by the role-A waves
// only (8 pieces each, `s_waitcnt vmcnt(8)` before the barrier), role B taking its residual from the input plane with plain
// loads and storing in the same body through stores hidden from the waitcnt pass -- bit-identical, 45.4-45.5 ms against 42.8:
// the tile time does not come from the one-tile prefetch distance either, and the eight DMA issues per tile on the waves that
// also carry 60 % of the MFMAs cost more than the deeper prefetch returns.
//
// Geometry (unbordered coordinates; the tensors carry a one-pixel zero border, element (f, t) sits at bordered (f + 1, t + 1)):
//   workgroup = 512 threads = 8 waves, owns output rows f0 .. f0+3 of one window and walks tiles of 60 frames, t0 = 60 tt
//   patch  8 rows x 64 pixels x 64 B: bordered rows f0-1 .. f0+6, bordered columns t0-1 .. t0+62 (clamped into the plane: what the
//          clamp changes only feeds mid positions outside the image, and those are set to zero); a ring of THREE (tile k for the
//          residual, tile k+1 for conv_a, tile k+2 arriving)
//   mid    6 rows x 64 pixels: rows f0-1 .. f0+4, columns t0-1 .. t0+62 (columns 62, 63 are never used); ZERO outside the image --
//          the zero border the second convolution sees in the unfused path; TWO buffers (tile k read, tile k+1 written)
//   LDS    patches 3 x 32 768 + mid 2 x 24 576 (+ pads) = 147 840 B: one workgroup per CU, two waves per SIMD
//
// Body k of the walk (all waves; then `s_waitcnt vmcnt(0) lgkmcnt(0)` and the barrier):
//   all  LDS-DMA of patch k+2 into ring slot (k+2) % 3 -- last read in body k-1 (residual of tile k-1) and k-2 (conv_a of tile k-1)
//   A    conv_a(k+1) from ring slot (k+1) % 3 (requested in body k-1, landed before that body's barrier) -> mid[(k+1) & 1], last
//        read by conv_b(k-1) in body k-1
//   B    epilogue(k-1) from registers: its stores are issued at the top of the body, right behind the DMA, so the vmcnt(0) at the
//        end of the body finds them drained; conv_b(k) from mid[k & 1] (written in body k-1); residual vectors of tile k
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
constexpr int CB_OFF_MID = CB_OFF_P0 + 3 * CB_PATCH + 128;        // 128 B: the two pixels garbage m-tile positions read past a buffer
constexpr int CB_MID_PITCH = CB_MID + 128;
constexpr int CB_LDS = CB_OFF_MID + 2 * CB_MID_PITCH;

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

// 9 taps of one convolution for NM consecutive m-tiles of this wave.  w: the weight fragments (registers); img: patch or mid at the
// wave's first m-tile; a[kw]: this lane's swizzled byte offset at column shift kw, row shift 0.  The next m-tile is 16 pixels =
// 1 024 B further (m-tiles are numbered along rows and a row is four of them), a row further down is + CB_ROW; neither changes
// bit 2 of the pixel index, so the swizzle is the same and both are immediate offsets of the read.  The pixel fragments of
// tap + 1 are read before the MFMAs of tap are issued.
template <int NM>
__device__ inline void cb_conv9(const uint4 (&w)[9][2], const char* img, const unsigned (&a)[3], f32x4_t (&acc)[NM][2]) {
  uint4 xf[2][NM];
  auto read_frags = [&](int tap, int buf) __attribute__((always_inline)) {
    const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
    for (int m = 0; m < NM; ++m) xf[buf][m] = *(const uint4*)(img + a[kw] + m * 1024 + kh * CB_ROW);
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
    cb_dma4(off, in_b, __builtin_amdgcn_readfirstlane(lds_base + CB_OFF_P0 + (k % 3) * CB_PATCH + wave * CB_ROW));
  };
  issue(0);
  if (n_tiles > 1) issue(1);

  // ---- this wave's role and its weight set, in registers: lane (li, lg) supplies, for n-tile j, the row of channel
  // (li >> 2) * 8 + j * 4 + (li & 3), k chunk lg (global layout [tap][channel][64 B]); its 8 channels are then lg * 8 .. + 7
  // (accumulator (j, r) is channel lg * 8 + j * 4 + r)
  const bool role_a = wave < 4;
  uint4 w[9][2];
  float bias_r[8];
  {
    const char* gw = (const char*)(role_a ? p.wa : p.wb);
    const float* gb = role_a ? p.ba : p.bb;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int j = 0; j < 2; ++j) w[tap][j] = *(const uint4*)(gw + (tap * CB_NT + (li >> 2) * 8 + j * 4 + (li & 3)) * 64 + lg * 16);
    const float4 b0 = *(const float4*)(gb + lg * 8), b1 = *(const float4*)(gb + lg * 8 + 4);
    bias_r[0] = b0.x; bias_r[1] = b0.y; bias_r[2] = b0.z; bias_r[3] = b0.w; bias_r[4] = b1.x; bias_r[5] = b1.y; bias_r[6] = b1.z; bias_r[7] = b1.w;
  }
  char* mid0 = cb_smem + CB_OFF_MID;
  cb_wait_all();                       // patches 0 (and 1) landed
  __syncthreads();

  if (role_a) {
    // ======== role A: first convolution, one tile ahead.  m-tiles 6 w .. 6 w + 5 of (mid row r, 16-pixel group mi) = divmod(idx, 4)
    // (m-tile idx sits at pixel 16 idx of the row-major [6][64] mid image: consecutive m-tiles are 1 024 B apart)
    unsigned aoff[3];                      // this lane's fragment offset at the wave's first m-tile, per column shift
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) aoff[kw] = cb_swz((unsigned)(wave * 6 * 16 + li + kw), lg);
    const unsigned moff = cb_swz((unsigned)(wave * 6 * 16 + li), lg);      // where this lane's mid vector of m-tile 0 goes
    auto conv_a = [&](int k) __attribute__((always_inline)) {      // mid(r, px) = sum_taps wa[tap] . patch(r + kh, px + kw), tile k
      const int t0 = (tt0 + k) * CB_OT;
      const char* patch = cb_smem + CB_OFF_P0 + (k % 3) * CB_PATCH;
      char* mid = mid0 + (k & 1) * CB_MID_PITCH;
#pragma unroll
      for (int h = 0; h < 2; ++h) {            // two halves of three m-tiles: 24 accumulator + 24 fragment registers live at a time
        f32x4_t acc[3][2];
        cb_conv9<3>(w, patch + h * 3 * 1024, aoff, acc);
#pragma unroll
        for (int mm = 0; mm < 3; ++mm) {
          const int m = h * 3 + mm, idx = wave * 6 + m;
          const int f = f0 - 1 + (idx >> 2), t = t0 - 1 + (idx & 3) * 16 + li;
          const bool inside = f >= 0 && f < p.F && t >= 0 && t < p.T;            // outside: the zero border conv_b must see
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = inside ? fmaxf(acc[mm][e >> 2][e & 3] + bias_r[e], 0.f) : 0.f;
          *(uint4*)(mid + moff + m * 1024) = make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
        }
      }
    };
    conv_a(0);
    cb_wait_lds();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int k = 0; k < n_tiles; ++k) {
      if (k + 2 < n_tiles) issue(k + 2);
      if (k + 1 < n_tiles) conv_a(k + 1);
      cb_wait_all();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  } else {
    // ======== role B: second convolution of output row w - 4 (its four 16-pixel groups), residual, stores
    const int brow = wave - 4;
    unsigned boff[3];                      // fragment offset in mid at the row's first m-tile, per column shift
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) boff[kw] = cb_swz((unsigned)(brow * CB_PT + li + kw), lg);
    const unsigned roff = cb_swz((unsigned)((brow + 2) * CB_PT + li + 2), lg);      // residual: x(f, t) = patch(row + 2, column + 2)
    const int fo = f0 + brow;                                  // this wave's output row
    const unsigned frow_off = (unsigned)(min(fo, p.F - 1) + 1) * TP;
    f32x4_t acc2[2][2][2];                 // [half][m-tile of the half][n-tile]
    cb_u32x4 rp[4];
    auto epilogue = [&](int k) __attribute__((always_inline)) {    // bias + residual + ReLU in fp32, one 16-byte store per lane and m-tile
      const int t0 = (tt0 + k) * CB_OT;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc2[m >> 1][m & 1][e >> 2][e & 3] + bias_r[e];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * e] += __uint_as_float(rp[m][e] << 16);
          v[2 * e + 1] += __uint_as_float(rp[m][e] & 0xffff0000u);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        const int o = m * 16 + li, t = t0 + o;
        if (o < CB_OT && t < p.T && fo < p.F)
          *(uint4*)(out_b + ((size_t)(frow_off + t + 1) * CB_NT + lg * 8) * 2) =
              make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
      }
    };
    cb_wait_lds();
    __builtin_amdgcn_s_barrier();          // role A's conv_a(0)
    asm volatile("" ::: "memory");
    for (int k = 0; k < n_tiles; ++k) {
      if (k + 2 < n_tiles) issue(k + 2);
      if (k > 0) epilogue(k - 1);
      // out(row, o) = sum_taps wb[tap] . mid(row + kh, o + kw), tile k; then its residual vectors out of patch k
#pragma unroll
      for (int h = 0; h < 2; ++h) cb_conv9<2>(w, mid0 + (k & 1) * CB_MID_PITCH + h * 2 * 1024, boff, acc2[h]);      // two halves: 16 fragment registers live
      const char* patch = cb_smem + CB_OFF_P0 + (k % 3) * CB_PATCH;
#pragma unroll
      for (int m = 0; m < 4; ++m) rp[m] = *(const cb_u32x4*)(patch + roff + m * 1024);
      cb_wait_all();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    epilogue(n_tiles - 1);
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
