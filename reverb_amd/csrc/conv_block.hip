// A whole stride-1 BasicBlock of the ResNet34 trunk's 32-channel stage in ONE kernel (bf16):
//
//     mid = relu(conv3x3_a(x) + b_a)          kept in LDS, never written to HBM
//     out = relu(conv3x3_b(mid) + b_b + x)    the residual x comes out of the input patch that is in LDS anyway
//
// Why (round 5; VERDICT r4 "next" #2): as two conv_stream launches a block moves five tensor passes through HBM (x read, mid
// written, mid read, x read again as the residual, out written) for 37 kFLOP per pixel: 281 GB per hour of audio at 5.2 TB/s --
// that form IS bound by HBM (profiles/r05_call7_conv_block_counters.txt).  Here x is read once (8 patch rows for 4 output rows, the
// 4 halo rows served by the L2: 1.03 x the plane from HBM) and out written once: 113 GB per hour.
//
// Geometry (unbordered coordinates; the tensors carry a one-pixel zero border, element (f, t) sits at bordered (f + 1, t + 1)):
//   workgroup = 256 threads = 4 waves, owns output rows f0 .. f0+3 of one window and walks tiles of 60 frames, t0 = 60 tt
//   patch  8 rows x 64 pixels x 64 B: bordered rows f0-1 .. f0+6, bordered columns t0-1 .. t0+62 (clamped into the plane: what the
//          clamp changes only feeds mid positions outside the image, and those are set to zero); ONE buffer
//   mid    6 rows x 64 pixels: rows f0-1 .. f0+4, columns t0-1 .. t0+62 (columns 62, 63 are never used); ZERO outside the image --
//          the zero border the second convolution sees in the unfused path
//   conv_a 24 m-tiles of 16 pixels, 6 per wave (two halves of three);  conv_b output row w, 4 m-tiles (two halves of two)
//   LDS    patch 32 768 + mid 24 576 (+ pads) = 57 600 B and 248 VGPRs: TWO workgroups per CU, two waves per SIMD
//
// What makes it fast (each measured on the way, bit-identical throughout; "ms" = the stage per hour of audio, two launches: 54-56):
//   * operand roles swapped: A = weights (M = channels), B = pixels (N = pixels), and the weight ROWS a lane supplies are chosen
//     so that accumulator row (lane >> 4) * 4 + r of n-tile j is channel (lane >> 4) * 8 + j * 4 + r: a lane ends up holding 8
//     CONSECUTIVE channels of ONE pixel -- exactly the 16 bytes it writes (to mid in LDS, or to the output in HBM: a wave stores
//     1 KiB contiguous).  The first form (pixels x channels accumulators through a per-wave LDS slab, 61 LDS cycles per m-tile,
//     weights and slabs in LDS: 61 ms, SLOWER than two launches) showed that these stages are bound by their LDS operations once
//     HBM is out of the way;
//   * both weight sets live in REGISTERS (2 x 9 taps x 2 n-tiles x 16 B per lane = 144 VGPRs), loaded once per workgroup: LDS holds
//     only pixels, and no fragment of a weight is ever re-read;
//   * the 16-byte chunks of a pixel are XOR-swizzled with bit 2 of the pixel index (chunk ^= 2 for pixels 4-7 mod 8), on the
//     LDS-DMA's source side for the patch and on the write side for mid: the plain [pixel][64 B] layout serves a ds_read_b128
//     with a 2-way bank conflict in every lane group (lanes {0-3, 12-15} of a group sit 12 pixels = 768 B = 0 mod 256 apart);
//     bank-conflict cycles 44 % -> 10 % of the LDS-active cycles.  Together: 42.5-43.0 ms (8 waves, two patch buffers, all waves in
//     the same phase, two barriers per tile);
//   * the same with the two convolutions as ROLES of different waves one tile apart (one barrier per tile): 42.8 ms -- equal; with
//     the patches requested two tiles ahead by the role-A waves: 45.4 ms -- slower (profiles/r05_call5_*, r05_call8_*).  Neither
//     phase serialisation inside a workgroup nor the prefetch distance was what was left: a tile took 3.0 us of which 1.2 us are
//     MFMA time (MFMA busy 41 %, waves 33 % issuing / 43 % stalled at issue / 24 % parked, 3.25 VALU instructions per MFMA);
//   * what did help is the ordinary remedy: FOUR waves and ONE patch buffer per workgroup, so that two workgroups share a CU and
//     fill each other's barriers, DMA waits and VALU phases -- no prefetch across tiles at all: **39.3-39.5 ms**
//     (profiles/r05_call9_conv_block_two_workgroups.txt).
//
// Iteration k (tile k of the walk):
//   conv_a on the patch; bias + ReLU + zeroing; mid written; the wave's residual vectors read out of the patch; barrier B (mid
//   visible, the patch free); LDS-DMA of patch k+1 into the patch buffer; conv_b on mid; `s_waitcnt vmcnt(0)` (this wave's pieces of
//   patch k+1 have landed; the stores of tile k-1, issued a tile ago, have drained) and barrier A (patch k+1 visible, mid free);
//   epilogue in registers: bias + residual + ReLU, one 16-byte store per lane and m-tile, draining under iteration k+1.
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
constexpr int CB_OFF_MID = CB_PATCH + 128;              // 128 B: the two pixels garbage m-tile positions read past a buffer
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


__global__ __launch_bounds__(256, 2) void conv_block32_kernel(ConvBlockArgs p, int tsplit) {
  extern __shared__ __attribute__((aligned(16))) char cb_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int FP = p.F + 2, TP = p.T + 2;
  const int tiles_f = (p.F + CB_OF - 1) / CB_OF, tiles_t = (p.T + CB_OT - 1) / CB_OT;
  const int per = (tiles_t + tsplit - 1) / tsplit;
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

  // wave w brings patch rows 2 w and 2 w + 1 (8 pieces); chunk swizzle on the source side as in the 8-wave kernel
  const unsigned rowoff0 = (unsigned)(min(max(f0 - 1 + 2 * wave, 0), FP - 1) * TP) * (CB_NT * 2);
  const unsigned rowoff1 = (unsigned)(min(max(f0 + 2 * wave, 0), FP - 1) * TP) * (CB_NT * 2);
  const int ppx = lane >> 2;
  const unsigned piece_b = (unsigned)((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 16;
  auto issue = [&](int k) __attribute__((always_inline)) {
    const int t0 = (tt0 + k) * CB_OT;
    unsigned px[4], off[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) px[g] = (unsigned)min(max(t0 - 1 + g * 16 + ppx, 0), TP - 1) * (CB_NT * 2) + piece_b;
    const unsigned dst = lds_base + 2 * wave * CB_ROW;
#pragma unroll
    for (int g = 0; g < 4; ++g) off[g] = rowoff0 + px[g];
    cb_dma4(off, in_b, __builtin_amdgcn_readfirstlane(dst));
#pragma unroll
    for (int g = 0; g < 4; ++g) off[g] = rowoff1 + px[g];
    cb_dma4(off, in_b, __builtin_amdgcn_readfirstlane(dst + CB_ROW));
  };
  issue(0);
  uint4 wa[9][2], wb[9][2];
  float ba_r[8], bb_r[8];
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
    const float4 a0 = *(const float4*)(p.ba + lg * 8), a1 = *(const float4*)(p.ba + lg * 8 + 4);
    const float4 b0 = *(const float4*)(p.bb + lg * 8), b1 = *(const float4*)(p.bb + lg * 8 + 4);
    ba_r[0] = a0.x; ba_r[1] = a0.y; ba_r[2] = a0.z; ba_r[3] = a0.w; ba_r[4] = a1.x; ba_r[5] = a1.y; ba_r[6] = a1.z; ba_r[7] = a1.w;
    bb_r[0] = b0.x; bb_r[1] = b0.y; bb_r[2] = b0.z; bb_r[3] = b0.w; bb_r[4] = b1.x; bb_r[5] = b1.y; bb_r[6] = b1.z; bb_r[7] = b1.w;
  }
  unsigned aoff[3], boff[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    aoff[kw] = cb_swz((unsigned)(wave * 6 * 16 + li + kw), lg);
    boff[kw] = cb_swz((unsigned)(wave * CB_PT + li + kw), lg);
  }
  const unsigned moff = cb_swz((unsigned)(wave * 6 * 16 + li), lg);
  const unsigned roff = cb_swz((unsigned)((wave + 2) * CB_PT + li + 2), lg);
  const char* patch = cb_smem;
  char* mid = cb_smem + CB_OFF_MID;
  const int fo = f0 + wave;
  const unsigned frow_off = (unsigned)(min(fo, p.F - 1) + 1) * TP;
  cb_wait_all();
  __syncthreads();

  for (int k = 0; k < n_tiles; ++k) {
    const int t0 = (tt0 + k) * CB_OT;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4_t acc[3][2];
      cb_conv9<3>(wa, patch + h * 3 * 1024, aoff, acc);
#pragma unroll
      for (int mm = 0; mm < 3; ++mm) {
        const int m = h * 3 + mm, idx = wave * 6 + m;
        const int f = f0 - 1 + (idx >> 2), t = t0 - 1 + (idx & 3) * 16 + li;
        const bool inside = f >= 0 && f < p.F && t >= 0 && t < p.T;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = inside ? fmaxf(acc[mm][e >> 2][e & 3] + ba_r[e], 0.f) : 0.f;
        *(uint4*)(mid + moff + m * 1024) = make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
      }
    }
    cb_u32x4 rp[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) rp[m] = *(const cb_u32x4*)(patch + roff + m * 1024);
    cb_wait_lds();
    __builtin_amdgcn_s_barrier();                      // B: mid visible, the patch is free
    asm volatile("" ::: "memory");
    if (k + 1 < n_tiles) issue(k + 1);
    f32x4_t acc2[2][2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) cb_conv9<2>(wb, mid + h * 2 * 1024, boff, acc2[h]);
    cb_wait_all();
    __builtin_amdgcn_s_barrier();                      // A: patch k+1 visible, mid free
    asm volatile("" ::: "memory");
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = acc2[m >> 1][m & 1][e >> 2][e & 3] + bb_r[e];
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
  hipLaunchKernelGGL(conv_block32_kernel, dim3((unsigned)blocks), dim3(256), CB_LDS, s, a, tsplit);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace rvb
