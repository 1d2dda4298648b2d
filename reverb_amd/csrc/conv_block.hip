// A whole stride-1 BasicBlock of the ResNet34 trunk's 32-channel stage in ONE kernel (bf16):
//
//     mid = relu(conv3x3_a(x) + b_a)          kept in LDS, never written to HBM
//     out = relu(conv3x3_b(mid) + b_b + x)    the residual x comes out of the input rows that are in LDS anyway
//
// Why (round 5; VERDICT r4 "next" #2): as two conv_stream launches a block moves five tensor passes through HBM (x read, mid
// written, mid read, x read again as the residual, out written) for 37 kFLOP per pixel: 281 GB per hour of audio at 5.2 TB/s --
// that form IS bound by HBM (profiles/archive/r05_call7_conv_block_counters.txt).  Here x is read once and out written once.
//
// Structure (the "band" form): a workgroup of 4 waves owns a 60-frame BAND of one window and walks DOWN its rows, three output rows
// per step, with the input rows and the mid rows in two 8-row rings in LDS (64 pixels x 64 B per row; 2 x 32 KiB).  A step brings
// three NEW input rows (LDS-DMA, requested a step ahead), computes the three NEW mid rows and three output rows; wave w owns the
// 16-pixel column strip w of both convolutions (cb_slide).  No row of either convolution is computed twice, every input row is
// fetched once, ONE barrier per step; 65.8 KB of LDS and 240 VGPRs: two workgroups per CU.
//   bordered rows: input row pb = f + 1 (0 and F + 1 are the zero border); mid row mb = f + 1, ZERO for mb < 1 or mb > F -- the zero
//   border the second convolution sees in the unfused path; columns: patch column c = bordered column t0 - 1 + c, mid column c =
//   frame t0 - 1 + c (ZERO outside [0, T)), output o = frame t0 + o (o < 60)
//   step k = -1 .. K - 1 (K = ceil(F / 3)):   conv_a -> mid rows 3k + 2 .. 3k + 4 from input rows 3k + 1 .. 3k + 5
//                                             conv_b -> output rows (bordered) 3k + 1 .. 3k + 3 from mid rows 3k .. 3k + 4     (k >= 0)
//                                             LDS-DMA of input rows 3k + 6 .. 3k + 8 (step k + 1's new rows) at the top of step k
//   rings: row r lives in slot r & 7.  While a step reads input rows 3k + 1 .. 3k + 5, rows 3k + 6 .. 3k + 8 land: eight consecutive
//   rows, eight slots; a wave that is past the barrier and already writes mid rows 3k + 5 .. 3k + 7 (step k + 1) beside one that still
//   reads 3k .. 3k + 4: eight again.  Rows outside the tensor are clamped for the DMA; what they feed is zeroed or never stored.
//
// What makes it fast (each measured on the way, bit-identical throughout; "ms" = the stage per hour of audio, two launches: 54-56):
//   * operand roles swapped: A = weights (M = channels), B = pixels (N = pixels), and the weight ROWS a lane supplies are chosen
//     so that accumulator row (lane >> 4) * 4 + r of n-tile j is channel (lane >> 4) * 8 + j * 4 + r: a lane ends up holding 8
//     CONSECUTIVE channels of ONE pixel -- exactly the 16 bytes it writes (to mid in LDS, or to the output in HBM).  The first form
//     (pixels x channels accumulators through a per-wave LDS slab, weights and slabs in LDS: 61 ms, SLOWER than two launches)
//     showed that these stages are bound by their LDS operations once HBM is out of the way;
//   * both weight sets live in REGISTERS (2 x 9 taps x 2 n-tiles x 16 B per lane = 144 VGPRs), loaded once per workgroup;
//   * the 16-byte chunks of a pixel are XOR-swizzled with bit 2 of the pixel index (chunk ^= 2 for pixels 4-7 mod 8), on the
//     LDS-DMA's source side for the input rows and on the write side for mid: the plain [pixel][64 B] layout serves a ds_read_b128
//     with a 2-way bank conflict in every lane group; bank-conflict cycles 44 % -> 10 % of the LDS-active cycles.  Together, on
//     tiles of 4 rows x 60 frames (8 waves, two patch buffers): 42.5-43.0 ms;
//   * the two convolutions as ROLES of different waves one tile apart: 42.8 ms -- equal; patches two tiles ahead: 45.4 -- slower
//     (profiles/archive/r05_call5_*, r05_call8_*): neither phase serialisation inside a workgroup nor the prefetch distance was what was left;
//   * FOUR waves and ONE patch buffer per workgroup, so that two workgroups share a CU and fill each other's barriers, DMA waits
//     and VALU phases: 39.3-39.5 ms (profiles/archive/r05_call9_conv_block_two_workgroups.txt);
//   * column strips sliding down the rows of the tile (a fragment read once per input row instead of once per tap: 42 fragment reads
//     per tile and wave instead of 90): 37.7 ms -- the LDS reads were no longer what bound it (profiles/archive/r05_call15_*);
//   * what did: the tile's halo.  Four output rows need six mid rows, 64 columns give 60 outputs: 1.33 x the block's MFMAs.  Walking
//     down a band computes every row once (64 / 60 = 1.07 x): **30.8-31.0 ms** (profiles/archive/r05_call16_conv_block_band.txt).
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

constexpr int CB_OT = 60, CB_PT = 64, CB_NT = 32;         // output frames per band, patch / mid pixels per row, channels
constexpr int CB_ROW = CB_PT * 64;                      // one patch / mid row: 4 096 B
constexpr int CBB_R = 3, CBB_RING = 8;                  // output rows per step; rows per ring
constexpr int CBB_OFF_MID = CBB_RING * CB_ROW + 128;    // 128 B: the two pixels the last strip reads past a ring's last row
constexpr int CBB_LDS = CBB_OFF_MID + CBB_RING * CB_ROW + 128;

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

__device__ inline void cb_wait_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
__device__ inline void cb_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ inline const char* cb_uniform(const char* q) {
  const unsigned long long v = (unsigned long long)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}

__device__ inline void cb_dma1(unsigned off, const void* sbase, unsigned lds0) {
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

// One convolution for NR consecutive output rows of this wave's column strip, sliding down NR + 2 input rows of a ring (input row i
// of the slide is ring row row0 + i).  The fragment of input row i at column shift kw is the operand of tap (kh, kw) of output row
// i - kh for kh = 0, 1, 2: read ONCE, used for up to three output rows.  Every output still accumulates its taps in the order 0 .. 8
// (row i - kh meets input row i at tap row kh, and rows arrive in order).  w: the weight fragments (registers); a[kw]: this lane's
// swizzled byte offset at column shift kw in row 0 of the ring; done(r, acc) runs when row r has its last tap, between the MFMAs of
// the rows that are still open.
template <int NR, typename Done>
__device__ inline void cb_slide(const uint4 (&w)[9][2], const char* ring, const unsigned (&a)[3], int row0, Done&& done) {
  f32x4_t acc[NR][2];
  uint4 xf[2][3];
  auto read_row = [&](int i, int buf) __attribute__((always_inline)) {
    const unsigned ro = (unsigned)((row0 + i) & (CBB_RING - 1)) * CB_ROW;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) xf[buf][kw] = *(const uint4*)(ring + (a[kw] + ro));
  };
  read_row(0, 0);
#pragma unroll
  for (int i = 0; i < NR + 2; ++i) {
    const int cur = i & 1;
    if (i + 1 < NR + 2) read_row(i + 1, cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int kh = 2; kh >= 0; --kh) {
        const int r = i - kh;
        if (r < 0 || r >= NR) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (kh == 0 && kw == 0) acc[r][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
          cb_mma(w[kh * 3 + kw][j], xf[cur][kw], acc[r][j]);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
    if (i >= 2) done(i - 2, acc[i - 2]);
  }
}

__global__ __launch_bounds__(256, 2) void conv_block32_kernel(ConvBlockArgs p) {
  extern __shared__ __attribute__((aligned(16))) char cb_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int FP = p.F + 2, TP = p.T + 2;
  const int bands = (p.T + CB_OT - 1) / CB_OT;
  int lin;
  {
    const int nblk = (int)gridDim.x, q = nblk >> 3, r = nblk & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int band = lin % bands;
  const int b = lin / bands;
  const int t0 = band * CB_OT;
  const char* in_b = cb_uniform((const char*)p.in + (size_t)b * FP * TP * CB_NT * 2);
  char* out_b = (char*)p.out + (size_t)b * FP * TP * CB_NT * 2;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)cb_smem;

  // a step's 12 pieces (3 rows x 4 groups of 16 pixels): wave w brings pieces 3 w .. 3 w + 2; the column part of their source
  // offsets never changes (the band is fixed), the row part is uniform
  const int ppx = lane >> 2;
  const unsigned piece_b = (unsigned)((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 16;
  unsigned colterm[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int g = (wave * 3 + i) & 3;
    colterm[i] = (unsigned)min(max(t0 - 1 + g * 16 + ppx, 0), TP - 1) * (CB_NT * 2) + piece_b;
  }
  auto issue = [&](int k) __attribute__((always_inline)) {          // the new patch rows of step k: 3 k + 3 .. 3 k + 5
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int q = wave * 3 + i, pb = CBB_R * k + 3 + (q >> 2);
      const unsigned rowterm = (unsigned)(min(pb, FP - 1) * TP) * (CB_NT * 2);
      cb_dma1(colterm[i] + rowterm, in_b, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(pb & (CBB_RING - 1)) * CB_ROW + (unsigned)(q & 3) * 1024));
    }
  };
  issue(-1);
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
  unsigned soff[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) soff[kw] = cb_swz((unsigned)(wave * 16 + li + kw), lg);
  const unsigned moff = cb_swz((unsigned)(wave * 16 + li), lg);
  const unsigned roff = cb_swz((unsigned)(wave * 16 + li + 2), lg);
  const char* patch = cb_smem;
  char* mid = cb_smem + CBB_OFF_MID;
  const int o = wave * 16 + li;
  const int tm = t0 - 1 + o;                                // the mid column's frame
  const bool tm_in = tm >= 0 && tm < p.T;
  const int t = t0 + o;
  const bool t_ok = o < CB_OT && t < p.T;
  const int K = (p.F + CBB_R - 1) / CBB_R;
  cb_wait_all();
  __syncthreads();

  for (int k = -1; k < K; ++k) {
    if (k + 1 < K) issue(k + 1);
    const int m0 = CBB_R * k + 2;                           // first new mid row
    cb_slide<CBB_R>(wa, patch, soff, m0 - 1, [&](int r, f32x4_t (&acc)[2]) __attribute__((always_inline)) {
      const int mb = m0 + r;
      const bool inside = tm_in && mb >= 1 && mb <= p.F;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = inside ? fmaxf(acc[e >> 2][e & 3] + ba_r[e], 0.f) : 0.f;
      *(uint4*)(mid + (moff + (unsigned)(mb & (CBB_RING - 1)) * CB_ROW)) =
          make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
    });
    const int ob0 = CBB_R * k + 1;                          // first output row (bordered) of the step
    cb_u32x4 rp[CBB_R];
    if (k >= 0) {
#pragma unroll
      for (int q = 0; q < CBB_R; ++q) rp[q] = *(const cb_u32x4*)(patch + (roff + (unsigned)((ob0 + q) & (CBB_RING - 1)) * CB_ROW));
    }
    cb_wait_all();                                     // mid written, residuals read, this wave's pieces of the next rows landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (k < 0) continue;
    cb_slide<CBB_R>(wb, mid, soff, ob0 - 1, [&](int q, f32x4_t (&acc)[2]) __attribute__((always_inline)) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = acc[e >> 2][e & 3] + bb_r[e];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] += __uint_as_float(rp[q][e] << 16);
        v[2 * e + 1] += __uint_as_float(rp[q][e] & 0xffff0000u);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      const int ob = ob0 + q;
      if (t_ok && ob <= p.F)
        *(uint4*)(out_b + ((size_t)((unsigned)ob * TP + t + 1) * CB_NT + lg * 8) * 2) =
            make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
    });
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
  static bool attr_b = false;
  if (!attr_b) {
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)conv_block32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CBB_LDS));
    attr_b = true;
  }
  const int64_t wgs = (int64_t)a.B * cdiv(a.T, CB_OT);
  if (wgs >= ((int64_t)1 << 31)) { set_error("conv_block32: too many workgroups"); return E_ARG; }
  hipLaunchKernelGGL(conv_block32_kernel, dim3((unsigned)wgs), dim3(256), CBB_LDS, s, a);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace rvb
