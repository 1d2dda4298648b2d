// 3x3 / stride 1 / pad 1 convolutions of the 32- and 64-channel stages of the ResNet34 trunk (bf16) as a STREAM: a workgroup
// owns 4 mel rows of one window and walks their 62-frame tiles; the weights of all 9 taps stay in LDS for the whole walk,
// the 6 x 64 pixel input patches arrive by LDS-DMA one or two tiles ahead of the MFMAs, the residual rows of a tile are
// requested before its patch is multiplied, and the stores of a tile drain underneath the next one.
//
// Why (round 4): resnet.hip's conv_kernel gives every 4 x 64 pixel tile its own workgroup -- {load patch + 9 taps' weights
// into registers, ds_write, barrier, 72-288 MFMAs per wave, barrier, residual loads, transposes, stores} with nothing of
// one tile overlapping anything of the next except through the 2-3 workgroups a CU holds.  The 32-channel stage ran at
// 3.2 TB/s and 19 % of the MFMA peak, the 64-channel stage at 28 % (profiles/r04b_rocprofv3_kernel_stats_diar_1h.csv:
// 596 / 399 us per launch of 189 windows), and 40 % of the bytes the 32-channel kernel pulled through the L2 were the same
// 18 KiB of weights again for every tile.
//
// Items.  The unit of the stream is (tile, 64-byte channel chunk): 6 rows x 64 pixels x 64 B = 24 KiB = 6 DMA instructions
// per thread (row i of the patch, pixels [16 wave, +16), one 16-byte piece per lane -- 1 KiB per instruction, lane order,
// which is what `global_load_lds_dwordx4` writes).  A tile is 62 output frames wide so that its patch is exactly 64 pixels:
// the m-tiles still cover 64 positions, positions 62 / 63 read two pixels past the row (the next row's first pixels, or
// the pad behind the last buffer) and are never stored.  Pixels beyond the bordered plane are clamped to its last row /
// column: they only feed outputs that are not stored either.
//
//   NT = Cin = Cout = 32: 1 chunk,  2 patch buffers (one tile ahead), 76 928 B of LDS -> two workgroups per CU
//   NT = Cin = Cout = 64: 2 chunks, 3 patch buffers (two items ahead), 156 800 B     -> one workgroup per CU
//
// Ordering.  Every vector-memory instruction whose latency matters is inline asm, so hipcc's waitcnt pass sees none of them
// (it drains vmcnt(0) in front of a ds_read behind an LDS-DMA, and any wait it places for a load result also drains the
// stores issued before it -- one queue): the kernel counts vmcnt itself.  Iteration k =
//   {residual loads of this tile (last chunk only); DMA of item k + PD; multiply item k; vmcnt(6 (PD - 1)) -- everything
//    older than the DMA just issued has landed: item k + 1's pieces, the residual vectors, the previous tile's stores;
//    s_barrier; (last chunk) epilogue + stores}.
// RAW: a wave waits for its own pieces of item k + 1 before the barrier that ends iteration k; readers are past it.
// WAR: buffer (k + PD) % NBUF = (k - 1) % NBUF was last read in iteration k - 1, whose MFMAs consumed those reads before
// its barrier; the DMA is issued after that barrier.  The transposition slabs are private to a wave.
//
// Results: accumulation order (chunks outer, taps inner), operand values and rounding points are conv_kernel's --
// bit-identical (tests/test_diar_gpu.py compares the two).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace rvb {

namespace {

constexpr int CS_OT = 62, CS_PT = 64, CS_PF = 6, CS_OF = 4;
constexpr int CS_PATCH = CS_PF * CS_PT * 64;      // 24 576 B
constexpr int CS_SROW = 32 * 4 + 16;              // fp32 slab row of 32 channels, padded
constexpr int CS_SLAB = 16 * CS_SROW;             // per wave

typedef unsigned cs_u32x4 __attribute__((ext_vector_type(4)));

__device__ inline void cs_mma(const uint4& a, const uint4& b, f32x4_t& c) {
  union U { uint4 u; bf16x8_t v; };
  U ua, ub;
  ua.u = a; ub.u = b;
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, c, 0, 0, 0);
}

// six 1-KiB LDS-DMA pieces: patch rows 0..5 (4 KiB apart in LDS), per-lane 32-bit byte offsets from a scalar base
__device__ inline void cs_dma6(const unsigned (&off)[6], const void* sbase, unsigned lds0) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %8\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %7\n\t"
      "s_add_u32 m0, m0, 0x1000\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %7\n\t"
      "s_add_u32 m0, m0, 0x1000\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %3, %7\n\t"
      "s_add_u32 m0, m0, 0x1000\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %4, %7\n\t"
      "s_add_u32 m0, m0, 0x1000\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %5, %7\n\t"
      "s_add_u32 m0, m0, 0x1000\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %6, %7\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "s"(sbase), "s"(lds0)
      : "memory", "scc");
}
__device__ inline void cs_dma1(unsigned off, const void* sbase, unsigned lds) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(off), "s"(sbase), "s"(lds)
      : "memory");
}
__device__ inline cs_u32x4 cs_load16(unsigned off, const void* sbase) {
  cs_u32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(off), "s"(sbase) : "memory");
  return r;
}
// ... and every LDS read of this wave has returned (in front of a barrier behind which its buffer is requested again)
template <int N> __device__ inline void cs_wait_vmlgkm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
template <int N> __device__ inline void cs_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ inline const char* cs_uniform(const char* q) {
  const unsigned long long v = (unsigned long long)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}

// The residual vectors are loaded by inline asm and become valid at the counted wait: this empty statement, placed right
// behind that wait, makes every later use of them depend on it (one statement on the merged path -- with the registers as
// operands of two alternative wait statements the allocator copied them in front of one of the two, i.e. read them in flight)
template <int NH> __device__ inline void cs_touch_rp(cs_u32x4 (&rp)[4][NH]) {
  if constexpr (NH == 1)
    asm volatile("" : "+v"(rp[0][0]), "+v"(rp[1][0]), "+v"(rp[2][0]), "+v"(rp[3][0])::"memory");
  else
    asm volatile(""
                 : "+v"(rp[0][0]), "+v"(rp[1][0]), "+v"(rp[2][0]), "+v"(rp[3][0]), "+v"(rp[0][NH - 1]), "+v"(rp[1][NH - 1]), "+v"(rp[2][NH - 1]),
                   "+v"(rp[3][NH - 1])::"memory");
}

template <int NT, int NCH, int NBUF>
struct CsLds {
  static constexpr int W = NCH * 9 * NT * 64;
  static constexpr int BUF0 = W;
  static constexpr int SLAB0 = W + NBUF * CS_PATCH + 128;
  static constexpr int TOTAL = SLAB0 + 4 * CS_SLAB;
};

template <int NT, int NCH, int NBUF>
__global__ __launch_bounds__(256, NT == 32 ? 2 : 1) void conv_stream_kernel(ConvArgs p, int tsplit) {
  extern __shared__ __attribute__((aligned(16))) char cs_smem[];
  using L = CsLds<NT, NCH, NBUF>;
  constexpr int PD = NBUF - 1;                   // items in flight ahead of the one being multiplied
  constexpr int NJ = NT / 16, NH = NT / 32;
  constexpr int CIN = NCH * 32;
  static_assert(PD == 1 || PD == 2, "the counted waits below enumerate one and two items ahead");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int FP = p.Fi + 2, TP = p.Ti + 2;
  const int tiles_f = (p.Fo + CS_OF - 1) / CS_OF, tiles_t = (p.To + CS_OT - 1) / CS_OT;
  const int per = (tiles_t + tsplit - 1) / tsplit;

  // workgroup -> (window, mel-row tile, part of the time axis).  Each XCD (workgroup id mod 8) takes a contiguous run of
  // the linear order, so that the two workgroups that share a halo row run on the same L2 at about the same time.
  int lin;
  {
    const int nblk = (int)gridDim.x, q = nblk >> 3, r = nblk & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int sp = lin % tsplit;
  const int tf = (lin / tsplit) % tiles_f;
  const int b = lin / (tsplit * tiles_f);
  const int f0 = tf * CS_OF;
  const int tt0 = sp * per, tt1 = min(tiles_t, tt0 + per);
  const int n_items = (tt1 - tt0) * NCH;
  if (n_items <= 0) return;

  const char* in_b = cs_uniform((const char*)p.in + (size_t)b * FP * TP * CIN * 2);
  const char* res_b = p.res ? cs_uniform((const char*)p.res + (size_t)b * FP * TP * NT * 2) : nullptr;
  char* out_b = (char*)p.out + (size_t)b * FP * TP * NT * 2;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)cs_smem;

  // ---- weights: LDS [chunk][tap][n][64 B] <- global [tap][chunk][Cout][64 B] (Cout == NT), once per workgroup, by LDS-DMA:
  // piece pc = 16 rows x 64 B (inside one (chunk, tap) block, NT % 16 == 0); wave w takes pieces w, w + 4, ...
  {
    constexpr int NPC = NCH * 9 * NT / 16;
    const char* wg = cs_uniform((const char*)p.w);
#pragma unroll
    for (int i = 0; i < (NPC + 3) / 4; ++i) {
      const int pc = i * 4 + wave;
      if (pc < NPC) {
        const int row0 = pc * 16, ct = row0 / NT, n = row0 - ct * NT + (lane >> 2);
        const int tap = ct % 9, c = ct / 9;
        const unsigned off = (unsigned)((((tap * NCH + c) * NT + n) * 4 + (lane & 3)) * 16);
        cs_dma1(off, wg, __builtin_amdgcn_readfirstlane(lds_base + pc * 1024));
      }
    }
  }
  // ---- DMA coordinates of this thread: rows f0 .. f0+5 of the bordered plane (clamped), pixel tid >> 2, piece tid & 3
  unsigned rowoff[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) rowoff[i] = (unsigned)(min(f0 + i, FP - 1) * TP) * (CIN * 2);
  const int ppx = tid >> 2;
  const unsigned piece_b = (unsigned)(tid & 3) * 16;
  auto issue = [&](int k) __attribute__((always_inline)) {
    const int tt = tt0 + k / NCH, c = k % NCH;
    const unsigned pix = (unsigned)min(tt * CS_OT + ppx, TP - 1) * (CIN * 2) + (unsigned)c * 64 + piece_b;
    unsigned off[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) off[i] = rowoff[i] + pix;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + L::BUF0 + (k % NBUF) * CS_PATCH + wave * 1024);
    cs_dma6(off, in_b, dst);
  };

  // ---- epilogue coordinates: lane = (pixel spx of a 16-pixel slab, 8-channel segment)
  const int spx = lane >> 2, sch = (lane & 3) * 8;
  const int f = f0 + wave;
  float bias_r[NH][8];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
    if (p.bias) { b0 = *(const float4*)(p.bias + h * 32 + sch); b1 = *(const float4*)(p.bias + h * 32 + sch + 4); }
    bias_r[h][0] = b0.x; bias_r[h][1] = b0.y; bias_r[h][2] = b0.z; bias_r[h][3] = b0.w;
    bias_r[h][4] = b1.x; bias_r[h][5] = b1.y; bias_r[h][6] = b1.z; bias_r[h][7] = b1.w;
  }
  char* slab = cs_smem + L::SLAB0 + wave * CS_SLAB;
  const unsigned frow_off = (unsigned)(min(f, p.Fo - 1) + 1) * TP;        // bordered row of this wave's outputs (clamped)

  f32x4_t acc[4][NJ];
  cs_u32x4 rp[4][NH];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int h = 0; h < NH; ++h) rp[mi][h] = (cs_u32x4){0u, 0u, 0u, 0u};

  // ---- prologue: the first PD items requested; weights visible; item 0 landed
#pragma unroll
  for (int k = 0; k < PD; ++k)
    if (k < n_items) issue(k);
  if (PD == 2 && n_items > 1) cs_wait_vm<6>(); else cs_wait_vm<0>();
  __syncthreads();

  auto item = [&](auto cc, int k) __attribute__((always_inline)) {
    constexpr int c = decltype(cc)::value;
    constexpr bool first = c == 0, last = c == NCH - 1;
    const int tt = tt0 + k / NCH;
    const int t0 = tt * CS_OT;
    if constexpr (first) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[mi][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    // residual vectors of this tile: in flight under its last chunk's MFMAs, older than the DMA issued below
    if constexpr (last) {
      if (res_b) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const unsigned t = (unsigned)min(t0 + mi * 16 + spx, p.To - 1) + 1;
          const unsigned off = ((frow_off + t) * NT + sch) * 2;
#pragma unroll
          for (int h = 0; h < NH; ++h) rp[mi][h] = cs_load16(off + h * 64, res_b);
        }
      }
    }
    const bool more = k + PD < n_items;
    if (more) issue(k + PD);

    // ---- 9 taps of this chunk: the next tap's fragments are read before the current tap's MFMAs are issued
    {
      const char* sB = cs_smem + (c * 9) * NT * 64 + li * 64 + lg * 16;
      const char* sA = cs_smem + L::BUF0 + (k % NBUF) * CS_PATCH + (wave * CS_PT + li) * 64 + lg * 16;
      uint4 bf[2][NJ], af[2][4];
      auto read_frags = [&](int tap, int buf) __attribute__((always_inline)) {
        const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[buf][j] = *(const uint4*)(sB + (tap * NT + j * 16) * 64);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[buf][mi] = *(const uint4*)(sA + (kh * CS_PT + kw + mi * 16) * 64);
      };
      read_frags(0, 0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int cur = tap & 1;
        if (tap + 1 < 9) read_frags(tap + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);          // the reads of tap + 1 are issued before the MFMAs of tap, not among them
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int j = 0; j < NJ; ++j) cs_mma(af[cur][mi], bf[cur][j], acc[mi][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // everything older than the DMA just issued has landed (this wave's pieces of item k + 1, the residual vectors)
    if (PD == 2 && more) cs_wait_vmlgkm<6>(); else cs_wait_vmlgkm<0>();
    if constexpr (last) cs_touch_rp<NH>(rp);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    if constexpr (last) {
      // ---- epilogue: 16 pixels x 32 channels per pass through the wave's slab; a lane then owns 8 consecutive channels of
      // one pixel: bias + residual + ReLU in fp32, one 16-byte store
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) *(float*)(slab + (lg * 4 + r) * CS_SROW + (j * 16 + li) * 4) = acc[mi][h * 2 + j][r];
          __builtin_amdgcn_wave_barrier();
          const float4 x0 = *(const float4*)(slab + spx * CS_SROW + sch * 4);
          const float4 x1 = *(const float4*)(slab + spx * CS_SROW + sch * 4 + 16);
          float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bias_r[h][e];
          if (res_b) {
            const cs_u32x4 raw = rp[mi][h];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * e] += __uint_as_float(raw[e] << 16);
              v[2 * e + 1] += __uint_as_float(raw[e] & 0xffff0000u);
            }
          }
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          const int o = mi * 16 + spx, t = t0 + o;
          if (o < CS_OT && t < p.To && f < p.Fo)
            *(uint4*)(out_b + ((size_t)(frow_off + t + 1) * NT + h * 32 + sch) * 2) =
                make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
        }
    }
  };

  for (int k = 0; k < n_items; k += NCH) {
    item(std::integral_constant<int, 0>(), k);
    if constexpr (NCH == 2) item(std::integral_constant<int, 1>(), k + 1);
  }
  cs_wait_vm<0>();
}

template <int NT, int NCH, int NBUF>
int launch_stream(hipStream_t st, const ConvArgs& p, int tsplit) {
  using L = CsLds<NT, NCH, NBUF>;
  auto kern = conv_stream_kernel<NT, NCH, NBUF>;
  static bool attr_set = false;
  if (!attr_set) {
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L::TOTAL));
    attr_set = true;
  }
  const int tiles_t = cdiv(p.To, CS_OT);
  tsplit = std::max(1, std::min(tsplit, tiles_t));
  const int64_t blocks = (int64_t)p.B * cdiv(p.Fo, CS_OF) * tsplit;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), L::TOTAL, st, p, tsplit);
  RVB_HIP_CHECK(hipGetLastError());
  return OK;
}

int stream_mode() {       // RVD_CONV_STREAM: 0 = off, n >= 1 = on with the time axis of a row split over n workgroups
  const char* e = lab_env("RVD_CONV_STREAM");        // read per call: the tests switch it between engines
  return e ? atoi(e) : CONV_STREAM_DEFAULT;
}

}  // namespace

bool conv_stream_applicable(int dtype, const ConvArgs& p) {
  if (stream_mode() <= 0) return false;
  if (p.Cin == 64) { const char* e = lab_env("RVD_CONV_STREAM64"); if (!(e && atoi(e) == 1)) return false; }
  return dtype == DT_BF16 && p.taps == 9 && p.stride == 1 && p.Cin == p.Cout && (p.Cin == 32 || p.Cin == 64) && p.Fo == p.Fi && p.To == p.Ti &&
         (int64_t)(p.Fi + 2) * (p.Ti + 2) * p.Cin * 2 < ((int64_t)1 << 31) && (int64_t)p.B * cdiv(p.Fo, CS_OF) * 64 < ((int64_t)1 << 31);
}

int conv_stream(hipStream_t s, const ConvArgs& p) {
  if (p.B <= 0) return OK;
  const int ts = stream_mode();
  return p.Cin == 32 ? launch_stream<32, 1, 2>(s, p, ts) : launch_stream<64, 2, 3>(s, p, ts);
}

}  // namespace rvb
