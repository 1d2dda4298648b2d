// Centroid-linkage agglomerative clustering of the speaker embeddings on the GPU.
//
// pyannote's AgglomerativeClustering (the clustering step of the pipeline the reference runs,
// diarization/infer_pyannote3.0.py:40) calls scipy.cluster.hierarchy.linkage(X, "centroid", "euclidean")
// on up to 3 embeddings per 1 s hop: ~10^4 points for an hour of audio, where scipy needs ~7 s for pdist
// and ~3 s for the merge loop on one host core.  Here the fp64 distance matrix is built by a tiled kernel
// and the merge loop -- scipy's `fast_linkage` (Muellner's generic algorithm: nearest-neighbour
// candidates with lazy validation, Lance-Williams centroid update) restated step for step so that the
// dendrogram is the same -- runs as ONE persistent 1024-thread workgroup: every merge is a block-wide
// argmin over the candidate distances plus one fused pass over the two merged rows.
#include <cstdio>
#include <cstdlib>

#include "kernels.h"

namespace rvb {

// ------------------------------------------------------------------------------------ pairwise distances (fp64)
// D[i][j] = sqrt(sum_k (X[i][k] - X[j][k])^2), full symmetric n x n.  64x64 tile per block, 4x4 per thread.
__global__ __launch_bounds__(256) void pdist_kernel(const double* __restrict__ X, int n, int d, double* __restrict__ D) {
  __shared__ double sa[16][65], sb[16][65];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  if (j0 + 63 < i0) return;     // strictly-lower tiles are filled by symmetry
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (int k0 = 0; k0 < d; k0 += 16) {
    for (int v = threadIdx.x; v < 64 * 16; v += 256) {
      const int r = v >> 4, kk = v & 15;
      const int gi = i0 + r, gj = j0 + r, gk = k0 + kk;
      sa[kk][r] = (gi < n && gk < d) ? X[(size_t)gi * d + gk] : 0.0;
      sb[kk][r] = (gj < n && gk < d) ? X[(size_t)gj * d + gk] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      double av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = sa[kk][ty * 4 + a];
#pragma unroll
      for (int b = 0; b < 4; ++b) bv[b] = sb[kk][tx * 4 + b];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { const double df = av[a] - bv[b]; acc[a][b] = fma(df, df, acc[a][b]); }
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = i0 + ty * 4 + a, j = j0 + tx * 4 + b;
      if (i < n && j < n && i <= j) {
        const double v = i == j ? 0.0 : sqrt(acc[a][b]);
        D[(size_t)i * n + j] = v;
        D[(size_t)j * n + i] = v;
      }
    }
}

// ------------------------------------------------------------------------------------ block-wide (value, index) argmin
struct MinPair { double v; int i; };
__device__ inline MinPair min_pair(MinPair a, MinPair b) { return (b.v < a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
// wave-wide argmin without LDS: `__shfl_xor` is a ds_bpermute per 32-bit piece (18 LDS round trips for a (double, int) pair,
// on the merge loop's critical path six times per merge); here the four steps inside a 16-lane row are DPP moves
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror -- min is commutative, any pairing that unifies the row works) and
// the two steps across rows are gfx950's v_permlane16_swap / v_permlane32_swap.
template <int CTRL> __device__ inline MinPair dpp_pair(const MinPair& p) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(p.v);
  const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)bits, CTRL, 0xf, 0xf, false);
  const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(bits >> 32), CTRL, 0xf, 0xf, false);
  MinPair q;
  q.v = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  q.i = (int)__builtin_amdgcn_update_dpp(0u, (unsigned)p.i, CTRL, 0xf, 0xf, false);
  return q;
}
// The value is reduced first (fp64 min: two DPP moves + v_min_f64 per step), then the smallest index among the lanes that
// hold that value (integer min) -- the same (value, index) lexicographic minimum as comparing pairs, in about 30
// instructions instead of 66.  The merge loop is VALU-issue-bound on its one CU (16 waves share 4 SIMDs and every wave
// runs every reduction), so instruction count is what a reduction costs.
template <int CTRL> __device__ inline double dpp_f64(double v) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)bits, CTRL, 0xf, 0xf, false);
  const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(bits >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int CTRL> __device__ inline int dpp_i32(int v) { return (int)__builtin_amdgcn_update_dpp(0u, (unsigned)v, CTRL, 0xf, 0xf, false); }
__device__ inline double min_f64(double a, double b) { return a < b ? a : b; }      // no NaNs here: distances or +inf
// minimum over the 16 lanes of each DPP row (every lane of the row ends up with it)
__device__ inline MinPair row_argmin(MinPair p) {
  double m = p.v;
  m = min_f64(m, dpp_f64<0xB1>(m));       // quad_perm [1,0,3,2]
  m = min_f64(m, dpp_f64<0x4E>(m));       // quad_perm [2,3,0,1]
  m = min_f64(m, dpp_f64<0x141>(m));      // row_half_mirror
  m = min_f64(m, dpp_f64<0x140>(m));      // row_mirror
  int i = p.v == m ? p.i : 0x7fffffff;
  i = min(i, dpp_i32<0xB1>(i));
  i = min(i, dpp_i32<0x4E>(i));
  i = min(i, dpp_i32<0x141>(i));
  i = min(i, dpp_i32<0x140>(i));
  return MinPair{m, i};
}
__device__ inline MinPair wave_argmin(MinPair p) {
  p = row_argmin(p);
  // across the four rows: gfx950's v_permlane16_swap / v_permlane32_swap hand every lane both halves
  {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(p.v);
    auto a = __builtin_amdgcn_permlane16_swap((unsigned)bits, (unsigned)bits, false, false);
    auto b = __builtin_amdgcn_permlane16_swap((unsigned)(bits >> 32), (unsigned)(bits >> 32), false, false);
    auto c = __builtin_amdgcn_permlane16_swap((unsigned)p.i, (unsigned)p.i, false, false);
    MinPair u{__longlong_as_double((long long)(((unsigned long long)b[0] << 32) | a[0])), (int)c[0]};
    MinPair w{__longlong_as_double((long long)(((unsigned long long)b[1] << 32) | a[1])), (int)c[1]};
    p = min_pair(u, w);
  }
  {
    const unsigned long long bits2 = (unsigned long long)__double_as_longlong(p.v);
    auto a = __builtin_amdgcn_permlane32_swap((unsigned)bits2, (unsigned)bits2, false, false);
    auto b = __builtin_amdgcn_permlane32_swap((unsigned)(bits2 >> 32), (unsigned)(bits2 >> 32), false, false);
    auto c = __builtin_amdgcn_permlane32_swap((unsigned)p.i, (unsigned)p.i, false, false);
    MinPair u{__longlong_as_double((long long)(((unsigned long long)b[0] << 32) | a[0])), (int)c[0]};
    MinPair w{__longlong_as_double((long long)(((unsigned long long)b[1] << 32) | a[1])), (int)c[1]};
    p = min_pair(u, w);
  }
  return p;
}

__device__ inline MinPair block_argmin(MinPair p, MinPair* red) {
  p = wave_argmin(p);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wv] = p;
  __syncthreads();
  MinPair r = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = min_pair(r, red[w]);
  return r;
}

// one-barrier variant for back-to-back reductions: the caller alternates between two result buffers, so the writes of
// reduction k+1 cannot race with the reads of reduction k (those are separated by reduction k+1's own barrier from k+2).
// Second level: lane l reads wave (l mod 16)'s result and the 16 values are reduced inside each DPP row -- one LDS read
// and ~25 instructions per wave instead of a 16-step serial loop.  Buffers hold 16 entries; unused ones stay +inf.
__device__ inline MinPair block_argmin_alt(MinPair p, MinPair (*red)[16], int& phase) {
  p = wave_argmin(p);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  MinPair* buf = red[phase & 1];
  ++phase;
  if (lane == 0) buf[wv] = p;
  __syncthreads();
  return row_argmin(buf[lane & 15]);
}

// three reductions behind ONE barrier (the multi-workgroup loop: best valid candidate, smallest stale bound, and the share of the
// new cluster's nearest neighbour left over from the merge pass): a block reduction is ~0.7 us of DPP steps + LDS round trip +
// barrier, and a round had three of them back to back (profile: 1.8 us per round in "local minima", RVD_LINKAGE_PROF).
// Buffers hold 48 entries, the unused ones stay +inf.
// Only WAVE 0 holds the results afterwards (the one thread that publishes them reads them there): the second stage is 3 x 40
// VALU instructions that 16 waves would otherwise all issue, and a wave whose lanes are all +inf (most of them, for the stale
// bounds; all of them in the waves beyond the workgroup's own slots) skips its first stage on one ballot -- the merge loop is
// VALU-issue-bound between its barriers (profile: 1.9 us per round in these reductions).
__device__ inline void block_argmin3_alt(MinPair& a, MinPair& b, MinPair& c, MinPair (*red)[48], int& phase) {
  const MinPair none{INFINITY, 0x7fffffff};
  a = __ballot(a.v < INFINITY) ? wave_argmin(a) : none;
  b = __ballot(b.v < INFINITY) ? wave_argmin(b) : none;
  c = __ballot(c.v < INFINITY) ? wave_argmin(c) : none;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  MinPair* buf = red[phase & 1];
  ++phase;
  if (lane == 0) { buf[wv] = a; buf[16 + wv] = b; buf[32 + wv] = c; }
  __syncthreads();
  if (wv == 0) {
    a = row_argmin(buf[lane & 15]);
    b = row_argmin(buf[16 + (lane & 15)]);
    c = row_argmin(buf[32 + (lane & 15)]);
  }
}

// nearest neighbour of every x among y > x (scipy find_min_dist: first index on ties); one block per row
__global__ __launch_bounds__(256) void nn_init_kernel(const double* __restrict__ D, int n, int* __restrict__ neighbor,
                                                      double* __restrict__ min_dist) {
  __shared__ MinPair red[4];
  const int x = blockIdx.x;
  MinPair p{INFINITY, 0x7fffffff};
  for (int j = x + 1 + threadIdx.x; j < n; j += 256) p = min_pair(p, MinPair{D[(size_t)x * n + j], j});
  p = block_argmin(p, red);
  if (threadIdx.x == 0) { neighbor[x] = p.v < INFINITY ? p.i : -1; min_dist[x] = p.v; }
}

// Lance-Williams centroid update, scipy's `_centroid` with the two divisions of a merge hoisted: sub = nx ny d(x,y)^2 / (nx + ny)
// and inv_fs = 1 / (nx + ny) are formed once per merge.  One definition for both merge loops, contraction off, so that the
// single-workgroup kernel and the multi-workgroup kernel round identically (their dendrograms are compared bit for bit).
__device__ inline double lw_centroid(double fx, double fy, double dxi, double dyi, double sub, double inv_fs) {
#pragma clang fp contract(off)
  return sqrt((((fx * dxi * dxi) + (fy * dyi * dyi)) - sub) * inv_fs);
}

// ------------------------------------------------------------------------------------ merge loop (one workgroup)
// LDS_STATE: the per-cluster state (candidate distance fp64, candidate neighbour, cluster size) lives in LDS
// (14 bytes per point: up to ~11 500 points = one hour of audio at 3 speakers per hop), so an iteration
// touches HBM/L2 only for the two merged rows (and the row of a stale candidate).
//
// Ownership: thread t owns the slots z = t (mod 1024) -- it is the only writer of their state and the one that visits
// them in the merge pass and in a row rescan.  Each thread keeps the minimum of ITS candidates in registers (`loc`), so
// a candidate round is one block reduction of register values (one barrier) instead of a scan of the LDS table; the
// owner re-reads its <= n/1024 entries only when the entry that was its minimum goes up (dropped row, rescanned row,
// merged row).  Every thread derives (x, y, dist) from the same reduction result, so nothing is broadcast through
// shared variables, and state writes sit right after a reduction's barrier: 1 barrier per candidate round, per row
// rescan and per merge.
static constexpr int LK_E = 10;      // elements per thread and batch (registers: 2 x LK_E doubles in the merge pass)
// COMPACT (LDS_STATE, n <= LK_E * NTH): the slots a thread owns are re-dealt from the sorted list of live slots every 256
// merges (registers `myz`), so that the passes walk ~the live clusters instead of all n slots -- dead slots are skipped per
// lane anyway, but a wave instruction costs the same whether 64 or 20 of its lanes are alive, and the merge pass is
// VALU-issue-bound.  Order is preserved (ascending slots per list position), so ties break as before.
template <bool LDS_STATE, int NTH, bool COMPACT>
__global__ __launch_bounds__(NTH) void linkage_kernel(int getenv_prof, double* __restrict__ D, int n, uint16_t* __restrict__ g_size,
                                                       int* __restrict__ cluster_id, int* __restrict__ g_neighbor,
                                                       double* __restrict__ g_min_dist, double* __restrict__ Z) {
  extern __shared__ __attribute__((aligned(16))) char lk_smem[];
  __shared__ MinPair red[2][16];
  int rphase = 0;
  double* s_md = (double*)lk_smem;
  int* s_nb = (int*)(s_md + (LDS_STATE ? n : 0));
  uint16_t* s_sz = (uint16_t*)(s_nb + (LDS_STATE ? n : 0));
  auto MD = [&](int i) -> double& { if constexpr (LDS_STATE) return s_md[i]; else return g_min_dist[i]; };
  auto NB = [&](int i) -> int& { if constexpr (LDS_STATE) return s_nb[i]; else return g_neighbor[i]; };
  auto SZ = [&](int i) -> uint16_t& { if constexpr (LDS_STATE) return s_sz[i]; else return g_size[i]; };
  uint16_t* s_list = s_sz + (LDS_STATE ? n : 0);          // COMPACT: sorted live slots (rebuild scratch)
  __shared__ int s_wtot[16];
  const int tid = threadIdx.x;
  if (tid < 32) red[tid >> 4][tid & 15] = MinPair{INFINITY, 0x7fffffff};
  long long retries = 0;
  long long t_arg = 0, t_scan = 0, t_merge = 0, t0c = 0;
  const bool prof = getenv_prof & 1;
  if (LDS_STATE) {
    for (int i = tid; i < n; i += NTH) { s_md[i] = g_min_dist[i]; s_nb[i] = g_neighbor[i]; s_sz[i] = g_size[i]; }
  }
  __syncthreads();
  int myz[COMPACT ? LK_E : 1];                // COMPACT: this thread's slots (-1 = none); else slot = base + tid + e * NTH
  int ecount = LK_E;                          // COMPACT: batches of NTH list positions in use (uniform)
  auto slot = [&](int base, int e) -> int {
    if constexpr (COMPACT) { (void)base; return myz[e]; }
    else { const int z = base + tid + e * NTH; return z < n ? z : -1; }
  };
  const int span = COMPACT ? 1 : n;           // COMPACT: one batch covers everything
  auto own_min = [&]() {                      // dropped rows hold +inf; slot n-1 has no candidate (no y > n-1)
    MinPair m{INFINITY, 0x7fffffff};
    for (int base = 0; base < span; base += LK_E * NTH) {
#pragma unroll
      for (int e = 0; e < LK_E; ++e) {
        if (COMPACT && e >= ecount) break;
        const int z = slot(base, e);
        if (z >= 0 && z < n - 1) m = min_pair(m, MinPair{MD(z), z});
      }
    }
    return m;
  };
  auto owns = [&](int z) -> bool {
    if constexpr (COMPACT) {
      bool o = false;
#pragma unroll
      for (int e = 0; e < LK_E; ++e) o |= myz[e] == z;
      return o;
    } else {
      return tid == (z & (NTH - 1));
    }
  };
  MinPair loc{INFINITY, 0x7fffffff};
  auto rebuild = [&]() {                      // COMPACT: deal the live slots out again, ascending, position p -> thread p % NTH
    if constexpr (COMPACT) {
      __syncthreads();                        // the previous merge's owner writes
      const int lane = tid & 63, wv = tid >> 6;
      const int C = (n + NTH - 1) / NTH, z0 = tid * C;
      int c = 0;
      for (int j = 0; j < C; ++j) { const int z = z0 + j; if (z < n && SZ(z) != 0) ++c; }
      int incl = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
      if (lane == 63) s_wtot[wv] = incl;
      __syncthreads();
      int woff = 0, total = 0;
      for (int w = 0; w < NTH / 64; ++w) { const int t = s_wtot[w]; if (w < wv) woff += t; total += t; }
      int pos = woff + incl - c;
      for (int j = 0; j < C; ++j) { const int z = z0 + j; if (z < n && SZ(z) != 0) s_list[pos++] = (uint16_t)z; }
      __syncthreads();
      ecount = (total + NTH - 1) / NTH;
#pragma unroll
      for (int e = 0; e < LK_E; ++e) { const int p = tid + e * NTH; myz[e] = p < total ? (int)s_list[p] : -1; }
      __syncthreads();
    }
  };
  if constexpr (!COMPACT) loc = own_min();
  const long long cyc0 = clock64(), wc0 = wall_clock64();
  for (int k = 0; k < n - 1; ++k) {
    if (COMPACT && (k & 255) == 0) { rebuild(); loc = own_min(); }
    // ---- closest valid candidate pair (lazy validation of the nearest-neighbour guesses) ----
    int x = 0, y = -1;
    double dist = 0.0;
    for (int it = 0; it <= n - k; ++it) {
      if (prof) t0c = wall_clock64();
      const MinPair p = block_argmin_alt(loc, red, rphase);
      // scipy validates the candidate lazily with `dist == D[x, neighbor[x]]`; here that predicate is kept up to
      // date where D changes (the merge pass below), encoded in the sign of the neighbour: nb >= 0 valid,
      // nb <= -2 stale (neighbour -2-nb), -1 none -- no global read on the critical path
      x = p.i; dist = p.v;
      if (x >= n - 1) break;                  // no candidate at all: cannot happen while two clusters are alive
      const int nbx = NB(x);
      if (prof) { const long long t1 = wall_clock64(); t_arg += t1 - t0c; t0c = t1; }
      if (nbx >= 0) { y = nbx; break; }
      ++retries;
      const double* rowx = D + (size_t)x * n;
      MinPair q{INFINITY, 0x7fffffff};
      // all of a thread's loads are issued before the first is used (LK_E per batch): one memory latency per rescan,
      // not one per element -- with a runtime trip count the compiler serialises load -> compare -> next load
      for (int base = 0; base < span; base += LK_E * NTH) {
        double dv[LK_E];
        int jz[LK_E];
#pragma unroll
        for (int e = 0; e < LK_E; ++e) {
          const int j = (COMPACT && e >= ecount) ? -1 : slot(base, e);
          jz[e] = j;
          dv[e] = (j > x && SZ(j) > 0) ? rowx[j] : INFINITY;
        }
#pragma unroll
        for (int e = 0; e < LK_E; ++e) q = min_pair(q, MinPair{dv[e], jz[e] < 0 ? 0x7fffffff : jz[e]});
      }
      q = block_argmin_alt(q, red, rphase);
      if (owns(x)) { NB(x) = q.v < INFINITY ? q.i : -1; MD(x) = q.v; loc = own_min(); }
      if (prof) { const long long t1 = wall_clock64(); t_scan += t1 - t0c; }
    }
    if (y < 0) {                              // uniform across the block (every thread holds the same reduction results)
      if (tid == 0) g_min_dist[n - 1] = -1.0;
      return;
    }
    if (prof) t0c = wall_clock64();
    const int nx = SZ(x), ny = SZ(y);
    if (tid == 64) {      // dendrogram row as (slot x, slot y): stores only; the host turns slots into cluster ids
      Z[4 * (size_t)k + 0] = x; Z[4 * (size_t)k + 1] = y; Z[4 * (size_t)k + 2] = dist; Z[4 * (size_t)k + 3] = nx + ny;
    }
    // ---- one pass over the merged rows: Lance-Williams centroid update, candidate maintenance, and the
    // new cluster's own nearest neighbour ----
    const double* rx = D + (size_t)x * n;
    double* ry = D + (size_t)y * n;
    const double fx = (double)nx, fy = (double)ny, fs = (double)(nx + ny);
    const double sub = (fx * fy * dist * dist) / fs;
    // one exactly rounded reciprocal per merge instead of an exactly rounded division per element: distances differ
    // from scipy's by <= 1 ulp, far inside the 1e-9 the dendrogram is compared at, and equal inputs still give equal
    // outputs, so ties break as before
    const double inv_fs = 1.0 / fs;
    MinPair best{INFINITY, 0x7fffffff};
    for (int base = 0; base < span; base += LK_E * NTH) {
      // both rows' elements of this thread in registers first (see the rescan above); no element written below is read
      // by this pass (writes go to row y / column y at z, reads come from rows x and y at other z)
      double dxv[LK_E], dyv[LK_E];
      unsigned okm = 0;
#pragma unroll
      for (int e = 0; e < LK_E; ++e) {
        const int z = (COMPACT && e >= ecount) ? -1 : slot(base, e);
        const bool ok = z >= 0 && z != x && z != y && SZ(z) != 0;
        okm |= ok ? 1u << e : 0u;
        dxv[e] = ok ? rx[z] : 0.0;
        dyv[e] = ok ? ry[z] : 0.0;
      }
#pragma unroll
      for (int e = 0; e < LK_E; ++e) {
        if (!(okm >> e & 1)) continue;
        int z = slot(base, e);
        asm volatile("" : "+v"(z));              // addresses are formed here, not hoisted for all LK_E elements at once
        const double dxi = dxv[e], dyi = dyv[e];
        const double nd = lw_centroid(fx, fy, dxi, dyi, sub, inv_fs);
        ry[z] = nd;
        D[(size_t)z * n + y] = nd;
        if (z < y) {
          const int nb0 = NB(z);
          int nb = nb0;
          int dec = nb <= -2 ? -2 - nb : nb;                 // neighbour whatever the validity
          if (z < x && dec == x) dec = y;                    // scipy: "reassign neighbor candidates from x to y"
          if (dec == y) nb = (MD(z) == nd) ? y : -2 - y;     // D[z][y] just changed: re-evaluate `dist == D[z, neighbor]`
          if (nd < MD(z)) { nb = y; MD(z) = nd; loc = min_pair(loc, MinPair{nd, z}); }      // lower-bound update
          if (nb != nb0) NB(z) = nb;
        } else {
          best = min_pair(best, MinPair{nd, z});
        }
        __builtin_amdgcn_sched_barrier(0);       // keep the unrolled elements apart: interleaving them spills (128 VGPRs per wave)
      }
    }
    best = block_argmin_alt(best, red, rphase);
    // state of the two merged slots, by their owners; everybody has read SZ(x), SZ(y) before the barrier above
    bool mine = false;
    if (owns(x)) { SZ(x) = 0; MD(x) = INFINITY; mine = true; }
    if (owns(y)) {
      SZ(y) = (uint16_t)(nx + ny);
      if (y < n - 1) { NB(y) = best.v < INFINITY ? best.i : -1; MD(y) = best.v; }
      mine = true;
    }
    if (mine) loc = own_min();
    if (prof) t_merge += wall_clock64() - t0c;
  }
  if (tid == 0 && prof) printf("shader clock %.0f MHz over %.1f ms\n", (double)(clock64() - cyc0) / (double)(wall_clock64() - wc0) * 100.0, (wall_clock64() - wc0) * 1e-5);
  if (tid == 0 && prof) printf("linkage n=%d: argmin+validate %.1f ms, row rescans %.1f ms, merge pass %.1f ms (100 MHz wall clock), retries %lld\n", n, t_arg * 1e-5, t_scan * 1e-5, t_merge * 1e-5, retries);
  __syncthreads();
  if (tid == 0) g_min_dist[n - 1] = (double)retries;     // statistics: invalid candidates re-evaluated (slot n-1 is unused)
}

// ------------------------------------------------------------------------------------ merge loop (MB_G workgroups of ONE XCD)
// Round 4.  The one-workgroup loop above is bound by what a single CU can issue per merge (a fused fp64 pass over the live
// slots + 3-4 block reductions: ~10 us per merge at n = 9 200, ~75 us at n = 27 000 where its state no longer fits LDS) and it
// is replicated on every rank of a sharded run: 91 ms of a 430 ms hour, 1.6 s of the 2.6 s that config 5's three hours take.
// Here MB_G = 16 persistent workgroups share a merge:
//
//   * slot z belongs to workgroup (z / 16) mod MB_G (16 consecutive slots = one 128-byte line of a row of D); the owner keeps
//     the slot's nearest-neighbour candidate (lower bound, neighbour, validity in the neighbour's sign -- exactly the
//     one-workgroup loop's encoding of scipy's lazy `dist == D[x, neighbor[x]]`) in ITS LDS.  Cluster sizes (uint16) are
//     replicated in every workgroup's LDS: every workgroup applies every merge to its copy.
//   * a ROUND = every workgroup publishes 48 bytes -- its best VALID candidate (distance, x, neighbour), its smallest STALE lower
//     bound (distance, z), and its share of the nearest neighbour of the cluster the previous merge created (the minimum of the
//     distances it just wrote into that row, over its slots above it; the three block reductions share one barrier) -- the
//     last word carries the round's number; wave 0 of every workgroup polls the 16 tag words, loads the entries, reduces them in
//     registers (DPP) and hands the result to its workgroup through LDS, so all workgroups derive the same decision without a
//     second exchange (form A, the first version: an arrival counter, then every wave loads the entries -- one more L2 round
//     trip per round):
//       - the best valid candidate is lexicographically (distance, slot) below every stale bound -> that pair is merged (what
//         scipy's heap pops once its top is valid);
//       - otherwise the stale rows whose bound lies below the best valid candidate are rescanned, each by its owner, all
//         workgroups in parallel (scipy rescans the same rows, one heap pop at a time), and the round repeats.
//     A rescan is block-local: every entry of row z other than column y / row y of the merge in progress was written either by
//     the owner itself or at least one barrier ago, and the liveness of the columns is in the replicated sizes.
//     (A first version kept every candidate exact and rescanned eagerly: correct, 65 ms for the pipeline's hour, but 30-140
//     rescans per merge on high-dimensional noise, where a few hub points are everybody's neighbour: 0.8-3.2 s.)
//   * coherence without cache maintenance: all workgroups run on ONE XCD (they share its L2), shared data (D, the published
//     entries, the barrier word) is read and written with agent-scope relaxed atomics (sc1: past the CU's vector L1, served by
//     the XCD's L2), and a workgroup arrives at the barrier only after every thread has waited for its own stores
//     (`s_waitcnt vmcnt(0)`: acknowledged by the L2).  No release / acquire fences: on a multi-XCD part those write back /
//     invalidate the whole L2.  Workgroups find each other at start-up: 8 x (MB_G + 8) are launched (round robin over the 8
//     XCDs), the first to arrive names its XCD (HW_REG_XCC_ID), the first MB_G arrivals on that XCD take tickets, everybody
//     else returns at once.  Every spin is bounded; a barrier that does not complete sets an abort word, the kernel returns
//     with status -2 and centroid_linkage() falls back to the one-workgroup loop.
// Also measured and dropped (profiles/r04_call8_linkage_tagged_exchange.txt): publishing as the barrier -- the round's number in
// the last word of every entry, every WAVE polling the 16 tag words itself instead of thread 0 polling one arrival counter.
// Exact, but 256 polling waves on 16 cache lines slow the writers down more than the saved round trip is worth: 111-122 ms
// instead of 94-97 ms at n = 9 200, 531 instead of 422 ms at n = 27 000, 92 instead of 74 ms in the pipeline.
// (MB_NT = 512, round 5: local minima unchanged, merge pass -14 %, publish + barrier +22 % -- 63.9-69.0 instead of 60.6 ms per hour; kept at 1024)
static constexpr int MB_G = 16, MB_NT = 1024;
// Eager validation right after a merge of the stale rows whose bound is within 5 % of the merged distance (to save the extra
// round they would cost when they reach the top): measured and switched off -- on high-dimensional noise 33 rows per merge
// qualify (96 -> 135 ms at n = 9 200, 422 -> 648 ms at n = 27 000), on the pipeline's embeddings it changes nothing (74 ms);
// profiles/r04_call5_tr_epilogue_linkage_eager.txt.  MB_EAGER = 0 compiles the path away.
static constexpr int MB_EAGER = 0;                       // stale rows a workgroup may validate right after a merge
static constexpr double MB_EAGER_FACTOR = 1.05;          // ... when their bound is within 5 % of the distance just merged
struct LkEntry { double val_v; int val_x, val_y; double st_v; int st_z, pad0; double part_v; int part_z, pad1; };   // 48 bytes
// form C of the exchange: four 16-byte pieces, each with the round's number in its last word (a 16-byte aligned dwordx4 store /
// load is one request to the L2: a piece is never seen torn, and a reader that finds all four tags current has the whole entry)
struct LkEntry16 { double val_v; int val_x; unsigned tag0; int val_y, st_z, part_z; unsigned tag1; double st_v; unsigned pad2, tag2; double part_v; unsigned pad3, tag3; };
static_assert(sizeof(LkEntry16) == 64, "published entry, form C: four 16-byte pieces");
struct LkCtl { int xcc, tickets; unsigned bar; int abort; LkEntry ent[2][MB_G]; LkEntry16 ent16[2][MB_G]; };
static_assert(sizeof(LkCtl) <= 4096, "the control block's scratch allocation (rvd_centroid_linkage)");
static_assert(sizeof(LkEntry) == 48, "published entry: six 8-byte words");

__device__ inline unsigned long long mb_ld64(const void* p) {
  return __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void mb_st64(void* p, unsigned long long v) {
  __hip_atomic_store((unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double mb_ldf(const double* p) { return __longlong_as_double((long long)mb_ld64(p)); }
__device__ inline void mb_stf(double* p, double v) { mb_st64(p, (unsigned long long)__double_as_longlong(v)); }
__device__ inline void mb_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// 16-byte sc1 (agent-scope, past the L1) accesses: __hip_atomic_* stops at 8 bytes
typedef unsigned mb_u32x4 __attribute__((ext_vector_type(4)));
__device__ inline void mb_st128x4(void* p, mb_u32x4 a, mb_u32x4 b, mb_u32x4 c, mb_u32x4 d) {
  asm volatile(
      "global_store_dwordx4 %0, %1, off sc1\n\t"
      "global_store_dwordx4 %0, %2, off offset:16 sc1\n\t"
      "global_store_dwordx4 %0, %3, off offset:32 sc1\n\t"
      "global_store_dwordx4 %0, %4, off offset:48 sc1\n\t"
      "s_nop 1"
      :: "v"(p), "v"(a), "v"(b), "v"(c), "v"(d) : "memory");
}
__device__ inline void mb_ld128x4(const void* p, mb_u32x4& a, mb_u32x4& b, mb_u32x4& c, mb_u32x4& d) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\t"
      "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
      "global_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
}
__device__ inline bool lex_less(double av, int ai, double bv, int bi) { return av < bv || (av == bv && ai < bi); }

// all MB_G workgroups have arrived `gen` times; false = gave up (abort word set, by us or by somebody else)
__device__ inline bool mb_grid_barrier(LkCtl* ctl, unsigned gen, int G, int* s_flag) {
  __syncthreads();                       // every thread of the workgroup has waited for its stores (caller)
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&ctl->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = gen * (unsigned)G;
    unsigned spins = 0;
    int ok = 1;
    while ((int)(__hip_atomic_load(&ctl->bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255u) == 0) {
        if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || spins > (1u << 22)) {
          __hip_atomic_store(&ctl->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0;
}

__global__ __launch_bounds__(MB_NT) void linkage_mb_kernel(double* __restrict__ D, int n, const int* __restrict__ g_neighbor,
                                                           double* __restrict__ g_min_dist, double* __restrict__ Z, LkCtl* ctl, int G,
                                                           int prof) {
  // G = participating workgroups: a power of two <= MB_G (RVD_LINKAGE_G; 16 by default)
  extern __shared__ __attribute__((aligned(16))) char mb_smem[];
  __shared__ MinPair red[2][16];
  __shared__ MinPair red3[2][48];
  __shared__ int s_wtot[16];
  __shared__ int s_b, s_flag, s_rcount, s_acount, s_round_y;
  __shared__ MinPair s_round[3];
  const int tid = threadIdx.x, lane = tid & 63;
  const bool xchg = (prof & 2) != 0;           // exchange form B / C (see the round loop)
  const bool xchg16 = (prof & 4) != 0;         // form C
  prof &= 1;
  if (tid < 96) red3[tid / 48][tid % 48] = MinPair{INFINITY, 0x7fffffff};
  int rphase3 = 0;
  // ---- which workgroups take part
  if (tid == 0) {
    const int my = (int)(__builtin_amdgcn_s_getreg(63508) & 0xf) + 1;      // HW_REG_XCC_ID + 1
    int expected = 0;
    __hip_atomic_compare_exchange_strong(&ctl->xcc, &expected, my, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int chosen = expected == 0 ? my : expected;
    int b = -1;
    if (chosen == my) {
      b = __hip_atomic_fetch_add(&ctl->tickets, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (b >= G) b = -1;
    }
    s_b = b;
  }
  if (tid < 32) red[tid >> 4][tid & 15] = MinPair{INFINITY, 0x7fffffff};
  __syncthreads();
  const int b = s_b;
  if (b < 0) return;
  int rphase = 0;

  // ---- LDS: own candidates, rescan list, replicated sizes, sorted live list
  const int ngrp = (n + 15) >> 4;                           // 16-slot groups
  const int m = ((ngrp + G - 1) / G) << 4;                  // own slots (capacity)
  double* s_md = (double*)mb_smem;                          // [m] candidate distance: exact when valid, a lower bound when stale
  int* s_nb = (int*)(s_md + m);                             // [m] >= 0 valid neighbour, <= -2 stale (neighbour -2 - nb), -1 none
  int* s_rl = s_nb + m;                                     // [m] rows to rescan
  uint16_t* s_sz = (uint16_t*)(s_rl + m);                   // [n]
  uint16_t* s_al = s_sz + ((n + 1) & ~1);                   // [n] live slots, ascending (rebuilt every 256 merges)
  auto slot_of = [&](int li) -> int { return (((li >> 4) * G + b) << 4) + (li & 15); };
  auto owner_of = [&](int z) -> int { return (z >> 4) & (G - 1); };
  auto li_of = [&](int z) -> int { return (((z >> 4) / G) << 4) + (z & 15); };
  for (int i = tid; i < n; i += MB_NT) s_sz[i] = 1;
  for (int li = tid; li < m; li += MB_NT) {
    const int z = slot_of(li);
    const bool ok = z < n - 1;                              // slot n-1 has no candidate (no y > n-1)
    s_md[li] = ok ? g_min_dist[z] : INFINITY;
    s_nb[li] = ok ? g_neighbor[z] : -1;
  }
  auto rebuild = [&]() {                                    // sorted list of the live slots (uniform result: s_acount)
    __syncthreads();
    const int wv = tid >> 6;
    const int C = (n + MB_NT - 1) / MB_NT, z0 = tid * C;
    int c = 0;
    for (int j = 0; j < C; ++j) { const int z = z0 + j; if (z < n && s_sz[z] != 0) ++c; }
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
    if (lane == 63) s_wtot[wv] = incl;
    __syncthreads();
    int woff = 0, total = 0;
    for (int w = 0; w < MB_NT / 64; ++w) { const int t = s_wtot[w]; if (w < wv) woff += t; total += t; }
    int pos = woff + incl - c;
    for (int j = 0; j < C; ++j) { const int z = z0 + j; if (z < n && s_sz[z] != 0) s_al[pos++] = (uint16_t)z; }
    if (tid == 0) s_acount = total;
    __syncthreads();
  };
  // exact nearest live neighbour above z, block-local (see the header): candidate of row z becomes valid
  auto rescan_row = [&](int z) {
    const int na = s_acount;
    const double* rz = D + (size_t)z * n;
    MinPair q{INFINITY, 0x7fffffff};
    for (int p0 = 0; p0 < na; p0 += 4 * MB_NT) {
      double dv[4];
      int jv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {                         // four loads in flight per thread
        const int p = p0 + u * MB_NT + tid;
        const int j = p < na ? (int)s_al[p] : -1;
        const bool ok = j > z && s_sz[j] != 0;
        jv[u] = ok ? j : 0x7fffffff;
        dv[u] = ok ? mb_ldf(rz + j) : INFINITY;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) q = min_pair(q, MinPair{dv[u], jv[u]});
    }
    q = block_argmin_alt(q, red, rphase);
    if (tid == 0) { s_md[li_of(z)] = q.v; s_nb[li_of(z)] = q.v < INFINITY ? q.i : -1; }
  };
  long long rescans = 0;
  long long t_min = 0, t_bar = 0, t_read = 0, t_val = 0, t_merge = 0, tc = 0, rounds = 0;      // RVD_LINKAGE_PROF (workgroup 0)
  auto tick = [&](long long& acc) { if (prof) { const long long t = wall_clock64(); acc += t - tc; tc = t; } };
  if (prof) tc = wall_clock64();
  unsigned gen = 0;                                         // grid barriers passed
  int yprev = -1;
  MinPair part{INFINITY, 0x7fffffff};                       // this workgroup's share of the new row's nearest neighbour
  for (int k = 0; k < n - 1; ++k) {
    if ((k & 255) == 0) rebuild();
    int x = 0, y = -1;
    double dist = 0.0;
    for (;;) {
      // ---- local minima over the own candidates (dead / pending slots hold +inf): best valid, smallest stale bound
      MinPair lv{INFINITY, 0x7fffffff}, ls{INFINITY, 0x7fffffff};
      for (int li = tid; li < m; li += MB_NT) {
        const double md = s_md[li];
        if (md < INFINITY) {
          const MinPair c{md, slot_of(li)};
          if (s_nb[li] >= 0) lv = min_pair(lv, c); else ls = min_pair(ls, c);
        }
      }
      block_argmin3_alt(lv, ls, part, red3, rphase3);       // `part`: per-thread after a merge pass, already reduced (idempotent) otherwise
      tick(t_min);
      ++rounds;
      MinPair gv, gs, gp;
      int gy;
      if (xchg) {
        // Exchange form B (default; RVD_LINKAGE_XCHG=0 selects form A below): publishing is the barrier, polled by ONE wave per
        // workgroup.  Measured against form A on one box: 76-82 vs 96 ms on the 9 200-point benchmark sets, 336 vs 375 ms at
        // n = 27 000, 65 vs 70 ms in the pipeline (sum of the RVD_LINKAGE_PROF phases).  Thread 0 writes the
        // data words, waits for them, then the word that carries the round's number; wave 0 polls the G tag words until all show
        // this round, loads the data words, reduces and hands the result to the other waves through LDS: one L2 round trip less
        // than {arrive at a counter, poll the counter, every wave loads the entries} (the all-waves polling variant lost to its
        // 256 pollers; this one has as many pollers as the counter has).
        mb_stores_done();
        ++gen;
        __syncthreads();
        if (tid < 64 && xchg16) {
          // form C: the whole entry in four tagged 16-byte pieces -- no wait between data and tag on the writer's side, one
          // round trip on the reader's side once the pieces are there
          int ok = 1;
          if (tid == 0) {
            const int nbx = lv.v < INFINITY ? s_nb[li_of(lv.i)] : -1;
            const unsigned long long v0 = (unsigned long long)__double_as_longlong(lv.v), v2 = (unsigned long long)__double_as_longlong(ls.v),
                                     v3 = (unsigned long long)__double_as_longlong(part.v);
            mb_st128x4(&ctl->ent16[gen & 1][b], (mb_u32x4){(unsigned)v0, (unsigned)(v0 >> 32), (unsigned)lv.i, gen},
                       (mb_u32x4){(unsigned)nbx, (unsigned)ls.i, (unsigned)part.i, gen}, (mb_u32x4){(unsigned)v2, (unsigned)(v2 >> 32), 0u, gen},
                       (mb_u32x4){(unsigned)v3, (unsigned)(v3 >> 32), 0u, gen});
          }
          const LkEntry16* e = &ctl->ent16[gen & 1][lane & (G - 1)];
          mb_u32x4 p0, p1, p2, p3;
          unsigned spins = 0;
          for (;;) {
            mb_ld128x4(e, p0, p1, p2, p3);
            if (__all(p0.w == gen && p1.w == gen && p2.w == gen && p3.w == gen)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0) {
              if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || spins > (1u << 22)) {
                __hip_atomic_store(&ctl->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
              }
            }
          }
          const double vv = __longlong_as_double((long long)(((unsigned long long)p0.y << 32) | p0.x));
          const double sv = __longlong_as_double((long long)(((unsigned long long)p2.y << 32) | p2.x));
          const double pv = __longlong_as_double((long long)(((unsigned long long)p3.y << 32) | p3.x));
          const int vx = (int)p0.z, vy = (int)p1.x, sz = (int)p1.y, pz = (int)p1.z;
          const MinPair rv = row_argmin(MinPair{vv, vx}), rs = row_argmin(MinPair{sv, sz}), rp = row_argmin(MinPair{pv, pz});
          const unsigned long long win = __ballot(vv == rv.v && vx == rv.i);
          const int ry = __shfl(vy, __builtin_ctzll(win | (1ull << 63)));
          if (tid == 0) { s_round[0] = rv; s_round[1] = rs; s_round[2] = rp; s_round_y = ry; s_flag = ok; }
        } else if (tid < 64) {
          int ok = 1;
          if (tid == 0) {
            LkEntry* e = &ctl->ent[gen & 1][b];
            const int nbx = lv.v < INFINITY ? s_nb[li_of(lv.i)] : -1;
            mb_stf(&e->val_v, lv.v);
            mb_st64(&e->val_x, (unsigned long long)(unsigned)lv.i | ((unsigned long long)(unsigned)nbx << 32));
            mb_stf(&e->st_v, ls.v);
            mb_stf(&e->part_v, part.v);
            mb_st64(&e->part_z, (unsigned long long)(unsigned)part.i);
            mb_stores_done();
            mb_st64(&e->st_z, (unsigned long long)(unsigned)ls.i | ((unsigned long long)gen << 32));
          }
          const LkEntry* e = &ctl->ent[gen & 1][lane & (G - 1)];
          unsigned long long tagw;
          unsigned spins = 0;
          for (;;) {
            tagw = mb_ld64(&e->st_z);
            if (__all((unsigned)(tagw >> 32) == gen)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0) {
              if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || spins > (1u << 22)) {
                __hip_atomic_store(&ctl->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
              }
            }
          }
          const double vv = mb_ldf(&e->val_v);
          const unsigned long long vxy = mb_ld64(&e->val_x);
          const double sv = mb_ldf(&e->st_v);
          const double pv = mb_ldf(&e->part_v);
          const int pz = (int)(unsigned)mb_ld64(&e->part_z);
          const int sz = (int)(unsigned)tagw;
          const int vx = (int)(unsigned)vxy, vy = (int)(unsigned)(vxy >> 32);
          const MinPair rv = row_argmin(MinPair{vv, vx}), rs = row_argmin(MinPair{sv, sz}), rp = row_argmin(MinPair{pv, pz});
          const unsigned long long win = __ballot(vv == rv.v && vx == rv.i);
          const int ry = __shfl(vy, __builtin_ctzll(win | (1ull << 63)));
          if (tid == 0) { s_round[0] = rv; s_round[1] = rs; s_round[2] = rp; s_round_y = ry; s_flag = ok; }
        }
        __syncthreads();
        if (s_flag == 0) {
          if (b == 0 && tid == 0) g_min_dist[n - 1] = -2.0;
          return;
        }
        gv = s_round[0]; gs = s_round[1]; gp = s_round[2]; gy = s_round_y;
        tick(t_bar);
      } else {
      if (tid == 0) {
        LkEntry* e = &ctl->ent[gen & 1][b];
        const int nbx = lv.v < INFINITY ? s_nb[li_of(lv.i)] : -1;
        mb_stf(&e->val_v, lv.v);
        mb_st64(&e->val_x, (unsigned long long)(unsigned)lv.i | ((unsigned long long)(unsigned)nbx << 32));
        mb_stf(&e->st_v, ls.v);
        mb_st64(&e->st_z, (unsigned long long)(unsigned)ls.i);
        mb_stf(&e->part_v, part.v);
        mb_st64(&e->part_z, (unsigned long long)(unsigned)part.i);
      }
      mb_stores_done();                                     // the merge pass's stores to D and the entry: acknowledged by the L2
      ++gen;
      if (!mb_grid_barrier(ctl, gen, G, &s_flag)) {
        if (b == 0 && tid == 0) g_min_dist[n - 1] = -2.0;
        return;
      }
      tick(t_bar);
      // ---- every wave: the 16 entries, reduced in registers
      {
        const LkEntry* e = &ctl->ent[(gen - 1) & 1][lane & (G - 1)];      // (lanes beyond G re-read entries: min is idempotent)
        const double vv = mb_ldf(&e->val_v);
        const unsigned long long vxy = mb_ld64(&e->val_x);
        const double sv = mb_ldf(&e->st_v);
        const int sz = (int)(unsigned)mb_ld64(&e->st_z);
        const double pv = mb_ldf(&e->part_v);
        const int pz = (int)(unsigned)mb_ld64(&e->part_z);
        const int vx = (int)(unsigned)vxy, vy = (int)(unsigned)(vxy >> 32);
        gv = row_argmin(MinPair{vv, vx});
        gs = row_argmin(MinPair{sv, sz});
        gp = row_argmin(MinPair{pv, pz});
        const unsigned long long win = __ballot(vv == gv.v && vx == gv.i);
        gy = __shfl(vy, __builtin_ctzll(win | (1ull << 63)));
      }
      }
      tick(t_read);
      if (yprev >= 0) {
        // the cluster of the previous merge: its (exact) candidate is known only now; installed by its owner, one more competitor
        if (yprev < n - 1) {
          if (owner_of(yprev) == b && tid == 0) { s_md[li_of(yprev)] = gp.v; s_nb[li_of(yprev)] = gp.v < INFINITY ? gp.i : -1; }
          if (lex_less(gp.v, yprev, gv.v, gv.i)) { gv = MinPair{gp.v, yprev}; gy = gp.i; }
        }
        yprev = -1;
        part = MinPair{INFINITY, 0x7fffffff};
      }
      if (!(gv.v < INFINITY) && !(gs.v < INFINITY)) {       // uniform: every workgroup read the same entries
        if (b == 0 && tid == 0) g_min_dist[n - 1] = -1.0;
        return;
      }
      if (lex_less(gv.v, gv.i, gs.v, gs.i)) { x = gv.i; y = gy; dist = gv.v; break; }
      // ---- validation: the own stale rows whose bound lies below the best valid candidate (at least the smallest stale one)
      if (tid == 0) s_rcount = 0;
      __syncthreads();
      for (int li = tid; li < m; li += MB_NT) {
        const double md = s_md[li];
        if (md < INFINITY && s_nb[li] <= -2) {
          const int z = slot_of(li);
          if (lex_less(md, z, gv.v, gv.i) || (md == gs.v && z == gs.i)) s_rl[atomicAdd(&s_rcount, 1)] = z;
        }
      }
      __syncthreads();
      const int R = s_rcount;
      for (int r = 0; r < R; ++r) rescan_row(s_rl[r]);
      rescans += R;
      __syncthreads();                                      // candidate writes of thread 0 before the next local minima
      tick(t_val);
    }
    if (y < 0 || y >= n || x < 0 || x >= n) {               // uniform
      if (b == 0 && tid == 0) g_min_dist[n - 1] = -1.0;
      return;
    }
    const int nx = s_sz[x], ny = s_sz[y];
    __syncthreads();                                        // sizes read; the owner's candidate write for yprev is visible
    if (b == 0 && tid == 64) {
      Z[4 * (size_t)k + 0] = x; Z[4 * (size_t)k + 1] = y; Z[4 * (size_t)k + 2] = dist; Z[4 * (size_t)k + 3] = nx + ny;
    }
    if (tid == 0) { s_sz[x] = 0; s_sz[y] = (uint16_t)(nx + ny); s_rcount = 0; }
    if (tid == 1 && owner_of(x) == b) s_md[li_of(x)] = INFINITY;
    if (tid == 2 && owner_of(y) == b) s_md[li_of(y)] = INFINITY;       // pending until the next barrier
    __syncthreads();
    // ---- merge pass over the own slots: Lance-Williams update of row / column y, lazy candidate maintenance (as above)
    const double fx = (double)nx, fy = (double)ny, fs = (double)(nx + ny);
    const double sub = (fx * fy * dist * dist) / fs;
    const double inv_fs = 1.0 / fs;
    const double eager_thr = dist * MB_EAGER_FACTOR;
    const double* rx = D + (size_t)x * n;
    double* ry = D + (size_t)y * n;
    MinPair best{INFINITY, 0x7fffffff};
    for (int li = tid; li < m; li += MB_NT) {
      const int z = slot_of(li);
      if (z >= n || z == x || z == y || s_sz[z] == 0) continue;
      const double dxi = mb_ldf(rx + z), dyi = mb_ldf(ry + z);
      const double nd = lw_centroid(fx, fy, dxi, dyi, sub, inv_fs);
      mb_stf(ry + z, nd);
      mb_stf(D + (size_t)z * n + y, nd);
      if (z < y) {
        const int nb0 = s_nb[li];
        const double md = s_md[li];
        int nb = nb0;
        int dec = nb <= -2 ? -2 - nb : nb;                  // neighbour whatever the validity
        if (z < x && dec == x) dec = y;                     // scipy: "reassign neighbor candidates from x to y"
        if (dec == y) nb = (md == nd) ? y : -2 - y;         // D[z][y] just changed: re-evaluate `dist == D[z, neighbor]`
        if (nd < md) { nb = y; s_md[li] = nd; }             // lower-bound update
        if (nb != nb0) s_nb[li] = nb;
        // a stale candidate whose bound is about as small as the distance just merged will be the heap's top within a few
        // merges: validating it NOW (block-local, below) costs a row scan; waiting costs the same scan plus a whole extra
        // round of every workgroup.  Bounded: at most MB_EAGER rows per workgroup and merge, the rest stay lazy.
        if constexpr (MB_EAGER > 0) {
          if (nb <= -2 && fmin(md, nd) <= eager_thr) { const int pos = atomicAdd(&s_rcount, 1); if (pos < MB_EAGER) s_rl[pos] = z; }
        }
      } else {
        best = min_pair(best, MinPair{nd, z});
      }
    }
    if constexpr (MB_EAGER > 0) mb_stores_done();           // the rescans below read what this pass wrote (own rows)
    if constexpr (MB_EAGER > 0) part = block_argmin_alt(best, red, rphase);             // (barrier inside: the list is complete)
    else part = best;                                       // reduced together with the next round's local minima (block_argmin3_alt)
    if constexpr (MB_EAGER > 0) {
      const int R = min(s_rcount, MB_EAGER);
      for (int r = 0; r < R; ++r) rescan_row(s_rl[r]);
      rescans += R;
      if (R) __syncthreads();                               // candidate writes of thread 0 before the next local minima
    }
    yprev = y;
    tick(t_merge);
  }
  if (prof && b == 0 && tid == 0)
    printf("linkage_mb n=%d G=%d: %lld rounds (%.2f per merge); local minima %.1f ms, publish + barrier %.1f ms, read + decide %.1f ms, "
           "validation rescans %.1f ms (%lld rows here), merge pass %.1f ms (100 MHz wall clock, workgroup 0)\n", n, G, rounds,
           (double)rounds / (n - 1), t_min * 1e-5, t_bar * 1e-5, t_read * 1e-5, t_val * 1e-5, rescans, t_merge * 1e-5);
  if (b == 0 && tid == 0) g_min_dist[n - 1] = (double)rescans;       // statistics (workgroup 0's rescans; slot n-1 is unused)
}

int centroid_linkage(hipStream_t s, const double* X, int n, int d, double* D, uint16_t* size, int* cluster_id, int* neighbor,
                     double* min_dist, double* Z, void* scratch, int workgroups) {
  if (n < 2) return OK;
  const int t = cdiv(n, 64);
  const size_t lds = (size_t)n * 14 + 16, lds_c = (size_t)n * 16 + 16;      // + the sorted slot list of the compacting variant
  const bool force_global = lab_env("RVD_LINKAGE_GLOBAL") != nullptr;      // test hook: exercise the large-n variant on small inputs
  const bool no_compact = lab_env("RVD_LINKAGE_COMPACT") && atoi(lab_env("RVD_LINKAGE_COMPACT")) == 0;
  const int flags = (lab_env("RVD_LINKAGE_PROF") ? 1 : 0) | (lab_env("RVD_LINKAGE_XCHG") ? (atoi(lab_env("RVD_LINKAGE_XCHG")) == 0 ? 0 : atoi(lab_env("RVD_LINKAGE_XCHG")) == 2 ? 6 : 2) : 2);      // exchange form: 0 = A, 1 = B (default), 2 = C
  // RVD_LINKAGE_MB: 0 = never the multi-workgroup loop, 1 = always (tests: any n), unset = from 3 000 points on (below that one
  // CU's LDS-resident loop is as fast: a merge is a chain of latencies either way)
  const char* mbe = lab_env("RVD_LINKAGE_MB");
  const int mb_mode = mbe ? atoi(mbe) : -1;
  // workgroups of the multi-workgroup loop: the caller's hint (rvd_set_linkage_workgroups: 1 = the one-workgroup loop, a power of
  // two up to 16, 0 = default), RVD_LINKAGE_G overrides; a count whose per-workgroup state does not fit LDS is doubled until it does
  int G = MB_G;
  if (workgroups == 2 || workgroups == 4 || workgroups == 8 || workgroups == 16) G = workgroups;
  if (const char* ge = lab_env("RVD_LINKAGE_G")) { const int v = atoi(ge); if (v == 2 || v == 4 || v == 8 || v == 16) G = v; }
  while (G < MB_G && (size_t)((((n + 15) >> 4) + G - 1) / G << 4) * 16 + (size_t)((n + 1) & ~1) * 4 > 150 * 1024) G *= 2;
  const int ngrp = (n + 15) >> 4, m_own = ((ngrp + G - 1) / G) << 4;
  const size_t lds_mb = (size_t)m_own * 16 + (size_t)((n + 1) & ~1) * 4;
  const bool use_mb = scratch != nullptr && workgroups != 1 && !force_global && !no_compact && lds_mb <= 150 * 1024 &&
                      (mb_mode == 1 || (mb_mode < 0 && n >= 3000));
  static bool attr_set = false;
  if (!attr_set) {
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)linkage_kernel<true, 1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)linkage_kernel<true, 1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
    RVB_HIP_CHECK(hipFuncSetAttribute((const void*)linkage_mb_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set = true;
  }
  for (int attempt = 0; attempt < 2; ++attempt) {
    hipLaunchKernelGGL(pdist_kernel, dim3(t, t), dim3(256), 0, s, X, n, d, D);
    RVB_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(nn_init_kernel, dim3(n - 1), dim3(256), 0, s, D, n, neighbor, min_dist);
    RVB_HIP_CHECK(hipGetLastError());
    if (use_mb && attempt == 0) {
      RVB_HIP_CHECK(hipMemsetAsync(scratch, 0, sizeof(LkCtl), s));
      // at least 64 KiB of LDS per workgroup so that a CU holds one of them (the MB_G participants sit on MB_G different CUs)
      const size_t lds_launch = lds_mb < 96 * 1024 ? 96 * 1024 : lds_mb;
      hipLaunchKernelGGL(linkage_mb_kernel, dim3(8 * (G + 8)), dim3(MB_NT), lds_launch, s, D, n, neighbor, min_dist, Z, (LkCtl*)scratch, G, flags);
      RVB_HIP_CHECK(hipGetLastError());
      double status = 0.0;
      RVB_HIP_CHECK(hipMemcpyAsync(&status, min_dist + (n - 1), 8, hipMemcpyDeviceToHost, s));
      RVB_HIP_CHECK(hipStreamSynchronize(s));
      if (status != -2.0) return OK;               // done (or -1: no candidate pair, reported by the caller)
      // a barrier gave up (the workgroups did not all become resident on one XCD): start over on the one-workgroup loop --
      // unless the multi-workgroup loop was asked for by name (tests): then this is an error, never a silent fall-back
      if (mb_mode == 1) { set_error("centroid_linkage: the multi-workgroup merge loop (RVD_LINKAGE_MB=1) gave up at a grid barrier"); return E_STATE; }
      if (lab_env("RVD_LINKAGE_PROF")) fprintf(stderr, "linkage: multi-workgroup loop aborted, falling back to one workgroup\n");
      const double inf = INFINITY;
      RVB_HIP_CHECK(hipMemcpyAsync(min_dist + (n - 1), &inf, 8, hipMemcpyHostToDevice, s));
      RVB_HIP_CHECK(hipStreamSynchronize(s));
      continue;
    }
    if (!force_global && !no_compact && lds_c <= 158 * 1024 && n <= LK_E * 1024) {
      hipLaunchKernelGGL((linkage_kernel<true, 1024, true>), dim3(1), dim3(1024), lds_c, s, flags, D, n, size, cluster_id, neighbor, min_dist, Z);
    } else if (lds <= 158 * 1024 && !force_global) {
      hipLaunchKernelGGL((linkage_kernel<true, 1024, false>), dim3(1), dim3(1024), lds, s, flags, D, n, size, cluster_id, neighbor, min_dist, Z);
    } else {
      hipLaunchKernelGGL((linkage_kernel<false, 1024, false>), dim3(1), dim3(1024), 0, s, flags, D, n, size, cluster_id, neighbor, min_dist, Z);
    }
    RVB_HIP_CHECK(hipGetLastError());
    break;
  }
  return OK;
}

}  // namespace rvb
