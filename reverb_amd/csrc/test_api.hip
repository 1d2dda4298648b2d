// rvb_test_*: raw kernel entry points used by tests/ (host buffers in, host buffers out).  Each one
// uploads fp32 host data (rounded to the compute dtype with the same RNE conversion the engine
// uses), launches exactly the kernel the engine launches, and downloads the result as fp32.
#include "mp3.h"
#include <cstdlib>
#include <cstring>
#include <vector>

#include "test_api.h"
#include "kernels.h"
#include "search.h"

using namespace rvb;

namespace {
struct Dev {
  void* p = nullptr;
  ~Dev() { if (p) (void)hipFree(p); }
  int alloc(size_t n) { if (n == 0) n = 16; if (hipMalloc(&p, n) != hipSuccess) { set_error("hipMalloc failed in test api"); return E_NOMEM; } return OK; }
};
int up_T(Dev& d, int dtype, const float* src, size_t n) {
  if (!src) return OK;
  int r = d.alloc(n * dt_size(dtype));
  if (r != OK) return r;
  if (dtype == DT_F32) { RVB_HIP_CHECK(hipMemcpy(d.p, src, n * 4, hipMemcpyHostToDevice)); return OK; }
  std::vector<bf16_t> t(n);
  for (size_t i = 0; i < n; ++i) t[i] = f32_to_bf16(src[i]);
  RVB_HIP_CHECK(hipMemcpy(d.p, t.data(), n * 2, hipMemcpyHostToDevice));
  return OK;
}
int up_raw(Dev& d, const void* src, size_t bytes) {
  if (!src) return OK;
  int r = d.alloc(bytes);
  if (r != OK) return r;
  RVB_HIP_CHECK(hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice));
  return OK;
}
int down_T(const Dev& d, int dtype, bool as_f32, float* dst, size_t n) {
  if (dtype == DT_F32 || as_f32) { RVB_HIP_CHECK(hipMemcpy(dst, d.p, n * 4, hipMemcpyDeviceToHost)); return OK; }
  std::vector<bf16_t> t(n);
  RVB_HIP_CHECK(hipMemcpy(t.data(), d.p, n * 2, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) dst[i] = bf16_to_f32(t[i]);
  return OK;
}
int need_gpu() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device available: librvb has no CPU fallback"); return E_HIP; }
  return OK;
}
#define T_TRY(x) do { int _r = (x); if (_r != OK) return _r; } while (0)
}  // namespace

extern "C" {

int rvb_test_gemm(int dtype, const float* A, const float* W, const float* bias, const float* res, float* C, int M,
                  int N, int K, float alpha, int act, int out_f32, int conv, int cT1, int cF1, int cC, int cB) {
  T_TRY(need_gpu());
  Dev dA, dW, dB, dR, dC;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  size_t a_elems = (size_t)M * K;
  if (conv) {
    const int T2 = (cT1 - 3) / 2 + 1, F2 = (cF1 - 3) / 2 + 1;
    if (M != cB * T2 * F2 || K != 9 * cC) { set_error("rvb_test_gemm: conv shape mismatch"); return E_ARG; }
    a_elems = (size_t)cB * cT1 * cF1 * cC;
    g.conv = 1; g.cT1 = cT1; g.cF1 = cF1; g.cT2 = T2; g.cF2 = F2; g.cC = cC;
  }
  T_TRY(up_T(dA, dtype, A, a_elems));
  T_TRY(up_T(dW, dtype, W, (size_t)N * K));
  T_TRY(up_raw(dB, bias, (size_t)N * 4));
  T_TRY(up_raw(dR, res, (size_t)M * N * 4));
  const bool f32out = dtype == DT_F32 || out_f32;
  T_TRY(dC.alloc((size_t)M * N * (f32out ? 4 : 2)));
  g.A = dA.p; g.W = dW.p; g.bias = (const float*)dB.p; g.res = (const float*)dR.p; g.C = dC.p;
  g.M = M; g.N = N; g.K = K; g.lda = conv ? cC : K; g.ldw = K; g.ldc = N; g.ldres = N;
  g.alpha = alpha; g.act = act; g.out_f32 = out_f32;
  T_TRY(gemm(nullptr, dtype, g));
  RVB_HIP_CHECK(hipDeviceSynchronize());
  return down_T(dC, dtype, f32out, C, (size_t)M * N);
}

int rvb_test_gemm_rowadd(const float* A, const float* W, const float* bias, const float* add, float* C, int M, int N, int K,
                         int add_rows, int add_col0, int add_cols) {
  T_TRY(need_gpu());
  Dev dA, dW, dB, dP, dC;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  T_TRY(up_T(dA, DT_BF16, A, (size_t)M * K));
  T_TRY(up_T(dW, DT_BF16, W, (size_t)N * K));
  T_TRY(up_raw(dB, bias, (size_t)N * 4));
  T_TRY(up_T(dP, DT_BF16, add, (size_t)add_rows * add_cols));
  T_TRY(dC.alloc((size_t)M * N * 2));
  g.A = dA.p; g.W = dW.p; g.bias = (const float*)dB.p; g.C = dC.p;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.alpha = 1.f; g.act = ACT_NONE;
  g.rowadd = dP.p; g.rowadd_rows = add_rows; g.rowadd_ld = add_cols; g.rowadd_col0 = add_col0; g.rowadd_cols = add_cols;
  T_TRY(gemm(nullptr, DT_BF16, g));
  RVB_HIP_CHECK(hipDeviceSynchronize());
  return down_T(dC, DT_BF16, false, C, (size_t)M * N);
}

// bf16 GEMM with the ACT_GLU epilogue: W rows / bias / output columns interleaved as the engine packs them (row 2c = a_c, 2c + 1 = b_c);
// C [M, N / 2] = a * sigmoid(b)
int rvb_test_gemm_glu(const float* A, const float* W, const float* bias, float* C, int M, int N, int K) {
  T_TRY(need_gpu());
  Dev dA, dW, dB, dC;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  T_TRY(up_T(dA, DT_BF16, A, (size_t)M * K));
  T_TRY(up_T(dW, DT_BF16, W, (size_t)N * K));
  T_TRY(up_raw(dB, bias, (size_t)N * 4));
  T_TRY(dC.alloc((size_t)M * (N / 2) * 2));
  g.A = dA.p; g.W = dW.p; g.bias = (const float*)dB.p; g.C = dC.p;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N / 2; g.alpha = 1.f; g.act = ACT_GLU;
  if (!gemm_glu_supported(DT_BF16, g)) { set_error("rvb_test_gemm_glu: shape not supported by the ACT_GLU epilogue"); return E_UNSUPPORTED; }
  T_TRY(gemm(nullptr, DT_BF16, g));
  RVB_HIP_CHECK(hipDeviceSynchronize());
  return down_T(dC, DT_BF16, false, C, (size_t)M * (N / 2));
}

int64_t rvb_test_mp3_decode(const void* data, int64_t nbytes, int channel, float* out, int64_t capacity, int64_t* info9, int64_t* stats12, int threads) {
  try {
    rvb::mp3::Info i;
    rvb::mp3::Stats st;
    const int64_t r = rvb::mp3::decode((const uint8_t*)data, (size_t)nbytes, channel, out, capacity, &i, &st, threads);
    if (info9) { info9[0] = i.version; info9[1] = i.channels; info9[2] = i.sample_rate; info9[3] = i.audio_frames; info9[4] = i.samples_per_frame;
                 info9[5] = i.has_info_frame; info9[6] = i.start_skip; info9[7] = i.samples; info9[8] = i.bitrate_kbps; }
    if (stats12) { stats12[0] = st.granules; stats12[1] = st.huff_exact; stats12[2] = st.huff_short; stats12[3] = st.huff_overrun; stats12[4] = st.crc_checked;
                   stats12[5] = st.crc_failed; stats12[6] = st.reservoir_missing; stats12[7] = st.short_granules; stats12[8] = st.mixed_granules;
                   stats12[9] = st.ms_granules; stats12[10] = st.intensity_granules; stats12[11] = st.max_main_data_begin; }
    return r;
  } catch (const rvb::mp3::Error& e) {
    set_error(e.msg);
    return e.code;
  }
}
int rvb_test_mp3_hybrid(float* xr, float* overlap, int block_type, int mixed, float* out) { rvb::mp3::hybrid_granule(xr, overlap, block_type, mixed, 2, out); return OK; }
int rvb_test_mp3_polyphase(const float* sb, float* vbuf, int* voff, float* pcm) { rvb::mp3::polyphase_granule(sb, vbuf, voff, pcm); return OK; }
int rvb_test_mp3_window(float* out512) { memcpy(out512, rvb::mp3::synthesis_window(), 512 * sizeof(float)); return OK; }
int rvb_test_mp3_huffman(int t, uint16_t* codes, uint8_t* lens, int32_t* linbits32) {
  const uint16_t* c = nullptr; const uint8_t* l = nullptr;
  int lb[32];
  const int n = rvb::mp3::huffman_table(t, &c, &l, lb);
  if (linbits32) for (int i = 0; i < 32; ++i) linbits32[i] = lb[i];
  if (n > 0) { memcpy(codes, c, 2 * (size_t)n); memcpy(lens, l, (size_t)n); }
  return n;
}

int rvb_test_rownorm(int dtype, const float* x, const float* gamma, const float* beta, float eps, int mode, int silu,
                     const float* add, float* out, int out_f32, int M, int d) {
  T_TRY(need_gpu());
  Dev dx, dg, db, da, dout;
  const bool x16 = (mode & 256) != 0;            // bit 8 of `mode`: x is handed to the kernel as bf16 (bf16 engine, conv-module norm)
  mode &= 255;
  if (x16) T_TRY(up_T(dx, DT_BF16, x, (size_t)M * d)); else T_TRY(up_raw(dx, x, (size_t)M * d * 4));
  T_TRY(up_raw(dg, gamma, (size_t)d * 4));
  T_TRY(up_raw(db, beta, (size_t)d * 4));
  T_TRY(up_T(da, dtype, add, (size_t)M * d));
  const bool f32out = dtype == DT_F32 || out_f32;
  T_TRY(dout.alloc((size_t)M * d * (f32out ? 4 : 2)));
  NormArgs a;
  a.x = (const float*)dx.p; a.gamma = (const float*)dg.p; a.beta = (const float*)db.p; a.eps = eps; a.mode = mode;
  a.silu = silu; a.add = da.p; a.out = dout.p; a.out_f32 = out_f32; a.M = M; a.d = d;
  a.x_bf16 = x16 ? 1 : 0;
  T_TRY(rownorm(nullptr, dtype, a));
  RVB_HIP_CHECK(hipDeviceSynchronize());
  return down_T(dout, dtype, f32out, out, (size_t)M * d);
}

int rvb_test_conv1(int dtype, const float* feats, const float* mean, const float* istd, const float* w, const float* b,
                   float* out, int B, int T0, int F0, int d) {
  T_TRY(need_gpu());
  const int T1 = (T0 - 3) / 2 + 1, F1 = (F0 - 3) / 2 + 1;
  Dev df, dm, di, dw, db, dout;
  T_TRY(up_raw(df, feats, (size_t)B * T0 * F0 * 4));
  T_TRY(up_raw(dm, mean, (size_t)F0 * 4));
  T_TRY(up_raw(di, istd, (size_t)F0 * 4));
  std::vector<float> wt((size_t)d * 9);            // the caller passes conv.0.weight as the reference stores it, [d][1][3][3]
  for (int c = 0; c < d; ++c)
    for (int k = 0; k < 9; ++k) wt[(size_t)k * d + c] = w[(size_t)c * 9 + k];
  T_TRY(up_raw(dw, wt.data(), (size_t)d * 9 * 4));
  T_TRY(up_raw(db, b, (size_t)d * 4));
  const size_t n = (size_t)B * T1 * F1 * d;
  T_TRY(dout.alloc(n * dt_size(dtype)));
  T_TRY(subsample_conv1(nullptr, dtype, (const float*)df.p, (const float*)dm.p, (const float*)di.p, (const float*)dw.p,
                        (const float*)db.p, dout.p, B, T0, F0, d));
  RVB_HIP_CHECK(hipDeviceSynchronize());
  return down_T(dout, dtype, false, out, n);
}

// conv_block.hip on host floats: x [B][F][T][32] (unbordered NHWC), wa / wb [32][32][3][3] as torch stores Conv2d weights (BatchNorm
// already folded by the caller), ba / bb [32]; out [B][F][T][32].  Builds the bordered planes and conv2d's weight layout.
int rvb_test_conv_block32(const float* x, const float* wa, const float* ba, const float* wb, const float* bb, float* out, int B, int F, int T) {
  T_TRY(need_gpu());
  if (!x || !wa || !ba || !wb || !bb || !out || B < 1 || F < 1 || T < 1) { set_error("rvb_test_conv_block32: bad argument"); return E_ARG; }
  const int FP = F + 2, TP = T + 2;
  const size_t np = (size_t)B * FP * TP * 32;
  std::vector<bf16_t> xb(np, 0);
  for (int b = 0; b < B; ++b)
    for (int f = 0; f < F; ++f)
      for (int t = 0; t < T; ++t)
        for (int c = 0; c < 32; ++c) xb[(((size_t)b * FP + f + 1) * TP + t + 1) * 32 + c] = f32_to_bf16(x[(((size_t)b * F + f) * T + t) * 32 + c]);
  auto pack_w = [](const float* w) {                 // [o][ci][kh][kw] -> [tap][1][o][ci]
    std::vector<bf16_t> p((size_t)9 * 32 * 32);
    for (int o = 0; o < 32; ++o)
      for (int ci = 0; ci < 32; ++ci)
        for (int t = 0; t < 9; ++t) p[((size_t)t * 32 + o) * 32 + ci] = f32_to_bf16(w[((size_t)o * 32 + ci) * 9 + t]);
    return p;
  };
  const std::vector<bf16_t> pa = pack_w(wa), pb = pack_w(wb);
  Dev dx, dwa, dwb, dba, dbb, dout;
  T_TRY(up_raw(dx, xb.data(), np * 2)); T_TRY(up_raw(dwa, pa.data(), pa.size() * 2)); T_TRY(up_raw(dwb, pb.data(), pb.size() * 2));
  T_TRY(up_raw(dba, ba, 32 * 4)); T_TRY(up_raw(dbb, bb, 32 * 4));
  T_TRY(dout.alloc(np * 2 + 256)); RVB_HIP_CHECK(hipMemset(dout.p, 0, np * 2 + 256));
  if (!conv_block32_applicable(DT_BF16, 32, 32, 32, 1, 1, 9, 9, F, T)) { set_error("rvb_test_conv_block32: the fused block is switched off (RVD_CONV_BLOCK=0)"); return E_STATE; }
  ConvBlockArgs a{};
  a.in = dx.p; a.wa = dwa.p; a.ba = (const float*)dba.p; a.wb = dwb.p; a.bb = (const float*)dbb.p; a.out = dout.p; a.B = B; a.F = F; a.T = T;
  T_TRY(conv_block32(nullptr, a));
  RVB_HIP_CHECK(hipDeviceSynchronize());
  std::vector<bf16_t> ob(np);
  RVB_HIP_CHECK(hipMemcpy(ob.data(), dout.p, np * 2, hipMemcpyDeviceToHost));
  for (int b = 0; b < B; ++b)
    for (int f = 0; f < F; ++f)
      for (int t = 0; t < T; ++t)
        for (int c = 0; c < 32; ++c) out[(((size_t)b * F + f) * T + t) * 32 + c] = bf16_to_f32(ob[(((size_t)b * FP + f + 1) * TP + t + 1) * 32 + c]);
  // the zero border of the output plane must be untouched (the next convolution relies on it)
  for (int b = 0; b < B; ++b)
    for (int f = 0; f < FP; ++f)
      for (int t = 0; t < TP; ++t)
        if (f == 0 || f == FP - 1 || t == 0 || t == TP - 1)
          for (int c = 0; c < 32; ++c)
            if (ob[(((size_t)b * FP + f) * TP + t) * 32 + c] != 0) { set_error("rvb_test_conv_block32: the kernel wrote into the zero border"); return E_STATE; }
  return OK;
}

int rvb_test_conv_s2sc(const float* x, const float* w, const float* b, const float* wsc, const float* bsc, float* out, float* sc, int B, int Fi, int Ti) {
  T_TRY(need_gpu());
  if (!x || !w || !b || !wsc || !bsc || !out || !sc || B < 1 || Fi < 1 || Ti < 1) { set_error("rvb_test_conv_s2sc: bad argument"); return E_ARG; }
  const int Fo = (Fi - 1) / 2 + 1, To = (Ti - 1) / 2 + 1;
  const int FPi = Fi + 2, TPi = Ti + 2, FPo = Fo + 2, TPo = To + 2;
  const size_t ni = (size_t)B * FPi * TPi * 32, no = (size_t)B * FPo * TPo * 64;
  std::vector<bf16_t> xb(ni, 0);
  for (int bb = 0; bb < B; ++bb)
    for (int f = 0; f < Fi; ++f)
      for (int t = 0; t < Ti; ++t)
        for (int c = 0; c < 32; ++c) xb[(((size_t)bb * FPi + f + 1) * TPi + t + 1) * 32 + c] = f32_to_bf16(x[(((size_t)bb * Fi + f) * Ti + t) * 32 + c]);
  std::vector<bf16_t> pw((size_t)9 * 64 * 32), ps((size_t)64 * 32);      // [o][ci][kh][kw] -> [tap][1][o][ci];  [o][ci] as it is
  for (int o = 0; o < 64; ++o)
    for (int ci = 0; ci < 32; ++ci) {
      for (int t = 0; t < 9; ++t) pw[((size_t)t * 64 + o) * 32 + ci] = f32_to_bf16(w[((size_t)o * 32 + ci) * 9 + t]);
      ps[(size_t)o * 32 + ci] = f32_to_bf16(wsc[(size_t)o * 32 + ci]);
    }
  Dev dx, dw, ds, db, dbs, dout, dsc;
  T_TRY(up_raw(dx, xb.data(), ni * 2)); T_TRY(up_raw(dw, pw.data(), pw.size() * 2)); T_TRY(up_raw(ds, ps.data(), ps.size() * 2));
  T_TRY(up_raw(db, b, 64 * 4)); T_TRY(up_raw(dbs, bsc, 64 * 4));
  T_TRY(dout.alloc(no * 2 + 256)); RVB_HIP_CHECK(hipMemset(dout.p, 0, no * 2 + 256));
  T_TRY(dsc.alloc(no * 2 + 256)); RVB_HIP_CHECK(hipMemset(dsc.p, 0, no * 2 + 256));
  if (!conv_s2sc_applicable(DT_BF16, 32, 64, 2, 9, 32, 64, 2, 1, Fi, Ti, Fo, To)) { set_error("rvb_test_conv_s2sc: switched off (RVD_CONV_S2SC=0)"); return E_STATE; }
  ConvS2Args a{};
  a.in = dx.p; a.w = dw.p; a.bias = (const float*)db.p; a.wsc = ds.p; a.bsc = (const float*)dbs.p; a.out = dout.p; a.sc = dsc.p;
  a.B = B; a.Fi = Fi; a.Ti = Ti; a.Fo = Fo; a.To = To;
  T_TRY(conv_s2sc(nullptr, a));
  RVB_HIP_CHECK(hipDeviceSynchronize());
  std::vector<bf16_t> ob(no), sb(no);
  RVB_HIP_CHECK(hipMemcpy(ob.data(), dout.p, no * 2, hipMemcpyDeviceToHost));
  RVB_HIP_CHECK(hipMemcpy(sb.data(), dsc.p, no * 2, hipMemcpyDeviceToHost));
  for (int bb = 0; bb < B; ++bb)
    for (int f = 0; f < FPo; ++f)
      for (int t = 0; t < TPo; ++t)
        for (int c = 0; c < 64; ++c) {
          const size_t at = (((size_t)bb * FPo + f) * TPo + t) * 64 + c;
          if (f == 0 || f == FPo - 1 || t == 0 || t == TPo - 1) {
            // the zero border of both output planes must be untouched (the next convolution relies on it)
            if (ob[at] != 0 || sb[at] != 0) { set_error("rvb_test_conv_s2sc: the kernel wrote into the zero border"); return E_STATE; }
          } else {
            const size_t o = (((size_t)bb * Fo + f - 1) * To + t - 1) * 64 + c;
            out[o] = bf16_to_f32(ob[at]); sc[o] = bf16_to_f32(sb[at]);
          }
        }
  return OK;
}

int rvb_test_glu_dwconv(int dtype, const float* G, const float* pw1_bias, const float* dw_w, const float* dw_b,
                        const int32_t* lens, float* out, int B, int T, int d, int K, int causal, const float* hist,
                        int hist_rows) {
  T_TRY(need_gpu());
  Dev dG, dpb, dw, db, dl, dout, dh;
  T_TRY(up_T(dG, dtype, G, (size_t)B * T * 2 * d));
  T_TRY(up_raw(dpb, pw1_bias, (size_t)2 * d * 4));
  std::vector<float> wt((size_t)d * K);            // the caller passes depthwise_conv.weight as the reference stores it, [d][K]
  for (int c = 0; c < d; ++c)
    for (int k = 0; k < K; ++k) wt[(size_t)k * d + c] = dw_w[(size_t)c * K + k];
  T_TRY(up_raw(dw, wt.data(), (size_t)d * K * 4));
  T_TRY(up_raw(db, dw_b, (size_t)d * 4));
  T_TRY(up_raw(dl, lens, (size_t)B * 4));
  const bool o16 = (causal & 2) != 0;            // bit 1 of `causal`: bf16 output (bf16 engine)
  causal &= 1;
  T_TRY(dout.alloc((size_t)B * T * d * 4));
  GluDwArgs a;
  a.out_bf16 = o16 ? 1 : 0;
  a.G = dG.p; a.pw1_bias = (const float*)dpb.p; a.dw_w = (const float*)dw.p; a.dw_b = (const float*)db.p;
  a.lens = (const int*)dl.p; a.out = (float*)dout.p; a.B = B; a.T = T; a.d = d; a.K = K;
  a.causal = causal;
  if (hist && hist_rows > 0) {      // [K-1][2d], the last hist_rows rows are real frames
    T_TRY(up_T(dh, dtype, hist, (size_t)(K - 1) * 2 * d));
    a.hist = dh.p; a.hist_rows = hist_rows;
  }
  T_TRY(glu_dwconv(nullptr, dtype, a));
  RVB_HIP_CHECK(hipDeviceSynchronize());
  if (o16) return down_T(dout, DT_BF16, false, out, (size_t)B * T * d);
  RVB_HIP_CHECK(hipMemcpy(out, dout.p, (size_t)B * T * d * 4, hipMemcpyDeviceToHost));
  return OK;
}

int rvb_test_attention(int dtype, const float* q, const float* k, const float* v, const float* p, const float* bias_u,
                       const float* bias_v, float* out, int q_rows, int kv_rows, int p_rows, int heads, int dk,
                       const int32_t* q_start, const int32_t* q_len, const int32_t* kv_start, const int32_t* kv_len,
                       int nseq, int causal) {
  T_TRY(need_gpu());
  const int d = heads * dk;
  Dev dq, dkk, dv, dp, du, dvv, dout, qs, ql, ks, kl;
  T_TRY(up_T(dq, dtype, q, (size_t)q_rows * d));
  // bit 2 of `causal` (with bit 1, bf16): the keys go up PREFOLDED, K' = k + p[position of the key in its sequence] summed in fp32 and
  // rounded once -- what the qkv GEMM's epilogue writes in the engine (GemmArgs::rowadd); the kernel then runs with k_prefolded
  const bool prefolded = (causal & 6) == 6 && p && dtype == DT_BF16;
  if (prefolded) {
    std::vector<float> kp(k, k + (size_t)kv_rows * d);
    for (int i = 0; i < nseq; ++i)
      for (int j = 0; j < kv_len[i] && j < p_rows; ++j)
        for (int c = 0; c < d; ++c) kp[(size_t)(kv_start[i] + j) * d + c] += p[(size_t)j * d + c];
    T_TRY(up_T(dkk, dtype, kp.data(), (size_t)kv_rows * d));
  } else {
    T_TRY(up_T(dkk, dtype, k, (size_t)kv_rows * d));
  }
  T_TRY(up_T(dv, dtype, v, (size_t)kv_rows * d));
  T_TRY(up_T(dp, dtype, p, (size_t)p_rows * d));
  T_TRY(up_raw(du, bias_u, (size_t)d * 4));
  T_TRY(up_raw(dvv, bias_v, (size_t)d * 4));
  T_TRY(up_raw(qs, q_start, (size_t)nseq * 4));
  T_TRY(up_raw(ql, q_len, (size_t)nseq * 4));
  T_TRY(up_raw(ks, kv_start, (size_t)nseq * 4));
  T_TRY(up_raw(kl, kv_len, (size_t)nseq * 4));
  T_TRY(dout.alloc((size_t)q_rows * d * dt_size(dtype)));
  RVB_HIP_CHECK(hipMemset(dout.p, 0, (size_t)q_rows * d * dt_size(dtype)));
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.q = dq.p; a.k = dkk.p; a.v = dv.p; a.p = dp.p;
  a.q_stride = a.k_stride = a.v_stride = a.p_stride = a.o_stride = d;
  a.bias_u = (const float*)du.p; a.bias_v = (const float*)dvv.p; a.out = dout.p;
  a.q_start = (const int*)qs.p; a.q_len = (const int*)ql.p; a.kv_start = (const int*)ks.p; a.kv_len = (const int*)kl.p;
  // `causal`: bit 0 = causal mask; bit 1 = folded positional term; bit 2 = ... with prefolded keys (see above); bits 8..19 = streaming chunk size (0 = off); bits 20..31 = left chunks + 1 (0 = all)
  a.nseq = nseq; a.heads = heads; a.dk = dk; a.causal = causal & 1; a.sqrt_dk = sqrtf((float)dk);
  a.chunk = (causal >> 8) & 0xfff; a.left = ((causal >> 20) & 0xfff) - 1;
  int mq = 0;
  for (int i = 0; i < nseq; ++i) mq = q_len[i] > mq ? q_len[i] : mq;
  a.max_q = mq;
  // bit 1 of `causal`: the bf16 encoder form with the positional term folded into per-key constants (as the engine runs it)
  Dev dc;
  if ((causal & 2) && p && dtype == DT_BF16) {
    T_TRY(dc.alloc((size_t)heads * p_rows * 4));
    T_TRY(attention_pos_bias(nullptr, dp.p, p_rows, d, (const float*)du.p, (const float*)dvv.p, heads, dk, 1.44269504f / sqrtf((float)dk),
                             (float*)dc.p));
    a.pos_bias = (const float*)dc.p; a.pos_bias_stride = p_rows;
    int mk = 0;
    for (int i = 0; i < nseq; ++i) mk = kv_len[i] > mk ? kv_len[i] : mk;
    a.fold_kv_cap = (mk + 63) / 64 * 64;
    a.k_prefolded = prefolded ? 1 : 0;
  }
  T_TRY(attention(nullptr, dtype, a));
  RVB_HIP_CHECK(hipDeviceSynchronize());
  return down_T(dout, dtype, false, out, (size_t)q_rows * d);
}

// ragged / shared-prefix form of the decoder self attention: keys through an index list, queries that start at
// position q_pos0 of their key sequence, 16-query blocks from a work list
int rvb_test_attention_trie(int dtype, const float* q, const float* k, const float* v, float* out, int rows, int heads, int dk,
                            const int32_t* q_start, const int32_t* q_len, const int32_t* q_pos0, const int32_t* kv_start,
                            const int32_t* kv_len, const int32_t* kv_index, int n_index, int nseq, int q_block) {
  T_TRY(need_gpu());
  const int d = heads * dk;
  Dev dq, dkk, dv, dout, qs, ql, qp, ks, kl, ki, wk;
  T_TRY(up_T(dq, dtype, q, (size_t)rows * d));
  T_TRY(up_T(dkk, dtype, k, (size_t)rows * d));
  T_TRY(up_T(dv, dtype, v, (size_t)rows * d));
  T_TRY(up_raw(qs, q_start, (size_t)nseq * 4)); T_TRY(up_raw(ql, q_len, (size_t)nseq * 4)); T_TRY(up_raw(qp, q_pos0, (size_t)nseq * 4));
  T_TRY(up_raw(ks, kv_start, (size_t)nseq * 4)); T_TRY(up_raw(kl, kv_len, (size_t)nseq * 4));
  T_TRY(up_raw(ki, kv_index, (size_t)n_index * 4));
  T_TRY(dout.alloc((size_t)rows * d * dt_size(dtype)));
  RVB_HIP_CHECK(hipMemset(dout.p, 0, (size_t)rows * d * dt_size(dtype)));
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.q = dq.p; a.k = dkk.p; a.v = dv.p;
  a.q_stride = a.k_stride = a.v_stride = a.o_stride = d; a.out = dout.p;
  a.q_start = (const int*)qs.p; a.q_len = (const int*)ql.p; a.q_pos0 = (const int*)qp.p;
  a.kv_start = (const int*)ks.p; a.kv_len = (const int*)kl.p; a.kv_index = (const int*)ki.p;
  a.nseq = nseq; a.heads = heads; a.dk = dk; a.causal = 1; a.sqrt_dk = sqrtf((float)dk); a.q_block = q_block;
  int mq = 0;
  std::vector<int32_t> work;
  const int qb = q_block == 16 ? 16 : 128;
  for (int i = 0; i < nseq; ++i) {
    mq = q_len[i] > mq ? q_len[i] : mq;
    for (int q0 = 0; q0 < q_len[i]; q0 += qb) { work.push_back(i); work.push_back(q0); }
  }
  a.max_q = mq;
  if (q_block == 16) {      // the work-list launch (what the engine uses); q_block 0: the plain (x, z) grid
    T_TRY(up_raw(wk, work.data(), work.size() * 4));
    a.work = (const int*)wk.p; a.n_work = (int)work.size() / 2;
  }
  T_TRY(attention(nullptr, dtype, a));
  RVB_HIP_CHECK(hipDeviceSynchronize());
  return down_T(dout, dtype, false, out, (size_t)rows * d);
}

int rvb_test_logsoftmax_topk(const float* logits, int M, int V, int k, float blank_penalty, int blank_id,
                             float* topk_val, int32_t* topk_idx, float* logp) {
  T_TRY(need_gpu());
  Dev dl, dv, di, dp;
  T_TRY(up_raw(dl, logits, (size_t)M * V * 4));
  T_TRY(dv.alloc((size_t)M * k * 4));
  T_TRY(di.alloc((size_t)M * k * 4));
  if (logp) T_TRY(dp.alloc((size_t)M * V * 4));
  T_TRY(logsoftmax_topk(nullptr, (const float*)dl.p, M, V, V, k, blank_penalty, blank_id, (float*)dv.p, (int*)di.p,
                        (float*)dp.p));
  RVB_HIP_CHECK(hipDeviceSynchronize());
  RVB_HIP_CHECK(hipMemcpy(topk_val, dv.p, (size_t)M * k * 4, hipMemcpyDeviceToHost));
  RVB_HIP_CHECK(hipMemcpy(topk_idx, di.p, (size_t)M * k * 4, hipMemcpyDeviceToHost));
  if (logp) RVB_HIP_CHECK(hipMemcpy(logp, dp.p, (size_t)M * V * 4, hipMemcpyDeviceToHost));
  return OK;
}

int rvb_test_lse_gather(const float* logits, int R, int V, const int32_t* target, float* out) {
  T_TRY(need_gpu());
  Dev dl, dt, dout;
  T_TRY(up_raw(dl, logits, (size_t)R * V * 4));
  T_TRY(up_raw(dt, target, (size_t)R * 4));
  T_TRY(dout.alloc((size_t)R * 4));
  T_TRY(lse_gather(nullptr, (const float*)dl.p, R, V, V, (const int*)dt.p, (float*)dout.p));
  RVB_HIP_CHECK(hipDeviceSynchronize());
  RVB_HIP_CHECK(hipMemcpy(out, dout.p, (size_t)R * 4, hipMemcpyDeviceToHost));
  return OK;
}

int rvb_test_lse_gather_multi(const float* logits, int R, int V, const int32_t* ptr, const int32_t* target, int P, float* out) {
  T_TRY(need_gpu());
  Dev dl, dp, dt, dout;
  T_TRY(up_raw(dl, logits, (size_t)R * V * 4));
  T_TRY(up_raw(dp, ptr, (size_t)(R + 1) * 4));
  T_TRY(up_raw(dt, target, (size_t)P * 4));
  T_TRY(dout.alloc((size_t)(P > 0 ? P : 1) * 4));
  T_TRY(lse_gather_multi(nullptr, (const float*)dl.p, R, V, V, (const int*)dp.p, (const int*)dt.p, (float*)dout.p));
  RVB_HIP_CHECK(hipDeviceSynchronize());
  if (P > 0) RVB_HIP_CHECK(hipMemcpy(out, dout.p, (size_t)P * 4, hipMemcpyDeviceToHost));
  return OK;
}

// fp8 GEMM of gemm2.hip on host floats: A is quantised per tensor (a_scale), W per output channel, exactly as the engine
// does; a_deq / w_deq (nullable) receive the values the quantised operands stand for, so that the caller's fp64 reference
// isolates the kernel from the quantisation.  out_kind 0 bf16, 1 fp32, 2 fp8 (values are returned de-quantised).
int rvb_test_gemm_fp8(const float* A, const float* W, const float* bias, const float* res, float* C, int M, int N, int K,
                      float a_scale, float alpha, int act, int out_kind, float out_scale, float* a_deq, float* w_deq) {
  T_TRY(need_gpu());
  std::vector<uint8_t> qa((size_t)M * K), qw((size_t)N * K);
  std::vector<float> ws(N);
  for (size_t i = 0; i < qa.size(); ++i) { qa[i] = f32_to_fp8_host(A[i] / a_scale); if (a_deq) a_deq[i] = fp8_to_f32_host(qa[i]) * a_scale; }
  for (int n = 0; n < N; ++n) {
    float am = 0.f;
    for (int k = 0; k < K; ++k) am = fmaxf(am, fabsf(W[(size_t)n * K + k]));
    ws[n] = am > 0.f ? am / 448.f : 1.f;
    for (int k = 0; k < K; ++k) {
      qw[(size_t)n * K + k] = f32_to_fp8_host(W[(size_t)n * K + k] / ws[n]);
      if (w_deq) w_deq[(size_t)n * K + k] = fp8_to_f32_host(qw[(size_t)n * K + k]) * ws[n];
    }
  }
  Dev dA, dW, dS, dB, dR, dC;
  T_TRY(up_raw(dA, qa.data(), qa.size())); T_TRY(up_raw(dW, qw.data(), qw.size())); T_TRY(up_raw(dS, ws.data(), (size_t)N * 4));
  T_TRY(up_raw(dB, bias, (size_t)N * 4)); T_TRY(up_raw(dR, res, (size_t)M * N * 4));
  const size_t osz = out_kind == 1 ? 4 : out_kind == 2 ? 1 : 2;
  T_TRY(dC.alloc((size_t)M * N * osz));
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = dA.p; g.W = dW.p; g.bias = (const float*)dB.p; g.res = (const float*)dR.p; g.C = dC.p;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.ldres = N; g.alpha = alpha; g.act = act;
  g.out_f32 = out_kind == 1; g.out_fp8 = out_kind == 2; g.in_fp8 = 1; g.a_scale = a_scale; g.w_scale = (const float*)dS.p;
  g.out_inv_scale = 1.f / out_scale;
  T_TRY(gemm(nullptr, DT_BF16, g));
  RVB_HIP_CHECK(hipDeviceSynchronize());
  if (out_kind == 1) return down_T(dC, DT_F32, true, C, (size_t)M * N);
  if (out_kind == 0) return down_T(dC, DT_BF16, false, C, (size_t)M * N);
  std::vector<uint8_t> q((size_t)M * N);
  RVB_HIP_CHECK(hipMemcpy(q.data(), dC.p, q.size(), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < q.size(); ++i) C[i] = fp8_to_f32_host(q[i]) * out_scale;
  return OK;
}

// fp8 implicit-GEMM convolution of conv_gemm.hip (round 4 candidate) on host floats.  x [B][Fi][Ti][Cin] and w [Cout][9][Cin] are
// quantised as the engine would (per tensor / per output channel) and laid out as the kernel wants them (bordered NHWC); x_deq /
// w_deq (nullable) receive the values the quantised operands stand for.  res (nullable) [B][Fo][To][Cout] is rounded to bf16.
// Outputs (either nullable): out [B][Fo][To][Cout] from the kernel's bf16 tensor, out8 the same from its e4m3 tensor (de-quantised
// with out8_scale); *amax receives the running maximum the kernel recorded.
int rvb_test_conv_igemm_fp8(const float* x, const float* w, const float* bias, const float* res, float* out, float* out8, int B,
                            int Fi, int Ti, int Cin, int Cout, int stride, int relu, float a_scale, float out8_scale, float* x_deq,
                            float* w_deq, float* amax) {
  T_TRY(need_gpu());
  const int Fo = (Fi - 1) / stride + 1, To = (Ti - 1) / stride + 1;
  const size_t npi = (size_t)B * (Fi + 2) * (Ti + 2) * Cin, npo = (size_t)B * (Fo + 2) * (To + 2) * Cout;
  std::vector<uint8_t> qx(npi, 0), qw((size_t)Cout * 9 * Cin);
  for (int b = 0; b < B; ++b)
    for (int f = 0; f < Fi; ++f)
      for (int t = 0; t < Ti; ++t)
        for (int c = 0; c < Cin; ++c) {
          const size_t si = (((size_t)b * Fi + f) * Ti + t) * Cin + c;
          const uint8_t q = f32_to_fp8_host(x[si] / a_scale);
          qx[(((size_t)b * (Fi + 2) + f + 1) * (Ti + 2) + t + 1) * Cin + c] = q;
          if (x_deq) x_deq[si] = fp8_to_f32_host(q) * a_scale;
        }
  std::vector<float> ws(Cout);
  const size_t K = (size_t)9 * Cin;
  for (int n = 0; n < Cout; ++n) {
    float am = 0.f;
    for (size_t k = 0; k < K; ++k) am = fmaxf(am, fabsf(w[(size_t)n * K + k]));
    ws[n] = am > 0.f ? am / 448.f : 1.f;
    for (size_t k = 0; k < K; ++k) {
      qw[(size_t)n * K + k] = f32_to_fp8_host(w[(size_t)n * K + k] / ws[n]);
      if (w_deq) w_deq[(size_t)n * K + k] = fp8_to_f32_host(qw[(size_t)n * K + k]) * ws[n];
    }
  }
  Dev dx, dw, ds, db, dr, dout, dout8, dam;
  T_TRY(up_raw(dx, qx.data(), qx.size())); T_TRY(up_raw(dw, qw.data(), qw.size())); T_TRY(up_raw(ds, ws.data(), (size_t)Cout * 4));
  T_TRY(up_raw(db, bias, (size_t)Cout * 4));
  if (res) {
    std::vector<bf16_t> rb(npo, 0);
    for (int b = 0; b < B; ++b)
      for (int f = 0; f < Fo; ++f)
        for (int t = 0; t < To; ++t)
          for (int c = 0; c < Cout; ++c)
            rb[(((size_t)b * (Fo + 2) + f + 1) * (To + 2) + t + 1) * Cout + c] = f32_to_bf16(res[(((size_t)b * Fo + f) * To + t) * Cout + c]);
    T_TRY(up_raw(dr, rb.data(), npo * 2));
  }
  if (out) { T_TRY(dout.alloc(npo * 2)); RVB_HIP_CHECK(hipMemset(dout.p, 0, npo * 2)); }
  if (out8) { T_TRY(dout8.alloc(npo)); RVB_HIP_CHECK(hipMemset(dout8.p, 0, npo)); }
  T_TRY(dam.alloc(4)); RVB_HIP_CHECK(hipMemset(dam.p, 0, 4));
  ConvArgs a{};
  a.bias = (const float*)db.p; a.res = dr.p; a.out = dout.p;
  a.B = B; a.Fi = Fi; a.Ti = Ti; a.Cin = Cin; a.Fo = Fo; a.To = To; a.Cout = Cout; a.stride = stride; a.taps = 9; a.relu = relu;
  a.in8 = dx.p; a.w8 = dw.p; a.w8_scale = (const float*)ds.p; a.a_scale = a_scale; a.out8 = dout8.p; a.out8_inv_scale = 1.f / out8_scale;
  a.amax8 = (unsigned*)dam.p;
  if (!conv_igemm8_applicable(DT_BF16, a)) { set_error("rvb_test_conv_igemm_fp8: shape not supported (channels multiples of 128, 3x3, stride 1 | 2)"); return E_ARG; }
  T_TRY(conv_igemm8(nullptr, a));
  RVB_HIP_CHECK(hipDeviceSynchronize());
  if (amax) RVB_HIP_CHECK(hipMemcpy(amax, dam.p, 4, hipMemcpyDeviceToHost));
  if (out) {
    std::vector<bf16_t> ob(npo);
    RVB_HIP_CHECK(hipMemcpy(ob.data(), dout.p, npo * 2, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b)
      for (int f = 0; f < Fo; ++f)
        for (int t = 0; t < To; ++t)
          for (int c = 0; c < Cout; ++c)
            out[(((size_t)b * Fo + f) * To + t) * Cout + c] = bf16_to_f32(ob[(((size_t)b * (Fo + 2) + f + 1) * (To + 2) + t + 1) * Cout + c]);
  }
  if (out8) {
    std::vector<uint8_t> o8(npo);
    RVB_HIP_CHECK(hipMemcpy(o8.data(), dout8.p, npo, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b)
      for (int f = 0; f < Fo; ++f)
        for (int t = 0; t < To; ++t)
          for (int c = 0; c < Cout; ++c)
            out8[(((size_t)b * Fo + f) * To + t) * Cout + c] = fp8_to_f32_host(o8[(((size_t)b * (Fo + 2) + f + 1) * (To + 2) + t + 1) * Cout + c]) * out8_scale;
  }
  return OK;
}

// LayerNorm with fp8 outputs (first stage and / or the fused second LayerNorm); results returned de-quantised
int rvb_test_rownorm_fp8(const float* x, const float* gamma, const float* beta, float eps, int silu, int M, int d, float scale,
                         float* out, const float* gamma2, const float* beta2, float eps2, float scale2, float* out1_f32, float* out2) {
  T_TRY(need_gpu());
  Dev dx, dg, db, dg2, db2, dout, dout2;
  T_TRY(up_raw(dx, x, (size_t)M * d * 4)); T_TRY(up_raw(dg, gamma, (size_t)d * 4)); T_TRY(up_raw(db, beta, (size_t)d * 4));
  T_TRY(up_raw(dg2, gamma2, (size_t)d * 4)); T_TRY(up_raw(db2, beta2, (size_t)d * 4));
  NormArgs a;
  a.x = (const float*)dx.p; a.gamma = (const float*)dg.p; a.beta = (const float*)db.p; a.eps = eps; a.mode = NORM_LN; a.silu = silu;
  a.add = nullptr; a.M = M; a.d = d;
  const bool two = gamma2 != nullptr;
  if (two) {       // stage 1 fp32 (in place semantics of norm_final), stage 2 fp8
    T_TRY(dout.alloc((size_t)M * d * 4)); T_TRY(dout2.alloc((size_t)M * d));
    a.out = dout.p; a.out_f32 = 1; a.gamma2 = (const float*)dg2.p; a.beta2 = (const float*)db2.p; a.eps2 = eps2; a.out2 = dout2.p;
    a.out2_fp8 = 1; a.out2_inv_scale = 1.f / scale2;
  } else {
    T_TRY(dout.alloc((size_t)M * d));
    a.out = dout.p; a.out_f32 = 0; a.out_fp8 = 1; a.out_inv_scale = 1.f / scale;
  }
  T_TRY(rownorm(nullptr, DT_BF16, a));
  RVB_HIP_CHECK(hipDeviceSynchronize());
  std::vector<uint8_t> q((size_t)M * d);
  if (two) {
    RVB_HIP_CHECK(hipMemcpy(out1_f32, dout.p, (size_t)M * d * 4, hipMemcpyDeviceToHost));
    RVB_HIP_CHECK(hipMemcpy(q.data(), dout2.p, q.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < q.size(); ++i) out2[i] = fp8_to_f32_host(q[i]) * scale2;
  } else {
    RVB_HIP_CHECK(hipMemcpy(q.data(), dout.p, q.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < q.size(); ++i) out[i] = fp8_to_f32_host(q[i]) * scale;
  }
  return OK;
}

// host only: the joint_decoding state machine (search.cpp: JointSearch), driven frame by frame by a caller that supplies the
// attention log-probs -- the CPU tests do that with the oracle's decoder, the engine with its own (rvb_joint_decode)
void* rvb_test_joint_new(int beam, int pre_beam, int blank, int sos, double w_ctc, double w_dec, double bonus) {
  JointParams p;
  p.beam = beam; p.pre_beam = pre_beam; p.blank = blank; p.sos = sos; p.w_ctc = w_ctc; p.w_dec = w_dec; p.bonus = bonus;
  return new JointSearch(p);
}
void rvb_test_joint_free(void* h) { delete (JointSearch*)h; }
// -> 0 frame skipped, 1 processed.  decode[n_decode] = nodes whose decoder row is needed, pairs (pair_node[i], pair_tok[i])
int rvb_test_joint_begin(void* h, int t, const float* tv, const int32_t* ti, int K, float p_tok0, float p_blank, int32_t* decode,
                         int32_t* n_decode, int32_t* pair_node, int32_t* pair_tok, int32_t* n_pairs, int cap) {
  if (!h || !tv || !ti || !n_decode || !n_pairs) { set_error("rvb_test_joint_begin: bad argument"); return E_ARG; }
  std::vector<int> d, pn, pt;
  const bool ran = ((JointSearch*)h)->begin_frame(t, tv, (const int*)ti, K, p_tok0, p_blank, &d, &pn, &pt);
  if ((int)d.size() > cap || (int)pn.size() > cap) { set_error("rvb_test_joint_begin: capacity"); return E_ARG; }
  *n_decode = (int32_t)d.size(); *n_pairs = (int32_t)pn.size();
  for (size_t i = 0; i < d.size(); ++i) decode[i] = d[i];
  for (size_t i = 0; i < pn.size(); ++i) { pair_node[i] = pn[i]; pair_tok[i] = pt[i]; }
  return ran ? 1 : 0;
}
int rvb_test_joint_finish(void* h, const float* vals) {
  if (!h) { set_error("rvb_test_joint_finish: bad argument"); return E_ARG; }
  ((JointSearch*)h)->finish_frame(vals);
  return OK;
}
int rvb_test_joint_prefix(void* h, int node, int32_t* toks, int32_t* n) {
  if (!h || !toks || !n) { set_error("rvb_test_joint_prefix: bad argument"); return E_ARG; }
  std::vector<int> t;
  ((JointSearch*)h)->prefix(node, &t);
  *n = (int32_t)t.size();
  for (size_t i = 0; i < t.size(); ++i) toks[i] = t[i];
  return OK;
}
int rvb_test_joint_result(void* h, int32_t* tokens, int32_t* times, int32_t* end_times, double* conf, int32_t* n, double* score) {
  if (!h || !n || !score) { set_error("rvb_test_joint_result: bad argument"); return E_ARG; }
  JointResult r;
  ((JointSearch*)h)->result(&r);
  *n = (int32_t)r.tokens.size(); *score = r.score;
  for (size_t i = 0; i < r.tokens.size(); ++i) { tokens[i] = r.tokens[i]; times[i] = r.times[i]; end_times[i] = r.end_times[i]; conf[i] = r.tokens_confidence[i]; }
  return OK;
}

int rvb_test_prefix_beam(const float* topk_val, const int32_t* topk_idx, int T, int beam, int blank, int32_t* n_hyps,
                         int32_t* tokens, int32_t* lens, int32_t* times, int32_t* times_lens, double* scores) {
  if (!topk_val || !topk_idx || !n_hyps || T < 0 || beam < 1) { set_error("rvb_test_prefix_beam: bad argument"); return E_ARG; }
  PrefixResult pr;
  prefix_beam_search(topk_val, topk_idx, T, beam, beam, blank, &pr);
  *n_hyps = (int32_t)pr.nbest.size();
  const int ml = T > 0 ? T : 1;
  for (size_t i = 0; i < pr.nbest.size(); ++i) {
    if (lens) lens[i] = (int32_t)pr.nbest[i].size();
    if (times_lens) times_lens[i] = (int32_t)pr.times[i].size();
    if (scores) scores[i] = pr.scores[i];
    for (int j = 0; j < ml; ++j) {
      if (tokens) tokens[i * ml + j] = j < (int)pr.nbest[i].size() ? pr.nbest[i][j] : -1;
      if (times) times[i * ml + j] = j < (int)pr.times[i].size() ? pr.times[i][j] : -1;
    }
  }
  return OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// GEMM micro-benchmark: random operands generated on the device, `variant` timed with HIP events,
// result compared against the gemm.hip kernel (variant 1) on sampled rows.
// ------------------------------------------------------------------------------------------------
namespace {
template <typename T>
__global__ void fill_random(T* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    p[i] = Cvt<T>::from_f32(((float)(x & 0xffffff) / 8388608.0f - 1.0f) * scale);
  }
}
template <typename T> void fill(void* p, size_t n, unsigned seed, float scale) {
  hipLaunchKernelGGL(fill_random<T>, dim3(2048), dim3(256), 0, nullptr, (T*)p, n, seed, scale);
}
}  // namespace

extern "C" int rvb_test_set_gemm_variant(int v) { g_gemm_variant = v; return OK; }
extern "C" int rvb_test_set_gemm2_opts(int flags, int group_m) { g_gemm2_flags = flags; g_gemm2_group_m = group_m; return OK; }

// per-workgroup phase timestamps of one bf16 gemm2 launch (scripts/gemm_timeline.py): out[6 * wg + {0..3}] = start / stage 0
// landed / main loop done / stores drained (10 ns ticks of the constant clock), [4] = HW_ID, [5] = XCC_ID
extern "C" int rvb_test_gemm_timeline(int M, int N, int K, int act, int out_f32, int with_res, long long* out, int cap, int* n_wg) {
  T_TRY(need_gpu());
  Dev dA, dW, dB, dR, dC, dT;
  T_TRY(dA.alloc((size_t)M * K * 2)); T_TRY(dW.alloc((size_t)N * K * 2)); T_TRY(dB.alloc((size_t)N * 4));
  const int padc = getenv("RVB_BENCH_PADC") ? atoi(getenv("RVB_BENCH_PADC")) : 0;      // probe: output / residual row stride off the power of two
  const int ldc = N + padc;
  T_TRY(dR.alloc((size_t)M * ldc * 4)); T_TRY(dC.alloc((size_t)M * ldc * (out_f32 ? 4 : 2)));
  fill<bf16_t>(dA.p, (size_t)M * K, 1u, 1.0f); fill<bf16_t>(dW.p, (size_t)N * K, 2u, 1.0f / sqrtf((float)K));
  fill<float>(dB.p, N, 3u, 1.0f); fill<float>(dR.p, (size_t)M * ldc, 4u, 1.0f);
  const int wgs = ((M + 255) / 256) * ((N + 255) / 256);
  if (wgs > cap) { set_error("rvb_test_gemm_timeline: output too small"); return E_ARG; }
  T_TRY(dT.alloc((size_t)wgs * 6 * 8));
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = dA.p; g.W = dW.p; g.bias = (const float*)dB.p; g.res = with_res ? (const float*)dR.p : nullptr; g.C = dC.p;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = ldc; g.ldres = ldc; g.alpha = 0.5f; g.act = act; g.out_f32 = out_f32;
  if (!gemm2_applicable(DT_BF16, g)) { set_error("rvb_test_gemm_timeline: shape not handled by gemm2"); return E_ARG; }
  for (int i = 0; i < 3; ++i) T_TRY(gemm2(nullptr, DT_BF16, g));
  g.dbg = (long long*)dT.p;
  T_TRY(gemm2(nullptr, DT_BF16, g));
  RVB_HIP_CHECK(hipDeviceSynchronize());
  RVB_HIP_CHECK(hipMemcpy(out, dT.p, (size_t)wgs * 6 * 8, hipMemcpyDeviceToHost));
  *n_wg = wgs;
  return OK;
}

extern "C" int rvb_test_gemm_bench(int dtype, int M, int N, int K, int variant, int iters, int act, int out_f32,
                                   int with_res, double* ms_out, double* max_abs_diff) {
  T_TRY(need_gpu());
  Dev dA, dW, dB, dR, dC, dC1;
  const size_t es = dt_size(dtype);
  const bool f32out = dtype == DT_F32 || out_f32;
  // RVB_BENCH_PAD=<elements>: pad the leading dimensions of A and W (probe for channel camping of 2^n row strides)
  const int pad = getenv("RVB_BENCH_PAD") ? atoi(getenv("RVB_BENCH_PAD")) : 0;
  const int ldk = K + pad;
  T_TRY(dA.alloc((size_t)M * ldk * es)); T_TRY(dW.alloc((size_t)N * ldk * es)); T_TRY(dB.alloc((size_t)N * 4));
  T_TRY(dR.alloc((size_t)M * N * 4)); T_TRY(dC.alloc((size_t)M * N * (f32out ? 4 : 2)));
  T_TRY(dC1.alloc((size_t)M * N * (f32out ? 4 : 2)));
  if (dtype == DT_BF16) { fill<bf16_t>(dA.p, (size_t)M * ldk, 1u, 1.0f); fill<bf16_t>(dW.p, (size_t)N * ldk, 2u, 1.0f / sqrtf((float)K)); }
  else { fill<float>(dA.p, (size_t)M * ldk, 1u, 1.0f); fill<float>(dW.p, (size_t)N * ldk, 2u, 1.0f / sqrtf((float)K)); }
  fill<float>(dB.p, N, 3u, 1.0f);
  fill<float>(dR.p, (size_t)M * N, 4u, 1.0f);
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = dA.p; g.W = dW.p; g.bias = (const float*)dB.p; g.res = with_res ? (const float*)dR.p : nullptr; g.C = dC.p;
  g.M = M; g.N = N; g.K = K; g.lda = ldk; g.ldw = ldk; g.ldc = N; g.ldres = N; g.alpha = 0.5f; g.act = act; g.out_f32 = out_f32;
  const int saved = g_gemm_variant;
  g_gemm_variant = variant;
  int r = gemm(nullptr, dtype, g);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  if (r == OK) {
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters && r == OK; ++i) r = gemm(nullptr, dtype, g);
    (void)hipEventRecord(e1, nullptr);
    if (hipDeviceSynchronize() != hipSuccess) { set_error("gemm bench kernel failed"); r = E_HIP; }
  }
  float ms = 0.f;
  if (r == OK) (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (ms_out) *ms_out = iters > 0 ? ms / iters : 0.0;
  if (r == OK && max_abs_diff) {
    g_gemm_variant = 1;
    g.C = dC1.p;
    r = gemm(nullptr, dtype, g);
    if (r == OK && hipDeviceSynchronize() != hipSuccess) r = E_HIP;
    if (r == OK) {
      const int rows = M < 512 ? M : 512;
      std::vector<float> a((size_t)rows * N), b((size_t)rows * N);
      double md = 0.0;
      for (int part = 0; part < 2 && r == OK; ++part) {
        const size_t off = part == 0 ? 0 : (size_t)(M - rows) * N;
        Dev va, vb;   // views
        va.p = (char*)dC.p + off * (f32out ? 4 : 2); vb.p = (char*)dC1.p + off * (f32out ? 4 : 2);
        r = down_T(va, dtype, f32out, a.data(), a.size());
        if (r == OK) r = down_T(vb, dtype, f32out, b.data(), b.size());
        va.p = nullptr; vb.p = nullptr;
        for (size_t i = 0; i < a.size(); ++i) { const double d = fabs((double)a[i] - (double)b[i]); if (d > md || d != d) md = d; }
      }
      *max_abs_diff = md;
    }
  }
  g_gemm_variant = saved;
  return r;
}
