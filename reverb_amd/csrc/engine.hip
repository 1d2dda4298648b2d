// librvb engine: weight packing, workspace and the encode / search / rescore pipeline.
// Orchestrates the kernels of gemm.hip, attention.hip, elementwise.hip, fbank.hip so that one call
// processes a whole batch of 20.51 s chunks (the reference loops chunk by chunk with batch 1,
// asr/wenet/cli/reverb.py:220-253).  Reference structure followed per stage is cited inline.
#include "engine.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace rvb {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* last_error() { return g_err.c_str(); }

int DevBuf::ensure(size_t n) {
  if (n <= bytes && p) return OK;
  if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
  if (n == 0) n = 16;
  hipError_t err = hipMalloc(&p, n);
  if (err != hipSuccess) {
    p = nullptr;
    (void)hipGetLastError();      // HIP's thread-local last error is sticky: a caller that retries with a smaller size must not
                                  // trip over this failure at its next hipGetLastError() check
    set_error("hipMalloc of " + std::to_string(n) + " bytes failed: " + hipGetErrorString(err));
    return E_NOMEM;
  }
  bytes = n;
  return OK;
}
void DevBuf::release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }

#define RVB_TRY(expr) do { int _r = (expr); if (_r != OK) return _r; } while (0)

// ------------------------------------------------------------------------------------ profiling
struct Scope {
  rvb_engine* e; hipEvent_t a = nullptr, b = nullptr; std::string name;
  Scope(rvb_engine* e_, const char* n, double flops = 0.0, double bytes = 0.0) : e(e_), name(n) {
    auto& pe = e->prof[name];
    pe.launches += 1; pe.flops += flops; pe.bytes += bytes;
    if (e->profiling == 0 || (e->profiling == 2 && name != "gemm" && name != "gemm_fp8")) return;
    auto get = [&]() { hipEvent_t ev; if (!e->event_pool.empty()) { ev = e->event_pool.back(); e->event_pool.pop_back(); } else (void)hipEventCreate(&ev); return ev; };
    a = get(); b = get();
    (void)hipEventRecord(a, e->stream);
  }
  ~Scope() {
    if (!a) return;
    (void)hipEventRecord(b, e->stream);
    e->pending.push_back({a, b, name});
  }
};
static void drain_prof(rvb_engine* e) {
  if (e->pending.empty()) return;
  (void)hipStreamSynchronize(e->stream);
  for (auto& p : e->pending) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, p.a, p.b);
    e->prof[p.name].ms += ms;
    e->event_pool.push_back(p.a); e->event_pool.push_back(p.b);
  }
  e->pending.clear();
}

// ------------------------------------------------------------------------------------ helpers
static int upload_f32(rvb_engine* e, DevBuf& dst, const float* src, size_t n) {
  RVB_TRY(dst.ensure(n * 4));
  RVB_HIP_CHECK(hipMemcpyAsync(dst.p, src, n * 4, hipMemcpyHostToDevice, e->stream));
  return OK;
}
static int upload_i32(rvb_engine* e, DevBuf& dst, const int32_t* src, size_t n) {
  RVB_TRY(dst.ensure(n * 4 + 16));
  if (n) RVB_HIP_CHECK(hipMemcpyAsync(dst.p, src, n * 4, hipMemcpyHostToDevice, e->stream));
  return OK;
}
// fp32 host matrix -> compute dtype on device
static int pack_T(rvb_engine* e, DevBuf& dst, const float* src, size_t n) {
  RVB_TRY(dst.ensure(n * dt_size(e->dtype)));
  if (e->dtype == DT_F32) {
    RVB_HIP_CHECK(hipMemcpyAsync(dst.p, src, n * 4, hipMemcpyHostToDevice, e->stream));
    return OK;
  }
  RVB_TRY(e->stage.ensure(n * 4));
  RVB_HIP_CHECK(hipMemcpyAsync(e->stage.p, src, n * 4, hipMemcpyHostToDevice, e->stream));
  return convert_f32(e->stream, e->dtype, e->stage.as<float>(), dst.p, n);
}
// f8: also keep an fp8 (OCP e4m3) copy with one scale per output channel: w8[n][k] = rne(w[n][k] / s_n), s_n = max|w[n]| / 448
static int pack_linear(rvb_engine* e, Linear& L, const float* w, const float* b, int out, int in, bool f8 = false) {
  L.out = out; L.in = in;
  RVB_TRY(pack_T(e, L.w, w, (size_t)out * in));
  if (b) RVB_TRY(upload_f32(e, L.b, b, out)); else L.b.release();
  if (f8 && e->fp8 && in % 128 == 0) {
    std::vector<uint8_t> q((size_t)out * in);
    std::vector<float> sc(out);
    for (int n = 0; n < out; ++n) {
      const float* row = w + (size_t)n * in;
      float am = 0.f;
      for (int k = 0; k < in; ++k) am = std::max(am, std::fabs(row[k]));
      const float sn = am > 0.f ? am / 448.f : 1.f;
      sc[n] = sn;
      const float inv = 1.f / sn;
      for (int k = 0; k < in; ++k) q[(size_t)n * in + k] = f32_to_fp8_host(row[k] * inv);
    }
    RVB_TRY(L.w8.ensure(q.size()));
    RVB_HIP_CHECK(hipMemcpyAsync(L.w8.p, q.data(), q.size(), hipMemcpyHostToDevice, e->stream));
    RVB_TRY(upload_f32(e, L.wscale, sc.data(), out));
    RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  }
  return OK;
}

static const HostTensor* find(rvb_engine* e, const std::string& name) {
  auto it = e->host.find(name);
  return it == e->host.end() ? nullptr : &it->second;
}
static int need(rvb_engine* e, const std::string& name, size_t numel, const HostTensor** out) {
  const HostTensor* t = find(e, name);
  if (!t) { set_error("missing tensor: " + name); return E_STATE; }
  if (t->numel() != numel) {
    set_error("tensor " + name + " has " + std::to_string(t->numel()) + " elements, expected " + std::to_string(numel));
    return E_ARG;
  }
  *out = t;
  return OK;
}
static int pack_named_linear(rvb_engine* e, Linear& L, const std::string& p, int out, int in, bool bias = true, bool f8 = false) {
  const HostTensor *w, *b = nullptr;
  RVB_TRY(need(e, p + ".weight", (size_t)out * in, &w));
  if (bias) RVB_TRY(need(e, p + ".bias", out, &b));
  return pack_linear(e, L, w->data.data(), b ? b->data.data() : nullptr, out, in, f8);
}
static int pack_norm(rvb_engine* e, LNorm& n, const std::string& p, int d, float eps) {
  const HostTensor *g, *b;
  RVB_TRY(need(e, p + ".weight", d, &g));
  RVB_TRY(need(e, p + ".bias", d, &b));
  n.eps = eps;
  RVB_TRY(upload_f32(e, n.g, g->data.data(), d));
  return upload_f32(e, n.b, b->data.data(), d);
}
// concatenate several [rows_i, in] linears along the output dim
static int pack_concat(rvb_engine* e, Linear& L, const std::vector<std::string>& names, int out_each, int in, bool f8 = false) {
  std::vector<float> w((size_t)names.size() * out_each * in), b((size_t)names.size() * out_each);
  for (size_t i = 0; i < names.size(); ++i) {
    const HostTensor *tw, *tb;
    RVB_TRY(need(e, names[i] + ".weight", (size_t)out_each * in, &tw));
    RVB_TRY(need(e, names[i] + ".bias", out_each, &tb));
    memcpy(w.data() + i * (size_t)out_each * in, tw->data.data(), (size_t)out_each * in * 4);
    memcpy(b.data() + i * (size_t)out_each, tb->data.data(), (size_t)out_each * 4);
  }
  int r = pack_linear(e, L, w.data(), b.data(), (int)names.size() * out_each, in, f8);
  if (r == OK) (void)hipStreamSynchronize(e->stream);   // w/b go out of scope
  return r;
}
// language-specific layers folded with the category weights: W = sum_i c_i W_i, b = sum_i c_i b_i
// (encoder_layer.py:378-390, decoder_layer.py:319-330 with 1-D cat_embs)
static int pack_lsl(rvb_engine* e, Linear& L, const std::string& p, int d, const float* cat, int ncat) {
  std::vector<float> w((size_t)d * d, 0.f), b(d, 0.f);
  for (int i = 0; i < ncat; ++i) {
    const HostTensor *tw, *tb;
    const std::string n = p + ".language_layers." + std::to_string(i);
    RVB_TRY(need(e, n + ".weight", (size_t)d * d, &tw));
    RVB_TRY(need(e, n + ".bias", d, &tb));
    const float c = cat[i];
    if (i == 0) {
      for (size_t k = 0; k < w.size(); ++k) w[k] = c * tw->data[k];
      for (int k = 0; k < d; ++k) b[k] = c * tb->data[k];
    } else {
      for (size_t k = 0; k < w.size(); ++k) w[k] = w[k] + c * tw->data[k];
      for (int k = 0; k < d; ++k) b[k] = b[k] + c * tb->data[k];
    }
  }
  int r = pack_linear(e, L, w.data(), b.data(), d, d);
  if (r == OK) (void)hipStreamSynchronize(e->stream);
  return r;
}

// sinusoid table, transformer/embedding.py:48-56 (float32 arithmetic as torch does it)
static void make_pe(int rows, int d, std::vector<float>* pe) {
  pe->assign((size_t)rows * d, 0.f);
  const float c = (float)(-(std::log(10000.0) / (double)d));
  for (int i = 0; i < d; i += 2) {
    const float div = std::exp((float)i * c);
    for (int pos = 0; pos < rows; ++pos) {
      const float ang = (float)pos * div;
      (*pe)[(size_t)pos * d + i] = std::sin(ang);
      if (i + 1 < d) (*pe)[(size_t)pos * d + i + 1] = std::cos(ang);
    }
  }
}

// Kaldi mel banks / povey window / FFT twiddles (see oracle/fbank_ref.py for the restatement)
static int make_fbank_tables(rvb_engine* e) {
  const int WIN = 400, NFFT = 512, NBIN = 257, NMEL = 80;
  const double PI = 3.14159265358979323846;
  std::vector<float> win(WIN), tw(2 * 256), melw((size_t)NMEL * NBIN, 0.f);
  std::vector<int32_t> lo(NMEL, NBIN), hi(NMEL, 0);
  for (int i = 0; i < WIN; ++i) win[i] = (float)std::pow(0.5 - 0.5 * std::cos(2.0 * PI * i / (WIN - 1)), 0.85);
  for (int k = 0; k < 256; ++k) { tw[2 * k] = (float)std::cos(2.0 * PI * k / NFFT); tw[2 * k + 1] = (float)(-std::sin(2.0 * PI * k / NFFT)); }
  auto mel = [](double f) { return 1127.0 * std::log(1.0 + f / 700.0); };
  const double mlo = mel(20.0), mhi = mel(8000.0), delta = (mhi - mlo) / (NMEL + 1);
  for (int m = 0; m < NMEL; ++m) {
    const double left = mlo + m * delta, center = left + delta, right = center + delta;
    for (int b = 0; b < NFFT / 2; ++b) {
      const double mf = mel(16000.0 / NFFT * b);
      const double up = (mf - left) / (center - left), down = (right - mf) / (right - center);
      const double w = std::max(0.0, std::min(up, down));
      if (w > 0.0) {
        melw[(size_t)m * NBIN + b] = (float)w;
        lo[m] = std::min(lo[m], b); hi[m] = std::max(hi[m], b + 1);
      }
    }
    if (hi[m] == 0) lo[m] = 0;
  }
  RVB_TRY(upload_f32(e, e->fb_window, win.data(), win.size()));
  RVB_TRY(upload_f32(e, e->fb_twiddle, tw.data(), tw.size()));
  RVB_TRY(upload_f32(e, e->fb_melw, melw.data(), melw.size()));
  RVB_TRY(upload_i32(e, e->fb_lo, lo.data(), lo.size()));
  RVB_TRY(upload_i32(e, e->fb_hi, hi.data(), hi.size()));
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  return OK;
}

static int out_frames(int T0) { const int T1 = (T0 - 3) / 2 + 1; return (T1 - 3) / 2 + 1; }

// ------------------------------------------------------------------------------------ gemm wrapper
// saturation counters of the fp8 activations: one row of 8 per block (+ a guard row: the last block's norm_final names "the next
// block"); zeroed whenever scales are (re)calibrated or installed
static int reset_f8sat(rvb_engine* e) {
  static const int on = lab_env("RVB_FP8_SAT") ? atoi(lab_env("RVB_FP8_SAT")) : 1;       // 0: no counters (A/B of their cost)
  if (!on) return OK;
  RVB_TRY(e->d_f8sat.ensure((e->enc.size() + 1) * 8 * 4));
  RVB_HIP_CHECK(hipMemsetAsync(e->d_f8sat.p, 0, (e->enc.size() + 1) * 8 * 4, e->stream));
  return OK;
}

// ALGORITHMIC HBM bytes of one GEMM launch: every operand once -- A [M,K] (the NHWC activation once for the implicit
// convolution, not its 9-fold gather), W [N,K], bias, C [M,N] written once, the fp32 residual read once
static double gemm_alg_bytes(const rvb_engine* e, const GemmArgs& g) {
  const double es = g.in_fp8 ? 1.0 : (double)dt_size(e->dtype);
  const double a = g.conv ? (double)(g.M / (g.cT2 * g.cF2)) * g.cT1 * g.cF1 * g.cC * es : (double)g.M * g.K * es;
  const double c = (double)g.M * (g.act == ACT_GLU ? g.N / 2 : g.N) * (g.out_fp8 ? 1.0 : (g.out_f32 ? 4.0 : (double)dt_size(e->dtype)));
  return a + (double)g.N * g.K * es + (g.bias ? 4.0 * g.N : 0.0) + c + (g.res ? 4.0 * (double)g.M * g.N : 0.0);
}

static constexpr int GLU_FUSE_DEFAULT = 1;      // pointwise_conv1 + GLU in the GEMM's epilogue (encoder_layer: lab switch RVB_GLU_FUSE)
static int run_gemm(rvb_engine* e, const void* A, int lda, const Linear& L, void* C, int ldc, int M, bool out_f32,
                    float alpha = 1.f, int act = ACT_NONE, const float* res = nullptr, int ldres = 0) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.W = L.w.p; g.bias = L.b.as<float>(); g.res = res; g.C = C;
  g.M = M; g.N = L.out; g.K = L.in; g.lda = lda; g.ldw = L.in; g.ldc = ldc; g.ldres = ldres;
  g.alpha = alpha; g.act = act; g.out_f32 = out_f32 ? 1 : 0;
  Scope sc(e, "gemm", 2.0 * M * (double)L.out * L.in, gemm_alg_bytes(e, g));
  return gemm(e->stream, e->dtype, g);
}
// out8 / out2_8 > 0: that output is fp8 bytes of value / scale (the calibrated per-tensor scale of the GEMM that reads it)
static int run_norm(rvb_engine* e, const float* x, const LNorm& n, void* out, bool out_f32, int M, int d,
                    int mode = NORM_LN, int silu = 0, const void* add = nullptr, const LNorm* second = nullptr,
                    void* out2 = nullptr, float out8 = 0.f, float out2_8 = 0.f, bool x_bf16 = false, unsigned* sat = nullptr,
                    unsigned* sat2 = nullptr) {
  NormArgs a;
  a.sat = sat; a.sat2 = sat2;
  a.x_bf16 = x_bf16 ? 1 : 0;
  a.x = x; a.gamma = n.g.as<float>(); a.beta = n.b.as<float>(); a.eps = n.eps; a.mode = mode; a.silu = silu;
  a.add = add; a.out = out; a.out_f32 = out_f32 ? 1 : 0; a.M = M; a.d = d;
  if (second) { a.gamma2 = second->g.as<float>(); a.beta2 = second->b.as<float>(); a.eps2 = second->eps; a.out2 = out2; }
  if (out8 > 0.f) { a.out_fp8 = 1; a.out_inv_scale = 1.f / out8; }
  if (out2_8 > 0.f) { a.out2_fp8 = 1; a.out2_inv_scale = 1.f / out2_8; }
  Scope sc(e, "rownorm");
  return rownorm(e->stream, e->dtype, a);
}
// fp8 GEMM: A8 [M, lda] bytes (values / a_scale), L.w8 / L.wscale; out_kind 0 = compute dtype, 1 = fp32, 2 = fp8 (/ out_scale)
static int run_gemm8(rvb_engine* e, const void* A8, int lda, const Linear& L, void* C, int ldc, int M, float a_scale, int out_kind,
                     float out_scale = 1.f, float alpha = 1.f, int act = ACT_NONE, const float* res = nullptr, int ldres = 0,
                     unsigned* sat = nullptr) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.sat = sat;
  g.A = A8; g.W = L.w8.p; g.bias = L.b.as<float>(); g.res = res; g.C = C;
  g.M = M; g.N = L.out; g.K = L.in; g.lda = lda; g.ldw = L.in; g.ldc = ldc; g.ldres = ldres;
  g.alpha = alpha; g.act = act; g.out_f32 = out_kind == 1; g.out_fp8 = out_kind == 2; g.in_fp8 = 1;
  g.a_scale = a_scale; g.w_scale = L.wscale.as<float>(); g.out_inv_scale = 1.f / out_scale;
  if (!L.w8.p) { set_error("fp8 GEMM on a layer without fp8 weights"); return E_STATE; }
  Scope sc(e, "gemm_fp8", 2.0 * M * (double)L.out * L.in, gemm_alg_bytes(e, g));
  return gemm(e->stream, e->dtype, g);
}

// ------------------------------------------------------------------------------------ finalize
static int pack_decoder(rvb_engine* e, Decoder& D, const std::string& p, int nblocks, const float* cat, int ncat) {
  const int d = e->cfg.d_model, V = e->cfg.vocab, ff = e->cfg.dec_ffn_dim;
  D.present = false;
  if (nblocks <= 0 || !find(e, p + ".embed.0.weight")) return OK;
  const HostTensor* emb;
  RVB_TRY(need(e, p + ".embed.0.weight", (size_t)V * d, &emb));
  RVB_TRY(upload_f32(e, D.embed, emb->data.data(), emb->data.size()));
  RVB_TRY(pack_norm(e, D.after, p + ".after_norm", d, 1e-5f));
  RVB_TRY(pack_named_linear(e, D.out, p + ".output_layer", V, d));
  D.layers.resize(nblocks);
  for (int j = 0; j < nblocks; ++j) {
    DecLayer& L = D.layers[j];
    const std::string q = p + ".decoders." + std::to_string(j);
    L.is_lsl = find(e, q + ".language_layers.0.weight") != nullptr;
    const float eps = L.is_lsl ? 1e-12f : 1e-5f;   // decoder_layer.py:56-58 vs :241-243
    RVB_TRY(pack_concat(e, L.self_qkv, {q + ".self_attn.linear_q", q + ".self_attn.linear_k", q + ".self_attn.linear_v"}, d, d));
    RVB_TRY(pack_named_linear(e, L.self_out, q + ".self_attn.linear_out", d, d));
    RVB_TRY(pack_named_linear(e, L.src_q, q + ".src_attn.linear_q", d, d));
    RVB_TRY(pack_concat(e, L.src_kv, {q + ".src_attn.linear_k", q + ".src_attn.linear_v"}, d, d));
    RVB_TRY(pack_named_linear(e, L.src_out, q + ".src_attn.linear_out", d, d));
    RVB_TRY(pack_named_linear(e, L.ff1, q + ".feed_forward.w_1", ff, d));
    RVB_TRY(pack_named_linear(e, L.ff2, q + ".feed_forward.w_2", d, ff));
    RVB_TRY(pack_norm(e, L.n1, q + ".norm1", d, eps));
    RVB_TRY(pack_norm(e, L.n2, q + ".norm2", d, eps));
    RVB_TRY(pack_norm(e, L.n3, q + ".norm3", d, eps));
    if (L.is_lsl) RVB_TRY(pack_lsl(e, L.lsl, q, d, cat, ncat));
  }
  D.present = true;
  return OK;
}

static int finalize_impl(rvb_engine* e, const float* cat, int ncat) {
  const rvb_model_cfg& c = e->cfg;
  const int d = c.d_model, ff = c.ffn_dim, K = c.cnn_kernel, V = c.vocab, F0 = c.input_dim;
  const int F1 = (F0 - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1;
  if (c.num_langs > 0 && ncat != c.num_langs) { set_error("finalize: cat_embs length must equal num_langs"); return E_ARG; }
  RVB_HIP_CHECK(hipSetDevice(e->device));

  if (!e->finalized) {
    const HostTensor *t, *t2;
    RVB_TRY(need(e, "encoder.global_cmvn.mean", F0, &t));
    RVB_TRY(upload_f32(e, e->cmvn_mean, t->data.data(), F0));
    RVB_TRY(need(e, "encoder.global_cmvn.istd", F0, &t));
    RVB_TRY(upload_f32(e, e->cmvn_istd, t->data.data(), F0));
    RVB_TRY(need(e, "encoder.embed.conv.0.weight", (size_t)d * 9, &t));
    {   // tap-major [9][d]: a thread of conv1_kernel reads its 8 channels of a tap as 32 contiguous bytes
      std::vector<float> wt((size_t)d * 9);
      for (int c = 0; c < d; ++c)
        for (int k = 0; k < 9; ++k) wt[(size_t)k * d + c] = t->data[(size_t)c * 9 + k];
      RVB_TRY(upload_f32(e, e->conv1_w, wt.data(), (size_t)d * 9));
    }
    RVB_TRY(need(e, "encoder.embed.conv.0.bias", d, &t));
    RVB_TRY(upload_f32(e, e->conv1_b, t->data.data(), d));
    {  // conv2 [co][ci][kh][kw] -> [co][(kh*3+kw)*d + ci]  (K-contiguous rows for the implicit GEMM)
      RVB_TRY(need(e, "encoder.embed.conv.2.weight", (size_t)d * d * 9, &t));
      RVB_TRY(need(e, "encoder.embed.conv.2.bias", d, &t2));
      std::vector<float> w((size_t)d * 9 * d);
      for (int co = 0; co < d; ++co)
        for (int ci = 0; ci < d; ++ci)
          for (int k = 0; k < 9; ++k) w[((size_t)co * 9 + k) * d + ci] = t->data[((size_t)co * d + ci) * 9 + k];
      RVB_TRY(pack_linear(e, e->conv2, w.data(), t2->data.data(), d, 9 * d, true));      // + fp8 copy in RVB_FP8 mode (policy bit 5)
      RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
    }
    {  // out.0 [o][c*F2+f] -> [o][f*d+c]: our conv2 output is (t, f, c) not the reference's (t, c, f)
      RVB_TRY(need(e, "encoder.embed.out.0.weight", (size_t)d * d * F2, &t));
      RVB_TRY(need(e, "encoder.embed.out.0.bias", d, &t2));
      std::vector<float> w((size_t)d * d * F2);
      for (int o = 0; o < d; ++o)
        for (int cc = 0; cc < d; ++cc)
          for (int f = 0; f < F2; ++f) w[(size_t)o * d * F2 + (size_t)f * d + cc] = t->data[(size_t)o * d * F2 + (size_t)cc * F2 + f];
      RVB_TRY(pack_linear(e, e->embed_out, w.data(), t2->data.data(), d, d * F2));
      RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
    }
    RVB_TRY(pack_norm(e, e->enc_after, "encoder.after_norm", d, 1e-5f));
    RVB_TRY(pack_named_linear(e, e->ctc, "ctc.ctc_lo", V, d));

    // sinusoid table for encoder positions and decoder positions
    const int Tmax = out_frames(c.chunk_frames);
    // 5000 rows = the reference's positional table (`max_len`, embedding.py:33,130): streaming offsets index it absolutely
    e->pe_rows = std::max(Tmax + 2, 5000);
    std::vector<float> pe;
    make_pe(e->pe_rows, d, &pe);
    RVB_TRY(upload_f32(e, e->pe_f32, pe.data(), pe.size()));
    DevBuf pe_T;
    RVB_TRY(pack_T(e, pe_T, pe.data(), pe.size()));
    RVB_HIP_CHECK(hipStreamSynchronize(e->stream));

    e->enc.resize(c.num_blocks);
    for (int i = 0; i < c.num_blocks; ++i) {
      EncLayer& L = e->enc[i];
      const std::string p = "encoder.encoders." + std::to_string(i);
      L.is_lsl = find(e, p + ".language_layers.0.weight") != nullptr;
      // fp8 mode: the GEMMs whose A operand is written by a LayerNorm or by a GEMM epilogue (95 % of a block's GEMM work);
      // the language-specific blocks keep their second feed-forward in bf16 (its input is the mixed projection y)
      RVB_TRY(pack_named_linear(e, L.ffm1, p + ".feed_forward_macaron.w_1", ff, d, true, true));
      RVB_TRY(pack_named_linear(e, L.ffm2, p + ".feed_forward_macaron.w_2", d, ff, true, true));
      RVB_TRY(pack_named_linear(e, L.ff1, p + ".feed_forward.w_1", ff, d, true, !L.is_lsl));
      RVB_TRY(pack_named_linear(e, L.ff2, p + ".feed_forward.w_2", d, ff, true, !L.is_lsl));
      RVB_TRY(pack_concat(e, L.qkv, {p + ".self_attn.linear_q", p + ".self_attn.linear_k", p + ".self_attn.linear_v"}, d, d, true));
      RVB_TRY(pack_named_linear(e, L.att_out, p + ".self_attn.linear_out", d, d));
      RVB_TRY(pack_named_linear(e, L.pw1, p + ".conv_module.pointwise_conv1", 2 * d, d, true, true));
      if (e->dtype == DT_BF16) {      // the same weights with rows (c, c + d) next to each other: columns 2c / 2c + 1 of the ACT_GLU GEMM
        const HostTensor *tw, *tb;
        RVB_TRY(need(e, p + ".conv_module.pointwise_conv1.weight", (size_t)2 * d * d, &tw));
        RVB_TRY(need(e, p + ".conv_module.pointwise_conv1.bias", (size_t)2 * d, &tb));
        std::vector<float> wi((size_t)2 * d * d), bi((size_t)2 * d);
        for (int c = 0; c < d; ++c) {
          memcpy(&wi[(size_t)(2 * c) * d], &tw->data[(size_t)c * d], (size_t)d * 4);
          memcpy(&wi[(size_t)(2 * c + 1) * d], &tw->data[(size_t)(d + c) * d], (size_t)d * 4);
          bi[2 * c] = tb->data[c]; bi[2 * c + 1] = tb->data[d + c];
        }
        RVB_TRY(pack_linear(e, L.pw1_glu, wi.data(), bi.data(), 2 * d, d, false));
        RVB_HIP_CHECK(hipStreamSynchronize(e->stream));      // wi / bi go out of scope
      }
      RVB_TRY(pack_named_linear(e, L.pw2, p + ".conv_module.pointwise_conv2", d, d, true, true));
      RVB_TRY(need(e, p + ".self_attn.pos_bias_u", d, &t));
      RVB_TRY(upload_f32(e, L.bias_u, t->data.data(), d));
      RVB_TRY(need(e, p + ".self_attn.pos_bias_v", d, &t));
      RVB_TRY(upload_f32(e, L.bias_v, t->data.data(), d));
      RVB_TRY(need(e, p + ".conv_module.depthwise_conv.weight", (size_t)d * K, &t));
      {   // tap-major [K][d] on the device: the lanes of glu_dw_kernel own adjacent channels, so a tap is one coalesced load
        std::vector<float> wt((size_t)d * K);
        for (int c = 0; c < d; ++c)
          for (int k = 0; k < K; ++k) wt[(size_t)k * d + c] = t->data[(size_t)c * K + k];
        RVB_TRY(upload_f32(e, L.dw_w, wt.data(), (size_t)d * K));
      }
      RVB_TRY(need(e, p + ".conv_module.depthwise_conv.bias", d, &t));
      RVB_TRY(upload_f32(e, L.dw_b, t->data.data(), d));
      RVB_TRY(pack_norm(e, L.n_ffm, p + ".norm_ff_macaron", d, 1e-5f));
      RVB_TRY(pack_norm(e, L.n_mha, p + ".norm_mha", d, 1e-5f));
      RVB_TRY(pack_norm(e, L.n_conv, p + ".norm_conv", d, 1e-5f));
      RVB_TRY(pack_norm(e, L.n_ff, p + ".norm_ff", d, 1e-5f));
      RVB_TRY(pack_norm(e, L.n_final, p + ".norm_final", d, 1e-5f));
      if (c.cnn_norm == 0) {
        RVB_TRY(pack_norm(e, L.n_cnn, p + ".conv_module.norm", d, 1e-5f));
      } else {  // BatchNorm1d (eval) folded to y = x*g' + b'
        const HostTensor *g, *b, *rm, *rv;
        RVB_TRY(need(e, p + ".conv_module.norm.weight", d, &g));
        RVB_TRY(need(e, p + ".conv_module.norm.bias", d, &b));
        RVB_TRY(need(e, p + ".conv_module.norm.running_mean", d, &rm));
        RVB_TRY(need(e, p + ".conv_module.norm.running_var", d, &rv));
        std::vector<float> gg(d), bb(d);
        for (int k = 0; k < d; ++k) {
          gg[k] = g->data[k] / std::sqrt(rv->data[k] + 1e-5f);
          bb[k] = b->data[k] - rm->data[k] * gg[k];
        }
        RVB_TRY(upload_f32(e, L.n_cnn.g, gg.data(), d));
        RVB_TRY(upload_f32(e, L.n_cnn.b, bb.data(), d));
        RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
      }
      // positional keys P = linear_pos(pe[:Tmax]) -- input independent (attention.py:374, embedding.py:145)
      Linear lp;
      RVB_TRY(pack_named_linear(e, lp, p + ".self_attn.linear_pos", d, d, false));
      RVB_TRY(L.pos_keys.ensure((size_t)e->pe_rows * d * dt_size(e->dtype)));
      RVB_TRY(run_gemm(e, pe_T.p, d, lp, L.pos_keys.p, d, e->pe_rows, false));
      if (e->dtype == DT_BF16) {
        // the positional product folded into a per-key constant (attention.hip FOLD): (v - u) . p_j, in the exp2 domain of the kernel
        const int dk_enc = d / c.heads;
        RVB_TRY(L.pos_bias.ensure((size_t)c.heads * e->pe_rows * 4));
        RVB_TRY(attention_pos_bias(e->stream, L.pos_keys.p, e->pe_rows, d, L.bias_u.as<float>(), L.bias_v.as<float>(), c.heads, dk_enc,
                                   1.44269504f / std::sqrt((float)dk_enc), L.pos_bias.as<float>()));
      }
      RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
      lp.w.release(); lp.b.release();
    }
    pe_T.release();
    RVB_TRY(make_fbank_tables(e));
  }
  // (re)fold the language-specific layers with the requested category weights
  for (int i = 0; i < c.num_blocks; ++i) {
    EncLayer& L = e->enc[i];
    if (L.is_lsl) RVB_TRY(pack_lsl(e, L.lsl, "encoder.encoders." + std::to_string(i), d, cat, ncat));
  }
  if (!e->finalized) {
    RVB_TRY(pack_decoder(e, e->dec_l, "decoder.left_decoder", c.dec_blocks, cat, ncat));
    RVB_TRY(pack_decoder(e, e->dec_r, "decoder.right_decoder", c.dec_r_blocks, cat, ncat));
  } else {
    for (int side = 0; side < 2; ++side) {
      Decoder& D = side ? e->dec_r : e->dec_l;
      const std::string p = side ? "decoder.right_decoder" : "decoder.left_decoder";
      if (!D.present) continue;
      for (size_t j = 0; j < D.layers.size(); ++j)
        if (D.layers[j].is_lsl) RVB_TRY(pack_lsl(e, D.layers[j].lsl, p + ".decoders." + std::to_string(j), d, cat, ncat));
    }
  }
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  if (e->fp8) {
    for (auto& L : e->enc)
      if (!L.ffm1.w8.p || !L.ffm2.w8.p || !L.qkv.w8.p || !L.pw1.w8.p || !L.pw2.w8.p || (!L.is_lsl && (!L.ff1.w8.p || !L.ff2.w8.p))) {
        set_error("fp8 mode needs encoder_conf.output_size and linear_units to be multiples of 128 (one fp8 K step)");
        return E_UNSUPPORTED;
      }
  }
  if (!e->finalized) {   // keep only what a later re-finalize needs
    for (auto it = e->host.begin(); it != e->host.end();) {
      if (it->first.find(".language_layers.") == std::string::npos) it = e->host.erase(it); else ++it;
    }
  }
  e->prof.clear();
  e->finalized = true;
  return OK;
}

// Which GEMM groups of which conformer blocks run in fp8 (rvb_set_fp8_policy).  Default: the two feed-forward modules
// (groups 1 | 16) of every block.  Measured on the bench hour against the unmodified reference (profiles/r03_fp8_policy_sweep.txt):
// all five groups 16.2 % greedy / 13.5 % rescored token errors at 123.4 ms; feed-forward only 9.3 % / 10.9 % at 130.8 ms
// (bf16: 4.7 % / 8.9 % at 143.3 ms; the reference's own bf16 autocast: 8.9 % / 9.1 %); qkv + pointwise only 16.4 % / 14.0 %:
// the operands of the softmax (qkv) and of the GLU gate / depthwise path (pointwise 1, 2) are where a 3-bit mantissa hurts.
static int set_fp8_policy_impl(rvb_engine* e, int groups, int first_block, int last_block) {
  const int nb = (int)e->enc.size();
  if (nb == 0) { set_error("rvb_set_fp8_policy before rvb_finalize"); return E_STATE; }
  unsigned mask = groups < 0 ? (lab_env("RVB_FP8_GROUPS") ? (unsigned)atoi(lab_env("RVB_FP8_GROUPS")) : 17u) : (unsigned)groups;
  if (groups < 0) {
    if (lab_env("RVB_FP8_FIRST")) first_block = atoi(lab_env("RVB_FP8_FIRST"));
    if (lab_env("RVB_FP8_LAST")) last_block = atoi(lab_env("RVB_FP8_LAST"));
  }
  if (last_block < 0) last_block = nb - 1;
  e->f8_groups.assign(nb, 0u);
  for (int l = 0; l < nb; ++l)
    if (l >= first_block && l <= last_block) e->f8_groups[l] = mask & 31u;
  e->f8_conv2 = (mask & 32u) != 0;           // not per block: the subsampling's conv2
  return OK;
}

// ------------------------------------------------------------------------------------ encoder
// One conformer block.  On entry e->xn already holds norm_ff_macaron(x) (written by the previous block's fused final
// norm, or by encode_impl for the first block); on exit the block has written `next`(x) to next_out the same way.
// `li` >= 0 selects the streaming form (forward_chunk, encoder.py:231-341): this chunk's keys / values are appended to
// layer li's cache and attention runs over cache + chunk, positional keys taken at the frames' absolute positions.
// fp8 mode (e->f8_state 2, offline only): the LayerNorms write fp8 operands at the calibrated per-tensor scales, the
// feed-forward / qkv / pointwise GEMMs run on the fp8 MFMA path, intermediate h stays fp8; state 1 is the calibration
// pass: the bf16 flow with the running max |.| of every tensor that will be quantised.
static int encoder_layer(rvb_engine* e, EncLayer& L, int lidx, int M, int B, int T, const LNorm& next, void* next_out,
                         float next8 = 0.f, int li = -1) {
  const int d = e->cfg.d_model, ff = e->cfg.ffn_dim, heads = e->cfg.heads, dk = d / heads;
  float* x = e->x.as<float>();
  const bool f8 = e->fp8 && e->f8_state == 2 && li < 0;
  const bool cal = e->fp8 && e->f8_state == 1 && li < 0;
  // which GEMM groups of this block run in fp8 (rvb_engine::f8_groups: bit 0 macaron feed-forward, 1 qkv, 2 pointwise conv 1,
  // 3 pointwise conv 2, 4 feed-forward); the others stay on the bf16 path, LayerNorm output included
  const unsigned grp = f8 ? e->f8_groups[lidx] : 0u;
  const bool f8_ffm = grp & 1u, f8_qkv = grp & 2u, f8_pw1 = grp & 4u, f8_pw2 = grp & 8u, f8_ff = (grp & 16u) && !L.is_lsl;
  const F8Scales sc8 = f8 ? e->f8[lidx] : F8Scales();
  // Folded rel-pos attention with the fold done by the qkv GEMM (round 6): (q+u).k + (q+v).p = (q+u).(k+p) + (v-u).p; the GEMM's
  // epilogue writes K' = k + p (positional key of the frame's place in its chunk, GemmArgs::rowadd) and the attention kernel
  // (attention.hip FOLD 2) multiplies once per key tile and starts from the per-key constants (v-u).p built at load time.
  // bf16 offline batches only (the streaming form caches k itself, and its positional rows move with the stream offset).
  const char* prefold_env = lab_env("RVB_ATTN_PREFOLD");       // read per call (not cached): the A/B test flips it inside one process
  const int prefold_on = prefold_env ? atoi(prefold_env) : 1;
  const bool prefold = prefold_on && e->dtype == DT_BF16 && li < 0 && !f8_qkv && L.pos_bias.p != nullptr && dk > 32 && dk <= 64 &&
                       T <= e->pe_rows && T <= 16384 && (d % 8) == 0;     // 16384: the keys whose constants the kernel holds in LDS
  auto note = [&](int slot, const void* t, size_t n) -> int {
    return cal ? amax_abs(e->stream, e->dtype, t, n, e->d_amax.as<float>() + (size_t)lidx * 8 + slot) : OK;
  };
  // saturation counter of activation slot `slot` of this block (same slot numbering as the scales: in_ffm1, h_ffm, in_qkv,
  // in_pw1, in_pw2, in_ff1, h_ff): the kernels that write an fp8 tensor add the values they had to clip at +-448
  auto satp = [&](int slot) -> unsigned* { return e->d_f8sat.p ? e->d_f8sat.as<unsigned>() + (size_t)lidx * 8 + slot : nullptr; };
  // macaron feed-forward: x += 0.5 * FFN(LN(x))          encoder_layer.py:199-206
  if (f8_ffm) {
    RVB_TRY(run_gemm8(e, e->xn.p, d, L.ffm1, e->h.p, ff, M, sc8.in_ffm1, 2, sc8.h_ffm, 1.f, ACT_SILU, nullptr, 0, satp(1)));
    RVB_TRY(run_gemm8(e, e->h.p, ff, L.ffm2, x, d, M, sc8.h_ffm, 1, 1.f, 0.5f, ACT_NONE, x, d));
  } else {
    RVB_TRY(note(0, e->xn.p, (size_t)M * d));
    RVB_TRY(run_gemm(e, e->xn.p, d, L.ffm1, e->h.p, ff, M, false, 1.f, ACT_SILU));
    RVB_TRY(note(1, e->h.p, (size_t)M * ff));
    RVB_TRY(run_gemm(e, e->h.p, ff, L.ffm2, x, d, M, true, 0.5f, ACT_NONE, x, d));
  }
  // rel-pos self attention: x += MHSA(LN(x))              encoder_layer.py:208-216
  if (f8_qkv) {
    RVB_TRY(run_norm(e, x, L.n_mha, e->xn.p, false, M, d, NORM_LN, 0, nullptr, nullptr, nullptr, sc8.in_qkv, 0.f, false, satp(2)));
    RVB_TRY(run_gemm8(e, e->xn.p, d, L.qkv, e->h.p, 3 * d, M, sc8.in_qkv, 0));
  } else {
    RVB_TRY(run_norm(e, x, L.n_mha, e->xn.p, false, M, d));
    RVB_TRY(note(2, e->xn.p, (size_t)M * d));
    if (prefold) {       // the K third of the output is written as K' = k + p (one rounding), see `prefold` above
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.A = e->xn.p; g.W = L.qkv.w.p; g.bias = L.qkv.b.as<float>(); g.C = e->h.p;
      g.M = M; g.N = 3 * d; g.K = d; g.lda = d; g.ldw = d; g.ldc = 3 * d; g.alpha = 1.f; g.act = ACT_NONE;
      g.rowadd = L.pos_keys.p; g.rowadd_rows = T; g.rowadd_ld = d; g.rowadd_col0 = d; g.rowadd_cols = d;
      Scope sc(e, "gemm", 2.0 * M * (double)g.N * g.K, gemm_alg_bytes(e, g));
      RVB_TRY(gemm(e->stream, e->dtype, g));
    } else {
      RVB_TRY(run_gemm(e, e->xn.p, d, L.qkv, e->h.p, 3 * d, M, false));
    }
  }
  {
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    const size_t es = dt_size(e->dtype);
    a.q = e->h.p; a.k = (const char*)e->h.p + (size_t)d * es; a.v = (const char*)e->h.p + (size_t)2 * d * es;
    a.p = L.pos_keys.p;
    {   // bf16: positional term folded into per-key constants (RVB_ATTN_FOLD=1; default: the two-product form)
      static const int fold = lab_env("RVB_ATTN_FOLD") ? atoi(lab_env("RVB_ATTN_FOLD")) : 0;     // measured slower (10.3 -> 10.8 ms per hour): opt-in
      const int cap = (T + 63) / 64 * 64;             // offline: every chunk's keys are its own T frames
      if (fold && li < 0 && L.pos_bias.p && cap <= 16384) { a.pos_bias = L.pos_bias.as<float>(); a.pos_bias_stride = e->pe_rows; a.fold_kv_cap = cap; }
      if (prefold && cap <= 16384) { a.pos_bias = L.pos_bias.as<float>(); a.pos_bias_stride = e->pe_rows; a.k_prefolded = 1; a.fold_kv_cap = cap; }
    }
    a.q_stride = a.k_stride = a.v_stride = 3 * d; a.p_stride = d; a.o_stride = d;
    a.bias_u = L.bias_u.as<float>(); a.bias_v = L.bias_v.as<float>();
    a.out = e->ao.p;
    a.q_start = e->d_seq_start.as<int>(); a.q_len = e->d_seq_len.as<int>();
    a.kv_start = e->d_seq_start.as<int>(); a.kv_len = e->cur_lens;
    a.nseq = B; a.heads = heads; a.dk = dk; a.max_q = T; a.causal = 0; a.sqrt_dk = std::sqrt((float)dk);
    a.chunk = e->dec_chunk; a.left = e->dec_left;          // add_optional_chunk_mask, encoder.py:140-145
    { static const int qb = lab_env("RVB_ATTN_QBLOCK") ? atoi(lab_env("RVB_ATTN_QBLOCK")) : 0; a.q_block = qb; }   // tuning: 64 / 128 queries per workgroup
    double keys = T;
    if (li >= 0) {
      // attention.py:361-369: k = cat(key_cache, k), v = cat(value_cache, v); pos_emb = position_encoding(offset -
      // cache_t1, cache_t1 + chunk) (encoder.py:305-306), no mask (att_mask is the fake (0,0,0) one)
      auto& st = e->stream_st;
      char* kvb = (char*)st.kv[li].p;
      RVB_HIP_CHECK(hipMemcpy2DAsync(kvb + (size_t)st.cache_len * 2 * d * es, (size_t)2 * d * es, (const char*)e->h.p + (size_t)d * es,
                                     (size_t)3 * d * es, (size_t)2 * d * es, M, hipMemcpyDeviceToDevice, e->stream));
      a.k = kvb; a.v = kvb + (size_t)d * es; a.k_stride = a.v_stride = 2 * d;
      a.p = (const char*)L.pos_keys.p + (size_t)(st.offset - st.cache_len) * d * es;
      if (a.pos_bias) a.pos_bias += (st.offset - st.cache_len);
      a.kv_start = e->d_stream_i32.as<int>(); a.kv_len = e->d_stream_i32.as<int>() + 1;
      a.chunk = 0; a.left = -1;
      keys = st.cache_len + M;
    }
    Scope sc(e, "attention", 6.0 * B * (double)T * keys * d);
    RVB_TRY(attention(e->stream, e->dtype, a));
  }
  RVB_TRY(run_gemm(e, e->ao.p, d, L.att_out, x, d, M, true, 1.f, ACT_NONE, x, d));
  // convolution module: x += Conv(LN(x))                   encoder_layer.py:218-229, convolution.py:89-144
  bool glu_fused = false;
  if (f8_pw1) {
    RVB_TRY(run_norm(e, x, L.n_conv, e->xn.p, false, M, d, NORM_LN, 0, nullptr, nullptr, nullptr, sc8.in_pw1, 0.f, false, satp(3)));
    RVB_TRY(run_gemm8(e, e->xn.p, d, L.pw1, e->h.p, 2 * d, M, sc8.in_pw1, 0));
  } else {
    RVB_TRY(run_norm(e, x, L.n_conv, e->xn.p, false, M, d));
    RVB_TRY(note(3, e->xn.p, (size_t)M * d));
    // pointwise_conv1 + GLU in one kernel (round 6): the GEMM runs on the interleaved copy of the weights and its epilogue stores
    // a * sigmoid(b) -- half the bytes written here and read by the depthwise kernel, the gate computed once per element instead
    // of once per staged element (halo rows twice).  Offline bf16 only (the streaming module caches pointwise OUTPUT rows).
    {
      const char* ge = lab_env("RVB_GLU_FUSE");      // read per call: the A/B test flips it inside one process
      const int glu_on = ge ? atoi(ge) : GLU_FUSE_DEFAULT;
      GemmArgs t;
      memset(&t, 0, sizeof(t));
      t.A = e->xn.p; t.W = L.pw1_glu.w.p; t.bias = L.pw1_glu.b.as<float>(); t.C = e->h.p; t.M = M; t.N = 2 * d; t.K = d; t.lda = d; t.ldw = d;
      t.ldc = d; t.alpha = 1.f; t.act = ACT_GLU;
      glu_fused = glu_on && li < 0 && L.pw1_glu.w.p != nullptr && !cal && gemm_glu_supported(e->dtype, t);
      if (glu_fused) {
        Scope sc(e, "gemm", 2.0 * M * (double)t.N * t.K, gemm_alg_bytes(e, t));
        RVB_TRY(gemm(e->stream, e->dtype, t));
      } else {
        RVB_TRY(run_gemm(e, e->xn.p, d, L.pw1, e->h.p, 2 * d, M, false));
      }
    }
  }
  {
    GluDwArgs g;
    g.gated = glu_fused ? 1 : 0;
    g.G = e->h.p; g.pw1_bias = L.pw1.b.as<float>(); g.dw_w = L.dw_w.as<float>(); g.dw_b = L.dw_b.as<float>();
    g.lens = e->cur_lens; g.out = e->dconv.as<float>(); g.B = B; g.T = T; g.d = d; g.K = e->cfg.cnn_kernel;
    g.causal = e->cfg.cnn_causal ? 1 : 0;
    g.out_bf16 = e->dtype == DT_BF16 ? 1 : 0;     // half the bytes to the norm that reads it next (the reference's bf16 autocast rounds here too)
    const int lorder = g.K - 1;
    const bool cached = li >= 0 && g.causal && lorder > 0;
    if (cached) { g.hist = e->stream_st.cnn[li].p; g.hist_rows = e->stream_st.cnn_rows; }
    {
      Scope sc(e, "glu_dwconv");
      RVB_TRY(glu_dwconv(e->stream, e->dtype, g));
    }
    if (cached) {
      // new_cache = cat(cache, x)[:, :, -lorder:] (convolution.py:116-121), kept as pointwise-conv1 OUTPUT rows: that
      // convolution is per frame, so what the reference recomputes from its cached inputs are these very rows
      auto& st = e->stream_st;
      const size_t es = dt_size(e->dtype), rb = (size_t)2 * d * es;
      if (M >= lorder) {
        RVB_HIP_CHECK(hipMemcpyAsync(st.cnn[li].p, (const char*)e->h.p + (size_t)(M - lorder) * rb, (size_t)lorder * rb,
                                     hipMemcpyDeviceToDevice, e->stream));
      } else {
        RVB_HIP_CHECK(hipMemcpyAsync(st.cnn2[li].p, (const char*)st.cnn[li].p + (size_t)M * rb, (size_t)(lorder - M) * rb,
                                     hipMemcpyDeviceToDevice, e->stream));
        RVB_HIP_CHECK(hipMemcpyAsync((char*)st.cnn2[li].p + (size_t)(lorder - M) * rb, e->h.p, (size_t)M * rb,
                                     hipMemcpyDeviceToDevice, e->stream));
        std::swap(st.cnn[li], st.cnn2[li]);
      }
    }
  }
  const int cmode = e->cfg.cnn_norm == 0 ? NORM_LN : NORM_AFFINE;
  const bool dw16 = e->dtype == DT_BF16;
  if (f8_pw2) {
    RVB_TRY(run_norm(e, e->dconv.as<float>(), L.n_cnn, e->xn.p, false, M, d, cmode, 1, nullptr, nullptr, nullptr, sc8.in_pw2, 0.f, dw16, satp(4)));
    RVB_TRY(run_gemm8(e, e->xn.p, d, L.pw2, x, d, M, sc8.in_pw2, 1, 1.f, 1.f, ACT_NONE, x, d));
  } else {
    RVB_TRY(run_norm(e, e->dconv.as<float>(), L.n_cnn, e->xn.p, false, M, d, cmode, 1, nullptr, nullptr, nullptr, 0.f, 0.f, dw16));
    RVB_TRY(note(4, e->xn.p, (size_t)M * d));
    RVB_TRY(run_gemm(e, e->xn.p, d, L.pw2, x, d, M, true, 1.f, ACT_NONE, x, d));
  }
  // feed-forward (+ language-specific mix), final norm     encoder_layer.py:231-244 / :372-402
  if (f8_ff) {
    RVB_TRY(run_norm(e, x, L.n_ff, e->xn.p, false, M, d, NORM_LN, 0, nullptr, nullptr, nullptr, sc8.in_ff1, 0.f, false, satp(5)));
    RVB_TRY(run_gemm8(e, e->xn.p, d, L.ff1, e->h.p, ff, M, sc8.in_ff1, 2, sc8.h_ff, 1.f, ACT_SILU, nullptr, 0, satp(6)));
    RVB_TRY(run_gemm8(e, e->h.p, ff, L.ff2, x, d, M, sc8.h_ff, 1, 1.f, 0.5f, ACT_NONE, x, d));
  } else {
    RVB_TRY(run_norm(e, x, L.n_ff, e->xn.p, false, M, d));
    const void* ffin = e->xn.p;
    if (L.is_lsl) {
      RVB_TRY(run_gemm(e, e->xn.p, d, L.lsl, e->y.p, d, M, false));
      ffin = e->y.p;
    } else {
      RVB_TRY(note(5, e->xn.p, (size_t)M * d));
    }
    RVB_TRY(run_gemm(e, ffin, d, L.ff1, e->h.p, ff, M, false, 1.f, ACT_SILU));
    if (!L.is_lsl) RVB_TRY(note(6, e->h.p, (size_t)M * ff));
    RVB_TRY(run_gemm(e, e->h.p, ff, L.ff2, x, d, M, true, 0.5f, ACT_NONE, x, d));
  }
  // x = norm_final(x) (+ y for the language-specific block, encoder_layer.py:400), and in the same pass the LayerNorm
  // that always reads it next: the following block's norm_ff_macaron, or the encoder's after_norm (encoder.py:147-148)
  // (the fp8 second output is the NEXT block's in_ffm1: slot 0 of block lidx + 1)
  RVB_TRY(run_norm(e, x, L.n_final, x, true, M, d, NORM_LN, 0, L.is_lsl ? e->y.p : nullptr, &next, next_out, 0.f, f8 ? next8 : 0.f, false,
                   nullptr, (f8 && next8 > 0.f && e->d_f8sat.p) ? e->d_f8sat.as<unsigned>() + (size_t)(lidx + 1) * 8 : nullptr));
  return OK;
}

static const int LOGIT_SLAB = 8192;   // rows of fp32 logits materialised at a time

static int wait_slices(rvb_engine* e, int i);

static int encode_impl(rvb_engine* e, const float* feats, int64_t first_chunk, const int32_t* lens, int B, int T0,
                       int beam, float blank_penalty) {
  const rvb_model_cfg& c = e->cfg;
  if (!e->finalized) { set_error("rvb_encode before rvb_finalize"); return E_STATE; }
  if (B <= 0 || B > c.max_chunks || T0 < 7 || T0 > c.chunk_frames) {
    set_error("rvb_encode: need 1 <= B <= max_chunks and 7 <= T0 <= chunk_frames"); return E_ARG;
  }
  if (beam < 1 || beam > 64 || beam > c.vocab) { set_error("rvb_encode: beam must be in [1,64]"); return E_ARG; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  RVB_TRY(wait_slices(e, -1));      // a previous batch may still be in flight
  e->stream_st.active = false;      // the offline path reuses the stream's output buffer: an open stream ends here
  const int d = c.d_model, F0 = c.input_dim, V = c.vocab;
  const int T1 = (T0 - 3) / 2 + 1, F1 = (F0 - 3) / 2 + 1, T2 = (T1 - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1;
  const int M = B * T2;
  const size_t es = dt_size(e->dtype);
  e->B = B; e->T0 = T0; e->T1 = T1; e->F1 = F1; e->T2 = T2; e->F2 = F2; e->beam = beam;
  e->dec_l.kv_ready = e->dec_r.kv_ready = false;
  e->last_blank_penalty = blank_penalty;
  e->in_lens.assign(lens, lens + B);
  e->enc_lens.resize(B);
  std::vector<int32_t> starts(B), qlens(B, T2);
  for (int b = 0; b < B; ++b) {
    if (lens[b] < 0 || lens[b] > T0) { set_error("rvb_encode: lens out of range"); return E_ARG; }
    // mask[:, :, 2::2][:, :, 2::2] (subsampling.py:226): frames 6+4j < len
    e->enc_lens[b] = lens[b] > 6 ? (lens[b] - 7) / 4 + 1 : 0;
    starts[b] = b * T2;
  }
  e->nbest.clear(); e->trie_l.clear(); e->rescored.clear();
  RVB_TRY(upload_i32(e, e->d_enc_lens, e->enc_lens.data(), B));
  RVB_TRY(upload_i32(e, e->d_seq_start, starts.data(), B));
  RVB_TRY(upload_i32(e, e->d_seq_len, qlens.data(), B));

  const float* d_feats;
  if (feats) {
    RVB_TRY(upload_f32(e, e->d_feats_in, feats, (size_t)B * T0 * F0));
    d_feats = e->d_feats_in.as<float>();
  } else {
    if (!e->feats.p || (first_chunk + B) * (int64_t)T0 > e->feat_rows) {
      set_error("rvb_encode: device features missing or too short (call rvb_fbank first)"); return E_STATE;
    }
    d_feats = e->feats.as<float>() + (size_t)first_chunk * T0 * F0;
  }
  // Sub-batch pipeline: the batch is encoded in up to 2 slices on the engine stream; each slice ends with
  // an async D2H copy of its per-frame top-k into pinned memory and an event.  rvb_encode returns once
  // everything is enqueued; the host CTC search of slice i (rvb_ctc_prefix_beam) then runs while the
  // GPU is still encoding slice i+1.  Workspaces are sized for one slice.
  // two slices: with 256x256 GEMM tiles a finer split leaves the N=1024 GEMMs with <2 waves of tiles per CU
  // (measured: 4 slices 818 TFLOP/s vs 923 un-sliced)
  // uneven split: the host search of the LAST slice is the part nothing overlaps, so that slice is the small one;
  // the first slice's search hides under the GPU time of the second.  For large batches the tail is exactly 32
  // chunks (32 x 512 frames = 64 row tiles = one full wave of 256x256 tiles over the 256 CUs for the N = 1024 GEMMs);
  // measured on the 176-chunk bench batch: first slice 96/112/128/144/160 -> 201.1/199.8/198.6/197.4/198.9 ms
  int SB = B >= 16 ? (B * 7 + 9) / 10 : B;            // chunks in the first (largest) slice
  if (B >= 64) SB = B - 32;
  if (const char* ov = lab_env("RVB_SLICE0")) { const int v = atoi(ov); if (v > 0 && v <= B) SB = v; }   // tuning override
  const int Ms = SB * T2;
  RVB_TRY(e->X1.ensure((size_t)SB * T1 * F1 * d * es));
  RVB_TRY(e->X2.ensure((size_t)SB * T2 * F2 * d * es));
  RVB_TRY(e->x.ensure((size_t)Ms * d * 4));
  RVB_TRY(e->xn.ensure((size_t)Ms * d * es));
  RVB_TRY(e->y.ensure((size_t)Ms * d * es));
  RVB_TRY(e->ao.ensure((size_t)Ms * d * es));
  RVB_TRY(e->dconv.ensure((size_t)Ms * d * 4));
  RVB_TRY(e->enc_out.ensure((size_t)M * d * es));
  RVB_TRY(e->h.ensure((size_t)Ms * std::max(c.ffn_dim, 3 * d) * es));
  const int Vld = (V + 3) & ~3;
  RVB_TRY(e->logits.ensure((size_t)LOGIT_SLAB * Vld * 4));
  RVB_TRY(e->topv.ensure((size_t)M * beam * 4));
  RVB_TRY(e->topi.ensure((size_t)M * beam * 4));
  if (e->h_top_cap < (size_t)M * beam) {
    if (e->h_topv) (void)hipHostFree(e->h_topv);
    if (e->h_topi) (void)hipHostFree(e->h_topi);
    e->h_topv = nullptr; e->h_topi = nullptr; e->h_top_cap = 0;
    RVB_HIP_CHECK(hipHostMalloc((void**)&e->h_topv, (size_t)M * beam * 4, hipHostMallocDefault));
    RVB_HIP_CHECK(hipHostMalloc((void**)&e->h_topi, (size_t)M * beam * 4, hipHostMallocDefault));
    e->h_top_cap = (size_t)M * beam;
  }
  e->slices.clear();
  if (e->fp8 && e->f8_state == 0) {     // first batch of an fp8 engine: bf16 pass that records the activation ranges
    RVB_TRY(e->d_amax.ensure((e->enc.size() + 1) * 8 * 4));
    RVB_HIP_CHECK(hipMemsetAsync(e->d_amax.p, 0, (e->enc.size() + 1) * 8 * 4, e->stream));
    RVB_TRY(reset_f8sat(e));
    e->f8_state = 1;
  }
  for (int c0 = 0; c0 < B; c0 += SB) {
    const int nb = std::min(SB, B - c0);               // first slice SB chunks, second slice the rest (<= SB)
    const int m = nb * T2;
    const int row0 = c0 * T2;
    e->cur_lens = e->d_enc_lens.as<int>() + c0;       // per-chunk arrays of this slice (starts are slice-relative)
    // Conv2dSubsampling4 (subsampling.py:201-226): cmvn+conv1 -> conv2 (implicit GEMM) -> linear * sqrt(d)
    // fp8 mode with conv2 in the policy (bit 5): conv1 writes e4m3 at the calibrated scale and conv2 runs on the fp8 phase loop
    const bool f8c2 = e->fp8 && e->f8_state == 2 && e->f8_conv2 && e->f8_x1 > 0.f && e->conv2.w8.p && d % 128 == 0;
    const bool calx = e->fp8 && e->f8_state == 1;
    {
      Scope sc(e, "subsample");
      RVB_TRY(subsample_conv1(e->stream, e->dtype, d_feats + (size_t)c0 * T0 * F0, e->cmvn_mean.as<float>(),
                              e->cmvn_istd.as<float>(), e->conv1_w.as<float>(), e->conv1_b.as<float>(), e->X1.p, nb, T0, F0, d,
                              f8c2 ? e->f8_x1 : 0.f, calx ? e->d_amax.as<unsigned>() + e->enc.size() * 8 : nullptr,
                              (f8c2 && e->d_f8sat.p) ? e->d_f8sat.as<unsigned>() + e->enc.size() * 8 + 1 : nullptr));
    }
    {
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.A = e->X1.p; g.W = e->conv2.w.p; g.bias = e->conv2.b.as<float>(); g.C = e->X2.p;
      g.M = nb * T2 * F2; g.N = d; g.K = 9 * d; g.lda = d; g.ldw = 9 * d; g.ldc = d;
      g.alpha = 1.f; g.act = ACT_RELU; g.conv = 1; g.cT1 = T1; g.cF1 = F1; g.cT2 = T2; g.cF2 = F2; g.cC = d;
      if (f8c2) { g.W = e->conv2.w8.p; g.in_fp8 = 1; g.a_scale = e->f8_x1; g.w_scale = e->conv2.wscale.as<float>(); }
      Scope sc(e, f8c2 ? "gemm_fp8" : "gemm", 2.0 * g.M * (double)g.N * g.K, gemm_alg_bytes(e, g));
      RVB_TRY(gemm(e->stream, e->dtype, g));
    }
    RVB_TRY(run_gemm(e, e->X2.p, F2 * d, e->embed_out, e->x.p, d, m, true, std::sqrt((float)d)));
    void* eo = (char*)e->enc_out.p + (size_t)row0 * d * es;
    const bool f8 = e->fp8 && e->f8_state == 2;
    RVB_TRY(run_norm(e, e->x.as<float>(), e->enc[0].n_ffm, e->xn.p, false, m, d, NORM_LN, 0, nullptr, nullptr, nullptr,
                     (f8 && (e->f8_groups[0] & 1u)) ? e->f8[0].in_ffm1 : 0.f, 0.f, false, e->d_f8sat.as<unsigned>()));
    for (size_t li = 0; li < e->enc.size(); ++li) {
      const bool last = li + 1 == e->enc.size();
      RVB_TRY(encoder_layer(e, e->enc[li], (int)li, m, nb, T2, last ? e->enc_after : e->enc[li + 1].n_ffm, last ? eo : e->xn.p,
                            (f8 && !last && (e->f8_groups[li + 1] & 1u)) ? e->f8[li + 1].in_ffm1 : 0.f));
    }
    // CTC head + log-softmax + per-frame top-k (ctc.py:106-114, search.py:155)
    for (int r0 = 0; r0 < m; r0 += LOGIT_SLAB) {
      const int rows = std::min(LOGIT_SLAB, m - r0);
      RVB_TRY(run_gemm(e, (const char*)eo + (size_t)r0 * d * es, d, e->ctc, e->logits.p, Vld, rows, true));
      Scope sc(e, "ctc_topk");
      RVB_TRY(logsoftmax_topk(e->stream, e->logits.as<float>(), rows, V, Vld, beam, blank_penalty, c.blank_id,
                              e->topv.as<float>() + (size_t)(row0 + r0) * beam, e->topi.as<int>() + (size_t)(row0 + r0) * beam, nullptr));
    }
    RVB_HIP_CHECK(hipMemcpyAsync(e->h_topv + (size_t)row0 * beam, e->topv.as<float>() + (size_t)row0 * beam, (size_t)m * beam * 4, hipMemcpyDeviceToHost, e->stream));
    RVB_HIP_CHECK(hipMemcpyAsync(e->h_topi + (size_t)row0 * beam, e->topi.as<int>() + (size_t)row0 * beam, (size_t)m * beam * 4, hipMemcpyDeviceToHost, e->stream));
    hipEvent_t ev;
    if (!e->slice_event_pool.empty()) { ev = e->slice_event_pool.back(); e->slice_event_pool.pop_back(); }
    else RVB_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    RVB_HIP_CHECK(hipEventRecord(ev, e->stream));
    e->slices.push_back({c0, nb, ev, false});
  }
  if (e->f8_state == 1) {
    // per-tensor scales: a power of two with headroom (2 * amax maps inside +-448; fp8 is floating point, so headroom
    // costs no relative precision); later batches saturate only beyond twice the calibration batch's maximum
    std::vector<float> am((e->enc.size() + 1) * 8);
    RVB_HIP_CHECK(hipMemcpyAsync(am.data(), e->d_amax.p, am.size() * 4, hipMemcpyDeviceToHost, e->stream));
    RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
    e->f8.resize(e->enc.size());
    auto sc = [](float a) { return a > 0.f ? std::exp2(std::ceil(std::log2(2.f * a / 448.f))) : 1.f; };
    for (size_t l = 0; l < e->enc.size(); ++l) {
      const float* a = am.data() + l * 8;
      e->f8[l] = {sc(a[0]), sc(a[1]), sc(a[2]), sc(a[3]), sc(a[4]), sc(a[5]), sc(a[6])};
    }
    e->f8_x1 = am[e->enc.size() * 8] > 0.f ? sc(am[e->enc.size() * 8]) : 0.f;       // conv1's output (>= 0: the float bits were max'ed as unsigned)
    if (e->f8_groups.size() != e->enc.size()) RVB_TRY(set_fp8_policy_impl(e, -1, 0, -1));     // default policy (or RVB_FP8_*)
    e->f8_state = 2;
  }
  return OK;
}

// wait until slice `i` (or every slice when i < 0) of the last rvb_encode has reached the host
static int wait_slices(rvb_engine* e, int i) {
  for (size_t k = 0; k < e->slices.size(); ++k) {
    if (i >= 0 && (int)k != i) continue;
    auto& sl = e->slices[k];
    if (sl.done) continue;
    RVB_HIP_CHECK(hipEventSynchronize(sl.ev));
    e->slice_event_pool.push_back(sl.ev);
    sl.done = true;
  }
  return OK;
}

// ------------------------------------------------------------------------------------ streaming encoder
// BaseEncoder.forward_chunk / forward_chunk_by_chunk (encoder.py:231-402) for one stream: the attention cache (keys
// and values of the frames already seen, per layer) lives in the engine; the reference hands it back and forth as
// a tensor.  Non-causal convolution modules carry no cnn cache (lorder = 0, convolution.py:118-123): the depthwise
// convolution sees zeros beyond the chunk, exactly as the reference's Conv1d padding does.  Causal ones (cnn_causal)
// keep, per block, the pointwise-conv1 outputs of the last K-1 frames (encoder_layer() below).
static int stream_begin_impl(rvb_engine* e) {
  if (!e->finalized) { set_error("rvb_stream_begin before rvb_finalize"); return E_STATE; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  RVB_TRY(wait_slices(e, -1));
  const int d = e->cfg.d_model;
  const size_t es = dt_size(e->dtype);
  auto& st = e->stream_st;
  st.kv.resize(e->enc.size()); st.kv2.resize(e->enc.size());
  for (size_t l = 0; l < e->enc.size(); ++l) {
    RVB_TRY(st.kv[l].ensure((size_t)e->pe_rows * 2 * d * es));
    RVB_TRY(st.kv2[l].ensure((size_t)e->pe_rows * 2 * d * es));
  }
  RVB_TRY(e->enc_out.ensure((size_t)e->pe_rows * d * es));
  if (e->cfg.cnn_causal && e->cfg.cnn_kernel > 1) {
    st.cnn.resize(e->enc.size()); st.cnn2.resize(e->enc.size());
    for (size_t l = 0; l < e->enc.size(); ++l) {
      RVB_TRY(st.cnn[l].ensure((size_t)(e->cfg.cnn_kernel - 1) * 2 * d * es));
      RVB_TRY(st.cnn2[l].ensure((size_t)(e->cfg.cnn_kernel - 1) * 2 * d * es));
    }
  }
  st.active = true; st.offset = 0; st.cache_len = 0; st.cnn_rows = 0;
  e->B = 0; e->nbest.clear(); e->trie_l.clear(); e->rescored.clear(); e->slices.clear();
  e->dec_l.kv_ready = e->dec_r.kv_ready = false;
  return OK;
}

static int stream_chunk_impl(rvb_engine* e, const float* feats, int T0, int required_cache_size, float* out, int32_t* n_out) {
  const rvb_model_cfg& c = e->cfg;
  auto& st = e->stream_st;
  if (!st.active) { set_error("rvb_stream_chunk before rvb_stream_begin"); return E_STATE; }
  if (T0 < 7) { set_error("rvb_stream_chunk: a chunk needs at least 7 input frames (Conv2dSubsampling4)"); return E_ARG; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  const int d = c.d_model, F0 = c.input_dim;
  const int T1 = (T0 - 3) / 2 + 1, F1 = (F0 - 3) / 2 + 1, T2 = (T1 - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1;
  const int M = T2;
  const size_t es = dt_size(e->dtype);
  if (st.offset + M > e->pe_rows) {
    set_error("rvb_stream_chunk: more than " + std::to_string(e->pe_rows) + " encoder frames in one stream (the reference's positional "
              "table has max_len 5000 rows, embedding.py:33)");
    return E_UNSUPPORTED;
  }
  RVB_TRY(e->X1.ensure((size_t)T1 * F1 * d * es));
  RVB_TRY(e->X2.ensure((size_t)T2 * F2 * d * es));
  RVB_TRY(e->x.ensure((size_t)M * d * 4));
  RVB_TRY(e->xn.ensure((size_t)M * d * es));
  RVB_TRY(e->y.ensure((size_t)M * d * es));
  RVB_TRY(e->ao.ensure((size_t)M * d * es));
  RVB_TRY(e->dconv.ensure((size_t)M * d * 4));
  RVB_TRY(e->h.ensure((size_t)M * std::max(c.ffn_dim, 3 * d) * es));
  RVB_TRY(upload_f32(e, e->d_feats_in, feats, (size_t)T0 * F0));
  const int32_t zero = 0, mm = M, kv[2] = {0, st.cache_len + M};
  RVB_TRY(upload_i32(e, e->d_seq_start, &zero, 1));
  RVB_TRY(upload_i32(e, e->d_seq_len, &mm, 1));
  RVB_TRY(upload_i32(e, e->d_enc_lens, &mm, 1));
  RVB_TRY(upload_i32(e, e->d_stream_i32, kv, 2));
  e->cur_lens = e->d_enc_lens.as<int>();
  {
    Scope sc(e, "subsample");
    RVB_TRY(subsample_conv1(e->stream, e->dtype, e->d_feats_in.as<float>(), e->cmvn_mean.as<float>(), e->cmvn_istd.as<float>(),
                            e->conv1_w.as<float>(), e->conv1_b.as<float>(), e->X1.p, 1, T0, F0, d));
  }
  {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = e->X1.p; g.W = e->conv2.w.p; g.bias = e->conv2.b.as<float>(); g.C = e->X2.p;
    g.M = T2 * F2; g.N = d; g.K = 9 * d; g.lda = d; g.ldw = 9 * d; g.ldc = d;
    g.alpha = 1.f; g.act = ACT_RELU; g.conv = 1; g.cT1 = T1; g.cF1 = F1; g.cT2 = T2; g.cF2 = F2; g.cC = d;
    Scope sc(e, "gemm", 2.0 * g.M * (double)g.N * g.K, gemm_alg_bytes(e, g));
    RVB_TRY(gemm(e->stream, e->dtype, g));
  }
  RVB_TRY(run_gemm(e, e->X2.p, F2 * d, e->embed_out, e->x.p, d, M, true, std::sqrt((float)d)));
  void* eo = (char*)e->enc_out.p + (size_t)st.offset * d * es;
  RVB_TRY(run_norm(e, e->x.as<float>(), e->enc[0].n_ffm, e->xn.p, false, M, d));
  for (size_t li = 0; li < e->enc.size(); ++li) {
    const bool last = li + 1 == e->enc.size();
    RVB_TRY(encoder_layer(e, e->enc[li], (int)li, M, 1, M, last ? e->enc_after : e->enc[li + 1].n_ffm, last ? eo : e->xn.p, 0.f, (int)li));
  }
  if (out) {
    if (e->dtype == DT_F32) {
      RVB_HIP_CHECK(hipMemcpyAsync(out, eo, (size_t)M * d * 4, hipMemcpyDeviceToHost, e->stream));
      RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
    } else {
      std::vector<bf16_t> tmp((size_t)M * d);
      RVB_HIP_CHECK(hipMemcpyAsync(tmp.data(), eo, tmp.size() * 2, hipMemcpyDeviceToHost, e->stream));
      RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
      for (size_t i = 0; i < tmp.size(); ++i) out[i] = bf16_to_f32(tmp[i]);
    }
  }
  // r_att_cache = new_att_cache[:, :, next_cache_start:, :] (encoder.py:307-312,331)
  const int key_size = st.cache_len + M;
  int start = 0;
  if (required_cache_size == 0) start = key_size;
  else if (required_cache_size > 0) start = std::max(key_size - required_cache_size, 0);
  const int keep = key_size - start;
  if (start > 0 && keep > 0) {
    for (size_t l = 0; l < e->enc.size(); ++l) {
      RVB_HIP_CHECK(hipMemcpyAsync(st.kv2[l].p, (const char*)st.kv[l].p + (size_t)start * 2 * d * es, (size_t)keep * 2 * d * es,
                                   hipMemcpyDeviceToDevice, e->stream));
      std::swap(st.kv[l], st.kv2[l]);
    }
  }
  st.cache_len = keep;
  st.cnn_rows = std::min(st.cnn_rows + M, std::max(c.cnn_kernel - 1, 0));
  st.offset += M;
  if (n_out) *n_out = M;
  return OK;
}

// CTC head + top-k over everything the stream produced: from here on the stream is one encoded "chunk" of st.offset
// frames and the search entry points work on it (ASRModel._forward_encoder with simulate_streaming, asr_model.py:301-306)
static int stream_finish_impl(rvb_engine* e, int beam, float blank_penalty) {
  const rvb_model_cfg& c = e->cfg;
  auto& st = e->stream_st;
  if (!st.active) { set_error("rvb_stream_finish before rvb_stream_begin"); return E_STATE; }
  if (beam < 1 || beam > 64 || beam > c.vocab) { set_error("rvb_stream_finish: beam must be in [1,64]"); return E_ARG; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  const int d = c.d_model, V = c.vocab, M = st.offset;
  const size_t es = dt_size(e->dtype);
  if (M <= 0) { set_error("rvb_stream_finish: the stream produced no encoder frame"); return E_STATE; }
  e->B = 1; e->T2 = M; e->beam = beam; e->T0 = 0;
  e->dec_l.kv_ready = e->dec_r.kv_ready = false;
  e->last_blank_penalty = blank_penalty;
  e->in_lens.assign(1, 0); e->enc_lens.assign(1, M);
  e->nbest.clear(); e->trie_l.clear(); e->rescored.clear();
  const int Vld = (V + 3) & ~3;
  RVB_TRY(e->logits.ensure((size_t)LOGIT_SLAB * Vld * 4));
  RVB_TRY(e->topv.ensure((size_t)M * beam * 4));
  RVB_TRY(e->topi.ensure((size_t)M * beam * 4));
  if (e->h_top_cap < (size_t)M * beam) {
    if (e->h_topv) (void)hipHostFree(e->h_topv);
    if (e->h_topi) (void)hipHostFree(e->h_topi);
    e->h_topv = nullptr; e->h_topi = nullptr; e->h_top_cap = 0;
    RVB_HIP_CHECK(hipHostMalloc((void**)&e->h_topv, (size_t)M * beam * 4, hipHostMallocDefault));
    RVB_HIP_CHECK(hipHostMalloc((void**)&e->h_topi, (size_t)M * beam * 4, hipHostMallocDefault));
    e->h_top_cap = (size_t)M * beam;
  }
  for (int r0 = 0; r0 < M; r0 += LOGIT_SLAB) {
    const int rows = std::min(LOGIT_SLAB, M - r0);
    RVB_TRY(run_gemm(e, (const char*)e->enc_out.p + (size_t)r0 * d * es, d, e->ctc, e->logits.p, Vld, rows, true));
    Scope sc(e, "ctc_topk");
    RVB_TRY(logsoftmax_topk(e->stream, e->logits.as<float>(), rows, V, Vld, beam, blank_penalty, c.blank_id,
                            e->topv.as<float>() + (size_t)r0 * beam, e->topi.as<int>() + (size_t)r0 * beam, nullptr));
  }
  RVB_HIP_CHECK(hipMemcpyAsync(e->h_topv, e->topv.p, (size_t)M * beam * 4, hipMemcpyDeviceToHost, e->stream));
  RVB_HIP_CHECK(hipMemcpyAsync(e->h_topi, e->topi.p, (size_t)M * beam * 4, hipMemcpyDeviceToHost, e->stream));
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  e->slices.clear();
  e->slices.push_back({0, 1, nullptr, true});
  st.active = false;
  return OK;
}

// ------------------------------------------------------------------------------------ search
// Host threads one engine may use for the search: RVB_SEARCH_THREADS, else this process's share of the cores when
// several ranks run on the node (LOCAL_WORLD_SIZE is set by torchrun), at most 32.
static unsigned search_threads() {
  if (const char* s = getenv("RVB_SEARCH_THREADS")) {
    const int v = atoi(s);
    if (v > 0) return (unsigned)v;
  }
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 4;
  if (const char* s = getenv("LOCAL_WORLD_SIZE")) {
    const int v = atoi(s);
    if (v > 1) hw = std::max(1u, hw / (unsigned)v);
  }
  return std::min(hw, 32u);
}

static void build_chunk_trie(rvb_engine* e, int b, bool reversed, int sos, int eos, TrieBatch* t);   // with the rescoring, below

static int prefix_beam_impl(rvb_engine* e, int beam) {
  if (e->B <= 0) { set_error("rvb_ctc_prefix_beam before rvb_encode"); return E_STATE; }
  // the search beam may be narrower than the top-k rvb_encode kept per frame (joint_decoding's pre-beam needs more): the first
  // `beam` entries of a frame's descending top-k ARE its top-`beam` (search.py:155 `logp.topk(beam_size)`)
  if (beam < 1 || beam > e->beam) { set_error("rvb_ctc_prefix_beam: beam exceeds the top-k kept by rvb_encode"); return E_ARG; }
  const int B = e->B, T = e->T2, K = e->beam;
  e->nbest.assign(B, PrefixResult());
  const bool prebuild = e->dec_l.present;
  e->trie_l.assign(prebuild ? B : 0, TrieBatch());
  const unsigned hw = search_threads();
  double busy_ms = 0.0;
  for (size_t si = 0; si < e->slices.size(); ++si) {
    RVB_TRY(wait_slices(e, (int)si));            // GPU keeps encoding the later slices meanwhile
    const auto t0 = std::chrono::steady_clock::now();
    const int c0 = e->slices[si].c0, nb = e->slices[si].nb;
    // ~0.5 ms of work per full chunk: two or more chunks per thread amortise the thread start; chunks are handed
    // out one at a time because their lengths (and so their cost) differ
    const unsigned nthr = std::max(1u, std::min<unsigned>(hw, (unsigned)(nb + 1) / 2));
    std::atomic<int> next_chunk(c0);
    const int sos = e->cfg.sos_id, eos = e->cfg.eos_id;
    auto work = [&, c0, nb]() {
      for (int b = next_chunk.fetch_add(1); b < c0 + nb; b = next_chunk.fetch_add(1)) {
        prefix_beam_search(e->h_topv + (size_t)b * T * K, e->h_topi + (size_t)b * T * K, e->enc_lens[b], K,
                           beam, e->cfg.blank_id, &e->nbest[b]);
        // the chunk's prefix trie for the left-to-right rescoring decoder, while the worker has the n-best list hot: for
        // every slice but the last this happens underneath the encoder of the next slice (rescore_impl only stitches)
        if (prebuild) build_chunk_trie(e, b, false, sos, eos, &e->trie_l[b]);
      }
    };
    e->pool.run(nthr, work);                     // the calling thread is one of the nthr
    busy_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  auto& pe = e->prof["search_host"];
  pe.ms += busy_ms;
  pe.launches += 1;
  return OK;
}

// ------------------------------------------------------------------------------------ rescoring
// The n-best hypotheses of a chunk share long prefixes (they come out of one prefix beam), and the decoder is causal:
// decoder row j of a hypothesis depends only on its tokens 0..j-1 and on the chunk's memory.  The reference runs the
// decoder on the padded [N, L] batch (search.py:391-412, asr_model.py:868-978), i.e. it recomputes a shared prefix N
// times; here every DISTINCT prefix of a chunk is one decoder row (a trie, built per decoder direction: the
// right-to-left decoder sees the reversed hypotheses, search.py:427-433).  Row results do not depend on the batch they
// are computed in (GEMM rows are independent, an attention row walks its own keys in order), so every hypothesis
// reads exactly the log-probs the padded batch would give it.  On the bench workload 13-35 % of the rows remain.
// seq(h, j) = j-th decoder input token AFTER <sos> of hypothesis h; target of pair (h, j) = seq(h, j) for j < len, else eos
template <typename SeqFn>
static void build_trie_range(const HypRef* hb, const HypRef* he, int chunk0, int B, int sos, int eos, SeqFn seq, TrieBatch* t) {
  *t = TrieBatch();
  std::vector<std::pair<int32_t, int32_t>> asks;   // (row, target) in pair order
  std::vector<std::vector<std::pair<int32_t, int32_t>>> kids;   // per row: (token, child row) -- fan-out is tiny
  t->crow_start.assign(B, 0); t->crow_len.assign(B, 0);
  int cur_chunk = -1, root = -1;
  for (const HypRef* hp = hb; hp != he; ++hp) {
    const HypRef& h = *hp;
    if (h.chunk - chunk0 != cur_chunk) {
      if (cur_chunk >= 0) t->crow_len[cur_chunk] = t->R - t->crow_start[cur_chunk];
      cur_chunk = h.chunk - chunk0;
      t->crow_start[cur_chunk] = t->R;
      root = -1;
    }
    const int first_new = t->R;
    int own_pos0 = -1;
    t->hkv_start.push_back((int32_t)t->path.size());
    int node = root;
    for (int j = 0; j <= h.len; ++j) {
      int next = -1;
      if (j == 0) {
        next = root;
      } else {
        const int tk = seq(h, j - 1);
        for (auto& kv : kids[node]) if (kv.first == tk) { next = kv.second; break; }
      }
      if (next < 0) {
        next = t->R++;
        t->tok.push_back(j == 0 ? sos : seq(h, j - 1));
        t->pos.push_back(j);
        kids.emplace_back();
        if (j == 0) root = next; else kids[node].push_back({seq(h, j - 1), next});
        if (own_pos0 < 0) own_pos0 = j;
      }
      node = next;
      t->path.push_back(node);
      asks.push_back({node, j < h.len ? seq(h, j) : eos});
    }
    const int n_own = t->R - first_new;             // new rows are a suffix of the path and contiguous
    t->hq_start.push_back(first_new); t->hq_len.push_back(n_own); t->hq_pos0.push_back(n_own ? own_pos0 : 0);
    t->hkv_len.push_back(h.len + 1);
    for (int q0 = 0; q0 < n_own; q0 += 16) { t->work.push_back((int32_t)t->hq_start.size() - 1); t->work.push_back(q0); }
  }
  if (cur_chunk >= 0) t->crow_len[cur_chunk] = t->R - t->crow_start[cur_chunk];
  for (int b = 0; b < B; ++b) t->max_chunk_rows = std::max(t->max_chunk_rows, t->crow_len[b]);
  // CSR of the asks by row (counting sort keeps pair order inside a row)
  t->P = (int)asks.size();
  t->tgt_ptr.assign(t->R + 1, 0);
  for (auto& a : asks) t->tgt_ptr[a.first + 1]++;
  for (int r = 0; r < t->R; ++r) t->tgt_ptr[r + 1] += t->tgt_ptr[r];
  std::vector<int32_t> fill(t->tgt_ptr.begin(), t->tgt_ptr.end() - 1);
  t->tgt.assign(t->P, 0); t->pair_slot.assign(t->P, 0);
  for (int p = 0; p < t->P; ++p) { const int slot = fill[asks[p].first]++; t->tgt[slot] = asks[p].second; t->pair_slot[p] = slot; }
}

// one decoder over the trie rows; logp[slot] = log p(target | prefix) for every ask (TrieBatch::pair_slot maps pairs)
// One chunk's trie in local numbering (rows, hypotheses, path entries and pairs counted from 0).
static void build_chunk_trie(rvb_engine* e, int b, bool reversed, int sos, int eos, TrieBatch* t) {
  const PrefixResult& pr = e->nbest[b];
  std::vector<HypRef> hyps(pr.nbest.size());
  for (size_t i = 0; i < hyps.size(); ++i) hyps[i] = {b, (int)i, (int)pr.nbest[i].size(), 0};
  if (reversed)
    build_trie_range(hyps.data(), hyps.data() + hyps.size(), b, 1, sos, eos,
                     [&](const HypRef& h, int j) { return pr.nbest[h.idx][h.len - 1 - j]; }, t);
  else
    build_trie_range(hyps.data(), hyps.data() + hyps.size(), b, 1, sos, eos, [&](const HypRef& h, int j) { return pr.nbest[h.idx][j]; }, t);
}

// The batch trie from the chunks' tries: the rows of a chunk are contiguous and only that chunk's hypotheses refer to them, so
// a chunk's local numbering differs from the global one by the running totals of the chunks before it -- rows, hypotheses,
// path entries, (hypothesis, position) pairs.  Bit-identical to build_trie_range over all hypotheses at once (which took
// 1.6 ms on one thread for the 176-chunk bench batch, with the device idle).
static void merge_tries(const std::vector<TrieBatch>& part, TrieBatch* t) {
  const int B = (int)part.size();
  *t = TrieBatch();
  size_t nR = 0, nH = 0, nPath = 0, nP = 0, nW = 0;
  for (const TrieBatch& c : part) { nR += c.R; nH += c.hq_start.size(); nPath += c.path.size(); nP += c.P; nW += c.work.size(); }
  t->tok.reserve(nR); t->pos.reserve(nR); t->path.reserve(nPath); t->work.reserve(nW);
  t->hq_start.reserve(nH); t->hq_len.reserve(nH); t->hq_pos0.reserve(nH); t->hkv_start.reserve(nH); t->hkv_len.reserve(nH);
  t->tgt.reserve(nP); t->pair_slot.reserve(nP); t->tgt_ptr.reserve(nR + 1);
  t->crow_start.assign(B, 0); t->crow_len.assign(B, 0);
  for (int b = 0; b < B; ++b) {
    const TrieBatch& c = part[b];
    const int32_t R0 = t->R, H0 = (int32_t)t->hq_start.size(), PATH0 = (int32_t)t->path.size(), P0 = t->P;
    t->crow_start[b] = c.R ? R0 : 0; t->crow_len[b] = c.R;      // a chunk without hypotheses keeps the zeros of the batch form
    t->max_chunk_rows = std::max(t->max_chunk_rows, c.R);
    t->tok.insert(t->tok.end(), c.tok.begin(), c.tok.end());
    t->pos.insert(t->pos.end(), c.pos.begin(), c.pos.end());
    for (int32_t v : c.path) t->path.push_back(v + R0);
    for (int32_t v : c.hq_start) t->hq_start.push_back(v + R0);
    t->hq_len.insert(t->hq_len.end(), c.hq_len.begin(), c.hq_len.end());
    t->hq_pos0.insert(t->hq_pos0.end(), c.hq_pos0.begin(), c.hq_pos0.end());
    for (int32_t v : c.hkv_start) t->hkv_start.push_back(v + PATH0);
    t->hkv_len.insert(t->hkv_len.end(), c.hkv_len.begin(), c.hkv_len.end());
    for (size_t k = 0; k + 1 < c.work.size(); k += 2) { t->work.push_back(c.work[k] + H0); t->work.push_back(c.work[k + 1]); }
    for (int r = 0; r < c.R; ++r) t->tgt_ptr.push_back(c.tgt_ptr[r] + P0);
    t->tgt.insert(t->tgt.end(), c.tgt.begin(), c.tgt.end());
    for (int32_t v : c.pair_slot) t->pair_slot.push_back(v + P0);
    t->R += c.R; t->P += c.P;
  }
  t->tgt_ptr.push_back(t->P);
}

// every chunk's trie on the host pool, then stitched
static void build_trie_parallel(rvb_engine* e, bool reversed, TrieBatch* t) {
  const int B = e->B, sos = e->cfg.sos_id, eos = e->cfg.eos_id;
  std::vector<TrieBatch> part(B);
  std::atomic<int> next(0);
  auto work = [&]() {
    for (int b = next.fetch_add(1); b < B; b = next.fetch_add(1)) build_chunk_trie(e, b, reversed, sos, eos, &part[b]);
  };
  e->pool.run(std::max(1u, std::min<unsigned>(search_threads(), (unsigned)B / 4)), work);
  merge_tries(part, t);
}

// Keys / values of the encoder output for every decoder layer (decoder_layer.py:112-119: `src_attn(x, memory, memory)`; the
// reference projects the memory once per hypothesis, asr_model.py:895): they depend on the encoder output alone, so they can
// be enqueued before the CTC search of the last slice has produced a single hypothesis (rvb_prepare_rescoring) -- the device
// computes them while the host searches.
static int decoder_memory_kv(rvb_engine* e, Decoder& D, int M) {
  const int d = e->cfg.d_model;
  const size_t es = dt_size(e->dtype);
  for (auto& L : D.layers) {
    RVB_TRY(L.kvmem.ensure((size_t)M * 2 * d * es));
    RVB_TRY(run_gemm(e, e->enc_out.p, d, L.src_kv, L.kvmem.p, 2 * d, M, false));
  }
  D.kv_ready = true;
  return OK;
}

static int decoder_forward(rvb_engine* e, Decoder& D, const TrieBatch& t, std::vector<float>* logp) {
  const rvb_model_cfg& c = e->cfg;
  const int d = c.d_model, heads = c.dec_heads, dk = d / heads, ff = c.dec_ffn_dim, V = c.vocab;
  const int M = e->B * e->T2, R = t.R, nhyp = (int)t.hq_start.size();
  const size_t es = dt_size(e->dtype);
  RVB_TRY(upload_i32(e, e->d_tok, t.tok.data(), R));
  RVB_TRY(upload_i32(e, e->d_pos, t.pos.data(), R));
  RVB_TRY(upload_i32(e, e->d_tgt, t.tgt.data(), t.P));
  RVB_TRY(upload_i32(e, e->d_tgt_ptr, t.tgt_ptr.data(), R + 1));
  RVB_TRY(upload_i32(e, e->d_path, t.path.data(), t.path.size()));
  RVB_TRY(upload_i32(e, e->d_work, t.work.data(), t.work.size()));
  RVB_TRY(upload_i32(e, e->d_hq_start, t.hq_start.data(), nhyp));
  RVB_TRY(upload_i32(e, e->d_hq_len, t.hq_len.data(), nhyp));
  RVB_TRY(upload_i32(e, e->d_hq_pos0, t.hq_pos0.data(), nhyp));
  RVB_TRY(upload_i32(e, e->d_hpath_start, t.hkv_start.data(), nhyp));
  RVB_TRY(upload_i32(e, e->d_hpath_len, t.hkv_len.data(), nhyp));
  RVB_TRY(upload_i32(e, e->d_hkv_start, t.crow_start.data(), e->B));
  RVB_TRY(upload_i32(e, e->d_hkv_len, t.crow_len.data(), e->B));
  RVB_TRY(e->dx.ensure((size_t)R * d * 4));
  RVB_TRY(e->dxn.ensure((size_t)R * d * es));
  RVB_TRY(e->dy.ensure((size_t)R * d * es));
  RVB_TRY(e->dao.ensure((size_t)R * d * es));
  RVB_TRY(e->dq.ensure((size_t)R * d * es));
  RVB_TRY(e->dqkv.ensure((size_t)R * 3 * d * es));
  RVB_TRY(e->dh.ensure((size_t)R * ff * es));
  if (!D.kv_ready) RVB_TRY(decoder_memory_kv(e, D, M));
  RVB_TRY(e->d_logp.ensure((size_t)t.P * 4));
  float* x = e->dx.as<float>();
  {
    Scope sc(e, "embed");
    RVB_TRY(embed_tokens(e->stream, D.embed.as<float>(), e->pe_f32.as<float>(), e->d_tok.as<int>(), e->d_pos.as<int>(),
                         x, R, d, std::sqrt((float)d)));
  }
  for (auto& L : D.layers) {
    // self attention (causal): a hypothesis' owned rows are the queries, the rows of its whole prefix path the keys
    // decoder_layer.py:91-110, decoder.py:150-156
    RVB_TRY(run_norm(e, x, L.n1, e->dxn.p, false, R, d));
    RVB_TRY(run_gemm(e, e->dxn.p, d, L.self_qkv, e->dqkv.p, 3 * d, R, false));
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = e->dqkv.p; a.k = (const char*)e->dqkv.p + (size_t)d * es; a.v = (const char*)e->dqkv.p + (size_t)2 * d * es;
    a.q_stride = a.k_stride = a.v_stride = 3 * d; a.o_stride = d; a.out = e->dao.p;
    a.q_start = e->d_hq_start.as<int>(); a.q_len = e->d_hq_len.as<int>(); a.q_pos0 = e->d_hq_pos0.as<int>();
    a.kv_start = e->d_hpath_start.as<int>(); a.kv_len = e->d_hpath_len.as<int>(); a.kv_index = e->d_path.as<int>();
    a.work = e->d_work.as<int>(); a.n_work = (int)t.work.size() / 2; a.q_block = 16;
    a.nseq = nhyp; a.heads = heads; a.dk = dk; a.max_q = 16; a.causal = 1; a.sqrt_dk = std::sqrt((float)dk);
    {
      Scope sc(e, "attention");
      RVB_TRY(attention(e->stream, e->dtype, a));
    }
    RVB_TRY(run_gemm(e, e->dao.p, d, L.self_out, x, d, R, true, 1.f, ACT_NONE, x, d));
    // cross attention over the chunk's encoder frames (memory K/V computed once, not per hypothesis:
    // the reference repeats the memory N times, asr_model.py:895)       decoder_layer.py:112-119
    RVB_TRY(run_norm(e, x, L.n2, e->dxn.p, false, R, d));
    RVB_TRY(run_gemm(e, e->dxn.p, d, L.src_q, e->dq.p, d, R, false));
    memset(&a, 0, sizeof(a));
    a.q = e->dq.p; a.k = L.kvmem.p; a.v = (const char*)L.kvmem.p + (size_t)d * es;
    a.q_stride = d; a.k_stride = a.v_stride = 2 * d; a.o_stride = d; a.out = e->dao.p;
    // all rows of a chunk attend to the same memory and there is no causal mask: they form ONE query sequence per
    // chunk, so the chunk's K/V tiles are staged once per 128 rows
    a.q_start = e->d_hkv_start.as<int>(); a.q_len = e->d_hkv_len.as<int>();
    a.kv_start = e->d_aux_i32.as<int>(); a.kv_len = e->d_aux_i32.as<int>() + e->B;
    a.nseq = e->B; a.heads = heads; a.dk = dk; a.max_q = t.max_chunk_rows; a.causal = 0; a.sqrt_dk = std::sqrt((float)dk);
    {
      Scope sc(e, "attention");
      RVB_TRY(attention(e->stream, e->dtype, a));
    }
    RVB_TRY(run_gemm(e, e->dao.p, d, L.src_out, x, d, R, true, 1.f, ACT_NONE, x, d));
    // feed forward (ReLU) with the language-specific mix     decoder_layer.py:121-127 / :313-333
    RVB_TRY(run_norm(e, x, L.n3, e->dxn.p, false, R, d));
    const void* ffin = e->dxn.p;
    if (L.is_lsl) {
      RVB_TRY(run_gemm(e, e->dxn.p, d, L.lsl, e->dy.p, d, R, false));
      ffin = e->dy.p;
    }
    RVB_TRY(run_gemm(e, ffin, d, L.ff1, e->dh.p, ff, R, false, 1.f, ACT_RELU));
    RVB_TRY(run_gemm(e, e->dh.p, ff, L.ff2, x, d, R, true, 1.f, ACT_NONE, x, d));
  }
  RVB_TRY(run_norm(e, x, D.after, e->dxn.p, false, R, d));
  const int Vld = (V + 3) & ~3;
  for (int r0 = 0; r0 < R; r0 += LOGIT_SLAB) {
    const int rows = std::min(LOGIT_SLAB, R - r0);
    RVB_TRY(run_gemm(e, (const char*)e->dxn.p + (size_t)r0 * d * es, d, D.out, e->logits.p, Vld, rows, true));
    Scope sc(e, "lse_gather");
    RVB_TRY(lse_gather_multi(e->stream, e->logits.as<float>(), rows, V, Vld, e->d_tgt_ptr.as<int>() + r0, e->d_tgt.as<int>(),
                             e->d_logp.as<float>()));
  }
  logp->resize(t.P);
  RVB_HIP_CHECK(hipMemcpyAsync(logp->data(), e->d_logp.p, (size_t)t.P * 4, hipMemcpyDeviceToHost, e->stream));
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  return OK;
}

static int rescore_impl(rvb_engine* e, double ctc_weight, double reverse_weight) {
  if ((int)e->nbest.size() != e->B || e->B <= 0) { set_error("rvb_attention_rescore before rvb_ctc_prefix_beam"); return E_STATE; }
  if (!e->dec_l.present) { set_error("model has no attention decoder"); return E_STATE; }
  RVB_TRY(wait_slices(e, -1));
  const bool use_r = reverse_weight > 0.0;
  if (use_r && !e->dec_r.present) { set_error("reverse_weight > 0 but model has no right-to-left decoder"); return E_STATE; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  const int B = e->B, T2 = e->T2, eos = e->cfg.eos_id, sos = e->cfg.sos_id;
  // every hypothesis of every chunk asks for len+1 log-probs ([sos] + tokens -> tokens + [eos]; add_sos_eos,
  // common.py:112-155, search.py:417-425)
  std::vector<HypRef> hyps;
  int P = 0;
  std::vector<int32_t> ckv(2 * (size_t)B);
  for (int b = 0; b < B; ++b) {
    const PrefixResult& pr = e->nbest[b];
    ckv[b] = b * T2; ckv[B + b] = e->enc_lens[b];
    for (size_t i = 0; i < pr.nbest.size(); ++i) {
      const int len = (int)pr.nbest[i].size();
      if (len + 1 > e->pe_rows) { set_error("hypothesis longer than the positional table"); return E_UNSUPPORTED; }
      hyps.push_back({b, (int)i, len, P});
      P += len + 1;
    }
  }
  RVB_TRY(upload_i32(e, e->d_aux_i32, ckv.data(), ckv.size()));
  TrieBatch tl, tr;
  const auto th0 = std::chrono::steady_clock::now();
  if ((int)e->trie_l.size() == B) merge_tries(e->trie_l, &tl);      // built by the prefix-beam workers, chunk by chunk
  else build_trie_parallel(e, false, &tl);
  {
    auto& pe = e->prof["rescore_trie_host"];
    pe.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th0).count();
    pe.launches += 1;
  }
  std::vector<float> lslot, rslot;
  const auto td0 = std::chrono::steady_clock::now();
  RVB_TRY(decoder_forward(e, e->dec_l, tl, &lslot));
  {
    auto& pe = e->prof["rescore_decoder_wall"];
    pe.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td0).count();
    pe.launches += 1;
  }
  e->xattn_max_rows = tl.max_chunk_rows;
  e->rescore_rows = tl.R; e->rescore_pairs = tl.P;
  if (use_r) {   // reversed input and targets, asr_model.py:896-953, search.py:427-433
    build_trie_parallel(e, true, &tr);
    RVB_TRY(decoder_forward(e, e->dec_r, tr, &rslot));
    e->rescore_rows += tr.R; e->rescore_pairs += tr.P;
  }
  const auto ta0 = std::chrono::steady_clock::now();
  std::vector<float> logp(P), rlogp(use_r ? P : 0);
  for (int p = 0; p < P; ++p) { logp[p] = lslot[tl.pair_slot[p]]; if (use_r) rlogp[p] = rslot[tr.pair_slot[p]]; }

  // score accumulation exactly as search.py:413-441: fp32 running sums (0-dim float32 tensors), strict '>' so the first
  // maximum wins; the python-float exp() of the confidences is evaluated for the winning hypothesis only (the reference
  // computes them for every hypothesis and keeps the winner's)
  e->rescored.assign(B, RescoreResult());
  std::vector<float> att_score(hyps.size());        // decoder score before the CTC term (the confidence is derived from it)
  std::vector<int> best_hyp(B, -1);
  for (size_t hi = 0; hi < hyps.size(); ++hi) {
    const HypRef& hr = hyps[hi];
    const PrefixResult& pr = e->nbest[hr.chunk];
    RescoreResult& rr = e->rescored[hr.chunk];
    if (rr.logp.empty()) { rr.logp.resize(pr.nbest.size()); rr.rlogp.resize(pr.nbest.size()); rr.score = -INFINITY; }
    const float* lp = logp.data() + hr.row0;
    rr.logp[hr.idx].assign(lp, lp + hr.len + 1);
    float score = 0.f;
    for (int j = 0; j < hr.len; ++j) score += lp[j];
    score += lp[hr.len];
    if (use_r) {
      const float* rp = rlogp.data() + hr.row0;
      rr.rlogp[hr.idx].assign(rp, rp + hr.len + 1);
      float r_score = 0.f;
      for (int j = 0; j < hr.len; ++j) r_score += rp[hr.len - j - 1];
      r_score += rp[hr.len];
      // python: tensor(fp32) * float(1 - rw) + tensor(fp32) * float(rw)
      score = score * (float)(1.0 - reverse_weight) + r_score * (float)reverse_weight;
    }
    att_score[hi] = score;
    score += (float)(pr.scores[hr.idx] * ctc_weight);
    if (best_hyp[hr.chunk] < 0 || score > rr.score) { rr.score = score; rr.best = hr.idx; best_hyp[hr.chunk] = (int)hi; }
  }
  for (int b = 0; b < B; ++b) {
    RescoreResult& rr = e->rescored[b];
    if (best_hyp[b] < 0) { rr.best = 0; rr.score = -INFINITY; continue; }
    const HypRef& hr = hyps[best_hyp[b]];
    const float* lp = logp.data() + hr.row0;
    rr.confidence = std::exp((double)(att_score[best_hyp[b]] / (float)(hr.len + 1)));
    rr.tok_conf.resize(hr.len);
    for (int j = 0; j < hr.len; ++j) rr.tok_conf[j] = std::exp((double)lp[j]);
    if (use_r) {
      const float* rp = rlogp.data() + hr.row0;
      for (int j = 0; j < hr.len; ++j) rr.tok_conf[j] = (rr.tok_conf[j] + std::exp((double)rp[hr.len - j - 1])) / 2.0;
    }
  }
  {
    auto& pe = e->prof["rescore_scores_host"];
    pe.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta0).count();
    pe.launches += 1;
  }
  return OK;
}


// ------------------------------------------------------------------------------------ attention beam search
// `attention` mode (search.py:251-360): autoregressive beam search with the left decoder.  The reference's
// forward_one_step (decoder.py:191-234) recomputes the keys/values of the whole prefix at every step; here every
// decoder layer keeps a K/V cache per hypothesis ([R][L][d], reordered by the beam's parent index after each step),
// the memory K/V of each chunk is projected once, and one step is one decoder row per hypothesis.  The beam
// bookkeeping follows the reference line by line in float32.
static int attention_decode_impl(rvb_engine* e, int N, float length_penalty) {
  const rvb_model_cfg& c = e->cfg;
  if (e->B <= 0) { set_error("rvb_attention_decode before rvb_encode"); return E_STATE; }
  if (!e->dec_l.present) { set_error("model has no attention decoder"); return E_STATE; }
  if (N < 1 || N > 64) { set_error("rvb_attention_decode: beam must be in 1..64"); return E_ARG; }
  RVB_TRY(wait_slices(e, -1));
  Decoder& D = e->dec_l;
  const int B = e->B, T2 = e->T2, d = c.d_model, heads = c.dec_heads, dk = d / heads, ff = c.dec_ffn_dim, V = c.vocab;
  const int eos = c.eos_id, sos = c.sos_id;
  const int R = B * N, L = T2, M = B * T2, NL = (int)D.layers.size();
  const size_t es = dt_size(e->dtype);
  if (L > e->pe_rows) { set_error("rvb_attention_decode: more steps than positional-table rows"); return E_UNSUPPORTED; }
  if (N > V) { set_error("rvb_attention_decode: beam larger than the vocabulary"); return E_ARG; }
  const int Vld = (V + 3) & ~3;

  e->kcache.resize(NL); e->vcache.resize(NL); e->kcache2.resize(NL); e->vcache2.resize(NL); e->memkv.resize(NL);
  const size_t cbytes = (size_t)R * L * d * es;
  for (int l = 0; l < NL; ++l) {
    RVB_TRY(e->kcache[l].ensure(cbytes)); RVB_TRY(e->vcache[l].ensure(cbytes));
    RVB_TRY(e->kcache2[l].ensure(cbytes)); RVB_TRY(e->vcache2[l].ensure(cbytes));
    RVB_TRY(e->memkv[l].ensure((size_t)M * 2 * d * es));
    RVB_TRY(run_gemm(e, e->enc_out.p, d, D.layers[l].src_kv, e->memkv[l].p, 2 * d, M, false));   // once per chunk, not per step
  }
  RVB_TRY(e->dx.ensure((size_t)R * d * 4));
  RVB_TRY(e->dxn.ensure((size_t)R * d * es));
  RVB_TRY(e->dy.ensure((size_t)R * d * es));
  RVB_TRY(e->dao.ensure((size_t)R * d * es));
  RVB_TRY(e->dq.ensure((size_t)R * d * es));
  RVB_TRY(e->dqkv.ensure((size_t)R * 3 * d * es));
  RVB_TRY(e->dh.ensure((size_t)R * ff * es));
  RVB_TRY(e->logits.ensure((size_t)std::min(R, LOGIT_SLAB) * Vld * 4));
  RVB_TRY(e->atopv.ensure((size_t)R * N * 4));
  RVB_TRY(e->atopi.ensure((size_t)R * N * 4));

  // sequence descriptors: self-attention = one query row per hypothesis against its cache rows [r*L, r*L + s];
  // cross-attention = the N hypotheses of a chunk form one query sequence against the chunk's valid frames
  std::vector<int32_t> q1(R), one(R, 1), kv0(R), kvl(R), cq(B), cn(B, N), ckv(2 * (size_t)B);
  for (int r = 0; r < R; ++r) { q1[r] = r; kv0[r] = r * L; }
  for (int b = 0; b < B; ++b) { cq[b] = b * N; ckv[b] = b * T2; ckv[B + b] = e->enc_lens[b]; }
  RVB_TRY(upload_i32(e, e->d_hq_start, q1.data(), R));
  RVB_TRY(upload_i32(e, e->d_hq_len, one.data(), R));
  RVB_TRY(upload_i32(e, e->d_seq_start, kv0.data(), R));
  RVB_TRY(upload_i32(e, e->d_hkv_start, cq.data(), B));
  RVB_TRY(upload_i32(e, e->d_hkv_len, cn.data(), B));
  RVB_TRY(upload_i32(e, e->d_aux_i32, ckv.data(), ckv.size()));
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));

  std::vector<std::vector<int>> hyps(R, std::vector<int>(1, sos));
  std::vector<float> scores(R, -INFINITY);
  for (int b = 0; b < B; ++b) scores[(size_t)b * N] = 0.f;            // search.py:289-292
  std::vector<char> end_flag(R, 0);
  // (a chunk without a single valid encoder frame is NOT special-cased: the reference decodes it against a fully masked
  // memory -- attention output zero, attention.py:112-114 -- and emits whatever the decoder's prior produces; the golden
  // case tiny_bn has such a 5-frame tail chunk)
  std::vector<int32_t> tok(R), pos(R), parent(R);
  std::vector<float> topv((size_t)R * N);
  std::vector<int32_t> topi((size_t)R * N);
  float* x = e->dx.as<float>();

  for (int s = 0; s < L; ++s) {                                        // i = s + 1 in search.py:296
    int n_end = 0;
    for (int r = 0; r < R; ++r) n_end += end_flag[r];
    if (n_end == R) break;
    for (int r = 0; r < R; ++r) { tok[r] = hyps[r].back(); pos[r] = s; kvl[r] = s + 1; }
    RVB_TRY(upload_i32(e, e->d_tok, tok.data(), R));
    RVB_TRY(upload_i32(e, e->d_pos, pos.data(), R));
    RVB_TRY(upload_i32(e, e->d_seq_len, kvl.data(), R));
    {
      Scope sc(e, "embed");
      RVB_TRY(embed_tokens(e->stream, D.embed.as<float>(), e->pe_f32.as<float>(), e->d_tok.as<int>(), e->d_pos.as<int>(), x, R, d,
                           std::sqrt((float)d)));
    }
    for (int l = 0; l < NL; ++l) {
      DecLayer& Ly = D.layers[l];
      RVB_TRY(run_norm(e, x, Ly.n1, e->dxn.p, false, R, d));
      RVB_TRY(run_gemm(e, e->dxn.p, d, Ly.self_qkv, e->dqkv.p, 3 * d, R, false));
      RVB_HIP_CHECK(hipMemcpy2DAsync((char*)e->kcache[l].p + (size_t)s * d * es, (size_t)L * d * es, (const char*)e->dqkv.p + (size_t)d * es,
                                     (size_t)3 * d * es, (size_t)d * es, R, hipMemcpyDeviceToDevice, e->stream));
      RVB_HIP_CHECK(hipMemcpy2DAsync((char*)e->vcache[l].p + (size_t)s * d * es, (size_t)L * d * es, (const char*)e->dqkv.p + (size_t)2 * d * es,
                                     (size_t)3 * d * es, (size_t)d * es, R, hipMemcpyDeviceToDevice, e->stream));
      AttnArgs a;
      memset(&a, 0, sizeof(a));
      a.q = e->dqkv.p; a.k = e->kcache[l].p; a.v = e->vcache[l].p;
      a.q_stride = 3 * d; a.k_stride = a.v_stride = d; a.o_stride = d; a.out = e->dao.p;
      a.q_start = e->d_hq_start.as<int>(); a.q_len = e->d_hq_len.as<int>();
      a.kv_start = e->d_seq_start.as<int>(); a.kv_len = e->d_seq_len.as<int>();
      a.nseq = R; a.heads = heads; a.dk = dk; a.max_q = 1; a.causal = 0; a.sqrt_dk = std::sqrt((float)dk);
      { Scope sc(e, "attention"); RVB_TRY(attention(e->stream, e->dtype, a)); }
      RVB_TRY(run_gemm(e, e->dao.p, d, Ly.self_out, x, d, R, true, 1.f, ACT_NONE, x, d));
      RVB_TRY(run_norm(e, x, Ly.n2, e->dxn.p, false, R, d));
      RVB_TRY(run_gemm(e, e->dxn.p, d, Ly.src_q, e->dq.p, d, R, false));
      a.q = e->dq.p; a.k = e->memkv[l].p; a.v = (const char*)e->memkv[l].p + (size_t)d * es;
      a.q_stride = d; a.k_stride = a.v_stride = 2 * d;
      a.q_start = e->d_hkv_start.as<int>(); a.q_len = e->d_hkv_len.as<int>();
      a.kv_start = e->d_aux_i32.as<int>(); a.kv_len = e->d_aux_i32.as<int>() + B;
      a.nseq = B; a.max_q = N;
      { Scope sc(e, "attention"); RVB_TRY(attention(e->stream, e->dtype, a)); }
      RVB_TRY(run_gemm(e, e->dao.p, d, Ly.src_out, x, d, R, true, 1.f, ACT_NONE, x, d));
      RVB_TRY(run_norm(e, x, Ly.n3, e->dxn.p, false, R, d));
      const void* ffin = e->dxn.p;
      if (Ly.is_lsl) { RVB_TRY(run_gemm(e, e->dxn.p, d, Ly.lsl, e->dy.p, d, R, false)); ffin = e->dy.p; }
      RVB_TRY(run_gemm(e, ffin, d, Ly.ff1, e->dh.p, ff, R, false, 1.f, ACT_RELU));
      RVB_TRY(run_gemm(e, e->dh.p, ff, Ly.ff2, x, d, R, true, 1.f, ACT_NONE, x, d));
    }
    RVB_TRY(run_norm(e, x, D.after, e->dxn.p, false, R, d));
    for (int r0 = 0; r0 < R; r0 += LOGIT_SLAB) {
      const int rows = std::min(LOGIT_SLAB, R - r0);
      RVB_TRY(run_gemm(e, (const char*)e->dxn.p + (size_t)r0 * d * es, d, D.out, e->logits.p, Vld, rows, true));
      Scope sc(e, "ctc_topk");
      RVB_TRY(logsoftmax_topk(e->stream, e->logits.as<float>(), rows, V, Vld, N, 0.f, 0, e->atopv.as<float>() + (size_t)r0 * N,
                              e->atopi.as<int>() + (size_t)r0 * N, nullptr));
    }
    RVB_HIP_CHECK(hipMemcpyAsync(topv.data(), e->atopv.p, (size_t)R * N * 4, hipMemcpyDeviceToHost, e->stream));
    RVB_HIP_CHECK(hipMemcpyAsync(topi.data(), e->atopi.p, (size_t)R * N * 4, hipMemcpyDeviceToHost, e->stream));
    RVB_HIP_CHECK(hipStreamSynchronize(e->stream));

    // ---- beam update, search.py:300-345 ----
    bool moved = false;
    std::vector<std::vector<int>> nh(R);
    std::vector<float> ns(R);
    std::vector<std::pair<float, int>> cand((size_t)N * N);
    for (int b = 0; b < B; ++b) {
      for (int n = 0; n < N; ++n) {
        const int r = b * N + n;
        for (int k = 0; k < N; ++k) {
          float lp = topv[(size_t)r * N + k];
          if (end_flag[r]) lp = k == 0 ? 0.f : -INFINITY;             // mask_finished_scores
          cand[(size_t)n * N + k] = {scores[r] + lp, n * N + k};
        }
      }
      // torch.topk: descending; equal values keep the lower index first
      std::stable_sort(cand.begin(), cand.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b2) { return a.first > b2.first; });
      for (int n = 0; n < N; ++n) {
        const int off = cand[n].second, pn = off / N, pk = off % N;
        const int pr = b * N + pn, r = b * N + n;
        const int pred = end_flag[pr] ? eos : topi[(size_t)pr * N + pk];   // mask_finished_preds
        nh[r] = hyps[pr];
        nh[r].push_back(pred);
        ns[r] = cand[n].first;
        parent[r] = pr;
        moved |= pr != r;
      }
    }
    hyps.swap(nh);
    scores.swap(ns);
    for (int r = 0; r < R; ++r) end_flag[r] = hyps[r].back() == eos;
    if (moved && s + 1 < L) {
      RVB_TRY(upload_i32(e, e->d_tgt, parent.data(), R));
      for (int l = 0; l < NL; ++l) {
        RVB_TRY(gather_cache(e->stream, e->kcache[l].p, e->kcache2[l].p, e->d_tgt.as<int>(), R, L, s + 1, (int)(d * es)));
        RVB_TRY(gather_cache(e->stream, e->vcache[l].p, e->vcache2[l].p, e->d_tgt.as<int>(), R, L, s + 1, (int)(d * es)));
        std::swap(e->kcache[l], e->kcache2[l]);
        std::swap(e->vcache[l], e->vcache2[l]);
      }
    }
  }
  // ---- best of the beam, search.py:347-360 (float32 like the tensors there) ----
  e->attn_tokens.assign(B, {});
  e->attn_scores.assign(B, 0.f);
  for (int b = 0; b < B; ++b) {
    float best = -INFINITY;
    int bi = 0;
    for (int n = 0; n < N; ++n) {
      const std::vector<int>& h = hyps[(size_t)b * N + n];
      int len = 0;
      for (int t : h) len += t != eos;
      const float sc = scores[(size_t)b * N + n] / std::pow((float)len, length_penalty);
      if (n == 0 || sc > best) { best = sc; bi = n; }
    }
    const std::vector<int>& h = hyps[(size_t)b * N + bi];
    for (size_t j = 1; j < h.size(); ++j) if (h[j] != eos) e->attn_tokens[b].push_back(h[j]);
    e->attn_scores[b] = best;
  }
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  return OK;
}


// ------------------------------------------------------------------------------------ joint_decoding
// `joint_decoding` (transformer/search.py:450-496 -> espnet/beam_search_timesync.py:86-508): time-synchronous joint CTC /
// attention beam search.  The reference runs one BeamSearchTimeSync per chunk and, inside it, the attention decoder on ONE
// new prefix at a time, re-feeding the whole prefix with the cached layer outputs of its parent (cached_score :185-224).
// Here every chunk of the batch advances in lockstep, one encoder frame per iteration:
//   * the CTC half of the frame and the joint scoring run on the host, per chunk (search.cpp JointSearch: the reference's
//     float64 arithmetic and dict semantics on a prefix trie);
//   * the prefixes whose decoder output is needed for the first time -- of ALL chunks -- form one batched decoder step:
//     one row per prefix (its last token), self-attention over the key / value rows of its ancestors (kept per decoder
//     layer for every decoded prefix, addressed through AttnArgs::kv_index), cross-attention against the chunk's memory
//     keys / values (projected once), output layer, log-softmax row kept on the device;
//   * the (prefix, next token) log-probs the joint scores need are gathered from those rows and copied back: a few
//     floats per chunk and frame.
// The memory is the chunk's valid frames, as the class's own `reset` expects ((1, len, d); the reference's call passes a
// 2-D tensor and fails there -- DESIGN.md, oracle/gen_golden_joint.py).
static int grow_rows(rvb_engine* e, DevBuf& b, size_t row_bytes, int64_t have, int64_t need) {
  if ((size_t)need * row_bytes <= b.bytes) return OK;
  DevBuf nb;
  const int64_t cap = std::max<int64_t>(need, (int64_t)(b.bytes / row_bytes) * 2);
  RVB_TRY(nb.ensure((size_t)cap * row_bytes));
  if (have > 0) RVB_HIP_CHECK(hipMemcpyAsync(nb.p, b.p, (size_t)have * row_bytes, hipMemcpyDeviceToDevice, e->stream));
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  b.release();
  b = nb;
  nb.p = nullptr; nb.bytes = 0;
  return OK;
}

// a few persistent host threads for the per-chunk halves of a joint_decoding frame (512 frames per batch: starting threads
// per frame would cost more than the work)
namespace {
class FramePool {
 public:
  explicit FramePool(unsigned n) {
    for (unsigned i = 1; i < n; ++i) th_.emplace_back([this, i] { loop((int)i); });
  }
  ~FramePool() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  // fn(i) for i in [0, n); item i always runs on participant i mod P (the caller is participant 0), so that a chunk's
  // search state is touched -- and its vectors are grown and freed -- by one thread only
  template <typename F> void run(int n, F&& fn) {
    if (th_.empty() || n < 8) { for (int i = 0; i < n; ++i) fn(i); return; }
    job_ = [&fn](int i) { fn(i); };
    { std::lock_guard<std::mutex> g(m_); n_ = n; busy_ = (int)th_.size(); ++gen_; }
    cv_.notify_all();
    work(0);
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this] { return busy_ == 0; });
  }

 private:
  void work(int id) { const int P = (int)th_.size() + 1; for (int i = id; i < n_; i += P) job_(i); }
  void loop(int id) {
    int seen = 0;
    for (;;) {
      { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [&] { return gen_ != seen; }); seen = gen_; if (stop_) return; }
      work(id);
      { std::lock_guard<std::mutex> g(m_); if (--busy_ == 0) done_.notify_one(); }
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  std::function<void(int)> job_;
  int n_ = 0, busy_ = 0, gen_ = 0;
  bool stop_ = false;
};
}  // namespace

static int joint_decode_impl(rvb_engine* e, int beam, double ctc_weight, double pre_beam_ratio, double length_bonus) {
  const rvb_model_cfg& c = e->cfg;
  if (e->B <= 0) { set_error("rvb_joint_decode before rvb_encode"); return E_STATE; }
  if (!e->dec_l.present) { set_error("model has no attention decoder"); return E_STATE; }
  const int pre_beam = (int)(pre_beam_ratio * beam);
  if (beam < 1 || pre_beam < 1) { set_error("rvb_joint_decode: beam and pre_beam_ratio * beam must be >= 1"); return E_ARG; }
  if (pre_beam > e->beam) {
    set_error("rvb_joint_decode: rvb_encode kept the top " + std::to_string(e->beam) + " CTC log-probs per frame, the pre-beam needs " +
              std::to_string(pre_beam));
    return E_ARG;
  }
  RVB_TRY(wait_slices(e, -1));
  Decoder& D = e->dec_l;
  const int B = e->B, T2 = e->T2, d = c.d_model, heads = c.dec_heads, dk = d / heads, ff = c.dec_ffn_dim, V = c.vocab;
  const int M = B * T2, NL = (int)D.layers.size(), K = e->beam;
  const size_t es = dt_size(e->dtype);
  const int Vld = (V + 3) & ~3;
  if (T2 + 1 > e->pe_rows) { set_error("rvb_joint_decode: more positions than positional-table rows"); return E_UNSUPPORTED; }

  // ---- per-frame log-prob of token 0 (the reference's blank-skip test reads p_ctc[0]) and of the blank
  std::vector<float> p0(M), pbl(M);
  {
    RVB_TRY(e->logits.ensure((size_t)LOGIT_SLAB * Vld * 4));
    RVB_TRY(e->d_tgt.ensure((size_t)LOGIT_SLAB * 4));
    RVB_TRY(e->d_logp.ensure((size_t)LOGIT_SLAB * 4));
    std::vector<int32_t> tgt(LOGIT_SLAB);
    for (int pass = 0; pass < (c.blank_id == 0 ? 1 : 2); ++pass) {
      std::fill(tgt.begin(), tgt.end(), pass == 0 ? 0 : c.blank_id);
      RVB_TRY(upload_i32(e, e->d_tgt, tgt.data(), LOGIT_SLAB));
      std::vector<float>& dst = pass == 0 ? p0 : pbl;
      for (int r0 = 0; r0 < M; r0 += LOGIT_SLAB) {
        const int rows = std::min(LOGIT_SLAB, M - r0);
        RVB_TRY(run_gemm(e, (const char*)e->enc_out.p + (size_t)r0 * d * es, d, e->ctc, e->logits.p, Vld, rows, true));
        // with a blank penalty the reference's CTC log-probs are the log-softmax of the PENALISED logits (ctc_logprobs,
        // asr_model.py:318-329; search.py:466): the same rows the top-k kernel produced for this batch
        RVB_TRY(lse_gather(e->stream, e->logits.as<float>(), rows, V, Vld, e->d_tgt.as<int>(), e->d_logp.as<float>(), e->last_blank_penalty,
                           c.blank_id));
        RVB_HIP_CHECK(hipMemcpyAsync(dst.data() + r0, e->d_logp.p, (size_t)rows * 4, hipMemcpyDeviceToHost, e->stream));
        RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
      }
    }
    if (c.blank_id == 0) pbl = p0;
  }

  // ---- memory keys / values of every chunk, once per decoder layer
  e->memkv.resize(NL); e->jkv.resize(NL);
  for (int l = 0; l < NL; ++l) {
    RVB_TRY(e->memkv[l].ensure((size_t)M * 2 * d * es));
    RVB_TRY(run_gemm(e, e->enc_out.p, d, D.layers[l].src_kv, e->memkv[l].p, 2 * d, M, false));
  }
  const int Rmax = B * std::max(beam, 1);
  RVB_TRY(e->dx.ensure((size_t)Rmax * d * 4));
  RVB_TRY(e->dxn.ensure((size_t)Rmax * d * es));
  RVB_TRY(e->dy.ensure((size_t)Rmax * d * es));
  RVB_TRY(e->dao.ensure((size_t)Rmax * d * es));
  RVB_TRY(e->dq.ensure((size_t)Rmax * d * es));
  RVB_TRY(e->dqkv.ensure((size_t)Rmax * 3 * d * es));
  RVB_TRY(e->dh.ensure((size_t)Rmax * ff * es));
  RVB_TRY(e->logits.ensure((size_t)std::max(Rmax, LOGIT_SLAB) * Vld * 4));
  RVB_TRY(e->atopv.ensure((size_t)Rmax * 4));
  RVB_TRY(e->atopi.ensure((size_t)Rmax * 4));

  JointParams jp;
  jp.beam = beam; jp.pre_beam = pre_beam; jp.blank = c.blank_id; jp.sos = c.sos_id;
  jp.w_ctc = ctc_weight; jp.w_dec = 1.0 - ctc_weight; jp.bonus = length_bonus; jp.log_thr = 0.0;
  std::vector<JointSearch> js;
  js.reserve(B);
  for (int b = 0; b < B; ++b) js.emplace_back(jp);
  int64_t next_row = 0;
  e->joint_rows = 0; e->joint_steps = 0;
  float* x = e->dx.as<float>();

  // one batched decoder step for the prefixes (chunk, node) in `req` (grouped by chunk, in order)
  struct Req { int chunk, node; };
  std::vector<int32_t> tok, pos, q1, one, pstart, plen, path, xq0, xqn, xkv;
  auto step = [&](const std::vector<Req>& req) -> int {
    const int R = (int)req.size();
    if (R == 0) return OK;
    if (R > Rmax) { set_error("rvb_joint_decode: more new prefixes in one frame than beam x chunks"); return E_STATE; }
    const int64_t row0 = next_row;
    for (int l = 0; l < NL; ++l) RVB_TRY(grow_rows(e, e->jkv[l], (size_t)2 * d * es, row0, row0 + R));
    RVB_TRY(grow_rows(e, e->jlogp, (size_t)V * 4, row0, row0 + R));
    tok.resize(R); pos.resize(R); q1.resize(R); one.assign(R, 1); pstart.resize(R); plen.resize(R);
    path.clear(); xq0.clear(); xqn.clear(); xkv.clear();
    std::vector<int32_t> xk0, xkn;
    int max_xq = 0;
    for (int r = 0; r < R; ++r) {
      JointSearch& J = js[req[r].chunk];
      const int node = req[r].node;
      J.set_tag(node, (int)(row0 + r));
      const int len = J.length(node);
      tok[r] = J.token(node); pos[r] = len - 1; q1[r] = r;
      pstart[r] = (int)path.size(); plen[r] = len;
      const size_t at = path.size();
      path.resize(at + len);
      for (int n = node, i = len - 1; n >= 0; n = J.parent(n), --i) path[at + i] = J.tag(n);   // ancestors are decoded
      if (r == 0 || req[r].chunk != req[r - 1].chunk) { xq0.push_back(r); xqn.push_back(0); xk0.push_back(req[r].chunk * T2); xkn.push_back(e->enc_lens[req[r].chunk]); }
      max_xq = std::max(max_xq, ++xqn.back());
    }
    const int nx = (int)xq0.size();
    xkv = xk0; xkv.insert(xkv.end(), xkn.begin(), xkn.end());
    RVB_TRY(upload_i32(e, e->d_tok, tok.data(), R));
    RVB_TRY(upload_i32(e, e->d_pos, pos.data(), R));
    RVB_TRY(upload_i32(e, e->d_hq_start, q1.data(), R));
    RVB_TRY(upload_i32(e, e->d_hq_len, one.data(), R));
    RVB_TRY(upload_i32(e, e->d_hpath_start, pstart.data(), R));
    RVB_TRY(upload_i32(e, e->d_hpath_len, plen.data(), R));
    RVB_TRY(upload_i32(e, e->d_path, path.data(), path.size()));
    RVB_TRY(upload_i32(e, e->d_hkv_start, xq0.data(), nx));
    RVB_TRY(upload_i32(e, e->d_hkv_len, xqn.data(), nx));
    RVB_TRY(upload_i32(e, e->d_aux_i32, xkv.data(), xkv.size()));
    {
      Scope sc(e, "embed");
      RVB_TRY(embed_tokens(e->stream, D.embed.as<float>(), e->pe_f32.as<float>(), e->d_tok.as<int>(), e->d_pos.as<int>(), x, R, d,
                           std::sqrt((float)d)));
    }
    for (int l = 0; l < NL; ++l) {
      DecLayer& Ly = D.layers[l];
      RVB_TRY(run_norm(e, x, Ly.n1, e->dxn.p, false, R, d));
      RVB_TRY(run_gemm(e, e->dxn.p, d, Ly.self_qkv, e->dqkv.p, 3 * d, R, false));
      // the new prefixes' key | value rows join the per-layer store (k and v are adjacent in the fused projection)
      RVB_HIP_CHECK(hipMemcpy2DAsync((char*)e->jkv[l].p + (size_t)row0 * 2 * d * es, (size_t)2 * d * es, (const char*)e->dqkv.p + (size_t)d * es,
                                     (size_t)3 * d * es, (size_t)2 * d * es, R, hipMemcpyDeviceToDevice, e->stream));
      AttnArgs a;
      memset(&a, 0, sizeof(a));
      a.q = e->dqkv.p; a.k = e->jkv[l].p; a.v = (const char*)e->jkv[l].p + (size_t)d * es;
      a.q_stride = 3 * d; a.k_stride = a.v_stride = 2 * d; a.o_stride = d; a.out = e->dao.p;
      a.q_start = e->d_hq_start.as<int>(); a.q_len = e->d_hq_len.as<int>();
      a.kv_start = e->d_hpath_start.as<int>(); a.kv_len = e->d_hpath_len.as<int>(); a.kv_index = e->d_path.as<int>();
      a.nseq = R; a.heads = heads; a.dk = dk; a.max_q = 1; a.causal = 0; a.sqrt_dk = std::sqrt((float)dk);
      { Scope sc(e, "attention"); RVB_TRY(attention(e->stream, e->dtype, a)); }
      RVB_TRY(run_gemm(e, e->dao.p, d, Ly.self_out, x, d, R, true, 1.f, ACT_NONE, x, d));
      RVB_TRY(run_norm(e, x, Ly.n2, e->dxn.p, false, R, d));
      RVB_TRY(run_gemm(e, e->dxn.p, d, Ly.src_q, e->dq.p, d, R, false));
      memset(&a, 0, sizeof(a));
      a.q = e->dq.p; a.k = e->memkv[l].p; a.v = (const char*)e->memkv[l].p + (size_t)d * es;
      a.q_stride = d; a.k_stride = a.v_stride = 2 * d; a.o_stride = d; a.out = e->dao.p;
      a.q_start = e->d_hkv_start.as<int>(); a.q_len = e->d_hkv_len.as<int>();
      a.kv_start = e->d_aux_i32.as<int>(); a.kv_len = e->d_aux_i32.as<int>() + nx;
      a.nseq = nx; a.heads = heads; a.dk = dk; a.max_q = max_xq; a.causal = 0; a.sqrt_dk = std::sqrt((float)dk);
      { Scope sc(e, "attention"); RVB_TRY(attention(e->stream, e->dtype, a)); }
      RVB_TRY(run_gemm(e, e->dao.p, d, Ly.src_out, x, d, R, true, 1.f, ACT_NONE, x, d));
      RVB_TRY(run_norm(e, x, Ly.n3, e->dxn.p, false, R, d));
      const void* ffin = e->dxn.p;
      if (Ly.is_lsl) { RVB_TRY(run_gemm(e, e->dxn.p, d, Ly.lsl, e->dy.p, d, R, false)); ffin = e->dy.p; }
      RVB_TRY(run_gemm(e, ffin, d, Ly.ff1, e->dh.p, ff, R, false, 1.f, ACT_RELU));
      RVB_TRY(run_gemm(e, e->dh.p, ff, Ly.ff2, x, d, R, true, 1.f, ACT_NONE, x, d));
    }
    RVB_TRY(run_norm(e, x, D.after, e->dxn.p, false, R, d));
    RVB_TRY(run_gemm(e, e->dxn.p, d, D.out, e->logits.p, Vld, R, true));
    {
      Scope sc(e, "ctc_topk");
      RVB_TRY(logsoftmax_topk(e->stream, e->logits.as<float>(), R, V, Vld, 1, 0.f, 0, e->atopv.as<float>(), e->atopi.as<int>(),
                              e->jlogp.as<float>() + (size_t)row0 * V));
    }
    next_row += R;
    e->joint_rows += R; e->joint_steps += 1;
    return OK;
  };

  // reset(): the decoder on <sos> for every chunk
  std::vector<Req> req;
  for (int b = 0; b < B; ++b) req.push_back({b, 0});
  RVB_TRY(step(req));

  int Tmax = 0;
  for (int b = 0; b < B; ++b) Tmax = std::max(Tmax, e->enc_lens[b]);
  std::vector<int> npairs(B), ran(B);
  std::vector<std::vector<int>> cdec(B), cpn(B), cpt(B);
  std::vector<size_t> pair_at(B + 1);
  FramePool pool(std::min<unsigned>(search_threads(), 16u));
  std::vector<int32_t> prow, ptok;
  std::vector<float> vals;
  for (int t = 0; t < Tmax; ++t) {
    req.clear(); prow.clear(); ptok.clear();
    pool.run(B, [&](int b) {                     // CTC half of the frame, chunk by chunk on the host threads
      ran[b] = 0; cdec[b].clear(); cpn[b].clear(); cpt[b].clear();
      if (t >= e->enc_lens[b]) return;
      const size_t f = (size_t)b * T2 + t;
      ran[b] = js[b].begin_frame(t, e->h_topv + f * K, e->h_topi + f * K, K, p0[f], pbl[f], &cdec[b], &cpn[b], &cpt[b]) ? 1 : 0;
    });
    for (int b = 0; b < B; ++b) {
      npairs[b] = 0;
      if (!ran[b]) continue;
      for (int n : cdec[b]) req.push_back({b, n});
      npairs[b] = (int)cpn[b].size();
      for (size_t i = 0; i < cpn[b].size(); ++i) { prow.push_back(-1 - cpn[b][i]); ptok.push_back(cpt[b][i]); }      // rows resolved after the step
    }
    RVB_TRY(step(req));
    {   // pair rows: the node's tag is known now
      size_t at = 0;
      for (int b = 0; b < B; ++b)
        for (int i = 0; i < npairs[b]; ++i, ++at) prow[at] = js[b].tag(-1 - prow[at]);
    }
    const int NP = (int)prow.size();
    vals.resize(std::max(NP, 1));
    if (NP > 0) {
      RVB_TRY(e->jpair_row.ensure((size_t)NP * 4)); RVB_TRY(e->jpair_tok.ensure((size_t)NP * 4)); RVB_TRY(e->jpair_out.ensure((size_t)NP * 4));
      RVB_TRY(upload_i32(e, e->jpair_row, prow.data(), NP));
      RVB_TRY(upload_i32(e, e->jpair_tok, ptok.data(), NP));
      RVB_TRY(gather_pairs(e->stream, e->jlogp.as<float>(), (size_t)V, e->jpair_row.as<int>(), e->jpair_tok.as<int>(), NP, e->jpair_out.as<float>()));
      RVB_HIP_CHECK(hipMemcpyAsync(vals.data(), e->jpair_out.p, (size_t)NP * 4, hipMemcpyDeviceToHost, e->stream));
      RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
    }
    pair_at[0] = 0;
    for (int b = 0; b < B; ++b) pair_at[b + 1] = pair_at[b] + (size_t)npairs[b];
    pool.run(B, [&](int b) { if (ran[b]) js[b].finish_frame(vals.data() + pair_at[b]); });
  }
  e->joint.assign(B, JointResult());
  for (int b = 0; b < B; ++b) js[b].result(&e->joint[b]);
  for (int b = 0; b < B; ++b)
    if (js[b].ties_cut() && K < V) {
      set_error("rvb_joint_decode: a frame of chunk " + std::to_string(b) + " has more than " + std::to_string(K - pre_beam) +
                " log-probs that tie exactly with the pre-beam threshold; keep more per frame (rvb_encode's beam argument, up to 64)");
      return E_UNSUPPORTED;
    }
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  return OK;
}

}  // namespace rvb

// ================================================================================================
//                                         C ABI
// ================================================================================================
using namespace rvb;

extern "C" {

const char* rvb_last_error(void) { return last_error(); }
const char* rvb_version(void) { return "librvb 0.1 (gfx950)"; }

}  // extern "C" (reopened below)
namespace rvb {
#ifdef RVB_TEST_API
const char* lab_env(const char* name) { return getenv(name); }      // librvb_test.so: the lab build reads its switches
#else
const char* lab_env(const char*) { return nullptr; }                // librvb.so: defaults only
#endif
}  // namespace rvb
extern "C" {

int rvb_model_cfg_size(void) { return (int)sizeof(rvb_model_cfg); }

// host only (no GPU needed): word alignment counts for reverb_amd/wer_evaluation/align.py (search.cpp edit_counts)
int rvb_wer_counts(const int32_t* ref, int64_t n_ref, const int32_t* hyp, int64_t n_hyp, int64_t* counts) {
  if ((!ref && n_ref > 0) || (!hyp && n_hyp > 0) || !counts || n_ref < 0 || n_hyp < 0) { set_error("rvb_wer_counts: bad argument"); return E_ARG; }
  edit_counts(ref, n_ref, hyp, n_hyp, counts);
  return OK;
}

int rvb_create(const rvb_model_cfg* cfg, int device, rvb_engine** out) {
  if (!cfg || !out) { set_error("rvb_create: null argument"); return E_ARG; }
  if (cfg->struct_size != (int32_t)sizeof(rvb_model_cfg)) {
    set_error("rvb_create: ABI mismatch: rvb_model_cfg.struct_size is " + std::to_string(cfg->struct_size) + ", this library's struct has " +
              std::to_string(sizeof(rvb_model_cfg)) + " bytes (bind it field by field from include/rvb.h)");
    return E_ARG;
  }
  if (cfg->dtype != RVB_F32 && cfg->dtype != RVB_BF16 && cfg->dtype != RVB_FP8) { set_error("rvb_create: bad dtype"); return E_ARG; }
  if (cfg->d_model <= 0 || cfg->heads <= 0 || cfg->d_model % cfg->heads || cfg->d_model % 8 || cfg->ffn_dim % 8 ||
      cfg->dec_ffn_dim % 8 || cfg->input_dim != 80 || cfg->vocab < 2 || cfg->num_blocks < 1 ||
      cfg->cnn_kernel < 1 || cfg->cnn_kernel > 31 || (!cfg->cnn_causal && (cfg->cnn_kernel % 2) == 0) ||
      cfg->chunk_frames < 7 || cfg->max_chunks < 1 ||
      (cfg->dec_blocks > 0 && (cfg->dec_heads <= 0 || cfg->d_model % cfg->dec_heads))) {
    set_error("rvb_create: unsupported model dimensions (need d, ffn dims % 8 == 0, d % heads == 0, input_dim 80, cnn kernel <= 31 and odd unless causal)");
    return E_ARG;
  }
  int ndev = 0;
  hipError_t err = hipGetDeviceCount(&ndev);
  if (err != hipSuccess || ndev <= 0) { set_error("no HIP device available: librvb has no CPU fallback"); return E_HIP; }
  if (device < 0 || device >= ndev) { set_error("rvb_create: device index out of range"); return E_ARG; }
  RVB_HIP_CHECK(hipSetDevice(device));
  rvb_engine* e = new rvb_engine();
  e->cfg = *cfg; e->device = device;
  e->fp8 = cfg->dtype == RVB_FP8;                 // the bf16 engine with the encoder's large GEMMs on the fp8 MFMA path
  e->dtype = e->fp8 ? (int)DT_BF16 : cfg->dtype;
  hipError_t se = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
  if (se != hipSuccess) { delete e; set_error("hipStreamCreate failed"); return E_HIP; }
  *out = e;
  return OK;
}

void rvb_destroy(rvb_engine* e) {
  if (!e) return;
  (void)rvb_comm_destroy(e);
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  // DevBuf members are released explicitly: list the big ones, the rest die with the process
  DevBuf* bufs[] = {&e->cmvn_mean, &e->cmvn_istd, &e->conv1_w, &e->conv1_b, &e->pe_f32, &e->stage, &e->pcm, &e->feats,
                    &e->d_feats_in, &e->X1, &e->X2, &e->x, &e->xn, &e->y, &e->h, &e->ao, &e->dconv, &e->enc_out,
                    &e->logits, &e->topv, &e->topi, &e->d_enc_lens, &e->d_seq_start, &e->d_seq_len, &e->d_aux_i32,
                    &e->dx, &e->dxn, &e->dy, &e->dh, &e->dqkv, &e->dq, &e->dao, &e->kvmem, &e->d_tok, &e->d_pos,
                    &e->d_tgt, &e->d_logp, &e->d_hq_start, &e->d_hq_len, &e->d_hkv_start, &e->d_hkv_len,
                    &e->d_hq_pos0, &e->d_hpath_start, &e->d_hpath_len, &e->d_path, &e->d_work, &e->d_tgt_ptr,
                    &e->fb_window, &e->fb_twiddle, &e->fb_melw, &e->fb_lo, &e->fb_hi,
                    &e->conv2.w, &e->conv2.b, &e->embed_out.w, &e->embed_out.b, &e->ctc.w, &e->ctc.b,
                    &e->enc_after.g, &e->enc_after.b};
  for (DevBuf* b : bufs) b->release();
  e->atopv.release(); e->atopi.release(); e->d_stream_i32.release(); e->d_amax.release(); e->d_f8sat.release();
  e->wave_f32.release(); e->wave_in.release(); e->rs_kernel.release();
  e->jlogp.release(); e->jpair_row.release(); e->jpair_tok.release(); e->jpair_out.release();
  for (auto& b : e->jkv) b.release();
  for (auto* v : {&e->stream_st.kv, &e->stream_st.kv2, &e->stream_st.cnn, &e->stream_st.cnn2}) for (auto& b : *v) b.release();
  for (auto* v : {&e->kcache, &e->vcache, &e->kcache2, &e->vcache2, &e->memkv}) for (auto& b : *v) b.release();
  auto rel_lin = [](Linear& l) { l.w.release(); l.b.release(); l.w8.release(); l.wscale.release(); };
  auto rel_n = [](LNorm& n) { n.g.release(); n.b.release(); };
  for (auto& L : e->enc) {
    for (Linear* l : {&L.ffm1, &L.ffm2, &L.ff1, &L.ff2, &L.qkv, &L.att_out, &L.pw1, &L.pw1_glu, &L.pw2, &L.lsl}) rel_lin(*l);
    for (LNorm* n : {&L.n_ffm, &L.n_mha, &L.n_conv, &L.n_ff, &L.n_final, &L.n_cnn}) rel_n(*n);
    L.pos_keys.release(); L.pos_bias.release(); L.bias_u.release(); L.bias_v.release(); L.dw_w.release(); L.dw_b.release();
  }
  for (Decoder* D : {&e->dec_l, &e->dec_r}) {
    D->embed.release(); rel_lin(D->out); rel_n(D->after);
    for (auto& L : D->layers) {
      for (Linear* l : {&L.self_qkv, &L.self_out, &L.src_q, &L.src_kv, &L.src_out, &L.ff1, &L.ff2, &L.lsl}) rel_lin(*l);
      for (LNorm* n : {&L.n1, &L.n2, &L.n3}) rel_n(*n);
      L.kvmem.release();
    }
  }
  if (e->h_topv) (void)hipHostFree(e->h_topv);
  if (e->h_topi) (void)hipHostFree(e->h_topi);
  for (auto& sl : e->slices) if (!sl.done) (void)hipEventDestroy(sl.ev);
  for (auto ev : e->slice_event_pool) (void)hipEventDestroy(ev);
  for (auto& p : e->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto ev : e->event_pool) (void)hipEventDestroy(ev);
  if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
  if (e->pcm_ready) (void)hipEventDestroy(e->pcm_ready);
  if (e->pcm_free) (void)hipEventDestroy(e->pcm_free);
  e->pcm_next.release();
  (void)hipStreamDestroy(e->stream);
  delete e;
}

int rvb_load_tensor(rvb_engine* e, const char* name, const float* host, const int64_t* shape, int ndim) {
  if (!e || !name || !host || ndim < 0 || (ndim > 0 && !shape)) { set_error("rvb_load_tensor: null argument"); return E_ARG; }
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) { if (shape[i] < 0) { set_error("negative dim"); return E_ARG; } t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
  t.data.assign(host, host + n);
  e->host[name] = std::move(t);
  return OK;
}

int rvb_finalize(rvb_engine* e, const float* cat_embs, int n_cat) {
  if (!e || (n_cat > 0 && !cat_embs)) { set_error("rvb_finalize: null argument"); return E_ARG; }
  return finalize_impl(e, cat_embs, n_cat);
}

int64_t rvb_num_frames(int64_t n) { return n < 400 ? 0 : 1 + (n - 400) / 160; }

// a synchronous upload replaces whatever rvb_upload_pcm_async left pending (the later call wins)
static int drop_pending_upload(rvb_engine* e) {
  if (e->pcm_pending) {
    RVB_HIP_CHECK(hipStreamSynchronize(e->copy_stream));
    e->pcm_pending = false;
  }
  return OK;
}

int rvb_upload_pcm(rvb_engine* e, const int16_t* pcm, int64_t n) {
  if (!e || (!pcm && n > 0) || n < 0) { set_error("rvb_upload_pcm: bad argument"); return E_ARG; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  RVB_TRY(drop_pending_upload(e));
  RVB_TRY(e->pcm.ensure((size_t)n * 2 + 16));
  if (n) RVB_HIP_CHECK(hipMemcpyAsync(e->pcm.p, pcm, (size_t)n * 2, hipMemcpyHostToDevice, e->stream));
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  e->n_samples = n;
  e->pcm_is_float = false;
  return OK;
}

// The double-buffered form: returns as soon as the copy is enqueued on the engine's copy stream (page-locked `pcm`: it then runs
// on a DMA engine underneath whatever the engine's compute stream is doing -- the decoding of the PREVIOUS recording); the
// samples become the engine's audio at the next rvb_fbank, which orders itself behind the copy.  `pcm` must stay valid and
// unchanged until then.  One upload may be pending at a time.
int rvb_upload_pcm_async(rvb_engine* e, const int16_t* pcm, int64_t n) {
  if (!e || (!pcm && n > 0) || n < 0) { set_error("rvb_upload_pcm_async: bad argument"); return E_ARG; }
  if (e->pcm_pending) { set_error("rvb_upload_pcm_async: an upload is already pending (consumed by the next rvb_fbank)"); return E_STATE; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  if (!e->copy_stream) {
    RVB_HIP_CHECK(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    RVB_HIP_CHECK(hipEventCreateWithFlags(&e->pcm_ready, hipEventDisableTiming));
    RVB_HIP_CHECK(hipEventCreateWithFlags(&e->pcm_free, hipEventDisableTiming));
  }
  else RVB_HIP_CHECK(hipStreamWaitEvent(e->copy_stream, e->pcm_free, 0));        // WAR: the last fbank may still read a PCM buffer
  RVB_TRY(e->pcm_next.ensure((size_t)n * 2 + 16));
  if (n) RVB_HIP_CHECK(hipMemcpyAsync(e->pcm_next.p, pcm, (size_t)n * 2, hipMemcpyHostToDevice, e->copy_stream));
  RVB_HIP_CHECK(hipEventRecord(e->pcm_ready, e->copy_stream));
  e->n_samples_next = n;
  e->pcm_pending = true;
  return OK;
}

int rvb_host_alloc(void** out, int64_t bytes) {
  if (!out || bytes < 0) { set_error("rvb_host_alloc: bad argument"); return E_ARG; }
  *out = nullptr;
  hipError_t err = hipHostMalloc(out, (size_t)std::max<int64_t>(bytes, 16), hipHostMallocDefault);
  if (err != hipSuccess) { *out = nullptr; set_error(std::string("hipHostMalloc: ") + hipGetErrorString(err)); return E_NOMEM; }
  return OK;
}
int rvb_host_free(void* p) {
  if (p) RVB_HIP_CHECK(hipHostFree(p));
  return OK;
}

int rvb_set_fp8_policy(rvb_engine* e, int groups, int first_block, int last_block) {
  if (!e) { set_error("rvb_set_fp8_policy: null engine"); return E_ARG; }
  if (!e->fp8) { set_error("rvb_set_fp8_policy: not an RVB_FP8 engine"); return E_STATE; }
  if (groups > 63 || first_block < 0) { set_error("rvb_set_fp8_policy: groups is a 6-bit mask (bit 5: the subsampling's conv2), first_block >= 0"); return E_ARG; }
  return set_fp8_policy_impl(e, groups, first_block, last_block);
}

int rvb_set_decoding_chunk(rvb_engine* e, int chunk_size, int num_left_chunks) {
  if (!e) { set_error("rvb_set_decoding_chunk: null engine"); return E_ARG; }
  if (chunk_size > 4095 || num_left_chunks > 4094) { set_error("rvb_set_decoding_chunk: value too large"); return E_ARG; }
  e->dec_chunk = chunk_size > 0 ? chunk_size : 0;
  e->dec_left = num_left_chunks < 0 ? -1 : num_left_chunks;
  return OK;
}

// torchaudio.transforms.Resample(sample_rate, 16000) of the uploaded samples (int16 in e->pcm or float in e->wave_in)
// into e->wave_f32 (cli/reverb.py:131-134)
static int resample_uploaded(rvb_engine* e, bool src_float, int64_t n, int sample_rate) {
  const int target = 16000;
  std::vector<float> ker;
  int orig, nw, width, K;
  resample_taps(sample_rate, target, &ker, &orig, &nw, &width, &K);
  const int64_t n_out = (nw * n + orig - 1) / orig;          // ceil(new * length / orig)
  RVB_TRY(e->rs_kernel.ensure(ker.size() * 4));
  RVB_TRY(e->wave_f32.ensure((size_t)std::max<int64_t>(n_out, 1) * 4));
  RVB_HIP_CHECK(hipMemcpyAsync(e->rs_kernel.p, ker.data(), ker.size() * 4, hipMemcpyHostToDevice, e->stream));
  {
    Scope sc(e, "resample");
    if (src_float) RVB_TRY(resample_f32(e->stream, e->wave_in.as<float>(), n, e->rs_kernel.as<float>(), orig, nw, width, K, e->wave_f32.as<float>(), n_out));
    else RVB_TRY(resample(e->stream, e->pcm.as<int16_t>(), n, e->rs_kernel.as<float>(), orig, nw, width, K, e->wave_f32.as<float>(), n_out));
  }
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  e->n_samples = n_out;
  e->pcm_is_float = true;
  return OK;
}

int rvb_upload_pcm_rate(rvb_engine* e, const int16_t* pcm, int64_t n, int sample_rate) {
  if (sample_rate == 16000) return rvb_upload_pcm(e, pcm, n);
  if (!e || (!pcm && n > 0) || n < 0 || sample_rate < 1000 || sample_rate > 384000) { set_error("rvb_upload_pcm_rate: bad argument"); return E_ARG; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  RVB_TRY(drop_pending_upload(e));
  RVB_TRY(e->pcm.ensure((size_t)n * 2 + 16));
  if (n) RVB_HIP_CHECK(hipMemcpyAsync(e->pcm.p, pcm, (size_t)n * 2, hipMemcpyHostToDevice, e->stream));
  return resample_uploaded(e, false, n, sample_rate);
}

int rvb_upload_wave_f32(rvb_engine* e, const float* wave, int64_t n, int sample_rate) {
  if (!e || (!wave && n > 0) || n < 0 || sample_rate < 1000 || sample_rate > 384000) { set_error("rvb_upload_wave_f32: bad argument"); return E_ARG; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  RVB_TRY(drop_pending_upload(e));
  rvb::DevBuf& dst = sample_rate == 16000 ? e->wave_f32 : e->wave_in;
  RVB_TRY(dst.ensure((size_t)std::max<int64_t>(n, 1) * 4));
  if (n) RVB_HIP_CHECK(hipMemcpyAsync(dst.p, wave, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
  if (sample_rate != 16000) return resample_uploaded(e, true, n, sample_rate);
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  e->n_samples = n;
  e->pcm_is_float = true;
  return OK;
}

int rvb_get_waveform(rvb_engine* e, float* out, int64_t* n) {
  if (!e) { set_error("rvb_get_waveform: null engine"); return E_ARG; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  if (n) *n = e->n_samples;
  if (!out || e->n_samples == 0) return OK;
  if (e->pcm_is_float) {
    RVB_HIP_CHECK(hipMemcpy(out, e->wave_f32.p, (size_t)e->n_samples * 4, hipMemcpyDeviceToHost));
  } else {
    std::vector<int16_t> tmp(e->n_samples);
    RVB_HIP_CHECK(hipMemcpy(tmp.data(), e->pcm.p, (size_t)e->n_samples * 2, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < e->n_samples; ++i) out[i] = (float)tmp[i];
  }
  return OK;
}

int rvb_fbank(rvb_engine* e, float* feats_out, int64_t* n_frames) {
  if (!e) { set_error("rvb_fbank: null engine"); return E_ARG; }
  if (!e->fb_window.p) { set_error("rvb_fbank before rvb_finalize"); return E_STATE; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  if (e->pcm_pending) {          // rvb_upload_pcm_async: the samples that went up underneath the previous decode become the audio
    RVB_HIP_CHECK(hipStreamWaitEvent(e->stream, e->pcm_ready, 0));
    std::swap(e->pcm, e->pcm_next);
    e->n_samples = e->n_samples_next;
    e->pcm_is_float = false;
    e->pcm_pending = false;
  }
  const int64_t nf = rvb_num_frames(e->n_samples);
  const int64_t T0 = e->cfg.chunk_frames;
  // zero padded so that ANY chunking with chunk_size <= chunk_frames finds whole chunks (feats_batcher pads the last
  // chunk with zeros, cli/reverb.py:165-175): ceil(nf / c) * c < nf + c <= nf + chunk_frames
  const int64_t rows = nf + T0;
  RVB_TRY(e->feats.ensure((size_t)std::max<int64_t>(rows, 1) * 80 * 4));
  // only the padding rows: the kernel writes rows [0, nf)
  RVB_HIP_CHECK(hipMemsetAsync((char*)e->feats.p + (size_t)nf * 80 * 4, 0, (size_t)std::max<int64_t>(rows - nf, 1) * 80 * 4, e->stream));
  FbankTables t{e->fb_window.as<float>(), e->fb_twiddle.as<float>(), e->fb_melw.as<float>(), e->fb_lo.as<int>(), e->fb_hi.as<int>()};
  {
    Scope sc(e, "fbank");
    if (e->pcm_is_float) RVB_TRY(fbank_f32(e->stream, e->wave_f32.as<float>(), nf, e->feats.as<float>(), t));
    else RVB_TRY(fbank(e->stream, e->pcm.as<int16_t>(), nf, e->feats.as<float>(), t));
  }
  if (e->copy_stream) RVB_HIP_CHECK(hipEventRecord(e->pcm_free, e->stream));     // the PCM buffers are idle from here on
  if (feats_out && nf) {
    RVB_HIP_CHECK(hipMemcpyAsync(feats_out, e->feats.p, (size_t)nf * 80 * 4, hipMemcpyDeviceToHost, e->stream));
    RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  }
  // features that stay in HBM: nothing to wait for, rvb_encode is ordered behind the kernel on the engine's stream
  e->n_frames = nf; e->feat_rows = rows;
  if (n_frames) *n_frames = nf;
  return OK;
}

int rvb_encode(rvb_engine* e, const float* feats, int64_t first_chunk, const int32_t* lens, int B, int T0, int beam,
               float blank_penalty) {
  if (!e || !lens) { set_error("rvb_encode: null argument"); return E_ARG; }
  return encode_impl(e, feats, first_chunk, lens, B, T0, beam, blank_penalty);
}

int rvb_stream_begin(rvb_engine* e) {
  if (!e) { set_error("rvb_stream_begin: null engine"); return E_ARG; }
  return stream_begin_impl(e);
}
int rvb_stream_chunk(rvb_engine* e, const float* feats, int n_frames, int required_cache_size, float* out, int32_t* n_out) {
  if (!e || !feats) { set_error("rvb_stream_chunk: null argument"); return E_ARG; }
  return stream_chunk_impl(e, feats, n_frames, required_cache_size, out, n_out);
}
int rvb_stream_state(rvb_engine* e, int32_t* offset, int32_t* cache_frames) {
  if (!e) { set_error("rvb_stream_state: null engine"); return E_ARG; }
  if (offset) *offset = e->stream_st.offset;
  if (cache_frames) *cache_frames = e->stream_st.cache_len;
  return OK;
}
int rvb_stream_finish(rvb_engine* e, int beam, float blank_penalty) {
  if (!e) { set_error("rvb_stream_finish: null engine"); return E_ARG; }
  return stream_finish_impl(e, beam, blank_penalty);
}

int rvb_encoder_frames(rvb_engine* e, int32_t* T_out) {
  if (!e || !T_out || e->B <= 0) { set_error("no encoded batch"); return E_STATE; }
  *T_out = e->T2; return OK;
}
int rvb_get_encoder_lens(rvb_engine* e, int32_t* lens) {
  if (!e || !lens || e->B <= 0) { set_error("no encoded batch"); return E_STATE; }
  memcpy(lens, e->enc_lens.data(), (size_t)e->B * 4); return OK;
}
int rvb_get_encoder_out(rvb_engine* e, float* out) {
  if (!e || !out || e->B <= 0) { set_error("no encoded batch"); return E_STATE; }
  const size_t n = (size_t)e->B * e->T2 * e->cfg.d_model;
  RVB_HIP_CHECK(hipSetDevice(e->device));
  RVB_TRY(wait_slices(e, -1));
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  if (e->dtype == DT_F32) {
    RVB_HIP_CHECK(hipMemcpy(out, e->enc_out.p, n * 4, hipMemcpyDeviceToHost));
  } else {
    std::vector<bf16_t> tmp(n);
    RVB_HIP_CHECK(hipMemcpy(tmp.data(), e->enc_out.p, n * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) out[i] = bf16_to_f32(tmp[i]);
  }
  return OK;
}
int rvb_get_ctc_logprobs(rvb_engine* e, int chunk, float* out) {
  if (!e || !out || e->B <= 0 || chunk < 0 || chunk >= e->B) { set_error("rvb_get_ctc_logprobs: bad chunk"); return E_ARG; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  const int d = e->cfg.d_model, V = e->cfg.vocab, T = e->T2, Vld = (V + 3) & ~3;
  if (T > LOGIT_SLAB) { set_error("chunk too long for the logit slab"); return E_UNSUPPORTED; }
  RVB_TRY(wait_slices(e, -1));
  DevBuf lp, tv, ti;
  RVB_TRY(lp.ensure((size_t)T * V * 4)); RVB_TRY(tv.ensure((size_t)T * 4)); RVB_TRY(ti.ensure((size_t)T * 4));
  int r = run_gemm(e, (const char*)e->enc_out.p + (size_t)chunk * T * d * dt_size(e->dtype), d, e->ctc, e->logits.p, Vld, T, true);
  if (r == OK) r = logsoftmax_topk(e->stream, e->logits.as<float>(), T, V, Vld, 1, 0.f, e->cfg.blank_id, tv.as<float>(), ti.as<int>(), lp.as<float>());
  if (r == OK && hipMemcpyAsync(out, lp.p, (size_t)T * V * 4, hipMemcpyDeviceToHost, e->stream) != hipSuccess) r = E_HIP;
  (void)hipStreamSynchronize(e->stream);
  lp.release(); tv.release(); ti.release();
  return r;
}
int rvb_get_ctc_topk(rvb_engine* e, float* vals, int32_t* idx) {
  if (!e || e->B <= 0) { set_error("no encoded batch"); return E_STATE; }
  const size_t n = (size_t)e->B * e->T2 * e->beam;
  RVB_TRY(wait_slices(e, -1));
  if (vals) memcpy(vals, e->h_topv, n * 4);
  if (idx) memcpy(idx, e->h_topi, n * 4);
  return OK;
}

int rvb_ctc_greedy(rvb_engine* e, int32_t* tokens, int32_t* ntok, int32_t* frames) {
  if (!e || !tokens || !ntok || e->B <= 0) { set_error("rvb_ctc_greedy: no encoded batch"); return E_STATE; }
  const int T = e->T2, beam = e->beam;
  RVB_TRY(wait_slices(e, -1));
  std::vector<int> tk, fr;
  for (int b = 0; b < e->B; ++b) {
    greedy_collapse(e->h_topi + (size_t)b * T * beam, e->enc_lens[b], beam, e->cfg.blank_id, &tk, &fr);
    ntok[b] = (int32_t)tk.size();
    for (size_t i = 0; i < tk.size(); ++i) { tokens[(size_t)b * T + i] = tk[i]; if (frames) frames[(size_t)b * T + i] = fr[i]; }
    for (size_t i = tk.size(); i < (size_t)T; ++i) { tokens[(size_t)b * T + i] = -1; if (frames) frames[(size_t)b * T + i] = -1; }
  }
  return OK;
}

int rvb_ctc_prefix_beam(rvb_engine* e, int beam) {
  if (!e) { set_error("null engine"); return E_ARG; }
  return prefix_beam_impl(e, beam);
}
int rvb_get_nbest_count(rvb_engine* e, int chunk, int32_t* n_hyps, int32_t* max_len) {
  if (!e || chunk < 0 || chunk >= (int)e->nbest.size()) { set_error("rvb_get_nbest_count: bad chunk / no search results"); return E_STATE; }
  const PrefixResult& pr = e->nbest[chunk];
  int ml = 0;
  for (auto& h : pr.nbest) ml = std::max(ml, (int)h.size());
  for (auto& t : pr.times) ml = std::max(ml, (int)t.size());
  if (n_hyps) *n_hyps = (int32_t)pr.nbest.size();
  if (max_len) *max_len = ml;
  return OK;
}
static void fill_nbest(const PrefixResult& pr, int ml, int32_t* tokens, int32_t* lens, int32_t* times, int32_t* times_lens, double* scores) {
  for (size_t i = 0; i < pr.nbest.size(); ++i) {
    if (lens) lens[i] = (int32_t)pr.nbest[i].size();
    if (times_lens) times_lens[i] = (int32_t)pr.times[i].size();
    if (scores) scores[i] = pr.scores[i];
    for (int j = 0; j < ml; ++j) {
      if (tokens) tokens[i * ml + j] = j < (int)pr.nbest[i].size() ? pr.nbest[i][j] : -1;
      if (times) times[i * ml + j] = j < (int)pr.times[i].size() ? pr.times[i][j] : -1;
    }
  }
}
int rvb_get_nbest(rvb_engine* e, int chunk, int32_t* tokens, int32_t* lens, int32_t* times, int32_t* times_lens, double* scores) {
  int32_t n, ml;
  int r = rvb_get_nbest_count(e, chunk, &n, &ml);
  if (r != OK) return r;
  fill_nbest(e->nbest[chunk], ml, tokens, lens, times, times_lens, scores);
  return OK;
}

int rvb_prepare_rescoring(rvb_engine* e, int right_to_left) {
  if (!e) { set_error("rvb_prepare_rescoring: null engine"); return E_ARG; }
  if (e->B <= 0) { set_error("rvb_prepare_rescoring before rvb_encode"); return E_STATE; }
  if (!e->dec_l.present) { set_error("model has no attention decoder"); return E_STATE; }
  if (right_to_left && !e->dec_r.present) { set_error("rvb_prepare_rescoring: model has no right-to-left decoder"); return E_STATE; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  // enqueued behind the encoder slices still in flight; nothing here waits for the device
  if (!e->dec_l.kv_ready) RVB_TRY(decoder_memory_kv(e, e->dec_l, e->B * e->T2));
  if (right_to_left && !e->dec_r.kv_ready) RVB_TRY(decoder_memory_kv(e, e->dec_r, e->B * e->T2));
  return OK;
}
int rvb_attention_rescore(rvb_engine* e, double ctc_weight, double reverse_weight) {
  if (!e) { set_error("null engine"); return E_ARG; }
  return rescore_impl(e, ctc_weight, reverse_weight);
}
int rvb_attention_decode(rvb_engine* e, int beam, float length_penalty) {
  if (!e) { set_error("rvb_attention_decode: null engine"); return E_ARG; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  return attention_decode_impl(e, beam, length_penalty);
}
int rvb_joint_decode(rvb_engine* e, int beam, double ctc_weight, double pre_beam_ratio, double length_bonus) {
  if (!e) { set_error("rvb_joint_decode: null engine"); return E_ARG; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  return joint_decode_impl(e, beam, ctc_weight, pre_beam_ratio, length_bonus);
}
int rvb_get_joint_result(rvb_engine* e, int chunk, int32_t* tokens, int32_t* times, int32_t* end_times, double* tokens_confidence,
                         int32_t* n_tokens, double* score) {
  if (!e || !n_tokens) { set_error("rvb_get_joint_result: null argument"); return E_ARG; }
  if (chunk < 0 || chunk >= (int)e->joint.size()) { set_error("rvb_get_joint_result: no joint_decoding result for this chunk"); return E_STATE; }
  const JointResult& r = e->joint[chunk];
  for (size_t i = 0; i < r.tokens.size(); ++i) {      // capacity: rvb_encoder_frames() entries each
    if (tokens) tokens[i] = r.tokens[i];
    if (times) times[i] = r.times[i];
    if (end_times) end_times[i] = r.end_times[i];
    if (tokens_confidence) tokens_confidence[i] = r.tokens_confidence[i];
  }
  *n_tokens = (int32_t)r.tokens.size();
  if (score) *score = r.score;
  return OK;
}
int rvb_get_joint_stats(rvb_engine* e, int64_t* decoder_rows, int64_t* steps) {
  if (!e) { set_error("rvb_get_joint_stats: null engine"); return E_ARG; }
  if (decoder_rows) *decoder_rows = e->joint_rows;
  if (steps) *steps = e->joint_steps;
  return OK;
}
int rvb_get_attention_result(rvb_engine* e, int chunk, int32_t* tokens, int32_t* n_tokens, float* score) {
  if (!e || !n_tokens) { set_error("rvb_get_attention_result: null argument"); return E_ARG; }
  if (chunk < 0 || chunk >= (int)e->attn_tokens.size()) { set_error("rvb_get_attention_result: no attention result for this chunk"); return E_STATE; }
  const std::vector<int>& t = e->attn_tokens[chunk];
  if (tokens) for (size_t i = 0; i < t.size(); ++i) tokens[i] = t[i];     // capacity: rvb_encoder_frames() entries
  *n_tokens = (int32_t)t.size();
  if (score) *score = e->attn_scores[chunk];
  return OK;
}

int rvb_get_rescored(rvb_engine* e, int chunk, int32_t* best_index, float* score, double* confidence, double* tokens_confidence) {
  if (!e || chunk < 0 || chunk >= (int)e->rescored.size()) { set_error("rvb_get_rescored: bad chunk / no rescoring results"); return E_STATE; }
  const RescoreResult& r = e->rescored[chunk];
  if (best_index) *best_index = r.best;
  if (score) *score = r.score;
  if (confidence) *confidence = r.confidence;
  if (tokens_confidence) for (size_t i = 0; i < r.tok_conf.size(); ++i) tokens_confidence[i] = r.tok_conf[i];
  return OK;
}
int rvb_get_rescored_batch(rvb_engine* e, int32_t* lens, int32_t* tokens, int32_t* times_lens, int32_t* times,
                           float* scores, double* confidences, double* tokens_confidence) {
  if (!e || (int)e->rescored.size() != e->B || e->B <= 0 || !lens || !tokens) { set_error("rvb_get_rescored_batch: no rescoring results"); return E_STATE; }
  const int T = e->T2;
  for (int b = 0; b < e->B; ++b) {
    const RescoreResult& r = e->rescored[b];
    const PrefixResult& pr = e->nbest[b];
    const std::vector<int>& tk = pr.nbest[r.best];
    const std::vector<int>& tm = pr.times[r.best];
    lens[b] = (int32_t)tk.size();
    if (times_lens) times_lens[b] = (int32_t)tm.size();
    if (scores) scores[b] = r.score;
    if (confidences) confidences[b] = r.confidence;
    for (int j = 0; j < T; ++j) {
      tokens[(size_t)b * T + j] = j < (int)tk.size() ? tk[j] : -1;
      if (times) times[(size_t)b * T + j] = j < (int)tm.size() ? tm[j] : -1;
      if (tokens_confidence) tokens_confidence[(size_t)b * T + j] = j < (int)r.tok_conf.size() ? r.tok_conf[j] : 0.0;
    }
  }
  return OK;
}
int rvb_fp8_recalibrate(rvb_engine* e) {
  if (!e) { set_error("rvb_fp8_recalibrate: null engine"); return E_ARG; }
  if (!e->fp8) { set_error("rvb_fp8_recalibrate: not an fp8 engine"); return E_STATE; }
  e->f8_state = 0;
  return OK;
}
int rvb_get_fp8_scales(rvb_engine* e, float* scales, int32_t* n) {
  if (!e || !n) { set_error("rvb_get_fp8_scales: null argument"); return E_ARG; }
  if (!e->fp8) { set_error("rvb_get_fp8_scales: not an RVB_FP8 engine"); return E_STATE; }
  if (e->f8_state != 2) { *n = 0; return OK; }                       // not calibrated yet
  *n = (int32_t)(e->f8.size() * 7 + 1);
  if (scales) {
    for (size_t l = 0; l < e->f8.size(); ++l) {
      const F8Scales& f = e->f8[l];
      const float v[7] = {f.in_ffm1, f.h_ffm, f.in_qkv, f.in_pw1, f.in_pw2, f.in_ff1, f.h_ff};
      for (int k = 0; k < 7; ++k) scales[l * 7 + k] = v[k];
    }
    scales[e->f8.size() * 7] = e->f8_x1;      // conv1's fp8 output (policy bit 5); 0 = not measured: conv2 stays bf16
  }
  return OK;
}
// Values that did not fit e4m3 at the installed scale (clipped to +-448 by the kernels that write the fp8 operands), per block
// and activation slot in the order of rvb_get_fp8_scales, summed since the scales were calibrated / installed / last reset:
// the fp8 mode's answer to "the calibration batch was not representative" -- until round 4 such clipping was silent.
int rvb_get_fp8_saturation(rvb_engine* e, uint32_t* counts, int32_t* n, int reset) {
  if (!e || !n) { set_error("rvb_get_fp8_saturation: null argument"); return E_ARG; }
  if (!e->fp8) { set_error("rvb_get_fp8_saturation: not an RVB_FP8 engine"); return E_STATE; }
  *n = (int32_t)(e->enc.size() * 7);
  if (!e->d_f8sat.p) { if (counts) memset(counts, 0, (size_t)*n * 4); return OK; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  std::vector<uint32_t> raw(e->enc.size() * 8);
  RVB_HIP_CHECK(hipMemcpyAsync(raw.data(), e->d_f8sat.p, raw.size() * 4, hipMemcpyDeviceToHost, e->stream));
  if (reset) RVB_HIP_CHECK(hipMemsetAsync(e->d_f8sat.p, 0, (e->enc.size() + 1) * 8 * 4, e->stream));
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  if (counts)
    for (size_t l = 0; l < e->enc.size(); ++l)
      for (int k = 0; k < 7; ++k) counts[l * 7 + k] = raw[l * 8 + k];
  return OK;
}
int rvb_get_fp8_subsample(rvb_engine* e, float* scale, uint32_t* clipped, int reset) {
  if (!e) { set_error("rvb_get_fp8_subsample: null engine"); return E_ARG; }
  if (!e->fp8) { set_error("rvb_get_fp8_subsample: not an RVB_FP8 engine"); return E_STATE; }
  if (scale) *scale = e->f8_state == 2 ? e->f8_x1 : 0.f;
  if (clipped) *clipped = 0;
  if (e->d_f8sat.p && (clipped || reset)) {
    unsigned* slot = e->d_f8sat.as<unsigned>() + e->enc.size() * 8 + 1;
    uint32_t v = 0;
    RVB_HIP_CHECK(hipMemcpyAsync(&v, slot, 4, hipMemcpyDeviceToHost, e->stream));
    if (reset) RVB_HIP_CHECK(hipMemsetAsync(slot, 0, 4, e->stream));
    RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
    if (clipped) *clipped = v;
  }
  return OK;
}
int rvb_set_fp8_scales(rvb_engine* e, const float* scales, int32_t n) {
  if (!e || !scales) { set_error("rvb_set_fp8_scales: null argument"); return E_ARG; }
  if (!e->fp8) { set_error("rvb_set_fp8_scales: not an RVB_FP8 engine"); return E_STATE; }
  const int32_t nb7 = (int32_t)(e->enc.size() * 7);
  if (!e->finalized || (n != nb7 && n != nb7 + 1)) {
    set_error("rvb_set_fp8_scales: need 7 scales per conformer block of a finalized engine (+ optionally the subsampling scale, as rvb_get_fp8_scales returns them)");
    return E_ARG;
  }
  for (int i = 0; i < nb7; ++i) if (!(scales[i] > 0.f) || !std::isfinite(scales[i])) { set_error("rvb_set_fp8_scales: scales must be positive and finite"); return E_ARG; }
  if (n == nb7 + 1) {     // the vector of rvb_get_fp8_scales: its last entry is conv1's output scale (0 = none -> conv2 in bf16)
    if (!(scales[nb7] >= 0.f) || !std::isfinite(scales[nb7])) { set_error("rvb_set_fp8_scales: the subsampling scale must be >= 0 and finite"); return E_ARG; }
    e->f8_x1 = scales[nb7];
  }
  e->f8.resize(e->enc.size());
  for (size_t l = 0; l < e->enc.size(); ++l) {
    const float* v = scales + l * 7;
    e->f8[l] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6]};
  }
  if (e->f8_groups.size() != e->enc.size()) RVB_TRY(set_fp8_policy_impl(e, -1, 0, -1));
  e->f8_state = 2;
  RVB_HIP_CHECK(hipSetDevice(e->device));
  RVB_TRY(reset_f8sat(e));
  return OK;
}
int rvb_get_rescore_stats(rvb_engine* e, int64_t* decoder_rows, int64_t* pairs) {
  if (!e) { set_error("rvb_get_rescore_stats: null engine"); return E_ARG; }
  if (decoder_rows) *decoder_rows = e->rescore_rows;
  if (pairs) *pairs = e->rescore_pairs;
  return OK;
}
int rvb_get_rescore_logp(rvb_engine* e, int chunk, int hyp, int right, float* out) {
  if (!e || !out || chunk < 0 || chunk >= (int)e->rescored.size()) { set_error("rvb_get_rescore_logp: bad chunk"); return E_STATE; }
  const RescoreResult& r = e->rescored[chunk];
  const auto& v = right ? r.rlogp : r.logp;
  if (hyp < 0 || hyp >= (int)v.size()) { set_error("rvb_get_rescore_logp: bad hyp"); return E_ARG; }
  for (size_t i = 0; i < v[hyp].size(); ++i) out[i] = v[hyp][i];
  return OK;
}

int rvb_set_profiling(rvb_engine* e, int level) { if (!e) return E_ARG; drain_prof(e); e->profiling = (level == 1 || level == 2) ? level : 0; return OK; }
int rvb_reset_timings(rvb_engine* e) { if (!e) return E_ARG; drain_prof(e); e->prof.clear(); return OK; }
int rvb_get_timing(rvb_engine* e, const char* name, double* ms, double* flops, int64_t* launches) {
  if (!e || !name) return E_ARG;
  drain_prof(e);
  auto it = e->prof.find(name);
  ProfEntry pe; if (it != e->prof.end()) pe = it->second;
  if (ms) *ms = pe.ms; if (flops) *flops = pe.flops; if (launches) *launches = pe.launches;
  return OK;
}
int rvb_get_timing_bytes(rvb_engine* e, const char* name, double* bytes) {
  if (!e || !name || !bytes) return E_ARG;
  auto it = e->prof.find(name);
  *bytes = it != e->prof.end() ? it->second.bytes : 0.0;
  return OK;
}

}  // extern "C"

#ifdef RVB_TEST_API      // librvb_test.so only (csrc/test_api.h): hooks into this file's static functions
#include "test_api.h"
// host only: the rescoring trie of given hypotheses (tests/test_search_native.py checks it against a Python trie).
// tokens: the hypotheses back to back; lens / chunk_of: per hypothesis (chunk ids ascending).  Outputs sized by the caller:
// rows <= P = sum(len + 1); tok, pos [rows]; path, tgt, pair_slot [P]; hq_start, hq_len, hq_pos0 [n_hyps]; tgt_ptr [rows + 1].
extern "C" int rvb_test_build_trie(const int32_t* tokens, const int32_t* lens, const int32_t* chunk_of, int n_hyps, int n_chunks, int sos,
                                   int eos, int reversed, int32_t* n_rows, int32_t* tok, int32_t* pos, int32_t* path, int32_t* hq_start,
                                   int32_t* hq_len, int32_t* hq_pos0, int32_t* tgt_ptr, int32_t* tgt, int32_t* pair_slot, int32_t* n_work) {
  if (!tokens || !lens || !chunk_of || !n_rows || n_hyps < 0) { set_error("rvb_test_build_trie: bad argument"); return E_ARG; }
  std::vector<HypRef> hyps;
  std::vector<int> first(n_hyps);
  int P = 0, off = 0;
  for (int i = 0; i < n_hyps; ++i) { hyps.push_back({chunk_of[i], i, lens[i], P}); first[i] = off; P += lens[i] + 1; off += lens[i]; }
  TrieBatch t;
  auto seq = [&](const HypRef& h, int j) { return tokens[first[h.idx] + (reversed ? h.len - 1 - j : j)]; };
  build_trie_range(hyps.data(), hyps.data() + hyps.size(), 0, n_chunks, sos, eos, seq, &t);
  {   // the engine builds the same trie chunk by chunk and stitches the parts (merge_tries): both forms must agree exactly
    std::vector<TrieBatch> part(std::max(n_chunks, 0));
    size_t a = 0;
    for (int b = 0; b < n_chunks; ++b) {
      size_t z = a;
      while (z < hyps.size() && hyps[z].chunk == b) ++z;
      build_trie_range(hyps.data() + a, hyps.data() + z, b, 1, sos, eos, seq, &part[b]);
      a = z;
    }
    TrieBatch m;
    merge_tries(part, &m);
    const bool same = m.R == t.R && m.P == t.P && m.max_chunk_rows == t.max_chunk_rows && m.tok == t.tok && m.pos == t.pos &&
                      m.path == t.path && m.hq_start == t.hq_start && m.hq_len == t.hq_len && m.hq_pos0 == t.hq_pos0 &&
                      m.hkv_start == t.hkv_start && m.hkv_len == t.hkv_len && m.crow_start == t.crow_start &&
                      m.crow_len == t.crow_len && m.tgt_ptr == t.tgt_ptr && m.tgt == t.tgt && m.pair_slot == t.pair_slot &&
                      m.work == t.work;
    if (a != hyps.size() || !same) {
      std::string which;
#define RVB_DIFF(f) if (!(m.f == t.f)) which += std::string(" ") + #f;
      RVB_DIFF(R) RVB_DIFF(P) RVB_DIFF(max_chunk_rows) RVB_DIFF(tok) RVB_DIFF(pos) RVB_DIFF(path) RVB_DIFF(hq_start) RVB_DIFF(hq_len)
      RVB_DIFF(hq_pos0) RVB_DIFF(hkv_start) RVB_DIFF(hkv_len) RVB_DIFF(crow_start) RVB_DIFF(crow_len) RVB_DIFF(tgt_ptr) RVB_DIFF(tgt)
      RVB_DIFF(pair_slot) RVB_DIFF(work)
#undef RVB_DIFF
      set_error("rvb_test_build_trie: the stitched per-chunk tries differ from the batch trie in:" + which);
      return E_STATE;
    }
  }
  *n_rows = t.R;
  if (n_work) *n_work = (int32_t)t.work.size() / 2;
  auto cp = [](int32_t* dst, const std::vector<int32_t>& v) { if (dst) memcpy(dst, v.data(), v.size() * 4); };
  cp(tok, t.tok); cp(pos, t.pos); cp(path, t.path); cp(hq_start, t.hq_start); cp(hq_len, t.hq_len); cp(hq_pos0, t.hq_pos0);
  cp(tgt_ptr, t.tgt_ptr); cp(tgt, t.tgt); cp(pair_slot, t.pair_slot);
  return OK;
}

// host only: HostPool runs `rounds` jobs of `n_threads` threads each; every job hands out `items` work items through an atomic
// counter (the pattern of the CTC search) and the call checks that each item was executed exactly once in every round.
extern "C" int rvb_test_host_pool(int n_threads, int items, int rounds) {
  if (n_threads < 1 || items < 0 || rounds < 1) { set_error("rvb_test_host_pool: bad argument"); return E_ARG; }
  HostPool pool;
  std::vector<std::atomic<int>> hits(items);
  for (int r = 0; r < rounds; ++r) {
    for (auto& h : hits) h.store(0);
    std::atomic<int> next(0), entered(0);
    const unsigned n = (unsigned)std::max(1, n_threads - (r % 3));     // the pool grows and is reused with fewer threads
    pool.run(n, [&] {
      entered.fetch_add(1);
      for (int i = next.fetch_add(1); i < items; i = next.fetch_add(1)) hits[i].fetch_add(1);
    });
    if (entered.load() != (int)n) { set_error("rvb_test_host_pool: a job was not run by the requested number of threads"); return E_STATE; }
    for (int i = 0; i < items; ++i)
      if (hits[i].load() != 1) { set_error("rvb_test_host_pool: work item executed " + std::to_string(hits[i].load()) + " times"); return E_STATE; }
  }
  return OK;
}

extern "C" int rvb_test_fbank(const int16_t* pcm, int64_t n_samples, float* feats) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available: librvb has no CPU fallback"); return E_HIP; }
  if (!pcm || !feats) { set_error("rvb_test_fbank: null argument"); return E_ARG; }
  rvb_engine e;   // default stream, only the fbank tables are used
  RVB_TRY(make_fbank_tables(&e));
  const int64_t nf = rvb_num_frames(n_samples);
  DevBuf dp, df;
  RVB_TRY(dp.ensure((size_t)n_samples * 2 + 16));
  RVB_TRY(df.ensure((size_t)std::max<int64_t>(nf, 1) * 80 * 4));
  RVB_HIP_CHECK(hipMemcpy(dp.p, pcm, (size_t)n_samples * 2, hipMemcpyHostToDevice));
  FbankTables t{e.fb_window.as<float>(), e.fb_twiddle.as<float>(), e.fb_melw.as<float>(), e.fb_lo.as<int>(), e.fb_hi.as<int>()};
  int r = fbank(nullptr, dp.as<int16_t>(), nf, df.as<float>(), t);
  if (r == OK && hipDeviceSynchronize() != hipSuccess) { set_error("fbank kernel failed"); r = E_HIP; }
  if (r == OK && nf && hipMemcpy(feats, df.p, (size_t)nf * 80 * 4, hipMemcpyDeviceToHost) != hipSuccess) r = E_HIP;
  dp.release(); df.release();
  for (DevBuf* b : {&e.fb_window, &e.fb_twiddle, &e.fb_melw, &e.fb_lo, &e.fb_hi}) b->release();
  return r;
}
#endif   // RVB_TEST_API
