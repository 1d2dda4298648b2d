// rvd_* : diarization engine (segmentation + embedding networks of the pyannote pipeline the reference
// runs in diarization/infer_pyannote3.0.py:33-42) on one MI355X.  Kernels: diar.hip, resnet.hip, gemm*.hip.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/rvd.h"
#include "engine.h"

using namespace rvb;

#define RVD_TRY(expr) do { int _r = (expr); if (_r != OK) return _r; } while (0)

namespace {
struct LstmLayer { Linear ih; DevBuf whh; int in_pad = 0; };
constexpr int SINC_K = 251, SINC_STRIDE = 10, CONV_K = 5;
// ResNet34 trunk: conv weights packed [tap][Cin/CK][Cout][CK] with BatchNorm folded in
struct ConvW { DevBuf w, b, w_ig; int cin = 0, cout = 0, taps = 9, stride = 1;      // w_ig: conv_gemm.hip's layout (optional)
               DevBuf w_ig_sc, b_sc;
               DevBuf w8, w8s; };            // RVD_EMB_FP8=1: e4m3 copy [cout][9][cin] + per-output-channel scales (conv_igemm8_kernel)      // second convolution of a block with a projection shortcut: w_ig rows + the shortcut's, summed biases
struct ResBlock { ConvW c1, c2, sc; bool has_sc = false; };
constexpr int FB_WIN = 400, FB_SHIFT = 160, FB_MEL = 80;
// windows per trunk pass; RVD_EMB_BATCH overrides (tuning).  768 since round 4: 1 h of audio 385-392 ms at 192, 381 at 384,
// 373 at 768 (profiles/r04_call17_persistent_batch.txt) -- fewer, longer launches; 27 GB of activations out of 288
static int emb_batch() {
  static int v = [] { const char* e = getenv("RVD_EMB_BATCH"); const int x = e ? atoi(e) : 0; return x > 0 ? x : 768; }();
  return v;
}
// windows per trunk pass (activations 35 MB per window in bf16); per engine since round 5: halved when the workspace does not fit
#define EMB_BATCH (e->emb_batch > 0 ? e->emb_batch : (e->emb_batch = emb_batch()))
}  // namespace

struct rvd_engine {
  rvd_model_cfg cfg;
  int device = 0, dtype = 0;
  int linkage_workgroups = 0;          // rvd_set_linkage_workgroups: hint for rvd_centroid_linkage's merge loop
  hipStream_t stream = nullptr;
  bool finalized = false;
  std::map<std::string, HostTensor> host;

  // derived frame counts of one window
  int f1 = 0, p1 = 0, f2 = 0, p2 = 0, f3 = 0, p3 = 0, cpad = 0;

  // segmentation weights
  DevBuf filt, fsum;
  float wn_gamma = 1.f, wn_beta = 0.f;
  LNorm norm[3];
  Linear conv2, conv3;
  std::vector<LstmLayer> lstm;
  std::vector<Linear> lin;
  DevBuf cls_w, cls_b;
  DevBuf stage;

  // audio
  DevBuf pcm, wave, craw;
  int64_t n_samples = 0, n_pad = 0, n_windows = 0, craw_frames = 0;

  // workspace of rvd_segment
  DevBuf stats, a1, c2, a2, c3, a3, xproj, hA, hB, l0, l1, logp, cls;
  int last_W = 0;
  const void* last_lstm = nullptr;

  // embedding model
  bool has_emb = false;
  DevBuf stem_w, stem_b;
  std::vector<std::vector<ResBlock>> stages;
  Linear seg1;
  DevBuf fb_window, fb_twiddle, fb_melw, fb_lo, fb_hi;
  DevBuf pcm_pad, emb_fb;                 // int16 [n_pad], fp32 [emb_frames][80] hamming log-mel of the whole file
  int64_t emb_frames = 0;
  int nfr = 0;                            // fbank frames per window (998)
  DevBuf act[4][4];                       // per stage: three rotating activation buffers + the shortcut
  // RVD_EMB_FP8=1 (round 4 candidate, not yet run on a GPU): stages 3-4 of the trunk on the fp8 implicit-GEMM kernel.  act8 = e4m3
  // copies of those stages' rotating buffers; scale8[((li - 2) * 8 + block) * 2 + {0: first convolution's output, 1: block output}],
  // calibrated by the first trunk pass (bf16, running maxima in d_amax8); emb_f8_state 0 not calibrated, 1 calibrating, 2 active
  int emb_batch = 0;                     // windows per trunk pass (0 = not chosen yet: RVD_EMB_BATCH or 768)
  bool emb_fp8 = false, want_fp8 = false;      // want_fp8: created with dtype RVB_FP8
  int emb_f8_state = 0;
  DevBuf act8[4][3], d_amax8, d_sat8;
  std::vector<float> scale8;
  int act_cap = 0;
  DevBuf e_win, e_mean, e_item_b, e_mask, e_stats, e_out;

  // profiling
  bool profiling = false;
  std::map<std::string, ProfEntry> prof;
  struct Pending { hipEvent_t a, b; std::string name; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;
};

namespace {

struct DScope {
  rvd_engine* e; hipEvent_t a = nullptr, b = nullptr; std::string name;
  DScope(rvd_engine* e_, const char* n, double flops = 0.0) : e(e_), name(n) {
    auto& pe = e->prof[name];
    pe.launches += 1; pe.flops += flops;
    if (!e->profiling) return;
    auto get = [&]() { hipEvent_t ev; if (!e->event_pool.empty()) { ev = e->event_pool.back(); e->event_pool.pop_back(); } else (void)hipEventCreate(&ev); return ev; };
    a = get(); b = get();
    (void)hipEventRecord(a, e->stream);
  }
  ~DScope() { if (a) { (void)hipEventRecord(b, e->stream); e->pending.push_back({a, b, name}); } }
};
void drain(rvd_engine* e) {
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

int up_f32(rvd_engine* e, DevBuf& dst, const float* src, size_t n) {
  RVD_TRY(dst.ensure(n * 4));
  RVB_HIP_CHECK(hipMemcpyAsync(dst.p, src, n * 4, hipMemcpyHostToDevice, e->stream));
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));   // src may be a temporary
  return OK;
}
int pack_T(rvd_engine* e, DevBuf& dst, const float* src, size_t n) {
  RVD_TRY(dst.ensure(n * dt_size(e->dtype)));
  if (e->dtype == DT_F32) return up_f32(e, dst, src, n);
  RVD_TRY(up_f32(e, e->stage, src, n));
  RVD_TRY(convert_f32(e->stream, e->dtype, e->stage.as<float>(), dst.p, n));
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  return OK;
}
int need(rvd_engine* e, const std::string& name, size_t numel, const HostTensor** out) {
  auto it = e->host.find(name);
  if (it == e->host.end()) { set_error("missing tensor: " + name); return E_STATE; }
  if (it->second.numel() != numel) {
    set_error("tensor " + name + " has " + std::to_string(it->second.numel()) + " elements, expected " + std::to_string(numel));
    return E_ARG;
  }
  *out = &it->second;
  return OK;
}
int pack_norm(rvd_engine* e, LNorm& n, const std::string& p, int d) {
  const HostTensor *g, *b;
  RVD_TRY(need(e, p + ".weight", d, &g));
  RVD_TRY(need(e, p + ".bias", d, &b));
  n.eps = 1e-5f;
  RVD_TRY(up_f32(e, n.g, g->data.data(), d));
  return up_f32(e, n.b, b->data.data(), d);
}

// asteroid_filterbanks ParamSincFB.filters(): 40 cos + 40 sin band-pass filters from (low_hz_, band_hz_)
void sinc_filters(const float* low_hz_, const float* band_hz_, int npair, int sr, std::vector<float>* out, std::vector<float>* fsum) {
  const int K = SINC_K, half = K / 2;
  out->assign((size_t)2 * npair * K, 0.f);
  fsum->assign((size_t)2 * npair, 0.f);
  const double min_low = 50.0, min_band = 50.0;
  for (int f = 0; f < npair; ++f) {
    const float low = (float)min_low + std::fabs(low_hz_[f]);
    float high = low + (float)min_band + std::fabs(band_hz_[f]);
    high = std::fmin(std::fmax(high, (float)min_low), (float)sr / 2.f);
    const float band = high - low;
    float* fc = out->data() + (size_t)f * K;
    float* fs = out->data() + (size_t)(npair + f) * K;
    for (int i = 0; i < half; ++i) {
      const float win = (float)(0.54 - 0.46 * std::cos(2.0 * M_PI * i / (K - 1)));
      const float n = (float)(2.0 * M_PI) * ((float)(i - half) / (float)sr);
      const float fl = low * n, fh = high * n;
      const float lc = ((std::sin(fh) - std::sin(fl)) / (n / 2.f)) * win;
      const float ls = ((std::cos(fl) - std::cos(fh)) / (n / 2.f)) * win;
      fc[i] = lc / (2.f * band); fc[K - 1 - i] = lc / (2.f * band);
      fs[i] = ls / (2.f * band); fs[K - 1 - i] = -ls / (2.f * band);
    }
    fc[half] = (2.f * band) / (2.f * band);
    fs[half] = 0.f;
  }
  for (int f = 0; f < 2 * npair; ++f) {
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += (*out)[(size_t)f * K + k];
    (*fsum)[f] = (float)s;
  }
}

// Conv1d weight [out][in][k] -> GEMM operand [out_pad][k][in_pad] (row m of the activation matrix is frame t,
// its K = k*in_pad columns run on into frames t+1..t+4, which are contiguous in memory)
int pack_conv1d(rvd_engine* e, Linear& L, const std::string& p, int out, int in, int out_pad, int in_pad) {
  const HostTensor *w, *b;
  RVD_TRY(need(e, p + ".weight", (size_t)out * in * CONV_K, &w));
  RVD_TRY(need(e, p + ".bias", out, &b));
  std::vector<float> pw((size_t)out_pad * CONV_K * in_pad, 0.f), pb(out_pad, 0.f);
  for (int o = 0; o < out; ++o) {
    pb[o] = b->data[o];
    for (int c = 0; c < in; ++c)
      for (int k = 0; k < CONV_K; ++k) pw[((size_t)o * CONV_K + k) * in_pad + c] = w->data[((size_t)o * in + c) * CONV_K + k];
  }
  L.out = out_pad; L.in = CONV_K * in_pad;
  RVD_TRY(pack_T(e, L.w, pw.data(), pw.size()));
  return up_f32(e, L.b, pb.data(), pb.size());
}

int run_gemm(rvd_engine* e, const char* name, const void* A, int lda, const Linear& L, void* C, int ldc, int64_t M, int act) {
  DScope sc(e, name, 2.0 * (double)M * L.out * L.in);
  GemmArgs g{};
  g.A = A; g.W = L.w.p; g.bias = L.b.p ? L.b.as<float>() : nullptr; g.res = nullptr; g.C = C;
  g.M = (int)M; g.N = L.out; g.K = L.in; g.lda = lda; g.ldw = L.in; g.ldc = ldc; g.ldres = 0;
  g.alpha = 1.f; g.act = act; g.out_f32 = 0; g.conv = 0;
  return gemm(e->stream, e->dtype, g);
}

// conv layers 2 / 3 of SincNet: the thin-GEMM kernel of diar.hip (bf16; lab: RVD_CONV1D5=0 = the generic GEMM)
int run_sincnet_conv(rvd_engine* e, const void* A, int cin, const Linear& L, void* C, int ldc, int64_t M) {
  const int on = lab_env("RVD_CONV1D5") ? atoi(lab_env("RVD_CONV1D5")) : 1;      // read per call: the A/B test flips it
  if (on && e->dtype == DT_BF16 && (cin == 80 || cin == 64) && L.out == 64 && ldc == 64 && L.in == 5 * cin && L.b.p) {
    DScope sc(e, "sincnet_conv", 2.0 * (double)M * L.out * L.in);
    return conv1d5(e->stream, e->dtype, A, cin, L.w.p, L.b.as<float>(), C, M);
  }
  return run_gemm(e, "sincnet_conv", A, cin, L, C, ldc, M, ACT_NONE);
}


int finalize_embedding(rvd_engine* e);

int finalize_impl(rvd_engine* e) {
  const rvd_model_cfg& c = e->cfg;
  const std::string S = "segmentation.";
  const HostTensor *t1, *t2;
  // --- SincNet ---
  RVD_TRY(need(e, S + "sincnet.wav_norm1d.weight", 1, &t1));
  RVD_TRY(need(e, S + "sincnet.wav_norm1d.bias", 1, &t2));
  e->wn_gamma = t1->data[0]; e->wn_beta = t2->data[0];
  const int npair = c.sinc_filters / 2;
  RVD_TRY(need(e, S + "sincnet.conv1d.0.filterbank.low_hz_", npair, &t1));
  RVD_TRY(need(e, S + "sincnet.conv1d.0.filterbank.band_hz_", npair, &t2));
  std::vector<float> filt, fsum;
  sinc_filters(t1->data.data(), t2->data.data(), npair, c.sample_rate, &filt, &fsum);
  RVD_TRY(up_f32(e, e->filt, filt.data(), filt.size()));
  RVD_TRY(up_f32(e, e->fsum, fsum.data(), fsum.size()));
  RVD_TRY(pack_norm(e, e->norm[0], S + "sincnet.norm1d.0", c.sinc_filters));
  RVD_TRY(pack_norm(e, e->norm[1], S + "sincnet.norm1d.1", c.sinc_channels));
  RVD_TRY(pack_norm(e, e->norm[2], S + "sincnet.norm1d.2", c.sinc_channels));
  RVD_TRY(pack_conv1d(e, e->conv2, S + "sincnet.conv1d.1", c.sinc_channels, c.sinc_filters, e->cpad, c.sinc_filters));
  RVD_TRY(pack_conv1d(e, e->conv3, S + "sincnet.conv1d.2", c.sinc_channels, c.sinc_channels, e->cpad, e->cpad));
  // --- LSTM: both directions' input projections as one [8H][in_pad] operand, bias = b_ih + b_hh ---
  const int H = c.lstm_hidden;
  e->lstm.resize(c.lstm_layers);
  for (int l = 0; l < c.lstm_layers; ++l) {
    const int in = l == 0 ? c.sinc_channels : 2 * H;
    const int in_pad = l == 0 ? e->cpad : 2 * H;
    std::vector<float> wih((size_t)8 * H * in_pad, 0.f), bias((size_t)8 * H, 0.f), whh((size_t)8 * H * H);
    for (int d = 0; d < 2; ++d) {
      const std::string suf = "_l" + std::to_string(l) + (d ? "_reverse" : "");
      const HostTensor *wi, *wh, *bi, *bh;
      RVD_TRY(need(e, S + "lstm.weight_ih" + suf, (size_t)4 * H * in, &wi));
      RVD_TRY(need(e, S + "lstm.weight_hh" + suf, (size_t)4 * H * H, &wh));
      RVD_TRY(need(e, S + "lstm.bias_ih" + suf, (size_t)4 * H, &bi));
      RVD_TRY(need(e, S + "lstm.bias_hh" + suf, (size_t)4 * H, &bh));
      // The projection's output columns are laid out for the recurrence kernel (diar.hip lstm_kernel): the eight pre-activations
      // one of its lanes starts a step from -- gates i, f, g, o of hidden units u and u + 16 -- are consecutive (one 16-byte load
      // per window and step instead of eight 2-byte ones): gate q of unit 32 v + 16 hf + c -> column 128 v + 8 c + 2 q + hf.
      for (int r = 0; r < 4 * H; ++r) {
        const int q = r / H, u = r % H, v = u / 32, hf = (u % 32) / 16, cc = u % 16;
        const int col = 128 * v + 8 * cc + 2 * q + hf;
        std::memcpy(&wih[((size_t)d * 4 * H + col) * in_pad], &wi->data[(size_t)r * in], (size_t)in * 4);
        bias[(size_t)d * 4 * H + col] = bi->data[r] + bh->data[r];
      }
      std::memcpy(&whh[(size_t)d * 4 * H * H], wh->data.data(), (size_t)4 * H * H * 4);
    }
    LstmLayer& L = e->lstm[l];
    L.in_pad = in_pad;
    L.ih.out = 8 * H; L.ih.in = in_pad;
    RVD_TRY(pack_T(e, L.ih.w, wih.data(), wih.size()));
    RVD_TRY(up_f32(e, L.ih.b, bias.data(), bias.size()));
    RVD_TRY(pack_T(e, L.whh, whh.data(), whh.size()));
  }
  // --- linears + classifier ---
  e->lin.resize(c.linear_layers);
  for (int i = 0; i < c.linear_layers; ++i) {
    const int in = i == 0 ? 2 * H : c.linear_dim;
    const std::string p = S + "linear." + std::to_string(i);
    RVD_TRY(need(e, p + ".weight", (size_t)c.linear_dim * in, &t1));
    RVD_TRY(need(e, p + ".bias", c.linear_dim, &t2));
    e->lin[i].out = c.linear_dim; e->lin[i].in = in;
    RVD_TRY(pack_T(e, e->lin[i].w, t1->data.data(), t1->data.size()));
    RVD_TRY(up_f32(e, e->lin[i].b, t2->data.data(), t2->data.size()));
  }
  const int cls_in = c.linear_layers ? c.linear_dim : 2 * H;
  RVD_TRY(need(e, S + "classifier.weight", (size_t)c.num_classes * cls_in, &t1));
  RVD_TRY(need(e, S + "classifier.bias", c.num_classes, &t2));
  RVD_TRY(up_f32(e, e->cls_w, t1->data.data(), t1->data.size()));
  RVD_TRY(up_f32(e, e->cls_b, t2->data.data(), t2->data.size()));
  if (c.emb_channels > 0) RVD_TRY(finalize_embedding(e));
  e->host.clear();
  e->stage.release();
  e->finalized = true;
  return OK;
}

int segment_impl(rvd_engine* e, int64_t first, int W, float* logp_out) {
  const rvd_model_cfg& c = e->cfg;
  if (!e->finalized || e->n_windows == 0) { set_error("rvd_segment: finalize the model and upload audio first"); return E_STATE; }
  if (first < 0 || W <= 0 || first + W > e->n_windows) { set_error("rvd_segment: window range outside the uploaded audio"); return E_ARG; }
  const size_t ts = dt_size(e->dtype);
  const int H = c.lstm_hidden, CP = e->cpad, NF = c.sinc_filters;
  const int64_t R1 = (int64_t)W * e->p1, R2 = (int64_t)W * e->p2, R3 = (int64_t)W * e->p3;
  if (R1 > 0x7fffffffLL / 4) { set_error("rvd_segment: too many windows in one call (limit 100k frames rows)"); return E_ARG; }
  RVD_TRY(e->stats.ensure((size_t)W * 8));
  RVD_TRY(e->a1.ensure((size_t)(R1 + 8) * NF * ts));
  RVD_TRY(e->c2.ensure((size_t)R1 * CP * ts));
  RVD_TRY(e->a2.ensure((size_t)(R2 + 8) * CP * ts));
  RVD_TRY(e->c3.ensure((size_t)R2 * CP * ts));
  RVD_TRY(e->a3.ensure((size_t)R3 * CP * ts));
  RVD_TRY(e->xproj.ensure((size_t)R3 * 8 * H * ts));
  RVD_TRY(e->hA.ensure((size_t)R3 * 2 * H * ts));
  RVD_TRY(e->hB.ensure((size_t)R3 * 2 * H * ts));
  RVD_TRY(e->l0.ensure((size_t)R3 * c.linear_dim * ts));
  RVD_TRY(e->l1.ensure((size_t)R3 * c.linear_dim * ts));
  RVD_TRY(e->logp.ensure((size_t)R3 * c.num_classes * 4));
  RVD_TRY(e->cls.ensure((size_t)R3));

  { DScope sc(e, "window_stats");
    RVD_TRY(window_stats(e->stream, e->wave.as<float>(), first, W, c.step_samples, c.window_samples, 1e-5f, e->stats.as<float>())); }
  PoolNormArgs pn{};
  pn.W = W; pn.eps = 1e-5f;
  { DScope sc(e, "pool_norm");
    pn.x = nullptr; pn.frames_in = e->f1; pn.C = NF; pn.ld_out = NF;
    pn.gamma = e->norm[0].g.as<float>(); pn.beta = e->norm[0].b.as<float>(); pn.out = e->a1.p;
    pn.craw = e->craw.p; pn.craw_frame0 = first * (c.step_samples / SINC_STRIDE);
    pn.craw_frames_per_step = c.step_samples / SINC_STRIDE;
    pn.stats = e->stats.as<float>(); pn.fsum = e->fsum.as<float>(); pn.wn_gamma = e->wn_gamma; pn.wn_beta = e->wn_beta;
    RVD_TRY(pool_norm(e->stream, e->dtype, pn)); }
  RVD_TRY(run_sincnet_conv(e, e->a1.p, NF, e->conv2, e->c2.p, CP, R1));
  { DScope sc(e, "pool_norm");
    pn.x = e->c2.p; pn.rows_in = e->p1; pn.ld_in = CP; pn.frames_in = e->f2; pn.C = c.sinc_channels; pn.ld_out = CP;
    pn.gamma = e->norm[1].g.as<float>(); pn.beta = e->norm[1].b.as<float>(); pn.out = e->a2.p;
    RVD_TRY(pool_norm(e->stream, e->dtype, pn)); }
  RVD_TRY(run_sincnet_conv(e, e->a2.p, CP, e->conv3, e->c3.p, CP, R2));
  { DScope sc(e, "pool_norm");
    pn.x = e->c3.p; pn.rows_in = e->p2; pn.ld_in = CP; pn.frames_in = e->f3; pn.C = c.sinc_channels; pn.ld_out = CP;
    pn.gamma = e->norm[2].g.as<float>(); pn.beta = e->norm[2].b.as<float>(); pn.out = e->a3.p;
    RVD_TRY(pool_norm(e->stream, e->dtype, pn)); }

  const void* x = e->a3.p;
  int ldx = CP;
  for (int l = 0; l < c.lstm_layers; ++l) {
    RVD_TRY(run_gemm(e, "lstm_inproj", x, ldx, e->lstm[l].ih, e->xproj.p, 8 * H, R3, ACT_NONE));
    void* out = (l & 1) ? e->hB.p : e->hA.p;
    { DScope sc(e, "lstm_recurrence", 2.0 * (double)R3 * 8 * H * H);
      RVD_TRY(lstm_recurrence(e->stream, e->dtype, e->xproj.p, e->lstm[l].whh.p, out, W, e->p3)); }
    x = out; ldx = 2 * H;
  }
  e->last_lstm = x;
  for (int i = 0; i < c.linear_layers; ++i) {
    void* out = (i & 1) ? e->l1.p : e->l0.p;
    RVD_TRY(run_gemm(e, "linear", x, ldx, e->lin[i], out, c.linear_dim, R3, ACT_LRELU));
    x = out; ldx = c.linear_dim;
  }
  { DScope sc(e, "classifier");
    RVD_TRY(classifier_logsoftmax(e->stream, e->dtype, x, ldx, e->cls_w.as<float>(), e->cls_b.as<float>(), e->logp.as<float>(),
                                  e->cls.as<uint8_t>(), R3, ldx, c.num_classes)); }
  e->last_W = W;
  if (logp_out) {
    DScope sc(e, "d2h");
    RVB_HIP_CHECK(hipMemcpyAsync(logp_out, e->logp.p, (size_t)R3 * c.num_classes * 4, hipMemcpyDeviceToHost, e->stream));
  }
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  return OK;
}

// ------------------------------------------------------------------------------------ embedding model
int make_hamming_fbank_tables(rvd_engine* e) {
  const int NFFT = 512, NBIN = 257;
  const double PI = 3.14159265358979323846;
  std::vector<float> win(FB_WIN), tw(2 * 256), melw((size_t)FB_MEL * NBIN, 0.f);
  std::vector<int32_t> lo(FB_MEL, NBIN), hi(FB_MEL, 0);
  for (int i = 0; i < FB_WIN; ++i) win[i] = (float)(0.54 - 0.46 * std::cos(2.0 * PI * i / (FB_WIN - 1)));
  for (int k = 0; k < 256; ++k) { tw[2 * k] = (float)std::cos(2.0 * PI * k / NFFT); tw[2 * k + 1] = (float)(-std::sin(2.0 * PI * k / NFFT)); }
  auto mel = [](double f) { return 1127.0 * std::log(1.0 + f / 700.0); };
  const double mlo = mel(20.0), mhi = mel(8000.0), delta = (mhi - mlo) / (FB_MEL + 1);
  for (int m = 0; m < FB_MEL; ++m) {
    const double left = mlo + m * delta, center = left + delta, right = center + delta;
    for (int b = 0; b < NFFT / 2; ++b) {
      const double mf = mel(16000.0 / NFFT * b);
      const double w = std::max(0.0, std::min((mf - left) / (center - left), (right - mf) / (right - center)));
      if (w > 0.0) { melw[(size_t)m * NBIN + b] = (float)w; lo[m] = std::min(lo[m], b); hi[m] = std::max(hi[m], b + 1); }
    }
    if (hi[m] == 0) lo[m] = 0;
  }
  RVD_TRY(up_f32(e, e->fb_window, win.data(), win.size()));
  RVD_TRY(up_f32(e, e->fb_twiddle, tw.data(), tw.size()));
  RVD_TRY(up_f32(e, e->fb_melw, melw.data(), melw.size()));
  RVD_TRY(up_f32(e, e->fb_lo, (const float*)lo.data(), lo.size()));   // raw 4-byte copies
  return up_f32(e, e->fb_hi, (const float*)hi.data(), hi.size());
}

// Conv2d weight [cout][cin][k][k] + eval-mode BatchNorm -> [tap][cin/CK][cout][CK] (T) and bias (fp32)
int pack_conv_bn(rvd_engine* e, ConvW& c, const std::string& conv, const std::string& bn, int cout, int cin, int k, int stride) {
  const HostTensor *w, *g, *b, *m, *v;
  RVD_TRY(need(e, conv + ".weight", (size_t)cout * cin * k * k, &w));
  RVD_TRY(need(e, bn + ".weight", cout, &g));
  RVD_TRY(need(e, bn + ".bias", cout, &b));
  RVD_TRY(need(e, bn + ".running_mean", cout, &m));
  RVD_TRY(need(e, bn + ".running_var", cout, &v));
  const int CK = 64 / (int)dt_size(e->dtype);
  if (cin % CK) { set_error("embedding conv " + conv + ": input channels must be a multiple of " + std::to_string(CK)); return E_UNSUPPORTED; }
  const int taps = k * k, nch = cin / CK;
  std::vector<float> pw((size_t)taps * cin * cout), pb(cout);
  for (int o = 0; o < cout; ++o) {
    const float sc = g->data[o] / std::sqrt(v->data[o] + 1e-5f);
    pb[o] = b->data[o] - m->data[o] * sc;
    for (int ci = 0; ci < cin; ++ci)
      for (int t = 0; t < taps; ++t)
        pw[(((size_t)t * nch + ci / CK) * cout + o) * CK + ci % CK] = w->data[((size_t)o * cin + ci) * taps + t] * sc;
  }
  c.cin = cin; c.cout = cout; c.taps = taps; c.stride = stride;
  RVD_TRY(pack_T(e, c.w, pw.data(), pw.size()));
  // implicit-GEMM kernel (conv_gemm.hip) for the 128- and 256-channel stride-1 convolutions: second weight layout
  // [cout][tap][cin].  Validated on hardware in round 2 (tests/test_diar_gpu.py: both kernels against the oracle and
  // against each other; 806 / 1150 TFLOP/s vs 555-598 for the direct kernel); RVD_CONV_IGEMM=0 selects the direct kernel.
  // Round 4: the stride-2 convolutions that open stages 3 and 4 go there too (RVD_CONV_IGEMM=1: stride 1 only, as in round 2).
  const char* ig = lab_env("RVD_CONV_IGEMM");
  const int ig_mode = ig ? atoi(ig) : 2;
  if (ig_mode != 0 && e->dtype == DT_BF16 && k == 3 && (stride == 1 || (stride == 2 && ig_mode >= 2)) && cin % 64 == 0 && cout % 128 == 0) {
    std::vector<float> pg((size_t)cout * taps * cin);
    for (int o = 0; o < cout; ++o) {
      const float sc = g->data[o] / std::sqrt(v->data[o] + 1e-5f);
      for (int t = 0; t < taps; ++t)
        for (int ci = 0; ci < cin; ++ci) pg[((size_t)o * taps + t) * cin + ci] = w->data[((size_t)o * cin + ci) * taps + t] * sc;
    }
    RVD_TRY(pack_T(e, c.w_ig, pg.data(), pg.size()));
    if (e->emb_fp8 && cin % 128 == 0) {          // e4m3 copy, one scale per output channel (as engine.hip's pack_linear)
      std::vector<uint8_t> q(pg.size());
      std::vector<float> ws(cout);
      const size_t K = (size_t)taps * cin;
      for (int o = 0; o < cout; ++o) {
        float am = 0.f;
        for (size_t kk = 0; kk < K; ++kk) am = std::max(am, std::fabs(pg[(size_t)o * K + kk]));
        ws[o] = am > 0.f ? am / 448.f : 1.f;
        const float inv = 1.f / ws[o];
        for (size_t kk = 0; kk < K; ++kk) q[(size_t)o * K + kk] = f32_to_fp8_host(pg[(size_t)o * K + kk] * inv);
      }
      RVD_TRY(c.w8.ensure(q.size()));
      RVB_HIP_CHECK(hipMemcpyAsync(c.w8.p, q.data(), q.size(), hipMemcpyHostToDevice, e->stream));
      RVD_TRY(up_f32(e, c.w8s, ws.data(), ws.size()));
      RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
    }
  }
  return up_f32(e, c.b, pb.data(), pb.size());
}

// Default since round 5 (measured: shortcut kernels 12.1 -> 5.2 ms per hour, the 256-channel stage 21.4 -> 25.0 ms, step 358-360
// -> 355-359 ms, profiles/archive/r05_call1.txt; and one rounding point fewer): the 1x1 / stride-2 projection shortcut of the blocks that open stages 3 and 4
// rides in the K loop of the block's second convolution (conv_gemm.hip, ConvArgs::in2) instead of being a kernel of its own whose
// output is written, read back as the residual and rounded to bf16 on the way: rows [cout][9 cout + cin_sc], bias = b_2 + b_sc.
int pack_fused_shortcut(rvd_engine* e, ResBlock& B, const std::string& p) {
  const char* f = lab_env("RVD_CONV_SC_FUSE");           // lab: 0 = the shortcut as a kernel of its own (until round 5)
  if ((f && atoi(f) == 0) || !B.has_sc || !B.c2.w_ig.p || B.sc.cin % 64 || B.sc.taps != 1) return OK;
  const int cout = B.c2.cout, cin = B.c2.cin, c2 = B.sc.cin;
  const HostTensor *w, *g, *b, *m, *v, *ws, *gs, *bs, *ms, *vs;
  RVD_TRY(need(e, p + ".conv2.weight", (size_t)cout * cin * 9, &w));
  RVD_TRY(need(e, p + ".bn2.weight", cout, &g));
  RVD_TRY(need(e, p + ".bn2.bias", cout, &b));
  RVD_TRY(need(e, p + ".bn2.running_mean", cout, &m));
  RVD_TRY(need(e, p + ".bn2.running_var", cout, &v));
  RVD_TRY(need(e, p + ".shortcut.0.weight", (size_t)cout * c2, &ws));
  RVD_TRY(need(e, p + ".shortcut.1.weight", cout, &gs));
  RVD_TRY(need(e, p + ".shortcut.1.bias", cout, &bs));
  RVD_TRY(need(e, p + ".shortcut.1.running_mean", cout, &ms));
  RVD_TRY(need(e, p + ".shortcut.1.running_var", cout, &vs));
  const size_t ld = (size_t)9 * cin + c2;
  std::vector<float> pg((size_t)cout * ld), pb(cout);
  for (int o = 0; o < cout; ++o) {
    const float sc = g->data[o] / std::sqrt(v->data[o] + 1e-5f), ss = gs->data[o] / std::sqrt(vs->data[o] + 1e-5f);
    pb[o] = (b->data[o] - m->data[o] * sc) + (bs->data[o] - ms->data[o] * ss);
    for (int t = 0; t < 9; ++t)
      for (int ci = 0; ci < cin; ++ci) pg[(size_t)o * ld + (size_t)t * cin + ci] = w->data[((size_t)o * cin + ci) * 9 + t] * sc;
    for (int ci = 0; ci < c2; ++ci) pg[(size_t)o * ld + (size_t)9 * cin + ci] = ws->data[(size_t)o * c2 + ci] * ss;
  }
  RVD_TRY(pack_T(e, B.c2.w_ig_sc, pg.data(), pg.size()));
  return up_f32(e, B.c2.b_sc, pb.data(), pb.size());
}

int finalize_embedding(rvd_engine* e) {
  const rvd_model_cfg& c = e->cfg;
  const std::string S = "embedding.resnet.";
  const int m = c.emb_channels;
  if (m != 32) { set_error("embedding: this build supports m_channels = 32 (ResNet34 of pyannote/wespeaker-voxceleb-resnet34-LM)"); return E_UNSUPPORTED; }
  // rvd_model_cfg.dtype = RVB_FP8: the bf16 engine with stages 3-4 of the trunk on e4m3 operands (lab: RVD_EMB_FP8=1 on a bf16 engine)
  { const char* f8 = lab_env("RVD_EMB_FP8"); e->emb_fp8 = (e->want_fp8 || (f8 && atoi(f8) == 1)) && e->dtype == DT_BF16; e->emb_f8_state = 0; }
  // stem: Conv2d(1, m, 3) + BN folded, fp32 weights [m][9]
  const HostTensor *w, *g, *b, *mu, *v;
  RVD_TRY(need(e, S + "conv1.weight", (size_t)m * 9, &w));
  RVD_TRY(need(e, S + "bn1.weight", m, &g));
  RVD_TRY(need(e, S + "bn1.bias", m, &b));
  RVD_TRY(need(e, S + "bn1.running_mean", m, &mu));
  RVD_TRY(need(e, S + "bn1.running_var", m, &v));
  std::vector<float> sw((size_t)m * 9), sb(m);
  for (int o = 0; o < m; ++o) {
    const float sc = g->data[o] / std::sqrt(v->data[o] + 1e-5f);
    sb[o] = b->data[o] - mu->data[o] * sc;
    for (int t = 0; t < 9; ++t) sw[(size_t)o * 9 + t] = w->data[(size_t)o * 9 + t] * sc;
  }
  RVD_TRY(up_f32(e, e->stem_w, sw.data(), sw.size()));
  RVD_TRY(up_f32(e, e->stem_b, sb.data(), sb.size()));
  const int nblk[4] = {3, 4, 6, 3};
  e->stages.assign(4, {});
  int cin = m;
  for (int li = 0; li < 4; ++li) {
    const int cout = m << li;
    e->stages[li].resize(nblk[li]);
    for (int bi = 0; bi < nblk[li]; ++bi) {
      ResBlock& B = e->stages[li][bi];
      const int stride = (bi == 0 && li > 0) ? 2 : 1;
      const std::string p = S + "layer" + std::to_string(li + 1) + "." + std::to_string(bi);
      RVD_TRY(pack_conv_bn(e, B.c1, p + ".conv1", p + ".bn1", cout, cin, 3, stride));
      RVD_TRY(pack_conv_bn(e, B.c2, p + ".conv2", p + ".bn2", cout, cout, 3, 1));
      B.has_sc = stride != 1 || cin != cout;
      if (B.has_sc) RVD_TRY(pack_conv_bn(e, B.sc, p + ".shortcut.0", p + ".shortcut.1", cout, cin, 1, stride));
      RVD_TRY(pack_fused_shortcut(e, B, p));
      cin = cout;
    }
  }
  const int stats = (FB_MEL / 8) * (m << 3) * 2;
  RVD_TRY(need(e, S + "seg_1.weight", (size_t)c.emb_dim * stats, &w));
  RVD_TRY(need(e, S + "seg_1.bias", c.emb_dim, &b));
  e->seg1.out = c.emb_dim; e->seg1.in = stats;
  RVD_TRY(pack_T(e, e->seg1.w, w->data.data(), w->data.size()));
  RVD_TRY(up_f32(e, e->seg1.b, b->data.data(), b->data.size()));
  RVD_TRY(make_hamming_fbank_tables(e));
  e->nfr = (c.window_samples - FB_WIN) / FB_SHIFT + 1;
  e->has_emb = true;
  return OK;
}

struct StageDims { int F, T, C; };
StageDims stage_dims(const rvd_engine* e, int li) {
  int F = FB_MEL, T = e->nfr;
  for (int i = 0; i < li; ++i) { F = (F - 1) / 2 + 1; T = (T - 1) / 2 + 1; }
  return {F, T, e->cfg.emb_channels << li};
}

int ensure_emb_workspace(rvd_engine* e, int B) {
  if (B <= e->act_cap) return OK;
  // lab hook (librvb_test.so only): pretend that more than RVD_FAKE_NOMEM_ABOVE windows do not fit, to exercise the retry below
  if (const char* f = lab_env("RVD_FAKE_NOMEM_ABOVE")) {
    if (B > atoi(f)) { set_error("hipMalloc refused (RVD_FAKE_NOMEM_ABOVE)"); return E_NOMEM; }
  }
  const size_t ts = dt_size(e->dtype);
  for (int li = 0; li < 4; ++li) {
    const StageDims d = stage_dims(e, li);
    const size_t bytes = (size_t)B * (d.F + 2) * (d.T + 2) * d.C * ts;
    for (int k = 0; k < 4; ++k) {
      if (k == 3 && li == 0) continue;            // stage 1 has no projection shortcut
      RVD_TRY(e->act[li][k].ensure(bytes));
      RVB_HIP_CHECK(hipMemsetAsync(e->act[li][k].p, 0, bytes, e->stream));   // the zero border is never written again
      if (e->emb_fp8 && li >= 2 && k < 3) {
        RVD_TRY(e->act8[li][k].ensure(bytes / ts));
        RVB_HIP_CHECK(hipMemsetAsync(e->act8[li][k].p, 0, bytes / ts, e->stream));
      }
    }
  }
  e->act_cap = B;
  return OK;
}

int run_conv(rvd_engine* e, const ConvW& c, const void* in, const StageDims& di, const void* res, void* out, const StageDims& dq, int B, int relu) {
  ConvArgs a{};
  a.in = in; a.w = c.w.p; a.bias = c.b.as<float>(); a.res = res; a.out = out;
  a.B = B; a.Fi = di.F; a.Ti = di.T; a.Cin = c.cin; a.Fo = dq.F; a.To = dq.T; a.Cout = c.cout;
  a.stride = c.stride; a.taps = c.taps; a.relu = relu;
  a.w_ig = c.w_ig.p;
  const std::string nm = std::string(c.taps == 1 ? "emb_conv_sc" : (c.stride == 2 ? "emb_conv_s2_" : "emb_conv_")) + (c.taps == 1 ? "" : std::to_string(c.cout));
  if (conv_igemm_applicable(e->dtype, a)) {
    e->prof["emb_conv_igemm"].launches += 1;                                             // how many went to conv_gemm.hip
    if (conv_igemm_wide(a)) e->prof["emb_conv_igemm_wide"].launches += 1;
  }
  else if (conv_row64_applicable(e->dtype, a)) e->prof["emb_conv_row64"].launches += 1;        // ... to conv_row64.hip
  else if (conv_stream_applicable(e->dtype, a)) e->prof["emb_conv_stream"].launches += 1;     // ... to conv_stream.hip
  DScope sc(e, nm.c_str(), 2.0 * (double)B * dq.F * dq.T * c.cout * c.cin * c.taps);
  return conv2d(e->stream, e->dtype, a);
}

// trunk on B distinct windows (device list e->e_win); leaves the stage-4 output in *trunk_out
int run_trunk(rvd_engine* e, int B, const void** trunk_out) {
  const rvd_model_cfg& c = e->cfg;
  const int fps = c.step_samples / FB_SHIFT;
  const StageDims d0 = stage_dims(e, 0);
  { DScope sc(e, "emb_stem", 2.0 * (double)B * d0.F * d0.T * d0.C * 9);
    RVD_TRY(emb_conv1(e->stream, e->dtype, e->emb_fb.as<float>(), e->e_win.as<int64_t>(), e->e_mean.as<float>(), e->stem_w.as<float>(),
                      e->stem_b.as<float>(), e->act[0][0].p, B, d0.F, d0.T, fps, d0.C)); }
  const void* x = e->act[0][0].p;
  StageDims dx = d0;
  int xi = 0;                                   // index of x among the stage's rotating buffers
  for (int li = 0; li < 4; ++li) {
    const StageDims d = stage_dims(e, li);
    for (size_t bi = 0; bi < e->stages[li].size(); ++bi) {
      const ResBlock& Bk = e->stages[li][bi];
      if (bi == 0 && li > 0) xi = -1;           // x lives in the previous stage's buffers
      const int ti = xi < 0 ? 0 : (xi + 1) % 3;
      const int oi = xi < 0 ? 1 : (xi + 2) % 3;
      void* tmp = e->act[li][ti].p;
      void* out = e->act[li][oi].p;
      if (!Bk.has_sc && dx.F == d.F && dx.T == d.T &&
          conv_block32_applicable(e->dtype, Bk.c1.cin, Bk.c1.cout, Bk.c2.cout, Bk.c1.stride, Bk.c2.stride, Bk.c1.taps, Bk.c2.taps, d.F, d.T)) {
        // the whole residual block in one kernel (conv_block.hip): x read once, out written once, the intermediate tensor in LDS
        ConvBlockArgs a{};
        a.in = x; a.wa = Bk.c1.w.p; a.ba = Bk.c1.b.as<float>(); a.wb = Bk.c2.w.p; a.bb = Bk.c2.b.as<float>(); a.out = out;
        a.B = B; a.F = d.F; a.T = d.T;
        e->prof["emb_conv_block"].launches += 1;
        { DScope sc(e, "emb_conv_32", 2.0 * 2.0 * (double)B * d.F * d.T * 32 * 32 * 9);
          RVD_TRY(conv_block32(e->stream, a)); }
        x = out; dx = d; xi = oi;
        continue;
      }
      const bool f8blk = e->emb_fp8 && li >= 2 && Bk.c2.w8.p && (bi == 0 || Bk.c1.w8.p);
      if (f8blk && e->emb_f8_state == 2) {
        // ---- fp8 path (candidate): block 0 of a stage keeps its stride-2 first convolution and its shortcut in bf16 (their input is the
        // previous stage's bf16 tensor) and quantises the result; every other 3x3 convolution of stages 3-4 reads e4m3 operands
        const size_t n_el = (size_t)B * (d.F + 2) * (d.T + 2) * d.C;
        const int si = ((li - 2) * 8 + (int)bi) * 2;
        const float s_tmp = e->scale8[si], s_out = e->scale8[si + 1];
        unsigned* sat = e->d_sat8.p ? e->d_sat8.as<unsigned>() : nullptr;
        void* tmp8 = e->act8[li][ti].p;
        void* out8 = e->act8[li][oi].p;
        const void* res = x;
        auto conv8 = [&](const ConvW& cw, const void* in8, float a_scale, const void* resid, void* o16, void* o8, float o8_scale) -> int {
          ConvArgs a{};
          a.bias = cw.b.as<float>(); a.res = resid; a.out = o16;
          a.B = B; a.Fi = d.F; a.Ti = d.T; a.Cin = cw.cin; a.Fo = d.F; a.To = d.T; a.Cout = cw.cout; a.stride = 1; a.taps = 9; a.relu = 1;
          a.in8 = in8; a.w8 = cw.w8.p; a.w8_scale = cw.w8s.as<float>(); a.a_scale = a_scale; a.out8 = o8; a.out8_inv_scale = 1.f / o8_scale; a.sat8 = sat;
          if (!conv_igemm8_applicable(e->dtype, a)) { set_error("fp8 trunk: shape not supported by conv_igemm8"); return E_UNSUPPORTED; }
          e->prof["emb_conv_fp8"].launches += 1;
          DScope sc(e, ("emb_conv_" + std::to_string(cw.cout)).c_str(), 2.0 * (double)B * d.F * d.T * cw.cout * cw.cin * 9);
          return conv_igemm8(e->stream, a);
        };
        if (bi == 0) {
          RVD_TRY(run_conv(e, Bk.c1, x, dx, nullptr, tmp, d, B, 1));
          RVD_TRY(act_quant_fp8(e->stream, tmp, tmp8, n_el, s_tmp, sat));
          if (Bk.has_sc) {
            RVD_TRY(run_conv(e, Bk.sc, x, dx, nullptr, e->act[li][3].p, d, B, 0));
            res = e->act[li][3].p;
          }
        } else {
          RVD_TRY(conv8(Bk.c1, e->act8[li][xi].p, e->scale8[si - 1], nullptr, nullptr, tmp8, s_tmp));      // x8 = the previous block's e4m3 output
        }
        RVD_TRY(conv8(Bk.c2, tmp8, s_tmp, res, out, out8, s_out));
        x = out; dx = d; xi = oi;
        continue;
      }
      bool sc_done = false;
      if (Bk.has_sc && !Bk.c2.w_ig_sc.p &&
          conv_s2sc_applicable(e->dtype, Bk.c1.cin, Bk.c1.cout, Bk.c1.stride, Bk.c1.taps, Bk.sc.cin, Bk.sc.cout, Bk.sc.stride, Bk.sc.taps, dx.F, dx.T, d.F, d.T)) {
        // the stride-2 convolution and the projection shortcut of the block that opens the 64-channel stage: one pass over x (conv_s2.hip)
        ConvS2Args a{};
        a.in = x; a.w = Bk.c1.w.p; a.bias = Bk.c1.b.as<float>(); a.wsc = Bk.sc.w.p; a.bsc = Bk.sc.b.as<float>();
        a.out = tmp; a.sc = e->act[li][3].p;
        a.B = B; a.Fi = dx.F; a.Ti = dx.T; a.Fo = d.F; a.To = d.T;
        e->prof["emb_conv_s2sc"].launches += 1;
        { DScope sc(e, ("emb_conv_s2_" + std::to_string(Bk.c1.cout)).c_str(),
                    2.0 * (double)B * d.F * d.T * Bk.c1.cout * ((double)Bk.c1.cin * 9 + Bk.sc.cin));
          RVD_TRY(conv_s2sc(e->stream, a)); }
        sc_done = true;
      } else {
        RVD_TRY(run_conv(e, Bk.c1, x, dx, nullptr, tmp, d, B, 1));
      }
      if (f8blk && e->emb_f8_state == 1)
        RVD_TRY(act_amax_bf16(e->stream, tmp, (size_t)B * (d.F + 2) * (d.T + 2) * d.C, e->d_amax8.as<unsigned>() + ((li - 2) * 8 + (int)bi) * 2));
      if (Bk.has_sc && Bk.c2.w_ig_sc.p) {        // the projection shortcut inside the second convolution's K loop (RVD_CONV_SC_FUSE=1)
        ConvArgs a{};
        a.in = tmp; a.w = Bk.c2.w.p; a.bias = Bk.c2.b_sc.as<float>(); a.res = nullptr; a.out = out;
        a.B = B; a.Fi = d.F; a.Ti = d.T; a.Cin = Bk.c2.cin; a.Fo = d.F; a.To = d.T; a.Cout = Bk.c2.cout;
        a.stride = 1; a.taps = 9; a.relu = 1;
        a.w_ig = Bk.c2.w_ig_sc.p;
        a.in2 = x; a.Cin2 = Bk.sc.cin; a.Fi2 = dx.F; a.Ti2 = dx.T; a.stride2 = Bk.sc.stride;
        if (!conv_igemm_applicable(e->dtype, a)) { set_error("fused shortcut: shape not supported by the implicit-GEMM kernel"); return E_UNSUPPORTED; }
        e->prof["emb_conv_igemm"].launches += 1;
        e->prof["emb_conv_sc_fused"].launches += 1;
        { DScope sc(e, ("emb_conv_" + std::to_string(Bk.c2.cout)).c_str(),
                    2.0 * (double)B * d.F * d.T * Bk.c2.cout * ((double)Bk.c2.cin * 9 + Bk.sc.cin));
          RVD_TRY(conv2d(e->stream, e->dtype, a)); }
        x = out; dx = d; xi = oi;
        continue;
      }
      const void* res = x;
      if (Bk.has_sc) {
        if (!sc_done) RVD_TRY(run_conv(e, Bk.sc, x, dx, nullptr, e->act[li][3].p, d, B, 0));
        res = e->act[li][3].p;
      }
      RVD_TRY(run_conv(e, Bk.c2, tmp, d, res, out, d, B, 1));
      if (f8blk && e->emb_f8_state == 1)
        RVD_TRY(act_amax_bf16(e->stream, out, (size_t)B * (d.F + 2) * (d.T + 2) * d.C, e->d_amax8.as<unsigned>() + ((li - 2) * 8 + (int)bi) * 2 + 1));
      x = out; dx = d; xi = oi;
    }
  }
  *trunk_out = x;
  return OK;
}

int embed_impl(rvd_engine* e, const int64_t* win, const float* mask, int n, float* emb_out) {
  const rvd_model_cfg& c = e->cfg;
  if (!e->has_emb) { set_error("rvd_embed: no embedding model was loaded (cfg.emb_channels == 0)"); return E_STATE; }
  if (e->n_windows == 0) { set_error("rvd_embed: upload audio first"); return E_STATE; }
  const int frames = e->p3;
  const StageDims d3 = stage_dims(e, 3);
  const int stats = 2 * d3.C * d3.F;
  int i0 = 0;
  while (i0 < n) {
    // next group of items covering at most EMB_BATCH distinct windows (items of one window are adjacent)
    std::vector<int64_t> uniq;
    std::vector<int32_t> item_b;
    int i1 = i0;
    while (i1 < n) {
      if (win[i1] < 0 || win[i1] >= e->n_windows) { set_error("rvd_embed: window index outside the uploaded audio"); return E_ARG; }
      int u = -1;
      for (int k = (int)uniq.size() - 1; k >= 0 && k >= (int)uniq.size() - 4; --k) if (uniq[k] == win[i1]) { u = k; break; }
      if (u < 0) {
        if ((int)uniq.size() == EMB_BATCH) break;
        uniq.push_back(win[i1]); u = (int)uniq.size() - 1;
      }
      item_b.push_back(u);
      ++i1;
    }
    const int B = (int)uniq.size(), ni = i1 - i0;
    {
      const int rc = ensure_emb_workspace(e, B);
      if (rc == E_NOMEM && B > 1) {                 // the activations of B windows do not fit beside what else lives on this GPU:
        e->emb_batch = std::max(1, std::min(EMB_BATCH, B) / 2);   // fewer windows per pass from here on, same results (ADVICE r4;
                                                    // r5: halve what was ASKED FOR, so that every retry really shrinks)
        for (auto& row : e->act) for (auto& b : row) b.release();
        for (auto& row : e->act8) for (auto& b : row) b.release();
        e->act_cap = 0;
        continue;                                   // regroup the same items under the smaller batch
      }
      if (rc != OK) return rc;
    }
    RVD_TRY(e->e_win.ensure((size_t)B * 8));
    RVD_TRY(e->e_item_b.ensure((size_t)ni * 4));
    RVD_TRY(e->e_mask.ensure((size_t)ni * frames * 4));
    RVD_TRY(e->e_stats.ensure((size_t)ni * stats * dt_size(e->dtype)));
    RVD_TRY(e->e_out.ensure((size_t)ni * c.emb_dim * 4));
    RVB_HIP_CHECK(hipMemcpyAsync(e->e_win.p, uniq.data(), (size_t)B * 8, hipMemcpyHostToDevice, e->stream));
    RVB_HIP_CHECK(hipMemcpyAsync(e->e_item_b.p, item_b.data(), (size_t)ni * 4, hipMemcpyHostToDevice, e->stream));
    RVB_HIP_CHECK(hipMemcpyAsync(e->e_mask.p, mask + (size_t)i0 * frames, (size_t)ni * frames * 4, hipMemcpyHostToDevice, e->stream));
    RVB_HIP_CHECK(hipStreamSynchronize(e->stream));    // uniq / item_b are locals
    const void* trunk = nullptr;
    if (e->emb_fp8 && e->emb_f8_state == 0) {          // the first trunk pass of an fp8 engine runs in bf16 and records the ranges
      RVD_TRY(e->d_amax8.ensure(32 * 4));
      RVD_TRY(e->d_sat8.ensure(4));
      RVB_HIP_CHECK(hipMemsetAsync(e->d_amax8.p, 0, 32 * 4, e->stream));
      RVB_HIP_CHECK(hipMemsetAsync(e->d_sat8.p, 0, 4, e->stream));
      e->emb_f8_state = 1;
    }
    RVD_TRY(run_trunk(e, B, &trunk));
    if (e->emb_f8_state == 1) {
      float am[32];
      RVB_HIP_CHECK(hipMemcpyAsync(am, e->d_amax8.p, sizeof(am), hipMemcpyDeviceToHost, e->stream));
      RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
      e->scale8.assign(32, 1.f);
      for (int i = 0; i < 32; ++i)
        if (am[i] > 0.f) e->scale8[i] = std::exp2(std::ceil(std::log2(2.f * am[i] / 448.f)));      // power of two, 2x headroom (as engine.hip)
      e->emb_f8_state = 2;
    }
    { DScope sc(e, "emb_pool");
      RVD_TRY(tstp_pool(e->stream, e->dtype, trunk, e->e_item_b.as<int>(), e->e_mask.as<float>(), frames, ni, d3.F, d3.T, d3.C, e->e_stats.p)); }
    { DScope sc(e, "emb_linear", 2.0 * (double)ni * stats * c.emb_dim);
      GemmArgs g{};
      g.A = e->e_stats.p; g.W = e->seg1.w.p; g.bias = e->seg1.b.as<float>(); g.C = e->e_out.p;
      g.M = ni; g.N = c.emb_dim; g.K = stats; g.lda = stats; g.ldw = stats; g.ldc = c.emb_dim; g.alpha = 1.f; g.act = ACT_NONE;
      g.out_f32 = 1;
      RVD_TRY(gemm(e->stream, e->dtype, g)); }
    RVB_HIP_CHECK(hipMemcpyAsync(emb_out + (size_t)i0 * c.emb_dim, e->e_out.p, (size_t)ni * c.emb_dim * 4, hipMemcpyDeviceToHost, e->stream));
    RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
    i0 = i1;
  }
  return OK;
}

// device T [rows][ld] -> host fp32 [rows][cols]
int fetch_T(rvd_engine* e, const void* src, int64_t rows, int ld, int cols, float* out) {
  const size_t ts = dt_size(e->dtype);
  std::vector<char> tmp((size_t)rows * ld * ts);
  RVB_HIP_CHECK(hipMemcpy(tmp.data(), src, tmp.size(), hipMemcpyDeviceToHost));
  for (int64_t r = 0; r < rows; ++r)
    for (int cc = 0; cc < cols; ++cc)
      out[r * cols + cc] = e->dtype == DT_BF16 ? bf16_to_f32(((const bf16_t*)tmp.data())[r * ld + cc]) : ((const float*)tmp.data())[r * ld + cc];
  return OK;
}

}  // namespace

extern "C" {

const char* rvd_last_error(void) { return rvb::last_error(); }

int rvd_model_cfg_size(void) { return (int)sizeof(rvd_model_cfg); }

int rvd_emb_windows_per_pass(rvd_engine* e) {
  if (!e) { set_error("rvd_emb_windows_per_pass: null engine"); return E_ARG; }
  return EMB_BATCH;
}

int rvd_create(const rvd_model_cfg* cfg, int device, rvd_engine** out) {
  if (!cfg || !out) { set_error("rvd_create: null argument"); return E_ARG; }
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(rvd_model_cfg)) {
    set_error("rvd_create: ABI mismatch: rvd_model_cfg.struct_size is " + std::to_string(cfg->struct_size) + ", this library's struct has " +
              std::to_string(sizeof(rvd_model_cfg)) + " bytes (bind it field by field from include/rvd.h)");
    return E_ARG;
  }
  if (cfg->dtype != DT_F32 && cfg->dtype != DT_BF16 && cfg->dtype != RVB_FP8) { set_error("rvd_create: dtype must be RVB_F32, RVB_BF16 or RVB_FP8"); return E_ARG; }
  if (cfg->lstm_hidden != 128 || cfg->sinc_filters != 80 || cfg->sinc_filters % 8 || cfg->sinc_channels < 1 || cfg->sinc_channels > 64 ||
      cfg->lstm_layers < 1 || cfg->linear_layers < 0 || cfg->linear_layers > 2 || (cfg->linear_layers && cfg->linear_dim % 8) ||
      cfg->linear_dim > 256 || cfg->num_classes < 1 || cfg->num_classes > 16 || cfg->sample_rate != 16000 ||
      cfg->step_samples % (SINC_STRIDE * 4) || cfg->step_samples <= 0 || cfg->window_samples < 4000) {
    set_error("rvd_create: unsupported model dimensions (this build: 80 sinc filters, <= 64 conv channels, LSTM hidden 128, 16 kHz)");
    return E_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device || device < 0) {
    set_error("rvd_create: no HIP device " + std::to_string(device) + " (this library has no CPU fallback)");
    return E_HIP;
  }
  RVB_HIP_CHECK(hipSetDevice(device));
  rvd_engine* e = new rvd_engine();
  e->cfg = *cfg; e->device = device;
  // RVB_FP8 = BASELINE configs[4] "fp8 MFMA GEMMs" on the diarization side: everything as in the bf16 engine except the 3x3
  // convolutions of the ResNet34 trunk's stages 3-4 (128 / 256 channels, the MFMA-bound half of the trunk), which run on
  // v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands (conv_igemm8_kernel); scales calibrated by the first rvd_embed call
  e->want_fp8 = cfg->dtype == RVB_FP8;
  e->dtype = e->want_fp8 ? (int)DT_BF16 : cfg->dtype;
  e->cfg.dtype = e->dtype;
  RVB_HIP_CHECK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  e->f1 = (cfg->window_samples - SINC_K) / SINC_STRIDE + 1; e->p1 = e->f1 / 3;
  e->f2 = e->p1 - (CONV_K - 1); e->p2 = e->f2 / 3;
  e->f3 = e->p2 - (CONV_K - 1); e->p3 = e->f3 / 3;
  e->cpad = 64;
  *out = e;
  return OK;
}

void rvd_destroy(rvd_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  drain(e);
  for (auto ev : e->event_pool) (void)hipEventDestroy(ev);
  DevBuf* bufs[] = {&e->filt, &e->fsum, &e->cls_w, &e->cls_b, &e->stage, &e->pcm, &e->wave, &e->craw, &e->stats, &e->a1, &e->c2,
                    &e->a2, &e->c3, &e->a3, &e->xproj, &e->hA, &e->hB, &e->l0, &e->l1, &e->logp, &e->cls, &e->conv2.w, &e->conv2.b,
                    &e->conv3.w, &e->conv3.b};
  for (auto* b : bufs) b->release();
  for (auto& n : e->norm) { n.g.release(); n.b.release(); }
  for (auto& l : e->lstm) { l.ih.w.release(); l.ih.b.release(); l.whh.release(); }
  for (auto& l : e->lin) { l.w.release(); l.b.release(); }
  DevBuf* ebufs[] = {&e->stem_w, &e->stem_b, &e->seg1.w, &e->seg1.b, &e->fb_window, &e->fb_twiddle, &e->fb_melw, &e->fb_lo, &e->fb_hi,
                     &e->pcm_pad, &e->emb_fb, &e->e_win, &e->e_mean, &e->e_item_b, &e->e_mask, &e->e_stats, &e->e_out};
  for (auto* b : ebufs) b->release();
  for (auto& st : e->stages)
    for (auto& blk : st)
      for (ConvW* cw : {&blk.c1, &blk.c2, &blk.sc}) { cw->w.release(); cw->b.release(); cw->w_ig.release(); cw->w_ig_sc.release(); cw->b_sc.release(); cw->w8.release(); cw->w8s.release(); }
  for (auto& row : e->act) for (auto& b : row) b.release();
  (void)hipStreamDestroy(e->stream);
  delete e;
}

int rvd_load_tensor(rvd_engine* e, const char* name, const float* host, const int64_t* shape, int ndim) {
  if (!e || !name || !host || ndim < 0 || (ndim && !shape)) { set_error("rvd_load_tensor: null argument"); return E_ARG; }
  if (e->finalized) { set_error("rvd_load_tensor: engine already finalized"); return E_STATE; }
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  t.data.assign(host, host + t.numel());
  e->host[name] = std::move(t);
  return OK;
}

int rvd_finalize(rvd_engine* e) {
  if (!e) { set_error("rvd_finalize: null engine"); return E_ARG; }
  if (e->finalized) { set_error("rvd_finalize: already finalized"); return E_STATE; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  return finalize_impl(e);
}

int64_t rvd_num_windows(const rvd_engine* e, int64_t n) {
  if (!e || n <= 0) return 0;
  const int64_t win = e->cfg.window_samples, step = e->cfg.step_samples;
  const int64_t full = n >= win ? (n - win) / step + 1 : 0;
  const bool tail = n < win || (n - win) % step > 0;
  return full + (tail ? 1 : 0);
}
int rvd_frames_per_window(const rvd_engine* e) { return e ? e->p3 : 0; }

// the file is in e->pcm (int16, n samples at the model's rate): waveform, SincNet front end, embedding fbank
static int prepare_audio(rvd_engine* e, int64_t n);

int rvd_upload_pcm(rvd_engine* e, const int16_t* pcm, int64_t n) {
  if (!e || !pcm || n <= 0) { set_error("rvd_upload_pcm: null or empty audio"); return E_ARG; }
  if (!e->finalized) { set_error("rvd_upload_pcm: finalize the model first"); return E_STATE; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  RVD_TRY(e->pcm.ensure((size_t)n * 2));
  { DScope sc(e, "h2d");
    RVB_HIP_CHECK(hipMemcpyAsync(e->pcm.p, pcm, (size_t)n * 2, hipMemcpyHostToDevice, e->stream)); }
  return prepare_audio(e, n);
}

int rvd_rerun_resident(rvd_engine* e) {
  if (!e) { set_error("rvd_rerun_resident: null engine"); return E_ARG; }
  if (e->n_samples <= 0 || !e->pcm.p) { set_error("rvd_rerun_resident: upload audio first"); return E_STATE; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  return prepare_audio(e, e->n_samples);
}

// pyannote's Audio resamples every file to the model's rate with torchaudio.functional.resample (its defaults: the kernel
// of the ASR front end, rvb_upload_pcm_rate).  The resampled waveform comes back as int16 -- both networks of this engine start
// from int16 PCM, and the host shards a recording by samples at the model's rate -- i.e. rounded by at most half an LSB (-96 dB).
// out == NULL: only *n_out (= ceil(n * rate_out / rate_in)) is written.
int rvd_resample_pcm(rvd_engine* e, const int16_t* pcm, int64_t n, int sample_rate, int16_t* out, int64_t* n_out) {
  if (!e || !pcm || n <= 0 || !n_out) { set_error("rvd_resample_pcm: null or empty audio"); return E_ARG; }
  if (sample_rate < 1000 || sample_rate > 384000) { set_error("rvd_resample_pcm: sample rate out of range"); return E_ARG; }
  const int target = e->cfg.sample_rate;
  std::vector<float> ker;
  int orig, nw, width, K;
  resample_taps(sample_rate, target, &ker, &orig, &nw, &width, &K);
  const int64_t m = ((int64_t)nw * n + orig - 1) / orig;
  if (!out) { *n_out = m; return OK; }
  if (*n_out < m) { set_error("rvd_resample_pcm: output buffer too small"); return E_ARG; }
  *n_out = m;
  RVB_HIP_CHECK(hipSetDevice(e->device));
  DevBuf src, taps, res, dst;
  int r = src.ensure((size_t)n * 2 + 16);
  if (r == OK) r = taps.ensure(ker.size() * 4);
  if (r == OK) r = res.ensure((size_t)m * 4);
  if (r == OK) r = dst.ensure((size_t)m * 2);
  if (r == OK && hipMemcpyAsync(src.p, pcm, (size_t)n * 2, hipMemcpyHostToDevice, e->stream) != hipSuccess) r = E_HIP;
  if (r == OK && hipMemcpyAsync(taps.p, ker.data(), ker.size() * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess) r = E_HIP;
  if (r == OK) { DScope sc(e, "resample"); r = resample(e->stream, src.as<int16_t>(), n, taps.as<float>(), orig, nw, width, K, res.as<float>(), m); }
  if (r == OK) r = round_to_i16(e->stream, res.as<float>(), m, dst.as<int16_t>());
  if (r == OK && hipMemcpyAsync(out, dst.p, (size_t)m * 2, hipMemcpyDeviceToHost, e->stream) != hipSuccess) r = E_HIP;
  if (hipStreamSynchronize(e->stream) != hipSuccess && r == OK) r = E_HIP;
  src.release(); taps.release(); res.release(); dst.release();
  if (r == E_HIP) set_error("rvd_resample_pcm: device copy or kernel failed");
  return r;
}

static int prepare_audio(rvd_engine* e, int64_t n) {
  const rvd_model_cfg& c = e->cfg;
  e->n_samples = n;
  e->n_windows = rvd_num_windows(e, n);
  e->n_pad = (e->n_windows - 1) * c.step_samples + c.window_samples;
  e->craw_frames = (e->n_pad - SINC_K) / SINC_STRIDE + 1;
  RVD_TRY(e->wave.ensure((size_t)e->n_pad * 4));
  RVD_TRY(e->craw.ensure((size_t)e->craw_frames * c.sinc_filters * dt_size(e->dtype)));
  { DScope sc(e, "pcm_to_float");
    RVD_TRY(pcm_to_float(e->stream, e->pcm.as<int16_t>(), n, e->wave.as<float>(), e->n_pad)); }
  { DScope sc(e, "sinc_conv", 2.0 * (double)e->craw_frames * c.sinc_filters * SINC_K);
    RVD_TRY(sinc_conv(e->stream, e->dtype, e->wave.as<float>(), e->filt.as<float>(), e->craw.p, e->craw_frames, c.sinc_filters,
                      SINC_K, SINC_STRIDE)); }
  if (e->has_emb) {
    // hamming log-mel of the zero-extended file, shared by all windows (frame j of window w = frame w*step/160 + j)
    e->emb_frames = (e->n_pad - FB_WIN) / FB_SHIFT + 1;
    RVD_TRY(e->pcm_pad.ensure((size_t)e->n_pad * 2));
    RVD_TRY(e->emb_fb.ensure((size_t)e->emb_frames * FB_MEL * 4));
    RVB_HIP_CHECK(hipMemsetAsync(e->pcm_pad.p, 0, (size_t)e->n_pad * 2, e->stream));
    RVB_HIP_CHECK(hipMemcpyAsync(e->pcm_pad.p, e->pcm.p, (size_t)n * 2, hipMemcpyDeviceToDevice, e->stream));
    DScope sc(e, "emb_fbank");
    FbankTables t{e->fb_window.as<float>(), e->fb_twiddle.as<float>(), e->fb_melw.as<float>(), e->fb_lo.as<int>(), e->fb_hi.as<int>()};
    RVD_TRY(fbank(e->stream, e->pcm_pad.as<int16_t>(), e->emb_frames, e->emb_fb.as<float>(), t));
    // per-window CMN means for every window of the file (one block each)
    RVD_TRY(e->e_mean.ensure((size_t)e->n_windows * FB_MEL * 4));
    RVD_TRY(emb_window_mean(e->stream, e->emb_fb.as<float>(), nullptr, (int)e->n_windows, c.step_samples / FB_SHIFT, e->nfr, e->e_mean.as<float>()));
  }
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  return OK;
}

int rvd_segment(rvd_engine* e, int64_t first_window, int n_windows, float* logp_out) {
  if (!e) { set_error("rvd_segment: null engine"); return E_ARG; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  return segment_impl(e, first_window, n_windows, logp_out);
}

int rvd_get_classes(rvd_engine* e, uint8_t* out) {
  if (!e || !out) { set_error("rvd_get_classes: null argument"); return E_ARG; }
  if (e->last_W <= 0) { set_error("rvd_get_classes: no rvd_segment call yet"); return E_STATE; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  RVB_HIP_CHECK(hipMemcpy(out, e->cls.p, (size_t)e->last_W * e->p3, hipMemcpyDeviceToHost));
  return OK;
}

int rvd_get_tap(rvd_engine* e, const char* name, float* out) {
  if (!e || !name || !out) { set_error("rvd_get_tap: null argument"); return E_ARG; }
  if (e->last_W <= 0) { set_error("rvd_get_tap: no rvd_segment call yet"); return E_STATE; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  const int64_t R3 = (int64_t)e->last_W * e->p3;
  if (!strcmp(name, "sincnet")) return fetch_T(e, e->a3.p, R3, e->cpad, e->cfg.sinc_channels, out);
  if (!strcmp(name, "lstm")) return fetch_T(e, e->last_lstm, R3, 2 * e->cfg.lstm_hidden, 2 * e->cfg.lstm_hidden, out);
  set_error(std::string("rvd_get_tap: unknown tap ") + name);
  return E_ARG;
}

int rvd_embed(rvd_engine* e, const int64_t* win, const float* mask, int n, float* emb_out) {
  if (!e || !win || !mask || !emb_out || n < 0) { set_error("rvd_embed: null argument"); return E_ARG; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  return embed_impl(e, win, mask, n, emb_out);
}
int rvd_get_emb_fbank(rvd_engine* e, int64_t window, float* out, int32_t* n_frames) {
  if (!e || !out) { set_error("rvd_get_emb_fbank: null argument"); return E_ARG; }
  if (!e->has_emb || e->n_windows == 0) { set_error("rvd_get_emb_fbank: needs an embedding model and uploaded audio"); return E_STATE; }
  if (window < 0 || window >= e->n_windows) { set_error("rvd_get_emb_fbank: window outside the uploaded audio"); return E_ARG; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  const int64_t f0 = window * (e->cfg.step_samples / FB_SHIFT);
  RVB_HIP_CHECK(hipMemcpy(out, e->emb_fb.as<float>() + f0 * FB_MEL, (size_t)e->nfr * FB_MEL * 4, hipMemcpyDeviceToHost));
  if (n_frames) *n_frames = e->nfr;
  return OK;
}

int rvd_centroid_linkage(rvd_engine* e, const double* X, int n, int d, double* Z) {
  if (!e || !X || !Z || n < 1 || d < 1) { set_error("rvd_centroid_linkage: bad argument"); return E_ARG; }
  if (n > 46000) { set_error("rvd_centroid_linkage: more than 46000 points"); return E_UNSUPPORTED; }   // n^2 index and uint16 sizes
  if (n == 1) return OK;
  RVB_HIP_CHECK(hipSetDevice(e->device));
  DevBuf dX, dD, dI, dM, dZ, dC;
  int rc = OK;
  do {
    if ((rc = dC.ensure(4096)) != OK) break;
    if ((rc = dX.ensure((size_t)n * d * 8)) != OK) break;
    if ((rc = dD.ensure((size_t)n * n * 8)) != OK) break;
    if ((rc = dI.ensure((size_t)3 * n * 4)) != OK) break;
    if ((rc = dM.ensure((size_t)n * 8)) != OK) break;
    if ((rc = dZ.ensure((size_t)(n - 1) * 4 * 8)) != OK) break;
    std::vector<int> init((size_t)3 * n);     // [cluster_id | neighbor | size as uint16]
    for (int i = 0; i < n; ++i) { init[i] = i; init[n + i] = -1; ((uint16_t*)&init[2 * (size_t)n])[i] = 1; }
    std::vector<double> inf(n, INFINITY);
    (void)hipMemcpyAsync(dX.p, X, (size_t)n * d * 8, hipMemcpyHostToDevice, e->stream);
    (void)hipMemcpyAsync(dI.p, init.data(), init.size() * 4, hipMemcpyHostToDevice, e->stream);
    (void)hipMemcpyAsync(dM.p, inf.data(), (size_t)n * 8, hipMemcpyHostToDevice, e->stream);
    (void)hipStreamSynchronize(e->stream);
    {
      DScope sc(e, "linkage");
      rc = centroid_linkage(e->stream, dX.as<double>(), n, d, dD.as<double>(), (uint16_t*)(dI.as<int>() + 2 * n), dI.as<int>(),
                            dI.as<int>() + n, dM.as<double>(), dZ.as<double>(), dC.p, e->linkage_workgroups);
    }
    if (rc != OK) break;
    if (hipMemcpyAsync(Z, dZ.p, (size_t)(n - 1) * 4 * 8, hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess) {
      set_error(std::string("rvd_centroid_linkage: ") + hipGetErrorString(hipGetLastError()));
      rc = E_HIP;
      break;
    }
    { double retries = 0.0;
      (void)hipMemcpy(&retries, dM.as<double>() + (n - 1), 8, hipMemcpyDeviceToHost);
      if (retries < 0.0) { set_error("rvd_centroid_linkage: the merge loop found no candidate pair (non-finite embeddings?)"); rc = E_ARG; break; }
      e->prof["linkage"].flops += retries; }      // reported through rvd_get_timing("linkage").flops
    // slots -> scipy cluster ids: the merged cluster lives on in slot y under the new id n + k
    std::vector<int> cid(n);
    for (int i = 0; i < n; ++i) cid[i] = i;
    for (int k = 0; k < n - 1; ++k) {
      const int x = (int)Z[4 * (size_t)k], y = (int)Z[4 * (size_t)k + 1];
      const int ix = cid[x], iy = cid[y];
      Z[4 * (size_t)k] = ix < iy ? ix : iy;
      Z[4 * (size_t)k + 1] = ix < iy ? iy : ix;
      cid[y] = n + k;
    }
  } while (0);
  dX.release(); dD.release(); dI.release(); dM.release(); dZ.release(); dC.release();
  return rc;
}

int rvd_get_emb_fp8(rvd_engine* e, int32_t* state, float* scales, int32_t* n, uint32_t* clipped) {
  if (!e) { set_error("rvd_get_emb_fp8: null engine"); return E_ARG; }
  if (state) *state = e->emb_fp8 ? e->emb_f8_state : 0;
  if (n) {
    const int cap = *n;
    *n = 32;
    if (scales && e->emb_f8_state == 2)
      for (int i = 0; i < 32 && i < cap; ++i) scales[i] = e->scale8[i];
  }
  if (clipped) {
    *clipped = 0;
    if (e->d_sat8.p) {
      RVB_HIP_CHECK(hipSetDevice(e->device));
      RVB_HIP_CHECK(hipMemcpyAsync(clipped, e->d_sat8.p, 4, hipMemcpyDeviceToHost, e->stream));
      RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
    }
  }
  return OK;
}

int rvd_set_emb_fp8_scales(rvd_engine* e, const float* scales, int32_t n) {
  if (!e || !scales || n != 32) { set_error("rvd_set_emb_fp8_scales: engine, 32 scales"); return E_ARG; }
  if (!e->emb_fp8) { set_error("rvd_set_emb_fp8_scales: not an fp8 engine"); return E_STATE; }
  for (int i = 0; i < 32; ++i)
    if (!(scales[i] > 0.f) || !std::isfinite(scales[i])) { set_error("rvd_set_emb_fp8_scales: scales must be positive and finite"); return E_ARG; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  RVD_TRY(e->d_sat8.ensure(4));
  RVB_HIP_CHECK(hipMemsetAsync(e->d_sat8.p, 0, 4, e->stream));
  e->scale8.assign(scales, scales + 32);
  e->emb_f8_state = 2;
  return OK;
}

int rvd_set_linkage_workgroups(rvd_engine* e, int workgroups) {
  if (!e || !(workgroups == 0 || workgroups == 1 || workgroups == 2 || workgroups == 4 || workgroups == 8 || workgroups == 16)) {
    set_error("rvd_set_linkage_workgroups: 0 (default), 1, 2, 4, 8 or 16");
    return E_ARG;
  }
  e->linkage_workgroups = workgroups;
  return OK;
}

int rvd_set_profiling(rvd_engine* e, int enabled) { if (!e) return E_ARG; drain(e); e->profiling = enabled != 0; return OK; }
int rvd_reset_timings(rvd_engine* e) { if (!e) return E_ARG; drain(e); e->prof.clear(); return OK; }
int rvd_get_timing(rvd_engine* e, const char* name, double* ms, double* flops, int64_t* launches) {
  if (!e || !name) return E_ARG;
  drain(e);
  auto it = e->prof.find(name);
  if (it == e->prof.end()) { if (ms) *ms = 0; if (flops) *flops = 0; if (launches) *launches = 0; return OK; }
  if (ms) *ms = it->second.ms;
  if (flops) *flops = it->second.flops;
  if (launches) *launches = it->second.launches;
  return OK;
}

}  // extern "C"
