// C-ABI collectives of librvb: the chunk-sharded path has ONE exchange step, the all-gather of the packed
// per-chunk results (~1.3 KB per chunk, latency-bound on xGMI).  This file binds RCCL directly -- ncclGetUniqueId /
// ncclCommInitRank / ncclAllGather resolved with dlopen at first use, so librvb has no link-time dependency on a particular
// librccl (PyTorch ships its own copy; whichever is already in the process is used) -- for hosts that do not want
// torch.distributed in the loop.  The 128-byte unique id travels by whatever side channel the host has (a file, MPI, a TCP
// socket; reverb_amd/dist.py's RvbComm broadcasts it over an existing torch.distributed group or takes it from a file).
#include <dlfcn.h>

#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include <cstring>

#include "engine.h"

#define RVB_TRY_RC(expr) do { int _r = (expr); if (_r != rvb::OK) return _r; } while (0)

namespace {

typedef int (*GetUniqueIdFn)(void*);
typedef int (*AllGatherFn)(const void*, void*, size_t, int /* ncclDataType_t */, void*, hipStream_t);
typedef int (*CommDestroyFn)(void*);
typedef int (*CommAbortFn)(void*);
typedef const char* (*GetErrorStringFn)(int);

}  // namespace
struct Id128 { char b[128]; };
namespace {

struct Rccl {
  void* lib = nullptr;
  GetUniqueIdFn get_id = nullptr;
  int (*init_rank)(void**, int, Id128, int) = nullptr;
  AllGatherFn all_gather = nullptr;
  CommDestroyFn destroy = nullptr;
  CommAbortFn abort = nullptr;           // optional (ncclCommAbort): tears a communicator down without waiting for its peers
  GetErrorStringFn err = nullptr;
};
Rccl g_rccl;
std::once_flag g_once;
std::string g_load_error;      // written once inside call_once (dlerror() reports a failure only once, to its first caller)

int load_rccl() {
  std::call_once(g_once, [] {
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (g_rccl.lib) break;
      const char* m = dlerror();
      g_load_error = m ? m : "dlopen failed";
    }
    if (!g_rccl.lib) return;
    g_load_error = "symbols missing";
    g_rccl.get_id = (GetUniqueIdFn)dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.init_rank = (int (*)(void**, int, Id128, int))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.all_gather = (AllGatherFn)dlsym(g_rccl.lib, "ncclAllGather");
    g_rccl.destroy = (CommDestroyFn)dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.abort = (CommAbortFn)dlsym(g_rccl.lib, "ncclCommAbort");
    g_rccl.err = (GetErrorStringFn)dlsym(g_rccl.lib, "ncclGetErrorString");
  });
  if (!g_rccl.lib || !g_rccl.get_id || !g_rccl.init_rank || !g_rccl.all_gather || !g_rccl.destroy) {
    rvb::set_error("RCCL (librccl.so) could not be loaded: " + g_load_error);
    return rvb::E_UNSUPPORTED;
  }
  return rvb::OK;
}
int nccl_fail(const char* what, int rc) {
  rvb::set_error(std::string(what) + ": " + (g_rccl.err ? g_rccl.err(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
  return rvb::E_HIP;
}

}  // namespace

using namespace rvb;

extern "C" {

int rvb_comm_unique_id(void* id128) {
  if (!id128) { set_error("rvb_comm_unique_id: null argument"); return E_ARG; }
  RVB_TRY_RC(load_rccl());
  const int rc = g_rccl.get_id(id128);
  return rc == 0 ? OK : nccl_fail("ncclGetUniqueId", rc);
}

// ---- stand-alone communicator: one per process / GPU, shared by the ASR engine and the diarization engine of that rank
struct rvb_comm {
  int device = 0, world = 1, rank = 0;
  void* nccl = nullptr;
  hipStream_t stream = nullptr;
  rvb::DevBuf send, recv;
  double timeout_s = 0.0;        // 0 = wait for ever (rvb_comm_set_timeout)
  bool dead = false;             // a collective timed out: the communicator was aborted and refuses further calls
  hipEvent_t done = nullptr;
};

// Wait for everything enqueued on the communicator's stream -- for ever, or (rvb_comm_set_timeout) until the deadline: a peer
// that died or hangs never joins the collective and the kernel RCCL launched would spin for ever.  On a timeout the
// communicator is aborted (ncclCommAbort where the library has it; its stream is abandoned, not synchronised) and marked
// dead: RVB_E_TIMEOUT now and RVB_E_STATE for every later call, so that the host can switch to its fall-back exchange
// (reverb_amd/dist.py: the rendezvous store) instead of hanging the job.
static int comm_wait(rvb_comm* c, const char* what) {
  if (c->timeout_s <= 0.0) { RVB_HIP_CHECK(hipStreamSynchronize(c->stream)); return rvb::OK; }
  if (!c->done) RVB_HIP_CHECK(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
  RVB_HIP_CHECK(hipEventRecord(c->done, c->stream));
  const auto start = std::chrono::steady_clock::now();
  const auto deadline = start + std::chrono::duration<double>(c->timeout_s);
  for (;;) {
    const hipError_t q = hipEventQuery(c->done);
    if (q == hipSuccess) return rvb::OK;
    if (q != hipErrorNotReady) { rvb::set_error(std::string(what) + ": " + hipGetErrorString(q)); return rvb::E_HIP; }
    const auto now = std::chrono::steady_clock::now();
    if (now > deadline) break;
    // a healthy collective of this size completes in tens of microseconds: poll without sleeping for the first 500 us so that
    // a timeout costs the barriers around a timed region nothing, then back off
    if (now - start > std::chrono::microseconds(500)) std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  c->dead = true;
  if (g_rccl.abort && c->nccl) {
    // ncclCommAbort makes the communicator's kernels leave their wait loops, so the stream drains: wait for that, or the copies
    // queued behind the collective could still land in host buffers the caller is about to free
    (void)g_rccl.abort(c->nccl);
    c->nccl = nullptr;
    (void)hipStreamSynchronize(c->stream);
  }
  rvb::set_error(std::string(what) + ": no completion within " + std::to_string(c->timeout_s) + " s (a peer is missing); communicator aborted");
  return rvb::E_TIMEOUT;
}
#define RVB_COMM_ALIVE(c, what) do { if ((c)->dead) { set_error(std::string(what) + ": the communicator was aborted after a timeout"); return E_STATE; } } while (0)

int rvb_comm_create(int device, int world, int rank, const void* id128, rvb_comm** out) {
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) { set_error("rvb_comm_create: bad argument"); return E_ARG; }
  RVB_TRY_RC(load_rccl());
  RVB_HIP_CHECK(hipSetDevice(device));
  Id128 id;
  memcpy(id.b, id128, 128);
  void* nccl = nullptr;
  const int rc = g_rccl.init_rank(&nccl, world, id, rank);
  if (rc != 0) return nccl_fail("ncclCommInitRank", rc);
  rvb_comm* c = new rvb_comm();
  c->device = device; c->world = world; c->rank = rank; c->nccl = nccl;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    g_rccl.destroy(nccl); delete c; set_error("rvb_comm_create: hipStreamCreate failed"); return E_HIP;
  }
  *out = c;
  return OK;
}

// every rank contributes `bytes` bytes (host memory); `recv` (host, world * bytes) receives them in rank order
int rvb_comm_allgather(rvb_comm* c, const void* send, int64_t bytes, void* recv) {
  if (!c || !send || !recv || bytes <= 0) { set_error("rvb_comm_allgather: bad argument"); return E_ARG; }
  RVB_COMM_ALIVE(c, "rvb_comm_allgather");
  RVB_HIP_CHECK(hipSetDevice(c->device));
  int r = c->send.ensure((size_t)bytes);
  if (r != OK) return r;
  r = c->recv.ensure((size_t)bytes * c->world);
  if (r != OK) return r;
  RVB_HIP_CHECK(hipMemcpyAsync(c->send.p, send, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
  const int rc = g_rccl.all_gather(c->send.p, c->recv.p, (size_t)bytes, 0 /* ncclInt8 */, c->nccl, c->stream);
  if (rc != 0) return nccl_fail("ncclAllGather", rc);
  RVB_HIP_CHECK(hipMemcpyAsync(recv, c->recv.p, (size_t)bytes * c->world, hipMemcpyDeviceToHost, c->stream));
  return comm_wait(c, "rvb_comm_allgather");
}

int rvb_comm_set_timeout(rvb_comm* c, double seconds) {
  if (!c || seconds < 0.0) { set_error("rvb_comm_set_timeout: bad argument"); return E_ARG; }
  if (seconds > 0.0) {
    // without ncclCommAbort a timed-out collective cannot be torn down: its kernel and the copy into the caller's host
    // buffer would stay queued and could land after the caller has freed that buffer -- refuse the timeout
    RVB_TRY_RC(load_rccl());
    if (!g_rccl.abort) { set_error("rvb_comm_set_timeout: this librccl has no ncclCommAbort; collectives cannot be given a deadline"); return E_UNSUPPORTED; }
  }
  c->timeout_s = seconds;
  return OK;
}

// barrier and max-reduction as 8-byte all-gathers on the same communicator: the launcher needs nothing else of a process
// group (bench.py's step timing: barrier, K steps, barrier, maximum over the ranks)
int rvb_comm_barrier(rvb_comm* c) {
  if (!c) { set_error("rvb_comm_barrier: null communicator"); return E_ARG; }
  double mine = 0.0;
  std::vector<double> all((size_t)c->world);
  return rvb_comm_allgather(c, &mine, 8, all.data());
}
int rvb_comm_max_f64(rvb_comm* c, double* value) {
  if (!c || !value) { set_error("rvb_comm_max_f64: null argument"); return E_ARG; }
  std::vector<double> all((size_t)c->world);
  const int r = rvb_comm_allgather(c, value, 8, all.data());
  if (r != OK) return r;
  double m = all[0];
  for (double v : all) m = v > m ? v : m;
  *value = m;
  return OK;
}

// The posterior exchange of SURVEY.md 8(e), device to device: every rank's per-frame top-k CTC log-probs and token ids of the
// last rvb_encode ([B, T, k] fp32 and int32, already in HBM) are gathered straight from the engine's buffers into the
// communicator's receive buffer -- [world][vals] followed by [world][ids] -- with no host staging.  `host_out` (nullable,
// 2 * world * bytes_per_rank_per_array) receives a copy for a host that searches centrally; *bytes_per_rank = B * T * k * 8.
// Every rank must hold the same batch shape.
int rvb_comm_allgather_topk(rvb_comm* c, rvb_engine* e, void* host_out, int64_t* bytes_per_rank) {
  if (!c || !e) { set_error("rvb_comm_allgather_topk: null argument"); return E_ARG; }
  RVB_COMM_ALIVE(c, "rvb_comm_allgather_topk");
  if (e->B <= 0 || !e->topv.p || !e->topi.p) { set_error("rvb_comm_allgather_topk: no encoded batch"); return E_STATE; }
  if (e->device != c->device) { set_error("rvb_comm_allgather_topk: engine and communicator are on different devices"); return E_ARG; }
  RVB_HIP_CHECK(hipSetDevice(c->device));
  const size_t part = (size_t)e->B * e->T2 * e->beam * 4;
  int r = c->recv.ensure(2 * part * c->world);
  if (r != OK) return r;
  if (!c->done) RVB_HIP_CHECK(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
  RVB_HIP_CHECK(hipEventRecord(c->done, e->stream));                 // the encoder's top-k kernels come first
  RVB_HIP_CHECK(hipStreamWaitEvent(c->stream, c->done, 0));
  int rc = g_rccl.all_gather(e->topv.p, c->recv.p, part, 0 /* ncclInt8 */, c->nccl, c->stream);
  if (rc == 0) rc = g_rccl.all_gather(e->topi.p, (char*)c->recv.p + part * c->world, part, 0, c->nccl, c->stream);
  if (rc != 0) return nccl_fail("ncclAllGather", rc);
  if (host_out) RVB_HIP_CHECK(hipMemcpyAsync(host_out, c->recv.p, 2 * part * c->world, hipMemcpyDeviceToHost, c->stream));
  if (bytes_per_rank) *bytes_per_rank = (int64_t)(2 * part);
  return comm_wait(c, "rvb_comm_allgather_topk");
}

// The collective alone, device buffer to device buffer: `iters` all-gathers of `bytes` bytes per rank between two HIP
// events on the communicator's stream (after one untimed call).  The send buffer holds the rank number in every byte and the
// last gather is checked on the host, so a figure is never reported for an exchange that did not happen.
int rvb_comm_time_allgather(rvb_comm* c, int64_t bytes, int iters, double* avg_ms) {
  if (!c || bytes <= 0 || iters < 1 || !avg_ms) { set_error("rvb_comm_time_allgather: bad argument"); return E_ARG; }
  RVB_COMM_ALIVE(c, "rvb_comm_time_allgather");
  RVB_HIP_CHECK(hipSetDevice(c->device));
  int r = c->send.ensure((size_t)bytes);
  if (r != OK) return r;
  r = c->recv.ensure((size_t)bytes * c->world);
  if (r != OK) return r;
  RVB_HIP_CHECK(hipMemsetAsync(c->send.p, c->rank & 0xff, (size_t)bytes, c->stream));
  RVB_HIP_CHECK(hipMemsetAsync(c->recv.p, 0xee, (size_t)bytes * c->world, c->stream));
  hipEvent_t t0, t1;
  RVB_HIP_CHECK(hipEventCreate(&t0));
  RVB_HIP_CHECK(hipEventCreate(&t1));
  int rc = g_rccl.all_gather(c->send.p, c->recv.p, (size_t)bytes, 0 /* ncclInt8 */, c->nccl, c->stream);
  if (rc == 0) {
    (void)hipEventRecord(t0, c->stream);
    for (int i = 0; i < iters && rc == 0; ++i) rc = g_rccl.all_gather(c->send.p, c->recv.p, (size_t)bytes, 0, c->nccl, c->stream);
    (void)hipEventRecord(t1, c->stream);
  }
  const hipError_t he = hipStreamSynchronize(c->stream);
  float ms = 0.f;
  if (rc == 0 && he == hipSuccess) (void)hipEventElapsedTime(&ms, t0, t1);
  (void)hipEventDestroy(t0);
  (void)hipEventDestroy(t1);
  if (rc != 0) return nccl_fail("ncclAllGather", rc);
  RVB_HIP_CHECK(he);
  // first and last byte of every rank's slot
  for (int k = 0; k < c->world; ++k) {
    unsigned char probe[2] = {0, 0};
    RVB_HIP_CHECK(hipMemcpy(&probe[0], (const char*)c->recv.p + (size_t)k * bytes, 1, hipMemcpyDeviceToHost));
    RVB_HIP_CHECK(hipMemcpy(&probe[1], (const char*)c->recv.p + (size_t)(k + 1) * bytes - 1, 1, hipMemcpyDeviceToHost));
    if (probe[0] != (k & 0xff) || probe[1] != (k & 0xff)) {
      set_error("rvb_comm_time_allgather: slot " + std::to_string(k) + " does not hold rank " + std::to_string(k) + "'s bytes");
      return E_STATE;
    }
  }
  *avg_ms = (double)ms / iters;
  return OK;
}

int rvb_comm_free(rvb_comm* c) {
  if (!c) return OK;
  (void)hipSetDevice(c->device);
  // a dead communicator whose library has no ncclCommAbort may never drain: its stream and buffers are then abandoned
  const bool drained = !c->dead || (g_rccl.abort != nullptr);
  if (c->stream && drained) (void)hipStreamSynchronize(c->stream);
  if (c->nccl) g_rccl.destroy(c->nccl);
  if (c->done) (void)hipEventDestroy(c->done);
  if (drained) { c->send.release(); c->recv.release(); }
  if (c->stream && drained) (void)hipStreamDestroy(c->stream);
  delete c;
  return OK;
}

// ---- the same collective bound to an ASR engine (its stream, its buffers)
int rvb_comm_init(rvb_engine* e, int world, int rank, const void* id128) {
  if (!e || !id128 || world < 1 || rank < 0 || rank >= world) { set_error("rvb_comm_init: bad argument"); return E_ARG; }
  if (e->comm) { set_error("rvb_comm_init: the engine already has a communicator"); return E_STATE; }
  RVB_TRY_RC(load_rccl());
  RVB_HIP_CHECK(hipSetDevice(e->device));
  Id128 id;
  memcpy(id.b, id128, 128);
  void* comm = nullptr;
  const int rc = g_rccl.init_rank(&comm, world, id, rank);
  if (rc != 0) return nccl_fail("ncclCommInitRank", rc);
  e->comm = comm; e->comm_world = world; e->comm_rank = rank;
  return OK;
}

// every rank contributes `bytes` bytes (host memory); `recv` (host, world * bytes) receives them in rank order
int rvb_allgather_results(rvb_engine* e, const void* send, int64_t bytes, void* recv) {
  if (!e || !send || !recv || bytes <= 0) { set_error("rvb_allgather_results: bad argument"); return E_ARG; }
  if (!e->comm) { set_error("rvb_allgather_results before rvb_comm_init"); return E_STATE; }
  RVB_HIP_CHECK(hipSetDevice(e->device));
  int r = e->comm_send.ensure((size_t)bytes);
  if (r != OK) return r;
  r = e->comm_recv.ensure((size_t)bytes * e->comm_world);
  if (r != OK) return r;
  RVB_HIP_CHECK(hipMemcpyAsync(e->comm_send.p, send, (size_t)bytes, hipMemcpyHostToDevice, e->stream));
  const int rc = g_rccl.all_gather(e->comm_send.p, e->comm_recv.p, (size_t)bytes, 0 /* ncclInt8 */, e->comm, e->stream);
  if (rc != 0) return nccl_fail("ncclAllGather", rc);
  RVB_HIP_CHECK(hipMemcpyAsync(recv, e->comm_recv.p, (size_t)bytes * e->comm_world, hipMemcpyDeviceToHost, e->stream));
  RVB_HIP_CHECK(hipStreamSynchronize(e->stream));
  return OK;
}

int rvb_comm_destroy(rvb_engine* e) {
  if (!e) { set_error("rvb_comm_destroy: null engine"); return E_ARG; }
  if (e->comm) {
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    g_rccl.destroy(e->comm);
    e->comm = nullptr;
  }
  e->comm_send.release(); e->comm_recv.release();
  return OK;
}

}  // extern "C"
