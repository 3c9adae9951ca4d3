// tor_multi.cpp -- everything between finished device rows and the caller's canvas:
//   download_rows        D2H through pinned staging, worker threads fill the pageable canvas while later chunks
//                        are still on the bus (the reference's render() writes canvas.pixels, render.nim:68)
//   render_multi_device  tor_render_opt with a device list: render.nim:55 `parallelFor row` across GPUs -- one
//                        host thread + stream per device, row-cyclic shards, then the framebuffer gather
//                        (single-process RCCL over xGMI | peer copies | straight to the host per device)
//   tor_comm_* / tor_render_gather_device   the same gather for one-process-per-GPU hosts (ncclCommInitRank)
// RCCL is loaded with dlopen on first use: hosts that never name a second device do not need it.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "tor_context.hpp"
#include "tor_knobs.hpp"

namespace tor {

// ---------------------------------------------------------------------------------------------
// RCCL, bound at run time
// ---------------------------------------------------------------------------------------------
namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // by SONAME first: inside a PyTorch process this is the copy torch already loaded
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : names) {
      api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.handle) break;
    }
    if (!api.handle) {
      api.error = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : "");
      return;
    }
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(api.handle, name);
      if (!p && api.error.empty()) api.error = std::string("librccl: missing symbol ") + name;
      return p;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.CommAbort = (decltype(api.CommAbort))sym("ncclCommAbort");
    api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  });
  return (api.handle && api.error.empty()) ? &api : nullptr;
}

int fail_rccl(ncclResult_t r, const char* what) {
  RcclApi* a = rccl();
  return fail(TOR_ERR_HIP, std::string(what) + ": " + (a && a->GetErrorString ? a->GetErrorString(r) : "RCCL error"));
}

#define RCCL_TRY(expr)                                        \
  do {                                                        \
    ncclResult_t r__ = (expr);                                \
    if (r__ != ncclSuccess) return fail_rccl(r__, #expr);      \
  } while (0)

static_assert(sizeof(ncclUniqueId) == 128, "tor_comm_unique_id hands out 128 bytes");

}  // namespace

struct RcclComm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

// ---------------------------------------------------------------------------------------------
// device rows -> host canvas
// ---------------------------------------------------------------------------------------------
int download_rows(TorContext* ctx, const void* d_src, int64_t n_rows, size_t row_bytes, const int32_t* rows, char* dst,
                  hipStream_t stream) {
  HIP_TRY(hipSetDevice(ctx->device));
  if (n_rows <= 0 || row_bytes == 0) {
    HIP_TRY(hipStreamSynchronize(stream));
    return TOR_OK;
  }
  const size_t total = (size_t)n_rows * row_bytes;
  HIP_TRY(ctx->staging.ensure(total));
  size_t chunk_target = (size_t)2 << 20;  // measured on C2 (tools/host_canvas_tune.py): 8 threads x 2 MiB
  int64_t rows_per_chunk = (int64_t)(chunk_target / row_bytes);
  if (rows_per_chunk < 1) rows_per_chunk = 1;
  const int64_t n_chunks = (n_rows + rows_per_chunk - 1) / rows_per_chunk;
  while ((int64_t)ctx->chunk_events.size() < n_chunks) {
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    ctx->chunk_events.push_back(ev);
  }
  char* stage = (char*)ctx->staging.ptr;
  for (int64_t c = 0; c < n_chunks; ++c) {
    const int64_t r0 = c * rows_per_chunk, r1 = (r0 + rows_per_chunk < n_rows) ? r0 + rows_per_chunk : n_rows;
    HIP_TRY(hipMemcpyAsync(stage + (size_t)r0 * row_bytes, (const char*)d_src + (size_t)r0 * row_bytes,
                           (size_t)(r1 - r0) * row_bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipEventRecord(ctx->chunk_events[(size_t)c], stream));
  }
  int n_workers = 8;
  if (n_workers < 1) n_workers = 1;
  if (n_workers > n_chunks) n_workers = (int)n_chunks;
  std::atomic<int> first_error{(int)hipSuccess};
  auto work = [&](int w) {
    if (w != 0) (void)hipSetDevice(ctx->device);
    for (int64_t c = w; c < n_chunks; c += n_workers) {
      const hipError_t e = hipEventSynchronize(ctx->chunk_events[(size_t)c]);
      if (e != hipSuccess) {
        int expected = (int)hipSuccess;
        first_error.compare_exchange_strong(expected, (int)e);
        return;
      }
      const int64_t r0 = c * rows_per_chunk, r1 = (r0 + rows_per_chunk < n_rows) ? r0 + rows_per_chunk : n_rows;
      if (!rows) {
        std::memcpy(dst + (size_t)r0 * row_bytes, stage + (size_t)r0 * row_bytes, (size_t)(r1 - r0) * row_bytes);
      } else {
        for (int64_t r = r0; r < r1; ++r)
          std::memcpy(dst + (size_t)rows[r] * row_bytes, stage + (size_t)r * row_bytes, row_bytes);
      }
    }
  };
  std::vector<std::thread> pool;
  for (int w = 1; w < n_workers; ++w) pool.emplace_back(work, w);
  work(0);
  for (std::thread& t : pool) t.join();
  if (first_error.load() != (int)hipSuccess) return fail_hip((hipError_t)first_error.load(), "download_rows");
  return TOR_OK;
}

// ---------------------------------------------------------------------------------------------
// tor_render_opt with a device list
// ---------------------------------------------------------------------------------------------
namespace {

std::mutex g_comm_mutex;
std::map<std::vector<int>, std::vector<ncclComm_t>> g_single_process_comms;  // per device list (ncclCommInitAll)
struct CommSetup;
std::map<std::vector<int>, std::shared_ptr<CommSetup>> g_comm_pending;       // creations in flight (helper threads), guarded by g_comm_mutex
// lists whose creation passed its deadline, and when: such a list is not tried again for kCommRetryAfterS seconds (a hung
// creation leaves a helper thread behind per attempt), then it gets another chance (one transient time-out used to
// disable the RCCL leg for the list for the rest of the process)
std::map<std::vector<int>, std::chrono::steady_clock::time_point> g_comm_timed_out;
constexpr int kCommRetryAfterS = 300;

int shard_max_rows(int32_t nrows, int32_t row_tile, int32_t count) {
  int m = 0;
  for (int k = 0; k < count; ++k) {
    const int r = tor_shard_rows(nrows, row_tile, k, count, nullptr);
    if (r > m) m = r;
  }
  return m;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// the three legs of the framebuffer gather, each complete in itself: on TOR_OK canvas holds the frame
// ---------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_last_note;

// TOR_FAULT_INJECT = comma list of rccl_init | rccl_xfer | peer: the named leg fails at that point (tests of the fallback
// chain; on a 1-GPU box `rccl_init` also makes AUTO consider RCCL for a device list with repeated ordinals)
struct FaultInjection { bool rccl_init = false, rccl_xfer = false, peer = false, rccl_hang = false; };
FaultInjection fault_injection() {
  FaultInjection f;
  if (const char* e = tor::knob("TOR_FAULT_INJECT")) {
    f.rccl_init = std::strstr(e, "rccl_init") != nullptr;
    f.rccl_xfer = std::strstr(e, "rccl_xfer") != nullptr;
    f.peer = std::strstr(e, "peer") != nullptr;
    f.rccl_hang = std::strstr(e, "rccl_hang") != nullptr;
  }
  return f;
}

std::map<std::vector<int>, bool> g_rccl_bad;  // device lists whose single-process communicator failed (guarded by g_comm_mutex)
bool rccl_known_bad(const std::vector<int>& key) {
  std::lock_guard<std::mutex> lock(g_comm_mutex);
  return g_rccl_bad.count(key) != 0;
}
void mark_rccl_bad(const std::vector<int>& key) {
  std::lock_guard<std::mutex> lock(g_comm_mutex);
  g_rccl_bad[key] = true;
}

const char* gather_name(int m) { return m == TOR_GATHER_RCCL ? "rccl" : (m == TOR_GATHER_PEER ? "peer" : "host"); }

struct GatherJob {
  const std::vector<TorContext*>* ctxs = nullptr;
  const std::vector<std::vector<int32_t>>* rows = nullptr;
  std::vector<int> key;
  size_t row_bytes = 0;
  int32_t nrows = 0, ncols = 0, row_tile = 1;
  int max_rows = 0;
  char* canvas = nullptr;
  int n() const { return (int)ctxs->size(); }
  size_t slot_bytes() const { return (size_t)(max_rows > 0 ? max_rows : 1) * (row_bytes > 0 ? row_bytes : 24); }
  size_t shard_bytes(int k) const { return (*rows)[(size_t)k].size() * row_bytes; }
};

// root buffers + de-interleave + D2H, shared by the two device-side legs (the shards already sit in root->gather)
int assemble_and_download(const GatherJob& j) {
  TorContext* root = (*j.ctxs)[0];
  HIP_TRY(hipSetDevice(root->device));
  HIP_TRY(launch_gather_rows((const double*)root->gather.ptr, (double*)root->frame.ptr, j.nrows, j.ncols, j.row_tile, j.n(),
                             (long long)(j.slot_bytes() / 8), root->stream));
  return download_rows(root, root->frame.ptr, j.nrows, j.row_bytes, nullptr, j.canvas, root->stream);
}

int ensure_root_buffers(const GatherJob& j) {
  TorContext* root = (*j.ctxs)[0];
  HIP_TRY(hipSetDevice(root->device));
  HIP_TRY(root->gather.ensure(j.slot_bytes() * (size_t)j.n()));
  HIP_TRY(root->frame.ensure((size_t)(j.nrows > 0 ? j.nrows : 1) * (j.row_bytes > 0 ? j.row_bytes : 24)));
  return TOR_OK;
}

// hipMemcpyPeerAsync over xGMI.  Peer access is switched on once per (root, peer) pair when the pair supports it -- the
// copy is then a direct xGMI transfer; without it the runtime stages through the host (still correct, noted).
int gather_peer(const GatherJob& j, const FaultInjection& fault) {
  if (fault.peer) return fail(TOR_ERR_HIP, "TOR_FAULT_INJECT=peer");
  int rc = ensure_root_buffers(j);
  if (rc != TOR_OK) return rc;
  TorContext* root = (*j.ctxs)[0];
  char* gbase = (char*)root->gather.ptr;
  static std::mutex peer_mutex;
  static std::map<std::pair<int, int>, bool> peer_enabled;
  for (int k = 0; k < j.n(); ++k) {
    TorContext* c = (*j.ctxs)[(size_t)k];
    const size_t bytes = j.shard_bytes(k);
    if (bytes == 0) continue;
    if (c->device == root->device) {
      HIP_TRY(hipMemcpyAsync(gbase + (size_t)k * j.slot_bytes(), c->scratch.ptr, bytes, hipMemcpyDeviceToDevice, root->stream));
      continue;
    }
    {
      std::lock_guard<std::mutex> lock(peer_mutex);
      const std::pair<int, int> pr(root->device, c->device);
      if (!peer_enabled.count(pr)) {
        int can = 0;
        bool on = false;
        if (hipDeviceCanAccessPeer(&can, root->device, c->device) == hipSuccess && can) {
          const hipError_t e = hipDeviceEnablePeerAccess(c->device, 0);  // current device = root (set by ensure_root_buffers)
          on = (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled);
          (void)hipGetLastError();
        }
        peer_enabled[pr] = on;
      }
    }
    HIP_TRY(hipMemcpyPeerAsync(gbase + (size_t)k * j.slot_bytes(), root->device, c->scratch.ptr, c->device, bytes, root->stream));
  }
  return assemble_and_download(j);
}

// no device-side gather: every device copies its rows straight into the canvas over its own PCIe link
int gather_host(const GatherJob& j) {
  const int N = j.n();
  std::vector<int> rcs((size_t)N, TOR_OK);
  std::vector<std::string> errs((size_t)N);
  auto one = [&](int k) {
    TorContext* c = (*j.ctxs)[(size_t)k];
    rcs[(size_t)k] = download_rows(c, c->scratch.ptr, (int64_t)(*j.rows)[(size_t)k].size(), j.row_bytes, (*j.rows)[(size_t)k].data(),
                                   j.canvas, c->stream);
    if (rcs[(size_t)k] != TOR_OK) errs[(size_t)k] = tor_last_error();
  };
  std::vector<std::thread> pool;
  for (int k = 1; k < N; ++k) pool.emplace_back(one, k);
  one(0);
  for (std::thread& t : pool) t.join();
  for (int k = 0; k < N; ++k)
    if (rcs[(size_t)k] != TOR_OK) return fail(rcs[(size_t)k], "device " + std::to_string((*j.ctxs)[(size_t)k]->device) + ": " + errs[(size_t)k]);
  return TOR_OK;
}

// ---- RCCL watchdog ----------------------------------------------------------------------------------------------
// A collective that goes wrong on a fresh node usually does not FAIL, it HANGS: ncclCommInitAll never returns, or the
// send/recv kernels spin for ever.  Neither gives AUTO's fallback chain anything to fall back from, so every wait of the
// RCCL leg has a deadline:
//   * device side: the streams are POLLED (hipStreamQuery) instead of synchronised; past the deadline the communicators
//     are aborted (ncclCommAbort makes RCCL's kernels leave), the streams are drained and the leg fails with a reason;
//   * host side: communicator creation + self-check run in a helper thread; if it does not come back in time it is
//     abandoned (it owns everything it touches) and the device list is marked bad.
// Deadlines: TOR_RCCL_TIMEOUT_MS for a transfer (default 10 s + 1 ms per MB), TOR_RCCL_INIT_TIMEOUT_MS for creation +
// self-check (default 120 s).  TOR_FAULT_INJECT=rccl_hang replaces the transfer by a kernel that never ends on its own
// (tests/test_gpu_round4.py).
long env_ms(const char* name, long dflt) {
  if (const char* e = tor::knob(name)) {
    char* endp = nullptr;
    const long v = std::strtol(e, &endp, 10);
    if (endp != e && v > 0) return v;
  }
  return dflt;
}

// all streams idle -> hipSuccess; past the deadline -> hipErrorNotReady; a stream in error -> that error
// (device, stream) of one participant of a gather: a context's own render stream, or -- the communicator's self-check, which
// may run in a helper thread that the caller has given up on -- a private stream that no render ever touches
struct GatherLane { int device; hipStream_t stream; };
std::vector<GatherLane> lanes_of(const std::vector<TorContext*>& ctxs) {
  std::vector<GatherLane> l;
  for (TorContext* c : ctxs) l.push_back({c->device, c->stream});
  return l;
}

hipError_t wait_streams(const std::vector<GatherLane>& ctxs, long deadline_ms) {
  using clk = std::chrono::steady_clock;
  const clk::time_point t0 = clk::now();
  size_t k = 0;
  unsigned spins = 0;
  while (k < ctxs.size()) {
    hipError_t e = hipSetDevice(ctxs[k].device);
    if (e == hipSuccess) e = hipStreamQuery(ctxs[k].stream);
    if (e == hipSuccess) { ++k; continue; }
    if (e != hipErrorNotReady) return e;
    (void)hipGetLastError();
    if (std::chrono::duration<double, std::milli>(clk::now() - t0).count() > (double)deadline_ms) return hipErrorNotReady;
    if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));  // (the first ~ms is polled hot: a 1080p gather takes 0.1 ms)
  }
  return hipSuccess;
}

// the injected hang: a kernel on the root's stream that leaves only when the host sets *flag (what ncclCommAbort does to
// RCCL's kernels) -- or after 20 s on its own, so that a broken test cannot hold the GPU
struct HangInjection {
  volatile unsigned* flag = nullptr;
  ~HangInjection() { if (flag) (void)hipHostFree((void*)flag); }
  hipError_t start(hipStream_t stream) {
    hipError_t e = hipHostMalloc((void**)&flag, 64, hipHostMallocDefault);
    if (e != hipSuccess) { flag = nullptr; return e; }
    *flag = 0u;
    return launch_spin_until(flag, 2000000000ull, stream);
  }
  void release() { if (flag) *flag = 1u; }
};

constexpr int TOR_RCCL_TIMED_OUT = -1000;  // internal: rccl_group_gather's deadline expired (the caller aborts the communicators)

// One grouped send/recv gather over the communicators of a single-process device list.  EVERY path out of here
// closes the group; the streams are idle on TOR_OK and on every error except TOR_RCCL_TIMED_OUT.
int rccl_group_gather(RcclApi* api, std::vector<ncclComm_t>& comms, const std::vector<GatherLane>& ctxs, const std::vector<const void*>& src,
                      const std::vector<size_t>& bytes, char* dst_base, size_t dst_stride, bool inject_failure, HangInjection* hang) {
  const int N = (int)ctxs.size();
  const GatherLane* root = &ctxs[0];
  ncclResult_t first = ncclSuccess;
  const char* where = "";
  auto note = [&](ncclResult_t r, const char* w) { if (r != ncclSuccess && first == ncclSuccess) { first = r; where = w; } };
  size_t total = 0;
  for (int k = 1; k < N; ++k) total += bytes[(size_t)k];
  if (hang) {
    HIP_TRY(hipSetDevice(root->device));
    HIP_TRY(hang->start(root->stream));
  } else {
    ncclResult_t r = api->GroupStart();
    if (r != ncclSuccess) return fail_rccl(r, "ncclGroupStart");
    for (int k = 1; k < N && first == ncclSuccess; ++k) {
      if (bytes[(size_t)k] == 0) continue;
      if (inject_failure && k == N - 1) { note(ncclInternalError, "TOR_FAULT_INJECT=rccl_xfer"); break; }
      note(api->Send(src[(size_t)k], bytes[(size_t)k], ncclChar, 0, comms[(size_t)k], ctxs[(size_t)k].stream), "ncclSend");
      if (first != ncclSuccess) break;
      note(api->Recv(dst_base + (size_t)k * dst_stride, bytes[(size_t)k], ncclChar, k, comms[0], root->stream), "ncclRecv");
    }
    note(api->GroupEnd(), "ncclGroupEnd");
  }
  // the sends are complete once the matching receives are; wait for all of it -- with a deadline
  const long deadline = env_ms("TOR_RCCL_TIMEOUT_MS", 10000 + (long)(total >> 20));
  const hipError_t he = wait_streams(ctxs, deadline);
  (void)hipSetDevice(root->device);
  if (he == hipErrorNotReady) {
    fail(TOR_ERR_HIP, "RCCL gather: not complete after " + std::to_string(deadline) + " ms (TOR_RCCL_TIMEOUT_MS); communicators aborted");
    return TOR_RCCL_TIMED_OUT;
  }
  if (first != ncclSuccess) return fail_rccl(first, where);
  if (he != hipSuccess) return fail_hip(he, "RCCL gather: stream query");
  return TOR_OK;
}

constexpr long kRcclDrainMs = 5000;  // how long the streams may take to become idle after an abort

// after a timed-out transfer: abort, then give the streams a moment to drain so that the next leg finds them idle
void abort_comms(RcclApi* api, std::vector<ncclComm_t>& comms, const std::vector<GatherLane>& ctxs, HangInjection* hang) {
  if (hang) hang->release();
  if (api && api->CommAbort)
    for (ncclComm_t& c : comms)
      if (c) { (void)api->CommAbort(c); c = nullptr; }
  (void)wait_streams(ctxs, kRcclDrainMs);
  (void)hipSetDevice(ctxs[0].device);
}

// ncclCommInitAll + self-check for one device list; runs in its own thread (see the watchdog comment).  Everything it
// touches is owned by this object -- the self-check's buffers and STREAMS included (it used to run on the contexts'
// render streams, which the caller is rendering on again once it has given up on this thread).
struct CommSetup {
  std::mutex m;
  std::condition_variable cv;
  bool done = false;
  std::atomic<bool> abandoned{false};  // the caller's deadline passed: publish nothing, destroy what was created, touch nothing shared
  int rc = TOR_OK;
  std::string err;
  std::vector<int> key;
  std::vector<int> devices;        // HIP ordinal of every entry of the list
  std::vector<ncclComm_t> comms;
};

void comm_setup_body(std::shared_ptr<CommSetup> st) {
  RcclApi* api = rccl();
  const int N = (int)st->key.size();
  int check = TOR_OK;
  std::vector<ncclComm_t> c((size_t)N, nullptr);
  const ncclResult_t r = api->CommInitAll(c.data(), N, st->key.data());
  if (r != ncclSuccess) check = fail_rccl(r, "ncclCommInitAll");
  auto drop = [&]() {  // nobody will ever use these communicators
    for (ncclComm_t& cc : c)
      if (cc) {
        if (api->CommAbort) (void)api->CommAbort(cc);
        else (void)api->CommDestroy(cc);
        cc = nullptr;
      }
  };
  // (an abandoned creation still marks the set-up done and wakes everybody: a second caller that joined the same pending set-up
  // must not sit out its own full deadline for a helper thread that has already left -- ADVICE r5)
  auto leave_abandoned = [&]() {
    drop();
    std::lock_guard<std::mutex> lock(st->m);
    st->rc = TOR_ERR_HIP;
    st->err = "RCCL: communicator creation was abandoned by the caller that started it (deadline passed)";
    st->done = true;
    st->cv.notify_all();
  };
  if (st->abandoned.load()) { leave_abandoned(); return; }
  // self-check: device k sends kCheck bytes of value (k * 37 + i) & 0xff through the very same grouped send/recv code, on
  // streams of its own
  constexpr size_t kCheck = 4096;
  std::vector<DeviceBuffer> pat((size_t)N);
  DeviceBuffer got;
  std::vector<const void*> src((size_t)N, nullptr);
  std::vector<size_t> bytes((size_t)N, kCheck);
  std::vector<unsigned char> host(kCheck);
  std::vector<GatherLane> lanes;
  for (int k = 0; k < N && check == TOR_OK; ++k) {
    for (size_t i = 0; i < kCheck; ++i) host[i] = (unsigned char)((k * 37 + (int)i) & 0xff);
    hipStream_t s = nullptr;
    hipError_t e = hipSetDevice(st->devices[(size_t)k]);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e == hipSuccess) lanes.push_back({st->devices[(size_t)k], s});
    if (e == hipSuccess) e = pat[(size_t)k].ensure(kCheck);
    if (e == hipSuccess) e = hipMemcpy(pat[(size_t)k].ptr, host.data(), kCheck, hipMemcpyHostToDevice);
    if (e != hipSuccess) check = fail_hip(e, "RCCL self-check: stream / pattern upload");
    src[(size_t)k] = pat[(size_t)k].ptr;
  }
  if (check == TOR_OK) {
    hipError_t e = hipSetDevice(st->devices[0]);
    if (e == hipSuccess) e = got.ensure(kCheck * (size_t)N);
    if (e == hipSuccess) e = hipMemset(got.ptr, 0, kCheck * (size_t)N);
    if (e != hipSuccess) check = fail_hip(e, "RCCL self-check: receive buffer");
  }
  if (check == TOR_OK) {
    check = rccl_group_gather(api, c, lanes, src, bytes, (char*)got.ptr, kCheck, false, nullptr);
    if (check == TOR_RCCL_TIMED_OUT) {
      const std::string why = tor_last_error();
      abort_comms(api, c, lanes, nullptr);
      check = fail(TOR_ERR_HIP, "RCCL self-check: " + why);
    }
  }
  if (check == TOR_OK) {
    std::vector<unsigned char> back(kCheck * (size_t)N);
    hipError_t e = hipMemcpy(back.data(), got.ptr, back.size(), hipMemcpyDeviceToHost);
    if (e != hipSuccess) check = fail_hip(e, "RCCL self-check: read back");
    for (int k = 1; k < N && check == TOR_OK; ++k)
      for (size_t i = 0; i < kCheck; ++i)
        if (back[(size_t)k * kCheck + i] != (unsigned char)((k * 37 + (int)i) & 0xff)) {
          check = fail(TOR_ERR_HIP, "RCCL self-check: rank " + std::to_string(k) + "'s pattern arrived corrupted");
          break;
        }
  }
  for (DeviceBuffer& b : pat) b.release();
  got.release();
  for (const GatherLane& l : lanes)
    if (hipSetDevice(l.device) == hipSuccess) (void)hipStreamDestroy(l.stream);
  if (check != TOR_OK) drop();
  std::lock_guard<std::mutex> lock(st->m);
  if (st->abandoned.load()) {  // (the flag is set under st->m: checked here, nothing can be published to a caller that has left)
    drop();
    st->rc = TOR_ERR_HIP;
    st->err = "RCCL: communicator creation was abandoned by the caller that started it (deadline passed)";
    st->done = true;
    st->cv.notify_all();
    return;
  }
  st->rc = check;
  if (check != TOR_OK) st->err = tor_last_error();
  else st->comms = std::move(c);
  st->done = true;
  st->cv.notify_all();
}

thread_local int g_last_gather_leg = -1, g_last_rccl_ranks = 0, g_last_devices = 0, g_last_distinct = 0;
thread_local std::vector<float> g_last_kernel_ms;

// Single-process RCCL (ncclCommInitAll): every device sends its shard to devices[0] over its own xGMI link (7 links
// in parallel, no ring).  A communicator is trusted only after a SELF-CHECK at creation: every device sends a 4 KB
// pattern through the very same grouped send/recv code and the root's copy must match byte for byte.
int gather_rccl(const GatherJob& j, const FaultInjection& fault) {
  if (fault.rccl_init) return fail(TOR_ERR_HIP, "TOR_FAULT_INJECT=rccl_init");
  RcclApi* api = rccl();
  if (!api && !fault.rccl_hang) return fail(TOR_ERR_HIP, "librccl.so.1 could not be loaded");
  int rc = ensure_root_buffers(j);
  if (rc != TOR_OK) return rc;
  const int N = j.n();
  const std::vector<TorContext*>& ctxs = *j.ctxs;
  TorContext* root = ctxs[0];
  std::vector<ncclComm_t>* comms = nullptr;
  std::vector<ncclComm_t> no_comms;  // (rccl_hang on a box without N distinct GPUs: no communicator, only the watchdog's path)
  if (fault.rccl_hang) {
    comms = &no_comms;
  } else {
    // The creation thread is waited for WITHOUT the registry's mutex (up to 120 s during which no other thread could
    // even look a communicator up); callers that ask for the same list meanwhile wait on the same CommSetup.  A list whose
    // creation timed out once is remembered: an explicit TOR_GATHER=rccl does not leave another hung thread behind per call.
    std::shared_ptr<CommSetup> st;
    {
      std::lock_guard<std::mutex> lock(g_comm_mutex);
      auto it = g_single_process_comms.find(j.key);
      if (it != g_single_process_comms.end()) comms = &it->second;
      else if (g_comm_timed_out.count(j.key) &&
               std::chrono::steady_clock::now() - g_comm_timed_out[j.key] < std::chrono::seconds(kCommRetryAfterS))
        return fail(TOR_ERR_HIP, "RCCL: communicator creation for this device list timed out less than " + std::to_string(kCommRetryAfterS) +
                                     " s ago in this process; not tried again yet");
      else {
        g_comm_timed_out.erase(j.key);
        auto pend = g_comm_pending.find(j.key);
        if (pend != g_comm_pending.end()) st = pend->second;
        else {
          st = std::make_shared<CommSetup>();
          st->key = j.key;
          for (TorContext* c : ctxs) st->devices.push_back(c->device);
          g_comm_pending[j.key] = st;
          std::thread(comm_setup_body, st).detach();
        }
      }
    }
    if (!comms) {
      const long deadline = env_ms("TOR_RCCL_INIT_TIMEOUT_MS", 120000);
      bool finished;
      {
        std::unique_lock<std::mutex> wait_lock(st->m);
        finished = st->cv.wait_for(wait_lock, std::chrono::milliseconds(deadline), [&] { return st->done; });
        if (!finished) {
          st->abandoned.store(true);   // under st->m: the thread checks it under the same mutex before publishing
          // ... and everybody else who waits on this set-up leaves now (the helper thread may be hung inside ncclCommInitAll for good)
          st->rc = TOR_ERR_HIP;
          st->err = "RCCL: communicator creation + self-check did not return within the first caller's deadline (TOR_RCCL_INIT_TIMEOUT_MS); abandoned";
          st->done = true;
          st->cv.notify_all();
        }
      }
      std::lock_guard<std::mutex> lock(g_comm_mutex);
      g_comm_pending.erase(j.key);
      if (!finished) {
        g_comm_timed_out[j.key] = std::chrono::steady_clock::now();
        return fail(TOR_ERR_HIP, "RCCL: communicator creation + self-check did not return within " + std::to_string(deadline) +
                                     " ms (TOR_RCCL_INIT_TIMEOUT_MS); abandoned (the helper thread destroys whatever it still creates)");
      }
      if (st->rc != TOR_OK) return fail(st->rc, st->err);
      auto it = g_single_process_comms.find(j.key);
      if (it == g_single_process_comms.end()) it = g_single_process_comms.emplace(j.key, std::move(st->comms)).first;   // (the first of several waiters publishes)
      comms = &it->second;
    }
  }
  HIP_TRY(hipSetDevice(root->device));
  char* gbase = (char*)root->gather.ptr;
  if (j.shard_bytes(0) > 0)
    HIP_TRY(hipMemcpyAsync(gbase, root->scratch.ptr, j.shard_bytes(0), hipMemcpyDeviceToDevice, root->stream));
  std::vector<const void*> src((size_t)N, nullptr);
  std::vector<size_t> bytes((size_t)N, 0);
  for (int k = 0; k < N; ++k) { src[(size_t)k] = ctxs[(size_t)k]->scratch.ptr; bytes[(size_t)k] = j.shard_bytes(k); }
  HangInjection hang;
  const std::vector<GatherLane> lanes = lanes_of(ctxs);
  rc = rccl_group_gather(api, *comms, lanes, src, bytes, gbase, j.slot_bytes(), fault.rccl_xfer, fault.rccl_hang ? &hang : nullptr);
  if (rc == TOR_RCCL_TIMED_OUT) {
    const std::string why = tor_last_error();
    abort_comms(api, *comms, lanes, fault.rccl_hang ? &hang : nullptr);
    if (!fault.rccl_hang) {
      std::lock_guard<std::mutex> lock(g_comm_mutex);
      g_single_process_comms.erase(j.key);  // (aborted communicators are gone)
    }
    return fail(TOR_ERR_HIP, why);
  }
  if (rc != TOR_OK) return rc;
  g_last_rccl_ranks = 0;
  if (api && api->CommCount && !comms->empty() && (*comms)[0]) {
    int cnt = 0;
    if (api->CommCount((*comms)[0], &cnt) == ncclSuccess) g_last_rccl_ranks = cnt;
  }
  return assemble_and_download(j);
}

}  // namespace

void set_last_note(const std::string& s) { g_last_note = s; }
const std::string& last_note() { return g_last_note; }

int render_multi_device(TorCanvas* canvas, const TorCamera* cam, TorHittableList world, int64_t max_depth,
                        const TorOptions& o, double timing_ms[5]) {
  using clk = std::chrono::steady_clock;
  auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
  const int N = o.device_count;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(TOR_ERR_NO_DEVICE, std::string("no HIP device available (") + hipGetErrorString(e) + "); libtor_mi355x has no CPU fallback");
  std::vector<TorContext*> ctxs((size_t)N, nullptr);
  bool distinct = true;
  for (int k = 0; k < N; ++k) {
    if (o.devices[k] >= count) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render: device ordinal out of range in TorOptions.devices");
    int replica = 0;
    for (int j = 0; j < k; ++j)
      if (o.devices[j] == o.devices[k]) { ++replica; distinct = false; }
    const int rc = default_context(o.devices[k], replica, &ctxs[(size_t)k]);
    if (rc != TOR_OK) return rc;
  }
  // Gather plan.  An explicit mode is tried alone (its failure is the caller's answer).  AUTO walks RCCL -> PEER -> HOST:
  // every leg only touches canvas->pixels once the whole frame is assembled (or, HOST, overwrites every row), so a leg
  // that fails part-way is simply followed by the next one -- never a wrong canvas.  RCCL is a candidate of AUTO only when
  // the ordinals are distinct (one RCCL rank per GPU), librccl loads and the communicator of this device list passed
  // its self-check (gather_rccl below).
  const FaultInjection fault = fault_injection();
  std::vector<int> plan;
  if (o.gather == TOR_GATHER_AUTO) {
    const std::vector<int> key(o.devices, o.devices + N);
    const bool injected = fault.rccl_init || fault.rccl_hang;  // (a 1-GPU box walks the RCCL leg's failure paths this way)
    if ((distinct || injected) && (rccl() != nullptr || injected) && !rccl_known_bad(key)) plan.push_back(TOR_GATHER_RCCL);
    plan.push_back(TOR_GATHER_PEER);
    plan.push_back(TOR_GATHER_HOST);
  } else {
    if (o.gather == TOR_GATHER_RCCL && !distinct)
      return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render: TOR_GATHER_RCCL needs distinct devices (one RCCL rank per GPU)");
    if (o.gather == TOR_GATHER_RCCL && rccl() == nullptr) return fail(TOR_ERR_HIP, "tor_render: TOR_GATHER_RCCL: librccl.so.1 could not be loaded");
    plan.push_back(o.gather);
  }
  // a plan that starts with HOST downloads inside the per-device jobs (every device over its own PCIe link, in parallel)
  const int mode = plan[0];

  const int32_t nrows = canvas->nrows, ncols = canvas->ncols;
  const size_t row_bytes = (size_t)(ncols > 0 ? ncols : 0) * 24;
  std::vector<std::vector<int32_t>> rows((size_t)N);
  for (int k = 0; k < N; ++k) {
    rows[(size_t)k].resize((size_t)(nrows > 0 ? nrows : 0));
    const int32_t n = tor_shard_rows(nrows, o.row_tile, k, N, rows[(size_t)k].data());
    rows[(size_t)k].resize((size_t)n);
  }
  const int max_rows = shard_max_rows(nrows, o.row_tile, N);

  // ---- phase 1: every device renders its shard (one host thread each, as render.nim:55 deals rows to workers) ----
  std::vector<int> rcs((size_t)N, TOR_OK);
  std::vector<std::string> errs((size_t)N);
  std::vector<double> t_upload((size_t)N, 0.0), t_render((size_t)N, 0.0), t_copy((size_t)N, 0.0);
  std::vector<char> hit((size_t)N, 0);
  auto device_job = [&](int k) {
    TorContext* ctx = ctxs[(size_t)k];
    auto body = [&]() -> int {
      clk::time_point t0 = clk::now();
      HIP_TRY(hipSetDevice(ctx->device));
      const int64_t hits_before = ctx->n_cache_hits;
      int rc = tor_scene_upload(ctx, world);
      if (rc != TOR_OK) return rc;
      hit[(size_t)k] = ctx->n_cache_hits > hits_before;
      rc = ensure_layouts(ctx, o.accel);
      if (rc != TOR_OK) return rc;
      t_upload[(size_t)k] = ms_since(t0);
      t0 = clk::now();
      HIP_TRY(ctx->scratch.ensure((size_t)(max_rows > 0 ? max_rows : 1) * (row_bytes > 0 ? row_bytes : 24)));
      TorOptions ok = o;
      ok.device_count = 0;
      ok.device = ctx->device;
      ok.shard_index = k;
      ok.shard_count = N;
      rc = tor_render_device(ctx, cam, nrows, ncols, canvas->samples_per_pixel, canvas->gamma_correction, max_depth, &ok,
                             (double*)ctx->scratch.ptr, ctx->stream);
      if (rc != TOR_OK) return rc;
      rc = rerender_if_stalled(ctx, cam, nrows, ncols, canvas->samples_per_pixel, canvas->gamma_correction, max_depth, &ok,
                               (double*)ctx->scratch.ptr, ctx->stream);
      if (rc != TOR_OK) return rc;
      if (mode == TOR_GATHER_HOST) {
        t_render[(size_t)k] = ms_since(t0);
        t0 = clk::now();
        rc = download_rows(ctx, ctx->scratch.ptr, (int64_t)rows[(size_t)k].size(), row_bytes, rows[(size_t)k].data(),
                           (char*)canvas->pixels, ctx->stream);
        t_copy[(size_t)k] = ms_since(t0);
        return rc;
      }
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      t_render[(size_t)k] = ms_since(t0);
      return TOR_OK;
    };
    rcs[(size_t)k] = body();
    if (rcs[(size_t)k] != TOR_OK) errs[(size_t)k] = tor_last_error();
  };
  {
    std::vector<std::thread> pool;
    for (int k = 1; k < N; ++k) pool.emplace_back(device_job, k);
    device_job(0);
    for (std::thread& t : pool) t.join();
  }
  for (int k = 0; k < N; ++k)
    if (rcs[(size_t)k] != TOR_OK) return fail(rcs[(size_t)k], "device " + std::to_string(o.devices[k]) + ": " + errs[(size_t)k]);
  // the dominant kernel's duration on every device (HIP events around the launch, on the launch's stream): tor_last_device_kernel_ms
  g_last_kernel_ms.assign((size_t)N, 0.f);
  for (int k = 0; k < N; ++k) {
    float ms = 0.f;
    if (ctxs[(size_t)k]->timing_valid && tor_last_kernel_ms(ctxs[(size_t)k], &ms, nullptr) == TOR_OK) g_last_kernel_ms[(size_t)k] = ms;
  }
  for (int k = 0; k < N; ++k) {
    if (t_upload[(size_t)k] > timing_ms[0]) timing_ms[0] = t_upload[(size_t)k];
    if (t_render[(size_t)k] > timing_ms[1]) timing_ms[1] = t_render[(size_t)k];
    if (t_copy[(size_t)k] > timing_ms[2]) timing_ms[2] = t_copy[(size_t)k];
  }
  timing_ms[4] = 1.0;
  for (int k = 0; k < N; ++k)
    if (!hit[(size_t)k]) timing_ms[4] = 0.0;
  g_last_devices = N;
  g_last_distinct = distinct ? 1 : 0;
  if (mode == TOR_GATHER_HOST) {
    set_last_note("gather: host");
    g_last_gather_leg = TOR_GATHER_HOST;
    g_last_rccl_ranks = 0;
    return TOR_OK;
  }

  // ---- phase 2: the shards travel to devices[0] (xGMI), one de-interleave kernel, one D2H ----------------------
  const clk::time_point t0 = clk::now();
  GatherJob job;
  job.ctxs = &ctxs;
  job.rows = &rows;
  job.row_bytes = row_bytes;
  job.nrows = nrows;
  job.ncols = ncols;
  job.row_tile = o.row_tile;
  job.max_rows = max_rows;
  job.key.assign(o.devices, o.devices + N);
  job.canvas = (char*)canvas->pixels;
  std::string notes;
  int rc = TOR_OK;
  for (size_t leg = 0; leg < plan.size(); ++leg) {
    const int m = plan[leg];
    rc = m == TOR_GATHER_RCCL ? gather_rccl(job, fault) : (m == TOR_GATHER_PEER ? gather_peer(job, fault) : gather_host(job));
    if (rc == TOR_OK) {
      set_last_note(notes + "gather: " + gather_name(m));
      g_last_gather_leg = m;
      if (m != TOR_GATHER_RCCL) g_last_rccl_ranks = 0;
      break;
    }
    notes += std::string(gather_name(m)) + " failed (" + tor_last_error() + "); ";
    // AUTO never tries a communicator again that failed once (an injected failure says nothing about the communicator)
    if (m == TOR_GATHER_RCCL && !fault.rccl_init && !fault.rccl_xfer && !fault.rccl_hang) mark_rccl_bad(job.key);
  }
  if (rc != TOR_OK) return fail(rc, "tor_render: framebuffer gather: " + notes);
  timing_ms[2] = ms_since(t0);
  return TOR_OK;
}

}  // namespace tor

// ---------------------------------------------------------------------------------------------
// one process per GPU: the gather inside the library
// ---------------------------------------------------------------------------------------------
using tor::fail;
using tor::fail_hip;
using tor::fail_rccl;

extern "C" {

const char* tor_last_note(void) { return tor::last_note().c_str(); }

int tor_last_gather_info(int32_t out[4]) {
  if (!out) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_last_gather_info: out is NULL");
  out[0] = tor::g_last_gather_leg;
  out[1] = tor::g_last_rccl_ranks;
  out[2] = tor::g_last_devices;
  out[3] = tor::g_last_distinct;
  return TOR_OK;
}

int32_t tor_last_device_kernel_ms(float* out, int32_t cap) {
  const int32_t n = (int32_t)tor::g_last_kernel_ms.size();
  for (int32_t k = 0; out && k < n && k < cap; ++k) out[k] = tor::g_last_kernel_ms[(size_t)k];
  return n;
}

int tor_comm_unique_id(uint8_t id_out[128]) {
  if (!id_out) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_comm_unique_id: NULL argument");
  tor::RcclApi* api = tor::rccl();
  if (!api) return fail(TOR_ERR_HIP, "tor_comm_unique_id: librccl.so.1 could not be loaded");
  ncclUniqueId id;
  RCCL_TRY(api->GetUniqueId(&id));
  std::memcpy(id_out, &id, 128);
  return TOR_OK;
}

int tor_comm_init_rank(TorContext* ctx, const uint8_t id[128], int32_t rank, int32_t world) {
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_comm_init_rank: bad argument");
  tor::RcclApi* api = tor::rccl();
  if (!api) return fail(TOR_ERR_HIP, "tor_comm_init_rank: librccl.so.1 could not be loaded");
  (void)tor_comm_destroy(ctx);
  HIP_TRY(hipSetDevice(ctx->device));
  ncclUniqueId uid;
  std::memcpy(&uid, id, 128);
  ncclComm_t comm = nullptr;
  RCCL_TRY(api->CommInitRank(&comm, world, uid, rank));
  ctx->comm = new tor::RcclComm{comm, rank, world};
  return TOR_OK;
}

int tor_comm_destroy(TorContext* ctx) {
  if (!ctx || !ctx->comm) return TOR_OK;
  tor::RcclApi* api = tor::rccl();
  if (api && ctx->comm->comm) (void)api->CommDestroy(ctx->comm->comm);
  delete ctx->comm;
  ctx->comm = nullptr;
  return TOR_OK;
}

int tor_comm_abort(TorContext* ctx) {
  if (!ctx || !ctx->comm) return TOR_OK;
  tor::RcclApi* api = tor::rccl();
  if (api && api->CommAbort && ctx->comm->comm) (void)api->CommAbort(ctx->comm->comm);
  delete ctx->comm;
  ctx->comm = nullptr;
  return TOR_OK;
}

int tor_comm_count(TorContext* ctx, int32_t* ranks_out) {
  if (!ctx || !ranks_out) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_comm_count: NULL argument");
  *ranks_out = 0;
  if (!ctx->comm || !ctx->comm->comm) return TOR_OK;
  tor::RcclApi* api = tor::rccl();
  if (!api || !api->CommCount) return fail(TOR_ERR_HIP, "tor_comm_count: librccl.so.1 could not be loaded");
  int n = 0;
  RCCL_TRY(api->CommCount(ctx->comm->comm, &n));
  *ranks_out = n;
  return TOR_OK;
}

int tor_render_gather_device(TorContext* ctx, const TorCamera* cam, int32_t nrows, int32_t ncols, int32_t spp,
                             float gamma_correction, int64_t max_depth, const TorOptions* opt, int32_t root, double* d_frame,
                             void* hip_stream) {
  if (!ctx || !cam) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_gather_device: NULL argument");
  TorOptions o;
  if (!tor::valid_options(opt, o, false)) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_gather_device: bad TorOptions: " + tor::options_why());
  const int world = ctx->comm ? ctx->comm->world : 1;
  const int rank = ctx->comm ? ctx->comm->rank : 0;
  if (root >= world) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_gather_device: root out of range");
  const bool receives = root < 0 || root == rank;
  if (receives && !d_frame) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_gather_device: d_frame is NULL on a receiving rank");
  if (nrows < 2 || ncols < 2) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_gather_device: need nrows >= 2, ncols >= 2");
  hipStream_t stream = (hipStream_t)hip_stream;
  HIP_TRY(hipSetDevice(ctx->device));
  o.device_count = 0;
  o.shard_index = rank;
  o.shard_count = world;
  const size_t row_bytes = (size_t)ncols * 24;
  const int max_rows = tor::shard_max_rows(nrows, o.row_tile, world);
  const size_t slot_bytes = (size_t)(max_rows > 0 ? max_rows : 1) * row_bytes;
  HIP_TRY(ctx->scratch.ensure(slot_bytes));
  int rc = tor_render_device(ctx, cam, nrows, ncols, spp, gamma_correction, max_depth, &o, (double*)ctx->scratch.ptr, stream);
  if (rc != TOR_OK) return rc;
  // A chain hand-off launch may flag itself INCOMPLETE (the stall escape of a launch that is not fully resident): such a shard
  // must not be gathered.  For those launches -- SEED_PIXEL with both accelerations, nothing else -- the call waits
  // for the shard and renders it again without the hand-off when the flag is set, as the blocking entry points do; every
  // other launch stays asynchronous.
  if (ctx->last_migrate) {
    rc = tor::rerender_if_stalled(ctx, cam, nrows, ncols, spp, gamma_correction, max_depth, &o, (double*)ctx->scratch.ptr, stream);
    if (rc != TOR_OK) return rc;
  }
  const double* gathered = (const double*)ctx->scratch.ptr;
  if (world > 1) {
    tor::RcclApi* api = tor::rccl();
    if (!api) return fail(TOR_ERR_HIP, "tor_render_gather_device: librccl.so.1 could not be loaded");
    if (receives) HIP_TRY(ctx->gather.ensure(slot_bytes * (size_t)world));
    char* gbase = (char*)ctx->gather.ptr;
    if (root < 0) {
      RCCL_TRY(api->AllGather(ctx->scratch.ptr, gbase, slot_bytes, ncclChar, ctx->comm->comm, stream));
    } else if (rank == root) {
      const size_t mine = (size_t)tor_shard_rows(nrows, o.row_tile, rank, world, nullptr) * row_bytes;
      if (mine > 0) HIP_TRY(hipMemcpyAsync(gbase + (size_t)rank * slot_bytes, ctx->scratch.ptr, mine, hipMemcpyDeviceToDevice, stream));
      // (every path closes the group: an open group would swallow the next call's RCCL operations)
      RCCL_TRY(api->GroupStart());
      ncclResult_t first = ncclSuccess;
      for (int k = 0; k < world && first == ncclSuccess; ++k) {
        if (k == rank) continue;
        const size_t bytes = (size_t)tor_shard_rows(nrows, o.row_tile, k, world, nullptr) * row_bytes;
        if (bytes > 0) first = api->Recv(gbase + (size_t)k * slot_bytes, bytes, ncclChar, k, ctx->comm->comm, stream);
      }
      const ncclResult_t ended = api->GroupEnd();
      if (first != ncclSuccess) return fail_rccl(first, "ncclRecv (framebuffer gather)");
      if (ended != ncclSuccess) return fail_rccl(ended, "ncclGroupEnd (framebuffer gather)");
    } else {
      const size_t mine = (size_t)tor_shard_rows(nrows, o.row_tile, rank, world, nullptr) * row_bytes;
      if (mine > 0) RCCL_TRY(api->Send(ctx->scratch.ptr, mine, ncclChar, root, ctx->comm->comm, stream));
    }
    gathered = (const double*)gbase;
  }
  if (receives)
    HIP_TRY(tor::launch_gather_rows(gathered, d_frame, nrows, ncols, o.row_tile, world, (long long)(slot_bytes / 8), stream));
  return TOR_OK;
}

}  // extern "C"
