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
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "tor_context.hpp"

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
  if (const char* e = std::getenv("TOR_COPY_CHUNK_KB")) chunk_target = (size_t)std::atoll(e) << 10;
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
  if (const char* e = std::getenv("TOR_COPY_THREADS")) n_workers = std::atoi(e);
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

int shard_max_rows(int32_t nrows, int32_t row_tile, int32_t count) {
  int m = 0;
  for (int k = 0; k < count; ++k) {
    const int r = tor_shard_rows(nrows, row_tile, k, count, nullptr);
    if (r > m) m = r;
  }
  return m;
}

}  // namespace

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
  int mode = o.gather;
  if (mode == TOR_GATHER_AUTO) mode = (distinct && rccl() != nullptr) ? TOR_GATHER_RCCL : TOR_GATHER_PEER;
  if (mode == TOR_GATHER_RCCL && !distinct)
    return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render: TOR_GATHER_RCCL needs distinct devices (one RCCL rank per GPU)");
  if (mode == TOR_GATHER_RCCL && rccl() == nullptr) return fail(TOR_ERR_HIP, "tor_render: TOR_GATHER_RCCL: librccl.so.1 could not be loaded");

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
  for (int k = 0; k < N; ++k) {
    if (t_upload[(size_t)k] > timing_ms[0]) timing_ms[0] = t_upload[(size_t)k];
    if (t_render[(size_t)k] > timing_ms[1]) timing_ms[1] = t_render[(size_t)k];
    if (t_copy[(size_t)k] > timing_ms[2]) timing_ms[2] = t_copy[(size_t)k];
  }
  timing_ms[4] = 1.0;
  for (int k = 0; k < N; ++k)
    if (!hit[(size_t)k]) timing_ms[4] = 0.0;
  if (mode == TOR_GATHER_HOST) return TOR_OK;

  // ---- phase 2: the shards travel to devices[0] (xGMI), one de-interleave kernel, one D2H ----------------------
  const clk::time_point t0 = clk::now();
  TorContext* root = ctxs[0];
  HIP_TRY(hipSetDevice(root->device));
  const size_t slot_bytes = (size_t)(max_rows > 0 ? max_rows : 1) * (row_bytes > 0 ? row_bytes : 24);
  HIP_TRY(root->gather.ensure(slot_bytes * (size_t)N));
  HIP_TRY(root->frame.ensure((size_t)(nrows > 0 ? nrows : 1) * (row_bytes > 0 ? row_bytes : 24)));
  char* gbase = (char*)root->gather.ptr;
  if (mode == TOR_GATHER_PEER) {
    for (int k = 0; k < N; ++k) {
      const size_t bytes = rows[(size_t)k].size() * row_bytes;
      if (bytes == 0) continue;
      if (ctxs[(size_t)k]->device == root->device)
        HIP_TRY(hipMemcpyAsync(gbase + (size_t)k * slot_bytes, ctxs[(size_t)k]->scratch.ptr, bytes, hipMemcpyDeviceToDevice, root->stream));
      else
        HIP_TRY(hipMemcpyPeerAsync(gbase + (size_t)k * slot_bytes, root->device, ctxs[(size_t)k]->scratch.ptr, ctxs[(size_t)k]->device, bytes,
                                   root->stream));
    }
  } else {
    RcclApi* api = rccl();
    std::vector<ncclComm_t>* comms = nullptr;
    {
      std::lock_guard<std::mutex> lock(g_comm_mutex);
      std::vector<int> key(o.devices, o.devices + N);
      auto it = g_single_process_comms.find(key);
      if (it == g_single_process_comms.end()) {
        std::vector<ncclComm_t> c((size_t)N, nullptr);
        RCCL_TRY(api->CommInitAll(c.data(), N, key.data()));
        it = g_single_process_comms.emplace(key, std::move(c)).first;
      }
      comms = &it->second;
    }
    // each peer sends over its own link to the root (7 links in parallel, no ring)
    if (!rows[0].empty())
      HIP_TRY(hipMemcpyAsync(gbase, root->scratch.ptr, rows[0].size() * row_bytes, hipMemcpyDeviceToDevice, root->stream));
    RCCL_TRY(api->GroupStart());
    for (int k = 1; k < N; ++k) {
      const size_t bytes = rows[(size_t)k].size() * row_bytes;
      if (bytes == 0) continue;
      RCCL_TRY(api->Send(ctxs[(size_t)k]->scratch.ptr, bytes, ncclChar, 0, (*comms)[(size_t)k], ctxs[(size_t)k]->stream));
      RCCL_TRY(api->Recv(gbase + (size_t)k * slot_bytes, bytes, ncclChar, k, (*comms)[0], root->stream));
    }
    RCCL_TRY(api->GroupEnd());
    HIP_TRY(hipSetDevice(root->device));
  }
  HIP_TRY(launch_gather_rows((const double*)root->gather.ptr, (double*)root->frame.ptr, nrows, ncols, o.row_tile, N,
                             (long long)(slot_bytes / 8), root->stream));
  const int rc = download_rows(root, root->frame.ptr, nrows, row_bytes, nullptr, (char*)canvas->pixels, root->stream);
  if (rc != TOR_OK) return rc;
  if (mode == TOR_GATHER_RCCL)
    for (int k = 1; k < N; ++k) {  // the sends are complete once the matching receives are, but leave the streams idle
      HIP_TRY(hipSetDevice(ctxs[(size_t)k]->device));
      HIP_TRY(hipStreamSynchronize(ctxs[(size_t)k]->stream));
    }
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

int tor_render_gather_device(TorContext* ctx, const TorCamera* cam, int32_t nrows, int32_t ncols, int32_t spp,
                             float gamma_correction, int64_t max_depth, const TorOptions* opt, int32_t root, double* d_frame,
                             void* hip_stream) {
  if (!ctx || !cam) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_gather_device: NULL argument");
  TorOptions o;
  if (!tor::valid_options(opt, o, false)) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_gather_device: bad TorOptions");
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
      RCCL_TRY(api->GroupStart());
      for (int k = 0; k < world; ++k) {
        if (k == rank) continue;
        const size_t bytes = (size_t)tor_shard_rows(nrows, o.row_tile, k, world, nullptr) * row_bytes;
        if (bytes > 0) RCCL_TRY(api->Recv(gbase + (size_t)k * slot_bytes, bytes, ncclChar, k, ctx->comm->comm, stream));
      }
      RCCL_TRY(api->GroupEnd());
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
