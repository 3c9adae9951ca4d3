// tor_api.cpp -- the C ABI of libtor_mi355x.so (include/tor_render.h): context, scene
// flattening (AoS HittableVariant -> device SoA), launch orchestration, error reporting.
// There is NO CPU fallback: without a HIP device every rendering entry point fails with
// TOR_ERR_NO_DEVICE.
#include "../../include/tor_render.h"

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "tor_context.hpp"
#include "tor_knobs.hpp"
#include "tor_screen.hpp"

// ---- layout guards: the structs must match what Nim's C backend emits --------
static_assert(sizeof(TorVec3) == 24, "Vec3 is 3 x float64 (vec3s.nim:12-14)");
static_assert(sizeof(TorMaterial) == 40 && offsetof(TorMaterial, u) == 8, "Material (core.nim:16-28)");
static_assert(sizeof(TorSphere) == 72, "Sphere (spheres.nim:15-18)");
static_assert(sizeof(TorMovingSphere) == 112, "MovingSphere (moving_spheres.nim:15-20)");
static_assert(sizeof(TorHittableVariant) == 120 && offsetof(TorHittableVariant, u) == 8,
              "HittableVariant (hittables_variants.nim:50-57)");
static_assert(sizeof(TorHittableList) == 16, "HittableList (hittables_lists.nim:20-24)");
static_assert(sizeof(TorCamera) == 192, "Camera (cameras.nim:15-22)");
static_assert(sizeof(TorCanvas) == 24, "Canvas (canvas.nim:20-28)");
static_assert(sizeof(tor::Camera) == sizeof(TorCamera), "device camera mirrors TorCamera");
static_assert(offsetof(TorOptions, device_count) == 32, "the round-1 TorOptions is a prefix of the current one");

namespace {
std::mutex g_handoff_mutex;                    // hand-off launches are chained per device (tor_render_device)
std::map<int, hipEvent_t> g_handoff_event;     // device -> end of the last hand-off launch of this process
thread_local std::string g_last_error = "";
thread_local double g_last_timing[5] = {0, 0, 0, 0, 0};
}  // namespace

namespace tor {

void set_last_error(const std::string& msg) { g_last_error = msg; }  // for the host-only sources (tor_mp4.cpp)

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

int fail_hip(hipError_t e, const char* what) {
  std::string m = std::string(what) + ": " + hipGetErrorString(e);
  if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
    return fail(TOR_ERR_NO_DEVICE, m + " (libtor_mi355x has no CPU fallback)");
  if (e == hipErrorOutOfMemory) return fail(TOR_ERR_OUT_OF_MEMORY, m);
  return fail(TOR_ERR_HIP, m);
}

hipError_t DeviceLayout::put(const HostLayout& lay, const std::vector<double>* cold_override) {
  const std::vector<double>& c = cold_override ? *cold_override : lay.cold;
  struct Part { const void* src; size_t bytes; size_t off; };
  Part parts[10] = {{lay.stat.data(), lay.stat.size() * 8, 0}, {lay.mov.data(), lay.mov.size() * 8, 0},
                   {lay.movy.data(), lay.movy.size() * 8, 0}, {lay.segs.data(), lay.segs.size() * 8, 0},
                   {c.data(), c.size() * 8, 0},               {lay.hot32.data(), lay.hot32.size() * 4, 0},
                   {lay.coop_trips.data(), lay.coop_trips.size() * 8, 0},
                   {lay.xhdr.data(), lay.xhdr.size() * 8, 0}, {lay.xrec.data(), lay.xrec.size() * 8, 0},
                    {lay.xpl.data(), lay.xpl.size() * 8, 0}};
  size_t total = 0;
  for (Part& p : parts) {
    p.off = total;
    total += (p.bytes + 255) / 256 * 256 + 256;  // every array keeps a little slack behind it
  }
  std::vector<unsigned char> host(total, 0);
  for (const Part& p : parts)
    if (p.bytes > 0) std::memcpy(host.data() + p.off, p.src, p.bytes);
  hipError_t e = blob.ensure(total);
  if (e == hipSuccess) e = hipMemcpy(blob.ptr, host.data(), total, hipMemcpyHostToDevice);
  const char* b = (const char*)blob.ptr;
  stat = (const double*)(b + parts[0].off);
  mov = (const double*)(b + parts[1].off);
  movy = (const double*)(b + parts[2].off);
  segs = (const double*)(b + parts[3].off);
  cold = (const double*)(b + parts[4].off);
  hot32 = (const float*)(b + parts[5].off);
  coop_trips = (const double*)(b + parts[6].off);
  xhdr = (const double*)(b + parts[7].off);
  xrec = (const double*)(b + parts[8].off);
  xpl = (const double*)(b + parts[9].off);
  n_segs = lay.n_segs;
  n_sorted = (int)lay.n_sorted;
  n_xrec = (int)lay.xrec.size();
  has_f32 = false;
  for (int s = 0; s < lay.n_segs; ++s) has_f32 = has_f32 || lay.segs[8 * (size_t)s] >= 5.0;
  return e;
}

static bool parse_device_list(const char* e, TorOptions& o) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return false;
  o.device_count = 0;
  if (std::strcmp(e, "all") == 0) {
    for (int d = 0; d < count && d < TOR_MAX_DEVICES; ++d) o.devices[o.device_count++] = d;
    return true;
  }
  const char* p = e;
  while (*p) {
    char* endp = nullptr;
    const long v = std::strtol(p, &endp, 10);
    if (endp == p || v < 0 || v >= count || o.device_count >= TOR_MAX_DEVICES) return false;
    o.devices[o.device_count++] = (int32_t)v;
    p = endp;
    if (*p == ',') ++p;
    else if (*p) return false;
  }
  return o.device_count > 0;
}

// (why the last valid_options() call said no; the callers put it into tor_last_error)
thread_local std::string g_options_why;
const std::string& options_why() { return g_options_why; }

bool valid_options(const TorOptions* opt, TorOptions& o, bool for_drop_in) {
  g_options_why.clear();
  auto no = [](const char* why) { g_options_why = why; return false; };
  o = TorOptions{};
  o.struct_size = sizeof(TorOptions);
  o.seeding = TOR_SEED_PIXEL;
  o.arith = TOR_ARITH_STRICT;
  o.device = -1;
  o.shard_index = 0;
  o.shard_count = 1;
  o.row_tile = 1;
  o.gather = TOR_GATHER_AUTO;
  constexpr uint32_t kV1Size = 32;  // round-1 layout: everything up to and including `accel`
  if (opt) {
    if (opt->struct_size == sizeof(TorOptions)) o = *opt;
    else if (opt->struct_size == kV1Size) std::memcpy(&o, opt, kV1Size);
    else return no("struct_size is neither sizeof(TorOptions) nor the 32-byte round-1 layout");
    o.struct_size = sizeof(TorOptions);
  } else if (for_drop_in) {
    // tor_render() has the reference's signature and no options: a host that cannot pass TorOptions (the Nim
    // shim of INTEGRATION.md) steers the library through the environment.  None of these changes a pixel.
    o.accel = TOR_ACCEL_BLOCKS | TOR_ACCEL_F32;  // exact accelerations on by default for the drop-in
    // (a malformed value is an error, not a silent fall-back to one GPU / the default: ADVICE r2)
    if (const char* e = tor::knob("TOR_DEFAULT_ACCEL")) {
      char* endp = nullptr;
      const long v = std::strtol(e, &endp, 10);
      if (endp != e && *endp == 0 && v >= 0 && v <= (TOR_ACCEL_BLOCKS | TOR_ACCEL_F32)) o.accel = (int32_t)v;
      else return no("TOR_DEFAULT_ACCEL must be 0, 1, 2 or 3");
    }
    // TOR_DEFAULT_SEEDING = pixel | sample: which random streams the drop-in uses.  `pixel` (default) is the reference's
    // render.nim:59-67 -- one stream per pixel, the spp samples in sequence.  `sample` is the counter-based stream of
    // BASELINE.json's north_star, seed(row, col, sample): render.nim:59-60 re-seeds per pixel "to be able to parallelize the
    // outer loops"; this is the same idea one level down, and it is what lets 8 GPUs share a 1080p frame.  It is
    // the one knob of this list that changes pixels (a different, equally valid sample set; oracle mode SAMPLE / QUANTIZED).
    if (const char* e = tor::knob("TOR_DEFAULT_SEEDING")) {
      if (!std::strcmp(e, "sample")) o.seeding = TOR_SEED_SAMPLE;
      else if (!std::strcmp(e, "pixel")) o.seeding = TOR_SEED_PIXEL;
      else return no("TOR_DEFAULT_SEEDING must be pixel or sample");
    }
    if (const char* e = tor::knob("TOR_DEVICES")) {
      TorOptions t = o;
      if (parse_device_list(e, t)) o = t;
      else return no("TOR_DEVICES must be \"all\" or a comma list of HIP device ordinals that exist (at most TOR_MAX_DEVICES)");
    }
    if (const char* e = tor::knob("TOR_GATHER")) {
      if (!std::strcmp(e, "rccl")) o.gather = TOR_GATHER_RCCL;
      else if (!std::strcmp(e, "peer")) o.gather = TOR_GATHER_PEER;
      else if (!std::strcmp(e, "host")) o.gather = TOR_GATHER_HOST;
      else if (!std::strcmp(e, "auto")) o.gather = TOR_GATHER_AUTO;
      else return no("TOR_GATHER must be auto, rccl, peer or host");
    }
  }
  if (o.seeding != TOR_SEED_PIXEL && o.seeding != TOR_SEED_SAMPLE) return no("seeding must be TOR_SEED_PIXEL or TOR_SEED_SAMPLE");
  if (o.arith == TOR_ARITH_FUSED)
    return no("TOR_ARITH_FUSED was removed in round 5: it was not the reference's rounding (README.md:82) and TOR_ARITH_STRICT behind the conservative "
              "FMA screen is faster -- pass TOR_ARITH_STRICT (0)");
  if (o.arith != TOR_ARITH_STRICT) return no("arith must be TOR_ARITH_STRICT");
  if (o.accel < 0 || o.accel > (TOR_ACCEL_BLOCKS | TOR_ACCEL_F32)) return no("accel holds unknown TOR_ACCEL_* bits");
  if (o.shard_count < 1) o.shard_count = 1;  // (0 in a zero-initialised struct: whole image, one row per tile)
  if (o.row_tile == 0) o.row_tile = 1;
  if (o.row_tile < 0) return no("row_tile must be >= 1");
  if (o.shard_index < 0 || o.shard_index >= o.shard_count) return no("shard_index must lie in [0, shard_count)");
  if (o.device_count < 0 || o.device_count > TOR_MAX_DEVICES) return no("device_count must lie in [0, TOR_MAX_DEVICES]");
  if (o.gather < TOR_GATHER_AUTO || o.gather > TOR_GATHER_HOST) return no("gather must be a TOR_GATHER_* value");
  if (o.pixel_kernel < TOR_PIXEL_KERNEL_AUTO || o.pixel_kernel > TOR_PIXEL_KERNEL_WAVE) return no("pixel_kernel must be a TOR_PIXEL_KERNEL_* value");
  if (o.device_count > 1) {
    // the device list IS the sharding and the placement: shard_index / shard_count / device stay at their defaults (tor_render.h)
    if (o.shard_count != 1 || o.shard_index != 0) return no("with a device list, shard_index / shard_count must be left at 0 / 1: entry k renders shard k");
    // (`device` is ignored with a device list -- a zero-initialised TorOptions says 0 there, as it says 0 for shard_count and
    // row_tile, which are tolerated too; only a value that names a device OUTSIDE the list is a contradiction: ADVICE r3)
    if (o.device > 0) {
      bool listed = false;
      for (int k = 0; k < o.device_count; ++k) listed = listed || o.devices[k] == o.device;
      if (!listed) return no("with a device list, `device` must be -1, 0 or one of the listed ordinals (it is ignored)");
    }
    o.device = -1;
    for (int k = 0; k < o.device_count; ++k)
      if (o.devices[k] < 0) return no("negative ordinal in the device list");
  } else if (o.device_count == 1) {
    if (o.devices[0] < 0) return no("negative ordinal in the device list");
    o.device = o.devices[0];
  }
  return true;
}

// ---- lazily built device layouts ---------------------------------------------------------------
int ensure_layouts(TorContext* ctx, int accel) {
  const TorHittableVariant* objs = (const TorHittableVariant*)ctx->scene_bytes.data();
  const int64_t n = ctx->n_objects;
  const bool want_f32 = (accel & TOR_ACCEL_F32) != 0;
  const bool want_blocks = (accel & TOR_ACCEL_BLOCKS) != 0;
  if (want_f32 && !ctx->f32_built) {
    ctx->f32 = f32_options_for(objs, n);
    ctx->f32_built = true;
  }
  auto build_flat = [&](int v) -> int {
    if (ctx->flat_built[v]) return TOR_OK;
    std::vector<int64_t> ids((size_t)n);
    for (int64_t i = 0; i < n; ++i) ids[(size_t)i] = i;
    HostLayout lay;
    std::string err;
    if (!build_layout(objs, ids, lay, err, v == 1 ? &ctx->f32 : nullptr)) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_scene_upload: " + err);
    if (n == 0) lay.n_segs = 0;
    HIP_TRY(ctx->flat[v].put(lay));
    ctx->flat_built[v] = true;
    ctx->n_layouts_built += 1;
    return TOR_OK;
  };
  auto build_blocks = [&](int v) -> int {
    if (ctx->accel_built[v]) return TOR_OK;
    build_accel(objs, n, ctx->accel[v], v == 1 ? &ctx->f32 : nullptr);
    if (!ctx->accel[v].available) {
      ctx->accel_built[v] = true;  // (built: this scene has no culling layout)
      return TOR_OK;
    }
    // the layout counts as built only once every upload below has succeeded: a failed HIP call leaves it unbuilt (the
    // next launch tries again) instead of marked-built with null or stale device pointers
    struct Rollback {
      TorContext* c; int v; bool ok = false;
      ~Rollback() { if (!ok) { c->d_accel[v].release(); c->accel[v] = tor::HostAccel{}; } }
    } rollback{ctx, v};
    auto put = [](DeviceBuffer& b, const void* src, size_t bytes) -> hipError_t {
      hipError_t e = b.ensure(bytes > 0 ? bytes : 8);
      if (e == hipSuccess && bytes > 0) e = hipMemcpy(b.ptr, src, bytes, hipMemcpyHostToDevice);
      return e;
    };
    HIP_TRY(ctx->d_accel[v].always.put(ctx->accel[v].always, &ctx->accel[v].cold));
    HIP_TRY(put(ctx->d_accel[v].hot, ctx->accel[v].hot.data(), ctx->accel[v].hot.size() * 8));
    HIP_TRY(put(ctx->d_accel[v].grp, ctx->accel[v].groups.data(), ctx->accel[v].groups.size() * 8));
    if (ctx->accel[v].sp32) HIP_TRY(put(ctx->d_accel[v].hot32, ctx->accel[v].hot32.data(), ctx->accel[v].hot32.size() * 4));
    const size_t n_bnd_p = (ctx->accel[v].n_blocks + kPad - 1) / kPad * kPad;
    const size_t n_super_p = (n_bnd_p / kPad + kPad - 1) / kPad * kPad;
    // per slot: the float64 boxes, then the same records as float32 (8 floats each)
    const size_t slot_bytes = (8 * (n_bnd_p + 1 + n_super_p + 1)) * (8 + 4);
    if (slot_bytes * TorContext::kRing > ctx->bnd_ring.bytes) {
      // growing the ring frees the old one: no launch may still be reading it
      if (ctx->bnd_ring.ptr) HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(ctx->bnd_ring.ensure(slot_bytes * TorContext::kRing));
    }
    if (slot_bytes > ctx->bnd_slot_bytes) ctx->bnd_slot_bytes = slot_bytes;
    rollback.ok = true;
    ctx->accel_built[v] = true;
    ctx->n_layouts_built += 1;
    return TOR_OK;
  };
  int v32 = want_f32 ? 1 : 0;
  if (want_blocks) {
    int rc = build_blocks(v32);
    if (rc != TOR_OK) return rc;
    // the float32 kernel variant needs float32 block records; otherwise the launch stays on the float64 layouts
    if (v32 && ctx->accel[1].available && !ctx->accel[1].sp32) {
      v32 = 0;
      rc = build_blocks(0);
      if (rc != TOR_OK) return rc;
    }
    if (ctx->accel[v32].available) return TOR_OK;  // the launch uses the always-layout of the accel variant
  }
  return build_flat(v32);
}

}  // namespace tor

using tor::DeviceBuffer;
using tor::fail;
using tor::fail_hip;
using tor::valid_options;

extern "C" {

const char* tor_last_error(void) { return g_last_error.c_str(); }

const char* tor_version(void) { return "tor_mi355x 0.4 (gfx950)"; }

int32_t tor_knob_count(void) { return tor::kKnobCount; }

int tor_knob_info(int32_t i, const char** name, const char** dflt, const char** range, const char** when, const char** what) {
  if (i < 0 || i >= tor::kKnobCount) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_knob_info: index out of range");
  const tor::Knob& k = tor::kKnobs[i];
  if (name) *name = k.name;
  if (dflt) *dflt = k.dflt;
  if (range) *range = k.range;
  if (when) *when = k.when;
  if (what) *what = k.what;
  return TOR_OK;
}

int tor_last_render_timing(double out[5]) {
  if (!out) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_last_render_timing: out is NULL");
  for (int k = 0; k < 5; ++k) out[k] = g_last_timing[k];
  return TOR_OK;
}

int tor_context_create(int32_t device, TorContext** out) {
  if (!out) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_context_create: out is NULL");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(TOR_ERR_NO_DEVICE,
                std::string("no HIP device available (") + hipGetErrorString(e) +
                    "); libtor_mi355x has no CPU fallback");
  if (device < 0) HIP_TRY(hipGetDevice(&device));
  if (device >= count) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_context_create: device ordinal out of range");
  HIP_TRY(hipSetDevice(device));
  TorContext* ctx = new (std::nothrow) TorContext();
  if (!ctx) return fail(TOR_ERR_OUT_OF_MEMORY, "tor_context_create: host allocation failed");
  ctx->device = device;
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    delete ctx;
    return fail_hip(e, "hipGetDeviceProperties");
  }
  ctx->num_cus = prop.multiProcessorCount;
  if (const char* w = tor::knob("TOR_WAVES_PER_SIMD")) ctx->waves_override = std::atoi(w);
  if (const char* l = tor::knob("TOR_LPT_MIN_SPP")) ctx->lpt_min_spp = std::atoi(l);
  if (const char* c = tor::knob("TOR_COOP_MAX_PIXELS")) ctx->coop_max_pixels = std::atoll(c);
  if (const char* c = tor::knob("TOR_HOT_FRAC")) ctx->hot_frac = (float)std::atof(c);
  if (const char* c = tor::knob("TOR_PRIO_SHIFT")) ctx->prio_shift = std::atoi(c);
  if (const char* c = tor::knob("TOR_MIGRATE")) ctx->mig_mode = std::atoi(c);
  if (const char* c = tor::knob("TOR_SRV_FRAC")) ctx->srv_frac = (float)std::atof(c);
  if (const char* c = tor::knob("TOR_SRV_PATIENCE_US")) ctx->srv_patience_us = std::atoi(c);
  if (const char* c = tor::knob("TOR_SRV_MIN_FRAC")) ctx->srv_min_frac = (float)std::atof(c);
  if (const char* c = tor::knob("TOR_PUSH_THETA")) ctx->push_theta = (float)std::atof(c);
  if (const char* c = tor::knob("TOR_FLOOR_THETA")) ctx->floor_theta = (float)std::atof(c);
  if (const char* c = tor::knob("TOR_CHAIN_THETA")) ctx->chain_theta = (float)std::atof(c);
  if (const char* c = tor::knob("TOR_TAIL_LANES")) ctx->mig_tail_lanes = std::atoi(c);
  if (const char* c = tor::knob("TOR_SCREEN")) ctx->screen = std::atoi(c) != 0;
  if (const char* c = tor::knob("TOR_PLANE")) ctx->plane_screen = std::atoi(c);  // 0 off, 1 gated (default), 2 on every segment that carries the table
  if (const char* c = tor::knob("TOR_TAIL_REST")) ctx->mig_tail_rest = std::atoi(c);
  if (const char* c = tor::knob("TOR_MIG_FLAGS")) ctx->mig_flags = (unsigned)std::strtoul(c, nullptr, 0);
  if (const char* b = tor::knob("TOR_BLOCKS_PER_CU"))
    for (int s = 0; s < 2; ++s) ctx->max_blocks_per_cu[s][0] = ctx->max_blocks_per_cu[s][1] = std::atoi(b);
  e = ctx->counters.ensure(TorContext::kRing * TorContext::kSlotWords * sizeof(unsigned long long));
  if (e == hipSuccess) e = ctx->cam_ring.ensure(TorContext::kRing * sizeof(TorCamera));
  for (int i = 0; i < TorContext::kRing && e == hipSuccess; ++i) {
    e = hipEventCreate(&ctx->ev_start[i]);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_stop[i]);
  }
  if (e != hipSuccess) {
    tor_context_destroy(ctx);
    return fail_hip(e, "tor_context_create");
  }
  *out = ctx;
  return TOR_OK;
}

int tor_context_destroy(TorContext* ctx) {
  if (!ctx) return TOR_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  (void)tor_comm_destroy(ctx);
  for (int v = 0; v < 2; ++v) {
    ctx->flat[v].release();
    ctx->d_accel[v].release();
  }
  ctx->bnd_ring.release();
  ctx->counters.release();
  ctx->cam_ring.release();
  ctx->wave_log.release();
  for (int i = 0; i < TorContext::kRing; ++i) {
    ctx->tile_order[i].release();
  }
  ctx->probe_buf.release();
  ctx->mig_rec.release();
  ctx->mig_flag.release();
  ctx->scratch.release();
  ctx->slice.release();
  ctx->gather.release();
  ctx->frame.release();
  ctx->staging.release();
  for (hipEvent_t ev : ctx->chunk_events) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : ctx->ev_call) if (ev) (void)hipEventDestroy(ev);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  for (int i = 0; i < TorContext::kRing; ++i) {
    if (ctx->ev_fork[i]) (void)hipEventDestroy(ctx->ev_fork[i]);
    if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
  }
  for (int i = 0; i < TorContext::kRing; ++i) {
    if (ctx->ev_start[i]) (void)hipEventDestroy(ctx->ev_start[i]);
    if (ctx->ev_stop[i]) (void)hipEventDestroy(ctx->ev_stop[i]);
  }
  delete ctx;
  return TOR_OK;
}

int tor_context_set_stats(TorContext* ctx, int32_t enable) {
  if (!ctx) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_context_set_stats: ctx is NULL");
  ctx->collect_stats = enable != 0;
  return TOR_OK;
}

int tor_context_scene_counters(TorContext* ctx, int64_t out[3]) {
  if (!ctx || !out) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_context_scene_counters: NULL argument");
  out[0] = ctx->n_uploads;
  out[1] = ctx->n_cache_hits;
  out[2] = ctx->n_layouts_built;
  return TOR_OK;
}

// AoS -> SoA.  Objects are partitioned into one static segment and one segment per distinct
// (time0, time1) pair; closest-hit is order independent (hittables_lists.nim:48-55; ties are
// broken by the original index carried in the cold record), so the reordering is exact.
// The context keeps a byte copy of the list: an identical list is a cache hit, and the device layouts are
// built from the copy when a launch first needs them (ensure_layouts).
int tor_scene_upload(TorContext* ctx, TorHittableList world) {
  if (!ctx) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_scene_upload: ctx is NULL");
  if (world.len < 0 || (world.len > 0 && !world.objects))
    return fail(TOR_ERR_INVALID_ARGUMENT, "tor_scene_upload: bad HittableList");
  if (world.len > (int64_t)1 << 24) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_scene_upload: too many objects");
  ctx->n_uploads += 1;
  const size_t bytes = (size_t)world.len * sizeof(TorHittableVariant);
  if (ctx->scene_ready && ctx->n_objects == world.len && ctx->scene_bytes.size() == bytes &&
      (bytes == 0 || std::memcmp(ctx->scene_bytes.data(), world.objects, bytes) == 0)) {
    ctx->n_cache_hits += 1;
    return TOR_OK;
  }
  for (int64_t i = 0; i < world.len; ++i) {
    const TorHittableVariant& h = world.objects[i];
    if (h.kind != TOR_SPHERE && h.kind != TOR_MOVING_SPHERE) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_scene_upload: unknown HittableVariant kind");
    const uint8_t mk = h.kind == TOR_SPHERE ? h.u.sphere.material.kind : h.u.moving_sphere.material.kind;
    if (mk > TOR_DIELECTRIC) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_scene_upload: unknown Material kind");
  }
  HIP_TRY(hipSetDevice(ctx->device));
  // the scene buffers are about to be overwritten / reallocated: no launch may still be reading them
  if (ctx->scene_ready) HIP_TRY(hipDeviceSynchronize());
  ctx->scene_ready = false;
  ctx->scene_bytes.assign((const unsigned char*)world.objects, (const unsigned char*)world.objects + bytes);
  ctx->n_objects = world.len;
  for (int v = 0; v < 2; ++v) ctx->flat_built[v] = ctx->accel_built[v] = false;
  ctx->f32_built = false;
  ctx->scene_ready = true;
  return TOR_OK;
}

int32_t tor_shard_rows(int32_t nrows, int32_t row_tile, int32_t shard_index, int32_t shard_count,
                       int32_t* rows_out) {
  if (nrows <= 0) return 0;
  if (shard_count < 1) shard_count = 1;
  if (row_tile < 1) row_tile = 1;
  if (shard_index < 0 || shard_index >= shard_count) return 0;
  int32_t n = 0;
  for (int32_t r = 0; r < nrows; ++r)
    if ((r / row_tile) % shard_count == shard_index) {
      if (rows_out) rows_out[n] = r;
      ++n;
    }
  return n;
}

int tor_render_device(TorContext* ctx, const TorCamera* cam, int32_t nrows, int32_t ncols,
                      int32_t spp, float gamma_correction, int64_t max_depth, const TorOptions* opt,
                      double* d_pixels, void* hip_stream) {
  if (!ctx || !cam || !d_pixels) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_device: NULL argument");
  if (!ctx->scene_ready) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_device: no scene uploaded");
  // The reference divides by (ncols-1) and (nrows-1) (render.nim:64-65) and by spp
  // (canvas.nim:49); degenerate sizes are rejected instead of producing inf/NaN canvases.
  if (nrows < 2 || ncols < 2 || spp < 1)
    return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_device: need nrows >= 2, ncols >= 2, samples_per_pixel >= 1");
  if (max_depth > 0x7fffffff) max_depth = 0x7fffffff;
  TorOptions o;
  if (!valid_options(opt, o, false)) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_device: bad TorOptions: " + tor::options_why());
  if (o.device_count > 1) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_device: one context renders on one device (device lists: tor_render_opt)");
  hipStream_t stream = (hipStream_t)hip_stream;
  HIP_TRY(hipSetDevice(ctx->device));

  const int32_t local_rows = tor_shard_rows(nrows, o.row_tile, o.shard_index, o.shard_count, nullptr);
  const long long npix = (long long)local_rows * ncols;
  const long long n_values = npix * 3;
  ctx->timing_valid = false;
  ctx->last_samples = 0;
  ctx->last_migrate = false;
  if (npix == 0) return TOR_OK;

  // a ring slot (events, camera, bounds, counters, tile schedule) is reused every kRing launches: its previous
  // launch must be done
  const int slot = (int)(ctx->launches % TorContext::kRing);
  if (ctx->launches >= TorContext::kRing) HIP_TRY(hipEventSynchronize(ctx->ev_stop[slot]));
  // One stream per context while launches are in flight (tor_render.h): the per-launch state (work counters, tile
  // schedule, probe buffer, hand-off queue) is ordered by the stream.  A launch on ANOTHER stream is accepted only once
  // the previous launch of this context has finished.
  if (ctx->launches > 0 && ctx->last_stream_valid && ctx->last_stream != hip_stream) {
    const hipError_t q = hipEventQuery(ctx->ev_stop[ctx->last_slot]);
    if (q == hipErrorNotReady)
      return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_device: the previous launch of this context is still running on a different stream -- "
                                            "launches of one context that may overlap must use ONE stream (or use one context per stream)");
    if (q != hipSuccess) return fail_hip(q, "hipEventQuery");
  }
  ctx->last_stream = hip_stream;
  ctx->last_stream_valid = true;
  unsigned long long* const slot_counters = (unsigned long long*)ctx->counters.ptr + (size_t)slot * TorContext::kSlotWords;
  HIP_TRY(hipMemsetAsync(slot_counters, 0, TorContext::kSlotWords * sizeof(unsigned long long), stream));
  if (max_depth <= 0 || ctx->n_objects < 0) {
    // render.nim:25: the bounce loop does not run -> every sample is black -> pow(0, g) = 0
    HIP_TRY(hipMemsetAsync(d_pixels, 0, (size_t)n_values * 8, stream));
    return TOR_OK;
  }
  if (o.seeding == TOR_SEED_SAMPLE) HIP_TRY(hipMemsetAsync(d_pixels, 0, (size_t)n_values * 8, stream));

  tor::KParams p{};
  // Small SEED_PIXEL frames: one wave per pixel (coop_pixel_kernel) -- the lane-per-pixel kernel would be bound by
  // the latency of the longest pixel chain.  Same canvas bit for bit; needs only the float64 flat layout.
  // Split fraction (measured: tools/split_sweep.py, with the tiles ordered by their longest chain): the wave-per-pixel
  // kernel saturates at ~285 Msamples/s, the lane kernel (with both exact accelerations) is bound by its longest chains;
  // the best share of the probed work for the wave kernel is ~20 000 / pixels, capped at 0.2 (C1 at 100 spp: 16-17 ms for
  // anything in 0.15-0.35, at 1000 spp 130 ms at 0.15 against 163 ms at 0.45) and floored at 0.02 (only the glass-sphere
  // tiles); from ~1.2 M pixels on the lane kernel keeps everything.
  float split_frac = ctx->split_frac;
  if (split_frac < 0.0f) {
    split_frac = 0.0f;
    if (o.accel == (TOR_ACCEL_BLOCKS | TOR_ACCEL_F32) && npix <= 1200000) {
      split_frac = 20000.0f / (float)npix;
      if (split_frac > 0.2f) split_frac = 0.2f;
      if (split_frac < 0.02f) split_frac = 0.02f;
    }
  }
  // Layout configuration of a launch: which device arrays the kernel walks for `accel`, the per-call block bounds and the
  // LDS staging decision.  A lambda because two launches of one call may use it: the frame's kernel with the caller's
  // accel bits and -- SEED_PIXEL -- the cost probe, which always runs with both exact accelerations when the scene has
  // them (it only COUNTS closest-hit queries per pixel; the count does not depend on how the hit is found).
  auto configure = [&](tor::KParams& q, int accel, std::vector<double>& bnd_host, std::vector<float>& bnd32_host, bool& use_accel, int& stage_wg,
                       const tor::HostAccel*& hacc_out) -> int {
    {
      const int rc = tor::ensure_layouts(ctx, accel);
      if (rc != TOR_OK) return rc;
    }
    // TOR_ACCEL_F32 layout variant; with TOR_ACCEL_BLOCKS it needs float32 block records (HostAccel::sp32),
    // otherwise the whole launch stays on the float64 layout
    int v32 = (accel & TOR_ACCEL_F32) ? 1 : 0;
    if (v32 && (accel & TOR_ACCEL_BLOCKS) && ctx->accel[1].available && !ctx->accel[1].sp32) v32 = 0;
    auto use_layout = [&](const tor::DeviceLayout& L) {
      q.stat = L.stat;
      q.mov = L.mov;
      q.movy = L.movy;
      q.segs = L.segs;
      q.cold = L.cold;
      q.hot32 = L.has_f32 ? L.hot32 : nullptr;
      q.xhdr = L.xhdr;
      q.xrec = L.xrec;
      q.xpl = ctx->plane_screen != 0 ? L.xpl : nullptr;
      q.plane_gate2 = ctx->plane_screen == 2 ? 0.0 : 4.0 * tor::kPlaneGate * tor::kPlaneGate;
      // stage two of the plane-screened segments reads its records per lane: from LDS when the table fits beside the per-wave
      // queues -- and, with sample streams, the camera-ray reservoirs -- at this launch's workgroups per CU (random_scene: 15.9 KB + 18.7 KB
      // + 18.8 KB of 53.3 KB) and inside the 64 KB a launch may
      // ask for without an opt-in, else through the vector cache.  Only the ARITH 2 variants stage it (brute-force layouts
      // behind the screen: tor_kernels.hip dynamic_lds), so the room is only counted there.
      q.xrec_lds_doubles = 0;
      if (q.xpl != nullptr && ctx->screen && accel == 0) {
        const size_t wgs = (size_t)std::max(1, ctx->max_blocks_per_cu[o.seeding][0]);
        const size_t need = (size_t)L.n_xrec * 8 + (size_t)tor::integrate_fixed_lds_bytes(0, 0, o.seeding);
        if (need <= (size_t)(160 * 1024) / wgs - 1024 && need <= (size_t)64 * 1024) q.xrec_lds_doubles = L.n_xrec;
      }
      q.n_segs = L.n_segs;
    };
    const bool blocks_avail = (accel & TOR_ACCEL_BLOCKS) && ctx->accel_built[v32] && ctx->accel[v32].available;
    if (!blocks_avail) use_layout(ctx->flat[v32]);
    for (int k = 0; k < 3; ++k) q.org[k] = ctx->f32.origin[k];
    q.bnd = nullptr;
    q.spatial_base = 0;
    q.shot = nullptr;
    q.sgrp = nullptr;
    q.shot_lds_doubles = 0;
    q.shot_stride = 8;
    use_accel = false;
    stage_wg = 0;  // > 0: compact records staged in LDS, at most this many workgroups per CU
    const tor::HostAccel& hacc = ctx->accel[v32];
    hacc_out = &hacc;
    if (blocks_avail) {
      // rays carry the camera's shutter times, or 0 after a metal / dielectric bounce (rays.nim:19)
      const double t_lo = std::fmin(0.0, std::fmin(cam->shutter_open, cam->shutter_close));
      const double t_hi = std::fmax(0.0, std::fmax(cam->shutter_open, cam->shutter_close));
      use_accel = tor::compute_block_bounds(hacc, t_lo, t_hi, bnd_host);
      if (!use_accel) {  // non-finite ray-time range: brute force over the flat layout
        const int rc = tor::ensure_layouts(ctx, accel & ~TOR_ACCEL_BLOCKS);
        if (rc != TOR_OK) return rc;
        use_layout(ctx->flat[v32]);
      }
    }
    if (use_accel) {
      use_layout(ctx->d_accel[v32].always);
      q.bnd = (const double*)((char*)ctx->bnd_ring.ptr + (size_t)slot * ctx->bnd_slot_bytes);  // filled below (async copy)
      q.spatial_base = (int)hacc.spatial_base;
      q.n_super = (int)(tor::accel_boxes_padded(hacc) / tor::kPad);
      q.two_level = hacc.two_level ? 1 : 0;
      q.shot = (const double*)ctx->d_accel[v32].hot.ptr;
      q.sgrp = (const double*)ctx->d_accel[v32].grp.ptr;
      q.shot_stride = hacc.hot_stride;
      // LDS staging of the compact records next to the per-wave queues (18 KB per workgroup, 160 KB per
      // CU): see the staging decision below.
      q.shot32 = nullptr;
      if (v32 && hacc.sp32) {
        q.shot32 = (const float*)ctx->d_accel[v32].hot32.ptr;
        q.shot32_stride = hacc.hot32_stride;
        q.shot32_block_stride = hacc.hot32_block_stride;
        q.sp_mc0max = hacc.sp_mc0max; q.sp_dcmax = hacc.sp_dcmax;
        q.sp_t0 = hacc.sp_t0; q.sp_dt = hacc.sp_dt;
      }
      // LDS staging of the block records next to the per-wave queues (10 KB per workgroup in these variants, 160 KB per CU): the
      // float32 pair records (TOR_ACCEL_F32; survivors are re-tested from the cold records -- staging the float64
      // records as well measured +2 % at best) or the float64 compact records.  It must fit at this mode's
      // workgroups/CU, else global loads (through L2).
      size_t hot_bytes = q.shot32 ? hacc.hot32.size() * 4 : hacc.hot.size() * 8;
      size_t bnd32_stage_floats = 0;
      if (q.shot32) {
        // float32 boxes for the slab tests; on two-level scenes the lanes read the block boxes themselves: stage them
        q.sp_bmax = tor::block_bounds_f32(bnd_host, ctx->f32.origin, bnd32_host, &q.sp_hmin);
        q.bnd32 = (const float*)((const char*)q.bnd + bnd_host.size() * 8);
        if (hacc.two_level) bnd32_stage_floats = 8 * tor::accel_boxes_padded(hacc);
      }
      const char* st = tor::knob("TOR_STAGE_LDS");
      const size_t hard_cap = st ? (size_t)std::atoll(st) : (size_t)1 << 30;
      int& wg = stage_wg;
      wg = 0;
      auto fits = [&](size_t bytes, int wgs) {
        return bytes <= hard_cap && bytes + (size_t)tor::integrate_fixed_lds_bytes(1, q.shot32 != nullptr ? 1 : 0) <= (size_t)(160 * 1024) / (size_t)wgs - 1024;
      };
      // Workgroups per CU come first: on the 1601-object animation frames 3 workgroups/CU reading the records through L2
      // render 2339 Msamples/s, 2 workgroups/CU with the records in LDS 1914.  So the launch keeps its full workgroup count
      // and stages what fits beside it: the block boxes of a two-level scene first (6.6 KB there; every lane reads 8 of
      // them per super box it expands), then the block records if there is still room; only a launch whose mode runs at 2
      // workgroups/CU anyway gets the bigger LDS share.
      const int full = ctx->max_blocks_per_cu[o.seeding][accel != 0];
      const bool stage_boxes = q.shot32 && bnd32_stage_floats > 0 && fits(bnd32_stage_floats * 4, full);
      const size_t box_bytes = stage_boxes ? bnd32_stage_floats * 4 : 0;
      bool stage_hot = fits(hot_bytes + box_bytes, full);
      wg = (stage_hot || stage_boxes) ? full : 0;
      if (!q.shot32 && !stage_hot && full > 2 && fits(hot_bytes, full - 1)) {
        // (the float64 compact records of TOR_ACCEL_BLOCKS alone are read by every lane for every block it enters: there
        // one workgroup fewer with the records in LDS wins, 1212 against 1140 Msamples/s on the same frames)
        stage_hot = true;
        wg = full - 1;
      }
      q.shot_lds_doubles = (stage_hot && !q.shot32) ? (int)hacc.hot.size() : 0;
      q.shot32_lds_floats = (stage_hot && q.shot32) ? (int)hacc.hot32.size() : 0;
      q.bnd32_lds_floats = stage_boxes ? (int)bnd32_stage_floats : 0;
    }
    return TOR_OK;
  };
  // Launch configuration, shared by the hand-off decision below and the launch itself.
  bool use_accel = false;
  int stage_wg = 0;  // > 0: compact records staged in LDS, at most this many workgroups per CU
  const tor::HostAccel* hacc_p = nullptr;
  bool configured = false;
  // launch shape: workgroups per CU for this mode (fewer when the LDS staging needs the room); the
  // kernel variant's register budget follows it
  int cap = 0, waves_per_simd = 0;
  long long resident_waves = 0;
  auto configure_frame = [&]() -> int {
    if (configured) return TOR_OK;
    const int rc = configure(p, o.accel, ctx->bnd_host[slot], ctx->bnd32_host[slot], use_accel, stage_wg, hacc_p);
    if (rc != TOR_OK) return rc;
    cap = ctx->max_blocks_per_cu[o.seeding][o.accel != 0];
    if (cap < 1) cap = 4;
    if (stage_wg > 0 && stage_wg < cap) cap = stage_wg;
    waves_per_simd = ctx->waves_override > 0 ? ctx->waves_override : ((cap >= 2 && cap <= 4) ? cap : 4);
    int bpc_eff = tor::integrate_blocks_per_cu(p, o.seeding, o.arith, waves_per_simd);
    if (bpc_eff > cap) bpc_eff = cap;
    resident_waves = (long long)ctx->num_cus * bpc_eff * (tor::kThreads / 64);
    configured = true;
    return TOR_OK;
  };
  const long long n_tiles = (npix + tor::kTilePixelsHost - 1) / tor::kTilePixelsHost;
  // Chain hand-off (DESIGN 4.7 (HISTORY 4.10)): the lane kernel pushes its long chains to server waves inside the same launch -- it
  // replaces both the whole-frame wave kernel and split mode wherever the launch's kernel variant carries the servers
  // (both exact accelerations, single-level culling layout with float32 records, <= 128 block boxes) and the probe runs.
  // Decided HERE, in full, before split mode and the wave-per-pixel kernel are ruled out: a frame whose hand-off cannot
  // run after all (non-finite shutter range -> no culling, a variant without servers, a tiny machine) keeps those two
  // (ADVICE r3).
  bool migrate = false;
  if (o.seeding == TOR_SEED_PIXEL && o.pixel_kernel == TOR_PIXEL_KERNEL_AUTO && ctx->mig_mode != 0 && o.accel == (TOR_ACCEL_BLOCKS | TOR_ACCEL_F32) &&
      ctx->lpt_min_spp > 0 && spp >= ctx->lpt_min_spp && npix > tor::kTilePixelsHost && !ctx->collect_stats && ctx->n_objects > 0) {
    const int rc = configure_frame();
    if (rc != TOR_OK) return rc;
    const tor::HostAccel& ha = ctx->accel[1];
    migrate = ctx->accel_built[1] && ha.available && ha.sp32 && !ha.two_level && tor::accel_boxes_padded(ha) <= 128 &&
              use_accel && n_tiles > 1 && tor::integrate_variant_serves_chains(p, o.seeding) && resident_waves >= 8;
  }
  const bool split_applies = !migrate && o.pixel_kernel == TOR_PIXEL_KERNEL_AUTO && split_frac > 0.0f && ctx->lpt_min_spp > 0 && spp >= ctx->lpt_min_spp &&
                             npix >= ctx->split_min_pixels && npix <= ctx->split_max_pixels && !ctx->collect_stats && ctx->n_objects > 0;
  const bool want_wave_kernel = o.pixel_kernel == TOR_PIXEL_KERNEL_WAVE ||
                                (o.pixel_kernel == TOR_PIXEL_KERNEL_AUTO && ctx->coop_max_pixels > 0 && npix <= ctx->coop_max_pixels && !split_applies && !migrate);
  if (o.seeding == TOR_SEED_PIXEL && !ctx->collect_stats && want_wave_kernel && ctx->n_objects > 0) {
    const int rc = tor::ensure_layouts(ctx, 0);
    if (rc != TOR_OK) return rc;
    const tor::DeviceLayout& L = ctx->flat[0];
    tor::KParams wp{};
    wp.cold = L.cold;
    wp.n_cold_slots = L.n_sorted;
    wp.coop_slots = (L.n_sorted + 63) / 64 * 64;
    wp.coop_trips = L.coop_trips;
    const int bpc = tor::coop_blocks_per_cu(wp, o.arith);
    if (bpc > 0) {
      wp.nrows = nrows; wp.ncols = ncols; wp.spp = spp; wp.max_depth = (int)max_depth;
      wp.shard_index = o.shard_index; wp.shard_count = o.shard_count; wp.row_tile = o.row_tile;
      wp.n_pixels = (unsigned)npix;
      wp.work_counter = slot_counters;
      wp.out = d_pixels;
      ctx->cam_host[slot] = *cam;
      wp.cam_dev = (const double*)((char*)ctx->cam_ring.ptr + (size_t)slot * sizeof(TorCamera));
      HIP_TRY(hipMemcpyAsync((void*)wp.cam_dev, &ctx->cam_host[slot], sizeof(TorCamera), hipMemcpyHostToDevice, stream));
      long long blocks = (npix + (tor::kThreads / 64) - 1) / (tor::kThreads / 64);
      const long long resident = (long long)ctx->num_cus * bpc;
      if (blocks > resident) blocks = resident;
      HIP_TRY(hipEventRecord(ctx->ev_start[slot], stream));
      HIP_TRY(tor::launch_coop(wp, o.arith, (int)blocks, stream));
      HIP_TRY(hipEventRecord(ctx->ev_stop[slot], stream));
      ctx->launches += 1;
      ctx->last_slot = slot;
      ctx->timing_valid = true;
      ctx->last_samples = (int64_t)npix * spp;
      HIP_TRY(tor::launch_finalize(d_pixels, n_values, 1.0 / (double)spp, 1.0 / (double)gamma_correction, stream));
      return TOR_OK;
    }
  }
  {
    const int rc = configure_frame();
    if (rc != TOR_OK) return rc;
  }
  const tor::HostAccel& hacc = *hacc_p;
  std::vector<double>& bnd_host = ctx->bnd_host[slot];
  p.nrows = nrows; p.ncols = ncols; p.spp = spp; p.max_depth = (int)max_depth;
  p.inv_spp = 1.0 / (double)spp; p.inv_ncols = 1.0 / (double)ncols; p.inv_row_tile = 1.0 / (double)(o.row_tile > 0 ? o.row_tile : 1);
  p.shard_index = o.shard_index; p.shard_count = o.shard_count; p.row_tile = o.row_tile;
  p.work_counter = slot_counters;
  p.stats = ctx->collect_stats ? slot_counters + 1 : nullptr;
  p.out = d_pixels;

  long long waves;
  p.n_pixels = (unsigned)npix;
  p.order = nullptr;
  p.pixel_cost = nullptr;
  ctx->last_probe_pixels = 0;
  p.sched = nullptr;
  if (o.seeding == TOR_SEED_PIXEL) {
    p.total_work = (unsigned long long)n_tiles * tor::kTilePixelsHost;  // tiles of 64 pixels
    p.chunk = tor::kTilePixelsHost;
    waves = n_tiles;
  } else {
    p.total_work = (unsigned long long)npix * (unsigned long long)spp;
    long long c = (long long)(p.total_work / (unsigned long long)(resident_waves * 16));
    // largest chunk of the guided schedule: up to 1024 samples, or one whole pixel when a pixel has more (<= 4096)
    const long long cap_c = (spp > 1024) ? (spp < 4096 ? spp : 4096) : 1024;
    if (c < 64) c = 64;
    if (c > cap_c) c = cap_c;
    p.chunk = (unsigned)((c >= spp) ? c : c / 64 * 64);
    waves = (long long)((p.total_work + 63) / 64);
  }
  if (waves > resident_waves) waves = resident_waves;
  int blocks = (int)((waves + (tor::kThreads / 64) - 1) / (tor::kThreads / 64));
  if (blocks < 1) blocks = 1;
  p.screen = ctx->screen ? 1 : 0;
  p.n_boxes = use_accel ? (int)tor::accel_boxes_padded(hacc) : 0;
  p.mig = nullptr;
  if (migrate) blocks = (int)(resident_waves / (tor::kThreads / 64));  // the whole machine: waves without a tile are servers at once
  p.n_waves = (unsigned)(blocks * (tor::kThreads / 64));
  p.wave_log = nullptr;
  if (ctx->collect_stats) {
    HIP_TRY(ctx->wave_log.ensure((size_t)p.n_waves * 64));
    HIP_TRY(hipMemsetAsync(ctx->wave_log.ptr, 0, (size_t)p.n_waves * 64, stream));
    p.wave_log = (unsigned long long*)ctx->wave_log.ptr;
    ctx->last_n_waves = (int64_t)p.n_waves;
  }

  // the camera travels in its own small device slot (one per in-flight launch)
  ctx->cam_host[slot] = *cam;
  p.cam_dev = (const double*)((char*)ctx->cam_ring.ptr + (size_t)slot * sizeof(TorCamera));
  HIP_TRY(hipMemcpyAsync((void*)p.cam_dev, &ctx->cam_host[slot], sizeof(TorCamera), hipMemcpyHostToDevice, stream));
  if (use_accel) {
    HIP_TRY(hipMemcpyAsync((void*)p.bnd, bnd_host.data(), bnd_host.size() * 8, hipMemcpyHostToDevice, stream));
    if (p.bnd32)
      HIP_TRY(hipMemcpyAsync((void*)p.bnd32, ctx->bnd32_host[slot].data(), ctx->bnd32_host[slot].size() * 4, hipMemcpyHostToDevice, stream));
  }
  if (o.seeding == TOR_SEED_PIXEL && ctx->lpt_min_spp > 0 && spp >= ctx->lpt_min_spp && n_tiles > 1) {
    // Cost-ordered schedule: a 2-spp probe (per-sample streams; it only counts closest-hit queries
    // per tile, it never touches the canvas) + a counting sort; ~2/spp of extra work.
    // (one buffer per context, not per ring slot: only the probe and the two sort kernels touch it, and the launches of a
    // context are ordered on one stream)
    DeviceBuffer& tile_cost = ctx->probe_buf;
    DeviceBuffer& tile_order = ctx->tile_order[slot];
    // per-pixel query counts of the probe, then per tile: sort key, probed work
    HIP_TRY(tile_cost.ensure(((size_t)npix + 2 * (size_t)n_tiles) * 4 + tor::kTileSortScratchBytes));
    HIP_TRY(tile_order.ensure((size_t)n_tiles * 4));
    HIP_TRY(hipMemsetAsync(tile_cost.ptr, 0, (size_t)npix * 4, stream));
    tor::KParams pp = p;
    if (ctx->probe_accel && (o.accel & TOR_ACCEL_BLOCKS) == 0) {
      // The probe only counts closest-hit queries per pixel -- the count does not depend on how a hit is found -- so a launch
      // without the block culling (whose ring slot of block bounds is free) still probes with both exact accelerations when
      // the scene has them: 2.5 ms instead of 7.7 ms in front of a float64 brute-force 1080p frame.
      tor::KParams pa = p;
      bool pa_accel = false;
      int pa_stage = 0;
      const tor::HostAccel* pa_h = nullptr;
      const int rc = configure(pa, TOR_ACCEL_BLOCKS | TOR_ACCEL_F32, ctx->probe_bnd_host[slot], ctx->probe_bnd32_host[slot], pa_accel, pa_stage, pa_h);
      if (rc != TOR_OK) return rc;
      if (pa_accel && pa.shot32 != nullptr && pa.bnd32 != nullptr) {
        HIP_TRY(hipMemcpyAsync((void*)pa.bnd, ctx->probe_bnd_host[slot].data(), ctx->probe_bnd_host[slot].size() * 8, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync((void*)pa.bnd32, ctx->probe_bnd32_host[slot].data(), ctx->probe_bnd32_host[slot].size() * 4, hipMemcpyHostToDevice, stream));
        pp = pa;
      }
    }
    pp.spp = ctx->probe_spp;
    pp.inv_spp = 1.0 / (double)ctx->probe_spp;
    pp.total_work = (unsigned long long)npix * (unsigned long long)ctx->probe_spp;
    ctx->last_probe_pixels = npix;
    pp.chunk = 256;
    pp.work_counter = slot_counters + 5;
    pp.stats = nullptr;
    pp.wave_log = nullptr;
    pp.out = nullptr;
    pp.pixel_cost = (unsigned*)tile_cost.ptr;
    long long pw = (long long)((pp.total_work + 63) / 64);
    const long long pres = (long long)ctx->num_cus * 3 * (tor::kThreads / 64);
    if (pw > pres) pw = pres;
    const int pblocks = (int)((pw + (tor::kThreads / 64) - 1) / (tor::kThreads / 64));
    pp.n_waves = (unsigned)(pblocks * (tor::kThreads / 64));
    HIP_TRY(tor::launch_probe(pp, pblocks, stream));
    // Split mode (mid-size frames): the lane kernel is bound by its longest pixel chains, the wave-per-pixel kernel by
    // its throughput -- so the most expensive tiles (split_frac of the probed cost) go to the wave kernel on a second
    // stream while the lane kernel renders the rest.  Disjoint pixels, same arithmetic: same canvas.
    tor::KParams wk{};
    bool split = false;
    if (split_applies) {
      const int rc = tor::ensure_layouts(ctx, 0);
      if (rc != TOR_OK) return rc;
      const tor::DeviceLayout& L = ctx->flat[0];
      wk = p;
      wk.cold = L.cold;
      wk.n_cold_slots = L.n_sorted;
      wk.coop_slots = (L.n_sorted + 63) / 64 * 64;
      wk.coop_trips = L.coop_trips;
      wk.order = (const unsigned*)tile_order.ptr;
      wk.split = slot_counters + 6;
      wk.work_counter = slot_counters + 7;
      wk.stats = nullptr;
      wk.wave_log = nullptr;
      split = tor::coop_blocks_per_cu(wk, o.arith) > 0;
    }
    unsigned* const tile_key = (unsigned*)tile_cost.ptr + npix;
    // schedule of the lane kernel (tor_kernels.hip): the hot chains -- a pixel chain is hot when it needs more than hot_frac of the
    // iterations an average wave runs in this frame
    //   average = total probed queries * spp / probe_spp / lanes
    p.prio_shift = ctx->prio_shift;
    p.sched = slot_counters + 8;
    tor::MigSchedule ms;
    ms.key_mode = ctx->key_mode;
    ms.probe_spp = ctx->probe_spp;
    if (migrate) {
      // queue of handed-over chains: 64-byte records + ready flags (zeroed per launch), control words in the ring slot
      const size_t cap = (size_t)(npix < (1ll << 20) ? npix : (1ll << 20));
      HIP_TRY(ctx->mig_rec.ensure(cap * 64));
      HIP_TRY(ctx->mig_flag.ensure(cap * 4));
      HIP_TRY(hipMemsetAsync(ctx->mig_flag.ptr, 0, cap * 4, stream));
      p.mig = slot_counters + TorContext::kMigWord0;
      p.mig_rec = (unsigned long long*)ctx->mig_rec.ptr;
      p.mig_flag = (unsigned*)ctx->mig_flag.ptr;
      p.mig_cap = (unsigned)cap;
      p.mig_tail_lanes = ctx->mig_tail_lanes;
      p.mig_flags = ctx->mig_flags;
      p.mig_patience = (unsigned)(ctx->srv_patience_us > 0 ? ctx->srv_patience_us : 0) * 100u;
      p.mig_tail_rest = ctx->mig_tail_rest;
      // (read per launch, not per context: the drop-in's cached contexts outlive any one caller's settings)
      double stall_s = ctx->mig_stall_s;
      if (const char* c = tor::knob("TOR_SRV_STALL_S")) stall_s = std::atof(c);
      p.mig_stall_ticks = stall_s > 0.0 ? (unsigned long long)(stall_s * 1e8) + 2ull : (stall_s < 0.0 ? 1ull : 0ull);  // (< 0, a test setting: every waiting server gives up at its first look)
      ms.mig = p.mig;
      ms.lavg_scale = (float)spp / (float)ctx->probe_spp / ((float)blocks * (float)tor::kThreads);
      ms.srv_frac = ctx->srv_frac;
      ms.srv_min_frac = ctx->srv_min_frac;
      ms.push_theta = ctx->push_theta;
      ms.chain_scale = (float)spp / (float)ctx->probe_spp / (float)npix;
      ms.chain_theta = ctx->chain_theta;
      ms.floor_theta = ctx->floor_theta;
      ms.blocks = blocks;
      ms.spp = spp;
      ms.max_depth = (int)max_depth;
    }
    HIP_TRY(tor::launch_tile_order((const unsigned*)tile_cost.ptr, (unsigned)npix, tile_key, tile_key + n_tiles, (unsigned*)tile_order.ptr,
                                   (int)n_tiles, split ? split_frac : 0.0f, split ? slot_counters + 6 : nullptr,
                                   split ? slot_counters : nullptr,
                                   ctx->hot_frac > 0.0f ? ctx->hot_frac * (float)spp / (float)ctx->probe_spp / (float)(resident_waves * 64) : -1.0f,
                                   p.sched, ms, stream));
    p.order = (const unsigned*)tile_order.ptr;
    if (split) {
      if (!ctx->stream2) HIP_TRY(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
      if (!ctx->ev_fork[slot]) {
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork[slot], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_join[slot], hipEventDisableTiming));
      }
      HIP_TRY(hipEventRecord(ctx->ev_start[slot], stream));
      HIP_TRY(hipEventRecord(ctx->ev_fork[slot], stream));
      HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork[slot], 0));
      const long long wave_blocks = (long long)ctx->num_cus * tor::coop_blocks_per_cu(wk, o.arith);
      HIP_TRY(tor::launch_coop(wk, o.arith, (int)wave_blocks, ctx->stream2));
      HIP_TRY(hipEventRecord(ctx->ev_join[slot], ctx->stream2));
      HIP_TRY(tor::launch_integrate(p, o.seeding, o.arith, waves_per_simd, blocks, stream));
      HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_join[slot], 0));
      HIP_TRY(hipEventRecord(ctx->ev_stop[slot], stream));
      ctx->launches += 1;
      ctx->last_slot = slot;
      ctx->timing_valid = true;
      ctx->last_samples = (int64_t)npix * spp;
      HIP_TRY(tor::launch_finalize(d_pixels, n_values, 1.0 / (double)spp, 1.0 / (double)gamma_correction, stream));
      return TOR_OK;
    }
  }
  ctx->last_migrate = migrate && p.mig != nullptr;
  if (ctx->last_migrate) {
    // A hand-off launch covers the whole machine and its waves WAIT for each other (servers for lane waves): two of them
    // interleaved on one GPU -- two contexts on one device, e.g. a device list that repeats an ordinal -- would each hold part
    // of the machine and wait for workgroups that can never start.  So hand-off launches of this process are chained per
    // device: each waits (on the device, not the host) for the previous one's end event.  Kernels without
    // inter-workgroup waits may still overlap it; they end on their own.
    std::lock_guard<std::mutex> lock(g_handoff_mutex);
    hipEvent_t& ev = g_handoff_event[ctx->device];
    if (ev) HIP_TRY(hipStreamWaitEvent(stream, ev, 0));
    else HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ctx->ev_start[slot], stream));
    HIP_TRY(tor::launch_integrate(p, o.seeding, o.arith, waves_per_simd, blocks, stream));
    HIP_TRY(hipEventRecord(ctx->ev_stop[slot], stream));
    HIP_TRY(hipEventRecord(ev, stream));
  } else {
    HIP_TRY(hipEventRecord(ctx->ev_start[slot], stream));
    HIP_TRY(tor::launch_integrate(p, o.seeding, o.arith, waves_per_simd, blocks, stream));
    HIP_TRY(hipEventRecord(ctx->ev_stop[slot], stream));
  }
  ctx->launches += 1;
  ctx->last_slot = slot;
  ctx->timing_valid = true;
  ctx->last_samples = (int64_t)npix * spp;
  // canvas.nim:47-54
  const double scale = 1.0 / (double)spp;
  const double gamma = 1.0 / (double)gamma_correction;
  HIP_TRY(tor::launch_finalize(d_pixels, n_values, scale, gamma, stream));
  return TOR_OK;
}

int tor_quantize_rgb8_device(TorContext* ctx, const double* d_pixels, int64_t n_values, uint8_t* d_rgb8,
                             void* hip_stream) {
  if (!ctx || !d_pixels || !d_rgb8 || n_values < 0)
    return fail(TOR_ERR_INVALID_ARGUMENT, "tor_quantize_rgb8_device: bad argument");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(tor::launch_quantize(d_pixels, n_values, d_rgb8, (hipStream_t)hip_stream));
  return TOR_OK;
}

// ---- video output stage (io/rgb.nim, io/color_conversions.nim, io/h264.nim) --------------------

int64_t tor_h264_frame_bytes(int32_t width, int32_t height) {
  // 4:2:0: even sizes.  A size that is not a multiple of 16 is coded with its last macroblock row / column padded by edge
  // replication and cropped away again by the SPS (tor_h264_stream_header): the standard's way (H.264 7.4.2.1.1).  The
  // reference has a TODO there (h264.nim:178): its SPS announces ceil(size / 16) macroblocks, its flushFrame writes
  // floor(size / 16) -- at 1920x1080, BASELINE configs[4]'s size, a stream no decoder accepts.
  if (width < 2 || height < 2 || (width & 1) || (height & 1) || width > 65520 || height > 65520) return TOR_ERR_INVALID_ARGUMENT;
  const int64_t n_mb = (int64_t)((width + 15) >> 4) * ((height + 15) >> 4);
  return n_mb * 386 + 8;  // slice header 9 + 384 per macroblock + 2 per macroblock after the first + stop byte
}

// initSPS (h264.nim:90-142) followed by the constant PPS (h264.nim:37), as H264Encoder.init writes
// them (h264.nim:174-176).  Returns the number of bytes written or a negative status.
int tor_h264_stream_header(int32_t width, int32_t height, uint8_t* out, int32_t cap) {
  if (!out || tor_h264_frame_bytes(width, height) < 0) return TOR_ERR_INVALID_ARGUMENT;
  std::vector<uint8_t> b = {0x00, 0x00, 0x00, 0x01};
  unsigned acc = 0;
  int nbits = 0;
  auto put = [&](int n, unsigned v) {  // MSB first
    for (int k = n - 1; k >= 0; --k) {
      acc = (acc << 1) | ((v >> k) & 1u);
      if (++nbits == 8) { b.push_back((uint8_t)acc); acc = 0; nbits = 0; }
    }
  };
  auto ue = [&](unsigned v) {  // putGolomb, h264.nim:73-78
    int size = 1;
    unsigned t = v + 1;
    while ((t >>= 1) != 0) ++size;
    put(2 * size - 1, v + 1);
  };
  put(1, 0); put(2, 3); put(5, 7);        // forbidden_zero_bit, nal_ref_idc, nal_unit_type = SPS
  put(8, 66);                             // baseline profile
  put(1, 0); put(1, 0); put(1, 0); put(1, 0); put(4, 0);  // constraint flags, reserved
  put(8, 10);                             // level_idc
  ue(0); ue(0); ue(0); ue(0);             // sps id, log2_max_frame_num-4, poc type, log2_max_poc_lsb-4
  ue(0); put(1, 0);                       // num_ref_frames, gaps_in_frame_num_value_allowed
  ue((unsigned)(((width + 15) >> 4) - 1));
  ue((unsigned)(((height + 15) >> 4) - 1));
  put(1, 1); put(1, 0);                   // frame_mbs_only, direct_8x8_inference
  const unsigned crop_r = (unsigned)((((width + 15) >> 4) << 4) - width) / 2, crop_b = (unsigned)((((height + 15) >> 4) << 4) - height) / 2;
  if (crop_r == 0 && crop_b == 0) {
    put(1, 0);                            // frame_cropping_flag: the reference never sets it (h264.nim:178 TODO) -- same bytes for multiples of 16
  } else {
    put(1, 1);                            // padded macroblocks are cropped away: offsets in units of 2 luma samples (4:2:0 frames)
    ue(0); ue(crop_r); ue(0); ue(crop_b);  // left, right, top, bottom
  }
  put(1, 0); put(1, 1);                   // vui_parameters_present, stop bit
  if (nbits > 0) b.push_back((uint8_t)(acc << (8 - nbits)));
  const uint8_t pps[8] = {0x00, 0x00, 0x00, 0x01, 0x68, 0xce, 0x38, 0x80};
  b.insert(b.end(), pps, pps + 8);
  if ((int)b.size() > cap) return TOR_ERR_INVALID_ARGUMENT;
  std::memcpy(out, b.data(), b.size());
  return (int)b.size();
}

int tor_encode_frame_device(TorContext* ctx, const double* d_pixels, int32_t nrows, int32_t ncols, uint8_t* d_slice,
                            uint8_t* d_y, uint8_t* d_cb, uint8_t* d_cr, void* hip_stream) {
  if (!ctx || !d_pixels || !d_slice) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_encode_frame_device: NULL argument");
  // h264.nim:178 "TODO cropping for non-multiple of 16": the reference silently drops the partial macroblock rows / columns
  // (flushFrame loops over `div 16`) although its SPS announces them; here they are padded and cropped (tor_h264_frame_bytes).
  if (tor_h264_frame_bytes(ncols, nrows) < 0)
    return fail(TOR_ERR_INVALID_ARGUMENT, "tor_encode_frame_device: width and height must be even (4:2:0) and at most 65520");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(tor::launch_encode_ipcm(d_pixels, nrows, ncols, d_slice, d_y, d_cb, d_cr, (hipStream_t)hip_stream));
  return TOR_OK;
}

// One turn of the animation driver's loop body (trace_of_radiance_animation.nim:181-196): render the
// frame, convert and pack it on the device, and hand back only the slice NAL unit (1.5 B per pixel
// over PCIe instead of the 24 B per pixel canvas).  Blocking.
int tor_render_frame_h264(TorContext* ctx, const TorCamera* cam, int32_t nrows, int32_t ncols, int32_t spp,
                          float gamma_correction, int64_t max_depth, const TorOptions* opt, uint8_t* slice_out,
                          int64_t cap) {
  if (!ctx || !cam || !slice_out) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_frame_h264: NULL argument");
  const int64_t n = tor_h264_frame_bytes(ncols, nrows);
  if (n < 0) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_frame_h264: width and height must be even (4:2:0) and at most 65520");
  if (cap < n) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_frame_h264: output buffer too small");
  if (opt && opt->shard_count > 1) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_frame_h264: whole frames only");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(ctx->scratch.ensure((size_t)nrows * ncols * 24));
  HIP_TRY(ctx->slice.ensure((size_t)n));
  int rc = tor_render_device(ctx, cam, nrows, ncols, spp, gamma_correction, max_depth, opt, (double*)ctx->scratch.ptr, nullptr);
  if (rc != TOR_OK) return rc;
  rc = tor::rerender_if_stalled(ctx, cam, nrows, ncols, spp, gamma_correction, max_depth, opt, (double*)ctx->scratch.ptr, nullptr);
  if (rc != TOR_OK) return rc;
  rc = tor_encode_frame_device(ctx, (const double*)ctx->scratch.ptr, nrows, ncols, (uint8_t*)ctx->slice.ptr, nullptr, nullptr,
                               nullptr, nullptr);
  if (rc != TOR_OK) return rc;
  HIP_TRY(hipMemcpy(slice_out, ctx->slice.ptr, (size_t)n, hipMemcpyDeviceToHost));
  return TOR_OK;
}

int tor_last_kernel_ms(TorContext* ctx, float* ms_out, int64_t* samples_out) {
  if (!ctx || !ms_out) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_last_kernel_ms: NULL argument");
  if (!ctx->timing_valid) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_last_kernel_ms: no timed launch");
  HIP_TRY(hipSetDevice(ctx->device));
  const int slot = (int)((ctx->launches - 1) % TorContext::kRing);
  HIP_TRY(hipEventSynchronize(ctx->ev_stop[slot]));
  HIP_TRY(hipEventElapsedTime(ms_out, ctx->ev_start[slot], ctx->ev_stop[slot]));
  if (samples_out) *samples_out = ctx->last_samples;
  // an asynchronous caller that comes here to learn that its launch is over also learns when the frame is NOT whole
  // (ADVICE r4): the hand-off's stall escape leaves holes, and only tor_context_handoff_stalled used to say so
  if (ctx->last_migrate && slot == ctx->last_slot) {
    unsigned long long w = 0;
    HIP_TRY(hipMemcpy(&w, (unsigned long long*)ctx->counters.ptr + (size_t)slot * TorContext::kSlotWords + TorContext::kMigWord0 + tor::kMigStalled,
                      sizeof w, hipMemcpyDeviceToHost));
    if (w != 0)
      return fail(TOR_ERR_INCOMPLETE, "tor_last_kernel_ms: the launch's chain hand-off stalled (not all of its workgroups were resident) and flagged the "
                                      "frame INCOMPLETE -- render it again (tor_context_handoff_stalled; the blocking entry points do so themselves)");
  }
  return TOR_OK;
}

int tor_kernel_ms_mean(TorContext* ctx, int32_t last_n, float* mean_ms_out, int32_t* n_used_out) {
  if (!ctx || !mean_ms_out || last_n < 1) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_kernel_ms_mean: bad argument");
  if (!ctx->timing_valid || ctx->launches < 1) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_kernel_ms_mean: no timed launch");
  HIP_TRY(hipSetDevice(ctx->device));
  int64_t n = last_n;
  if (n > ctx->launches) n = ctx->launches;
  if (n > TorContext::kRing) n = TorContext::kRing;
  double sum = 0.0;
  for (int64_t k = 0; k < n; ++k) {
    const int slot = (int)((ctx->launches - 1 - k) % TorContext::kRing);
    float ms = 0.f;
    HIP_TRY(hipEventSynchronize(ctx->ev_stop[slot]));
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_start[slot], ctx->ev_stop[slot]));
    sum += ms;
  }
  *mean_ms_out = (float)(sum / (double)n);
  if (n_used_out) *n_used_out = (int32_t)n;
  return TOR_OK;
}

int tor_last_wave_log(TorContext* ctx, uint64_t* out, int64_t cap_waves) {
  if (!ctx || !out) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_last_wave_log: NULL argument");
  if (!ctx->collect_stats || ctx->last_n_waves <= 0) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_last_wave_log: stats not enabled");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipDeviceSynchronize());
  const int64_t n = ctx->last_n_waves < cap_waves ? ctx->last_n_waves : cap_waves;
  HIP_TRY(hipMemcpy(out, ctx->wave_log.ptr, (size_t)n * 64, hipMemcpyDeviceToHost));
  return (int)n;
}

int64_t tor_last_pixel_cost(TorContext* ctx, uint32_t* out, int64_t cap_pixels) {
  if (!ctx || !out) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_last_pixel_cost: NULL argument");
  if (ctx->last_probe_pixels <= 0) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_last_pixel_cost: the last launch ran no cost probe");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipDeviceSynchronize());
  const int64_t n = ctx->last_probe_pixels < cap_pixels ? ctx->last_probe_pixels : cap_pixels;
  HIP_TRY(hipMemcpy(out, ctx->probe_buf.ptr, (size_t)n * 4, hipMemcpyDeviceToHost));
  return n;
}

int tor_last_handoff_counters(TorContext* ctx, uint64_t out[16]) {
  if (!ctx || !out) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_last_handoff_counters: NULL argument");
  if (ctx->launches < 1) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_last_handoff_counters: no launch yet");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long h[tor::kMigWords];
  HIP_TRY(hipMemcpy(h, (unsigned long long*)ctx->counters.ptr + (size_t)ctx->last_slot * TorContext::kSlotWords + TorContext::kMigWord0, sizeof h,
                    hipMemcpyDeviceToHost));
  const int idx[8] = {tor::kMigHead, tor::kMigTail, tor::kMigLaneWaves, tor::kMigSrvWgs, tor::kMigPush, tor::kMigServed, tor::kMigHotPushes, tor::kMigTailPushes};
  for (int k = 0; k < 8; ++k) out[k] = h[idx[k]];
  // [8..12]: microseconds after the kernel's start: work counter dry, last wave out of the lane loop, last hot chain served, last tail chain served; 0 = never
  const unsigned long long t0 = h[tor::kMigT0];
  auto us = [&](unsigned long long t) { return (t == 0 || t == ~0ull || t < t0) ? 0ull : (t - t0) / 100ull; };
  out[8] = us(h[tor::kMigTCounterDry]); out[9] = us(h[tor::kMigTLaneEnd]); out[10] = us(h[tor::kMigTHotDone]); out[11] = us(h[tor::kMigTTailDone]);
  out[12] = h[tor::kMigItsHot]; out[13] = h[tor::kMigItsTail]; out[14] = h[tor::kMigConverted]; out[15] = h[tor::kMigPushNow];
#ifdef TOR_SERVE_PROF
  // profiling build (make SERVE_PROF=1, tools/server_phases.py): shader-clock ticks per phase of the served bounces
  std::fprintf(stderr, "serve_prof query %llu lambertian %llu metal %llu dielectric %llu miss %llu setup %llu n_lambertian %llu n_metal %llu n_dielectric %llu "
               "n_miss %llu rounds %llu candidates %llu total %llu samples %llu q_slab %llu q_slots %llu q_load %llu q_test %llu q_min %llu\n", h[81], h[82], h[83], h[84], h[85], h[86], h[87], h[88], h[89], h[90], h[91], h[92], h[93], h[94], h[75], h[76], h[77], h[78], h[79]);
#endif
  return TOR_OK;
}

int tor_context_handoff_stalled(TorContext* ctx, int32_t* stalled_out, int64_t* total_out) {
  if (!ctx || !stalled_out) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_context_handoff_stalled: NULL argument");
  bool st = false;
  const int rc = tor::handoff_stalled(ctx, (hipStream_t)ctx->last_stream, &st);
  if (rc != TOR_OK) return rc;
  *stalled_out = st ? 1 : 0;
  if (total_out) *total_out = ctx->n_stalled_frames;
  return TOR_OK;
}

int tor_last_stats(TorContext* ctx, TorStats* out) {
  if (!ctx || !out) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_last_stats: NULL argument");
  if (!ctx->collect_stats) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_last_stats: stats not enabled");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long h[12];
  HIP_TRY(hipMemcpy(h, (unsigned long long*)ctx->counters.ptr + (size_t)ctx->last_slot * TorContext::kSlotWords, sizeof h, hipMemcpyDeviceToHost));
  out->hit_queries = h[1];
  out->object_tests = h[1] * (uint64_t)ctx->n_objects;
  out->candidates = h[2];
  out->wave_iterations = h[3];
  out->lane_slots = h[3] * 64;
  out->samples = h[4];
  // word 11 (= KParams.stats + 10): the cooperative variants count the objects their float32 block filter looked at apart from the
  // survivors that reach the exact float64 test; the per-lane block expansion (TOR_ACCEL_BLOCKS alone) runs the exact test on all 8
  out->block_tests = h[11];
  out->exact_tests = h[2] - h[11];
  return TOR_OK;
}

// ---- the drop-in: host canvas in, host canvas out ---------------------------------------

static std::mutex g_ctx_mutex;
static std::map<std::pair<int, int>, TorContext*> g_default_ctx;  // cached contexts per (device, replica)

}  // extern "C"

namespace tor {
// After a hand-off launch (ctx->last_migrate): waits for the stream and reads the launch's kMigStalled word.
int handoff_stalled(TorContext* ctx, hipStream_t stream, bool* stalled) {
  *stalled = false;
  if (!ctx->last_migrate) return TOR_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(stream));
  unsigned long long w = 0;
  HIP_TRY(hipMemcpy(&w, (unsigned long long*)ctx->counters.ptr + (size_t)ctx->last_slot * TorContext::kSlotWords + TorContext::kMigWord0 + tor::kMigStalled,
                    sizeof w, hipMemcpyDeviceToHost));
  *stalled = w != 0;
  return TOR_OK;
}

int rerender_if_stalled(TorContext* ctx, const TorCamera* cam, int32_t nrows, int32_t ncols, int32_t spp, float gamma_correction,
                        int64_t max_depth, const TorOptions* o, double* d_pixels, hipStream_t stream) {
  bool stalled = false;
  int rc = handoff_stalled(ctx, stream, &stalled);
  if (rc != TOR_OK || !stalled) return rc;
  ctx->n_stalled_frames += 1;
  const int saved = ctx->mig_mode;
  ctx->mig_mode = 0;
  rc = tor_render_device(ctx, cam, nrows, ncols, spp, gamma_correction, max_depth, o, d_pixels, stream);
  ctx->mig_mode = saved;
  return rc;
}

int default_context(int device, int replica, TorContext** out) {
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  TorContext*& ctx = g_default_ctx[{device, replica}];
  if (!ctx) {
    int rc = tor_context_create(device, &ctx);
    if (rc != TOR_OK) { ctx = nullptr; return rc; }
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { tor_context_destroy(ctx); ctx = nullptr; return fail_hip(e, "hipStreamCreate"); }
  }
  *out = ctx;
  return TOR_OK;
}
}  // namespace tor

extern "C" {

static std::mutex g_render_mutex;  // the drop-in is callable from one host thread at a time

int tor_render_opt(TorCanvas* canvas, const TorCamera* cam, TorHittableList world, int64_t max_depth,
                   const TorOptions* opt) {
  using clk = std::chrono::steady_clock;
  auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
  const clk::time_point t_call = clk::now();
  if (!canvas || !cam || !canvas->pixels) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render: NULL argument");
  TorOptions o;
  if (!valid_options(opt, o, true)) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render: bad options: " + tor::options_why());
  std::lock_guard<std::mutex> lock(g_render_mutex);
  for (double& t : g_last_timing) t = 0.0;
  if (o.device_count > 1) {
    const int rc = tor::render_multi_device(canvas, cam, world, max_depth, o, g_last_timing);
    g_last_timing[3] = ms_since(t_call);
    return rc;
  }
  int device = o.device;
  if (device < 0) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
      return fail(TOR_ERR_NO_DEVICE, std::string("no HIP device available (") + hipGetErrorString(e) +
                                         "); libtor_mi355x has no CPU fallback");
    HIP_TRY(hipGetDevice(&device));
  }
  TorContext* ctx = nullptr;
  int rc = tor::default_context(device, 0, &ctx);
  if (rc != TOR_OK) return rc;
  clk::time_point t0 = clk::now();
  const int64_t hits_before = ctx->n_cache_hits;
  rc = tor_scene_upload(ctx, world);
  if (rc != TOR_OK) return rc;
  g_last_timing[4] = ctx->n_cache_hits > hits_before ? 1.0 : 0.0;
  {
    // layouts are built here (not inside the timed kernel section) so that out[0] is the whole upload cost
    HIP_TRY(hipSetDevice(ctx->device));
    rc = tor::ensure_layouts(ctx, o.accel);
    if (rc != TOR_OK) return rc;
  }
  g_last_timing[0] = ms_since(t0);
  t0 = clk::now();
  const int32_t nrows = canvas->nrows, ncols = canvas->ncols;
  std::vector<int32_t> rows((size_t)(nrows > 0 ? nrows : 0));
  const int32_t local_rows = tor_shard_rows(nrows, o.row_tile, o.shard_index, o.shard_count, rows.data());
  const size_t row_bytes = (size_t)(ncols > 0 ? ncols : 0) * 24;
  HIP_TRY(ctx->scratch.ensure((size_t)(local_rows > 0 ? local_rows : 1) * (row_bytes > 0 ? row_bytes : 24)));
  // out[1] ("launch + kernels until the device is done") is measured on the device: one event in front of the launches, one
  // behind the last kernel; the D2H queues behind them without a host synchronisation in between, and out[2] is the rest
  // of the section (the part of the download that did not overlap nothing: copies + the host-side row placement)
  if (!ctx->ev_call[0]) {
    HIP_TRY(hipEventCreate(&ctx->ev_call[0]));
    HIP_TRY(hipEventCreate(&ctx->ev_call[1]));
  }
  HIP_TRY(hipEventRecord(ctx->ev_call[0], ctx->stream));
  rc = tor_render_device(ctx, cam, nrows, ncols, canvas->samples_per_pixel, canvas->gamma_correction,
                         max_depth, &o, (double*)ctx->scratch.ptr, ctx->stream);
  if (rc != TOR_OK) return rc;
  // (a hand-off launch whose servers gave up -- not all of its workgroups were resident -- left the frame incomplete: render it
  // again without the hand-off before anything reaches the caller's canvas)
  rc = tor::rerender_if_stalled(ctx, cam, nrows, ncols, canvas->samples_per_pixel, canvas->gamma_correction, max_depth, &o,
                                (double*)ctx->scratch.ptr, ctx->stream);
  if (rc != TOR_OK) return rc;
  HIP_TRY(hipEventRecord(ctx->ev_call[1], ctx->stream));
  rc = tor::download_rows(ctx, ctx->scratch.ptr, local_rows, row_bytes, o.shard_count > 1 ? rows.data() : nullptr,
                          (char*)canvas->pixels, ctx->stream);
  if (rc != TOR_OK) return rc;
  const double section = ms_since(t0);
  float dev_ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&dev_ms, ctx->ev_call[0], ctx->ev_call[1]));  // (both events are complete: download_rows waited for the stream)
  g_last_timing[1] = (double)dev_ms < section ? (double)dev_ms : section;
  g_last_timing[2] = section - g_last_timing[1];
  g_last_timing[3] = ms_since(t_call);
  return TOR_OK;
}

int tor_render(TorCanvas* canvas, const TorCamera* cam, TorHittableList world, int64_t max_depth) {
  return tor_render_opt(canvas, cam, world, max_depth, nullptr);
}

int tor_render_ptr(TorCanvas* canvas, const TorCamera* cam, const TorHittableList* world, int64_t max_depth) {
  if (!world) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_render_ptr: world is NULL");
  return tor_render_opt(canvas, cam, *world, max_depth, nullptr);
}

// Debug view of the TOR_ACCEL_BLOCKS layout (host only, no device needed): for the given object list and
// ray-time range, writes per spatial slot the original object index (-1 for padding) and per block its
// box {lo xyz, hi xyz}, then the super boxes.  Returns the number of blocks (0: no second level is built),
// or a negative status.  Used by the CPU test that checks the boxes are conservative.
int tor_debug_accel_layout(TorHittableList world, double t_lo, double t_hi, int64_t* slot_object, int64_t slot_cap,
                           double* block_boxes, double* super_boxes, int64_t box_cap, int32_t* two_level_out) {
  if (world.len < 0 || (world.len > 0 && !world.objects)) return TOR_ERR_INVALID_ARGUMENT;
  tor::HostAccel acc;
  tor::build_accel(world.objects, world.len, acc);
  if (!acc.available) return 0;
  std::vector<double> bnd;
  if (!tor::compute_block_bounds(acc, t_lo, t_hi, bnd)) return 0;
  const int64_t n_blocks = (int64_t)acc.n_blocks;
  const int64_t n_bnd_p = (n_blocks + tor::kPad - 1) / tor::kPad * tor::kPad;          // (per block, as this view reports them)
  const int64_t n_box_p = (int64_t)tor::accel_boxes_padded(acc), F = acc.fanout > 0 ? acc.fanout : 1;
  if (slot_cap < n_blocks * tor::kPad || box_cap < n_bnd_p) return TOR_ERR_INVALID_ARGUMENT;
  for (int64_t k = 0; k < n_blocks * tor::kPad; ++k) {
    int64_t orig = -1;
    if (acc.spatial[(size_t)k].valid) std::memcpy(&orig, &acc.cold[16 * (acc.spatial_base + (size_t)k) + 14], 8);
    slot_object[k] = orig;
  }
  // (a culling box stands for `fanout` consecutive blocks: block b is reported with the box that covers it; a super box covers
  // 8 boxes = 8 F blocks, reported per group of 8 BLOCKS as before -- the super box that covers the group's first block)
  for (int64_t b = 0; b < n_bnd_p; ++b)
    for (int c = 0; c < 6; ++c) block_boxes[6 * b + c] = (b / F < n_box_p) ? bnd[8 * (size_t)(b / F) + c] : std::nan("");
  const int64_t n_super = n_bnd_p / tor::kPad;
  for (int64_t sidx = 0; sidx < n_super; ++sidx)
    for (int c = 0; c < 6; ++c) {
      const int64_t sb = (sidx * tor::kPad / F) / tor::kPad;  // super box of the group's first block
      super_boxes[6 * sidx + c] = sb < n_box_p / tor::kPad ? bnd[8 * (size_t)(n_box_p + 1 + sb) + c] : std::nan("");
    }
  if (two_level_out) *two_level_out = acc.two_level ? 1 : 0;
  return (int)n_blocks;
}

// TOR_ACCEL_F32 self test (host build of tor_filter32.hpp, no device needed): ray i against sphere i.
// keep[i] = the float32 pre-filter's decision; need[i] = the float64 condition it must never miss,
// D > 0 and (half_b < 0 or c < 0) evaluated as spheres.nim:30-33 does (bit 1: a root > 0.001 exists).
int tor_selftest_filter32_host(int64_t n, const double* o, const double* d, const double* c0, const double* dc,
                               const int32_t* moving, const double* f, const double* r2, const double* origin,
                               int32_t* keep, int32_t* need) {
  if (n < 0 || !o || !d || !c0 || !dc || !moving || !f || !r2 || !origin || !keep || !need)
    return fail(TOR_ERR_INVALID_ARGUMENT, "tor_selftest_filter32_host: bad argument");
  for (int64_t i = 0; i < n; ++i) {
    const double* oo = o + 3 * i; const double* dd = d + 3 * i; const double* cc0 = c0 + 3 * i; const double* dcc = dc + 3 * i;
    const bool mv = moving[i] != 0;
    keep[i] = (int32_t)tor::filter_one(oo, dd, cc0, dcc, mv, f[i], r2[i], origin);
    double c[3] = {cc0[0], cc0[1], cc0[2]};
    if (mv) for (int k = 0; k < 3; ++k) c[k] = cc0[k] + dcc[k] * f[i];  // moving_spheres.nim:43
    const double ocx = oo[0] - c[0], ocy = oo[1] - c[1], ocz = oo[2] - c[2];
    const double a = dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2];
    const double hb = ocx * dd[0] + ocy * dd[1] + ocz * dd[2];
    const double cq = (ocx * ocx + ocy * ocy + ocz * ocz) - r2[i];
    const double disc = hb * hb - a * cq;
    int32_t nd = (disc > 0.0 && (hb < 0.0 || cq < 0.0)) ? 1 : 0;
    if (disc > 0.0) {
      const double root = std::sqrt(disc);
      const double s0 = (-hb - root) / a, s1 = (-hb + root) / a;
      if ((0.001 < s0 && s0 < INFINITY) || (0.001 < s1 && s1 < INFINITY)) nd |= 2;
    }
    need[i] = nd;
  }
  return TOR_OK;
}

// The conservative FMA screen of the strict object loop (tor_screen.hpp) on the HOST: ray i against sphere i, with the
// margins the kernel would use for a segment that holds only this object (the smallest margins the host ever hands out).
// keep[i] = the screen keeps the object; need[i] as in tor_selftest_filter32_host.
int tor_selftest_screen_host(int64_t n, const double* o, const double* d, const double* c0, const double* dc,
                             const int32_t* moving, const double* f, const double* r2, int32_t* keep, int32_t* need) {
  if (n < 0 || !o || !d || !c0 || !dc || !moving || !f || !r2 || !keep || !need)
    return fail(TOR_ERR_INVALID_ARGUMENT, "tor_selftest_screen_host: bad argument");
  auto up = [](double x) { return x * (1.0 + 0x1p-40); };
  for (int64_t i = 0; i < n; ++i) {
    const double* oo = o + 3 * i; const double* dd = d + 3 * i; const double* cc0 = c0 + 3 * i; const double* dcc = dc + 3 * i;
    const bool mv = moving[i] != 0;
    const double a = dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2];
    // tor_scene.cpp build_layout: segs[6], segs[7]
    const double reach = up(std::sqrt(cc0[0] * cc0[0] + cc0[1] * cc0[1] + cc0[2] * cc0[2]) + std::sqrt(std::fabs(r2[i])));
    const double travel = mv ? up(std::sqrt(dcc[0] * dcc[0] + dcc[1] * dcc[1] + dcc[2] * dcc[2])) : 0.0;
    const double s1 = std::fabs(oo[0]) + std::fabs(oo[1]) + std::fabs(oo[2]);
    const double d1 = std::fabs(dd[0]) + std::fabs(dd[1]) + std::fabs(dd[2]);
    double negmu, am;
    tor::screen_margins(s1 + reach + travel * std::fabs(mv ? f[i] : 0.0), d1, a, negmu, am);
    const double nf = -f[i];
    const double sx = mv ? tor::fma_(nf, dcc[0], oo[0] - cc0[0]) : oo[0] - cc0[0];
    const double sy = mv ? tor::fma_(nf, dcc[1], oo[1] - cc0[1]) : oo[1] - cc0[1];
    const double sz = mv ? tor::fma_(nf, dcc[2], oo[2] - cc0[2]) : oo[2] - cc0[2];
    keep[i] = tor::screen_filter(sx, sy, sz, dd[0], dd[1], dd[2], a, negmu, am, r2[i]) < 0 ? 1 : 0;
    double c[3] = {cc0[0], cc0[1], cc0[2]};
    if (mv) for (int k = 0; k < 3; ++k) c[k] = cc0[k] + dcc[k] * f[i];  // moving_spheres.nim:43
    const double ocx = oo[0] - c[0], ocy = oo[1] - c[1], ocz = oo[2] - c[2];
    const double hb = ocx * dd[0] + ocy * dd[1] + ocz * dd[2];
    const double cq = (ocx * ocx + ocy * ocy + ocz * ocz) - r2[i];
    const double disc = hb * hb - a * cq;
    int32_t nd = (disc > 0.0 && (hb < 0.0 || cq < 0.0)) ? 1 : 0;
    if (disc > 0.0) {
      const double root = std::sqrt(disc);
      const double s0 = (-hb - root) / a, s1r = (-hb + root) / a;
      if ((0.001 < s0 && s0 < INFINITY) || (0.001 < s1r && s1r < INFINITY)) nd |= 2;
    }
    need[i] = nd;
  }
  return TOR_OK;
}

// The SECOND form of the screen (tor_screen.hpp: expanded quadratic, normalised direction) on the host, same conventions as
// tor_selftest_screen_host.  variant 0: a static sphere through the general record (kind 10), a mover along y through kind 12,
// any other mover through the first form (as the kernel does); variant 1: static spheres through the common-height record (kind 11);
// variant 2: every pair through the PLANE screen (stage one) alone -- statics and movers along y by {cx, cz}, movers in general
// position by {c0x, c0z, dcx, dcz} (xkind 13).
int tor_selftest_screen2_host(int64_t n, const double* o, const double* d, const double* c0, const double* dc,
                              const int32_t* moving, const double* f, const double* r2, int32_t variant, int32_t* keep, int32_t* need) {
  if (n < 0 || !o || !d || !c0 || !dc || !moving || !f || !r2 || !keep || !need)
    return fail(TOR_ERR_INVALID_ARGUMENT, "tor_selftest_screen2_host: bad argument");
  int rc = tor_selftest_screen_host(n, o, d, c0, dc, moving, f, r2, keep, need);  // `need`, and `keep` of the pairs that stay on the first form
  if (rc != TOR_OK) return rc;
  auto up = [](double x) { return x * (1.0 + 0x1p-40); };
  for (int64_t i = 0; i < n; ++i) {
    const double* oo = o + 3 * i; const double* dd = d + 3 * i; const double* cc0 = c0 + 3 * i; const double* dcc = dc + 3 * i;
    const bool mv = moving[i] != 0;
    const bool general = mv && !(dcc[0] == 0.0 && dcc[2] == 0.0);
    if (general && variant != 2) continue;
    const double a = dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2];  // spheres.nim:30
    const double reach = up(std::sqrt(cc0[0] * cc0[0] + cc0[1] * cc0[1] + cc0[2] * cc0[2]) + std::sqrt(std::fabs(r2[i])));
    const double travel = mv ? up(std::sqrt(dcc[0] * dcc[0] + dcc[1] * dcc[1] + dcc[2] * dcc[2])) : 0.0;
    const tor::ScreenRay ray = tor::screen2_ray(oo[0], oo[1], oo[2], dd[0], dd[1], dd[2], a);
    int word;
    if (variant == 2) {  // stage one: the plane screen alone (R = this object's radius); a mover in general position through the 4-fma chain of xkind 13
      const tor::PlaneSeg ps = tor::plane_seg(ray, tor::plane_ray(ray), reach, travel, mv ? f[i] : 0.0, r2[i]);
      word = general ? tor::plane_word_mov(ps, cc0[0], cc0[2], dcc[0], dcc[2]) : tor::plane_word(ps, cc0[0], cc0[2]);
    } else if (mv) {
      const tor::ScreenSeg sg = tor::screen2_seg(ray, reach, travel, cc0[1], f[i]);
      word = tor::screen2_movy_y(sg, cc0[0], cc0[2], tor::screen2_Ky(cc0[0], cc0[2], r2[i]), dcc[1]);
    } else if (variant == 1) {
      const tor::ScreenSeg sg = tor::screen2_seg(ray, reach, 0.0, cc0[1], 0.0);
      word = tor::screen2_static_y(sg, cc0[0], cc0[2], tor::screen2_Ky(cc0[0], cc0[2], r2[i]));
    } else {
      const tor::ScreenSeg sg = tor::screen2_seg(ray, reach, 0.0, 0.0, 0.0);
      word = tor::screen2_static(sg, cc0[0], cc0[1], cc0[2], tor::screen2_K(cc0[0], cc0[1], cc0[2], r2[i]));
    }
    keep[i] = word < 0 ? 1 : 0;
  }
  return TOR_OK;
}

// TOR_ACCEL_F32 on the HOST over a whole scene: builds the same layout tor_scene_upload builds and walks its
// float32 segments (kinds 5/6/7) for each ray exactly as integrate_kernel does (same records, same
// tor_filter32.hpp code).  keep[ray * world.len + object] = 1 kept, 0 dropped, 2 object is on the float64
// loop (not eligible for the filter).  No device needed.
int tor_debug_filter32_scene(TorHittableList world, int64_t n_rays, const double* o, const double* d,
                             const double* time, int8_t* keep) {
  if (world.len < 0 || (world.len > 0 && !world.objects) || n_rays < 0 || !o || !d || !time || !keep)
    return fail(TOR_ERR_INVALID_ARGUMENT, "tor_debug_filter32_scene: bad argument");
  const tor::F32Options f32 = tor::f32_options_for(world.objects, world.len);
  std::vector<int64_t> ids((size_t)world.len);
  for (int64_t i = 0; i < world.len; ++i) ids[(size_t)i] = i;
  tor::HostLayout lay;
  std::string err;
  if (!tor::build_layout(world.objects, ids, lay, err, &f32)) return fail(TOR_ERR_INVALID_ARGUMENT, err);
  using tor::f2v;
  for (int64_t r = 0; r < n_rays; ++r) {
    int8_t* kr = keep + r * world.len;
    for (int64_t i = 0; i < world.len; ++i) kr[i] = 2;
    const double* oo = o + 3 * r; const double* dd = d + 3 * r;
    const double a = dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2];
    const tor::RayF32 r32 = tor::make_ray_f32(oo[0], oo[1], oo[2], dd[0], dd[1], dd[2], a, f32.origin[0], f32.origin[1], f32.origin[2]);
    for (int s = 0; s < lay.n_segs; ++s) {
      const double* sg = &lay.segs[8 * (size_t)s];
      const int kind = (int)sg[0];
      if (kind < 5) continue;
      const int begin = (int)sg[1], count = (int)sg[2] & 0xffffff, block0 = (int)sg[3];
      const double f64 = kind == 5 ? 0.0 : (time[r] - sg[4]) / sg[5];
      const tor::SegF32 s32 = tor::make_seg_f32(r32, f64, (float)sg[6], (float)sg[7]);
      const int stride = kind == 5 ? 10 : (kind == 6 ? 12 : 16);
      for (int i = 0; i < count; i += tor::kPad) {
        unsigned m = 0;
        for (int j = 0; j < tor::kPad / 2; ++j) {
          const float* rec = &lay.hot32[(size_t)begin + (size_t)(i / 2 + j) * stride];
          const f2v cx = {rec[0], rec[1]}, cy = {rec[2], rec[3]}, cz = {rec[4], rec[5]};
          f2v ocx = tor::oc_static32(r32.ox, cx), ocy = tor::oc_static32(r32.oy, cy), ocz = tor::oc_static32(r32.oz, cz);
          if (kind == 6) ocy = tor::oc_moving32(r32.oy, cy, (f2v){rec[10], rec[11]}, s32.nf);
          if (kind == 7) {
            ocx = tor::oc_moving32(r32.ox, cx, (f2v){rec[10], rec[11]}, s32.nf);
            ocy = tor::oc_moving32(r32.oy, cy, (f2v){rec[12], rec[13]}, s32.nf);
            ocz = tor::oc_moving32(r32.oz, cz, (f2v){rec[14], rec[15]}, s32.nf);
          }
          m = tor::filter_pair32(r32, s32, ocx, ocy, ocz, (f2v){rec[6], rec[7]}, (f2v){rec[8], rec[9]}, m);
        }
        m |= s32.wild;
        for (int j = 0; j < tor::kPad; ++j) {
          const size_t slot = (size_t)(block0 + i / tor::kPad) * tor::kPad + (size_t)j;
          int64_t orig;
          std::memcpy(&orig, &lay.cold[16 * slot + 14], 8);
          const bool padding = lay.cold[16 * slot + 15] < 0.0;
          const int bit = (m >> (7 - j)) & 1;
          if (!padding) kr[orig] = (int8_t)bit;
        }
      }
    }
  }
  return TOR_OK;
}

// The strict brute-force layout's SCREENED segments on the HOST over a whole scene: builds the layout tor_scene_upload builds
// (no float32 segments) and walks its xkinds 10-14 for each ray as integrate_kernel's ARITH 2 loop does when stage one runs --
// the plane screen first (the segment's plane table, its largest radius^2), then the segment's own test on the per-lane record:
// the second form for 10 / 11 / 12 (32 bytes), the first form for 13 / 14 (64 bytes).  keep[ray * world.len + object] = 0 dropped
// by the plane screen, 1 dropped by stage two, 2 a candidate of the exact test, 3 the object is on a segment without a table (a
// time group whose fraction is never finite).  kind_out[object] (nullable) = its segment's xkind (0, 10-14);
// pays_out[ray * n_segs_out + segment] (nullable, with n_segs_out) = plane_pays' vote of that ray.  No device needed.
int tor_debug_layout_segments(TorHittableList world, int32_t* out, int64_t max_segs, int64_t* n_segs_out) {
  if (world.len < 0 || (world.len > 0 && !world.objects) || max_segs < 0 || (max_segs > 0 && !out) || !n_segs_out)
    return fail(TOR_ERR_INVALID_ARGUMENT, "tor_debug_layout_segments: bad argument");
  std::vector<int64_t> ids((size_t)world.len);
  for (int64_t i = 0; i < world.len; ++i) ids[(size_t)i] = i;
  tor::HostLayout lay;
  std::string err;
  if (!tor::build_layout(world.objects, ids, lay, err, nullptr)) return fail(TOR_ERR_INVALID_ARGUMENT, err);
  *n_segs_out = lay.n_segs;
  for (int s = 0; s < lay.n_segs && s < max_segs; ++s) {
    const double* sg = &lay.segs[8 * (size_t)s];
    const int count = (int)sg[2] & 0xffffff, real = count - ((int)sg[2] >> 24);
    out[4 * s + 0] = (int32_t)lay.xsegs[8 * (size_t)s];
    out[4 * s + 1] = real;
    out[4 * s + 2] = count;
    out[4 * s + 3] = (int32_t)sg[3] * (int32_t)tor::kPad;
  }
  return TOR_OK;
}

int tor_debug_screen2_scene(TorHittableList world, int64_t n_rays, const double* o, const double* d, const double* time, int8_t* keep,
                            int32_t* kind_out, int8_t* pays_out, int64_t n_segs_out) {
  if (world.len < 0 || (world.len > 0 && !world.objects) || n_rays < 0 || !o || !d || !time || !keep)
    return fail(TOR_ERR_INVALID_ARGUMENT, "tor_debug_screen2_scene: bad argument");
  std::vector<int64_t> ids((size_t)world.len);
  for (int64_t i = 0; i < world.len; ++i) ids[(size_t)i] = i;
  tor::HostLayout lay;
  std::string err;
  if (!tor::build_layout(world.objects, ids, lay, err, nullptr)) return fail(TOR_ERR_INVALID_ARGUMENT, err);
  auto orig_of = [&](size_t slot) {
    int64_t v;
    std::memcpy(&v, &lay.cold[16 * slot + 14], 8);
    return v;
  };
  for (int64_t r = 0; r < n_rays; ++r) {
    int8_t* kr = keep + r * world.len;
    for (int64_t i = 0; i < world.len; ++i) kr[i] = 3;
    const double* oo = o + 3 * r; const double* dd = d + 3 * r;
    const double a = dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2];  // spheres.nim:30
    const tor::ScreenRay ray = tor::screen2_ray(oo[0], oo[1], oo[2], dd[0], dd[1], dd[2], a);
    const tor::PlaneRay pray = tor::plane_ray(ray);
    for (int s = 0; s < lay.n_segs; ++s) {
      const double* sg = &lay.segs[8 * (size_t)s];
      const double* xs = &lay.xsegs[8 * (size_t)s];
      const int xkind = (int)xs[0];
      const int count = (int)sg[2] & 0xffffff, real = count - ((int)sg[2] >> 24), block0 = (int)sg[3];
      if (r == 0 && kind_out)
        for (int i = 0; i < real; ++i) kind_out[orig_of((size_t)block0 * tor::kPad + (size_t)i)] = xkind;
      if (xkind < 10) continue;
      const double f = xkind >= 12 ? (time[r] - sg[4]) / sg[5] : 0.0;  // moving_spheres.nim:42
      const tor::ScreenSeg ss = tor::screen2_seg(ray, sg[6], sg[7], xs[2], f);
      const tor::PlaneSeg ps = tor::plane_seg(ray, pray, sg[6], sg[7], f, xs[4]);
      double negmu = 0.0, am = 0.0;  // first form (13 / 14): integrate_loop_plane.inc / integrate_loop_f64_movers.inc
      tor::screen_margins(ray.s1 + sg[6] + sg[7] * std::fabs(f), std::fabs(dd[0]) + std::fabs(dd[1]) + std::fabs(dd[2]), a, negmu, am);
      const size_t xsz = xkind >= 13 ? 8 : 4, pw = xkind == 13 ? 4 : 2;
      for (int i = 0; i < real; ++i) {
        const double* x = &lay.xrec[(size_t)xs[1] + xsz * (size_t)i];
        const double* pl = &lay.xpl[(size_t)xs[3] + pw * (size_t)i];
        const bool table_ok = xkind == 10   ? (pl[0] == x[0] && pl[1] == x[2])
                              : xkind <= 12 ? (pl[0] == x[0] && pl[1] == x[1])
                              : xkind == 13 ? (pl[0] == x[0] && pl[1] == x[2] && pl[2] == x[4] && pl[3] == x[6])
                                            : (pl[0] == x[0] && pl[1] == x[2] && x[4] == 0.0 && x[6] == 0.0);
        if (!table_ok) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_debug_screen2_scene: the plane table and the records disagree (layout bug)");
        int8_t verdict;
        const int w1 = xkind == 13 ? tor::plane_word_mov(ps, pl[0], pl[1], pl[2], pl[3]) : tor::plane_word(ps, pl[0], pl[1]);
        if (w1 >= 0) {
          verdict = 0;
        } else {
          int w2;
          if (xkind == 10) w2 = tor::screen2_static(ss, x[0], x[1], x[2], x[3]);
          else if (xkind == 11) w2 = tor::screen2_static_y(ss, x[0], x[1], x[2]);
          else if (xkind == 12) w2 = tor::screen2_movy_y(ss, x[0], x[1], x[2], x[3]);
          else w2 = tor::screen_filter(tor::fma_(-f, x[4], oo[0] - x[0]), tor::fma_(-f, x[5], oo[1] - x[1]), tor::fma_(-f, x[6], oo[2] - x[2]), dd[0], dd[1], dd[2], a, negmu, am, x[3]);
          verdict = w2 < 0 ? 2 : 1;
        }
        kr[orig_of((size_t)block0 * tor::kPad + (size_t)i)] = verdict;
      }
      if (pays_out && s < n_segs_out) pays_out[r * n_segs_out + s] = tor::plane_pays(pray, 4.0 * tor::kPlaneGate * tor::kPlaneGate, xs[4], xs[5], xs[6]) ? 1 : 0;
    }
  }
  return TOR_OK;
}

// Float32 slab test of the culling boxes on the HOST (same source as the kernel): ray i against box i.
// keep[i] = slab_bit32 on the float32 record block_bounds_f32 makes of the box; need[i] = the float64 slab test
// of the kernel's float64 path on the same box.  Correct iff need implies keep.
int tor_selftest_slab32_host(int64_t n, const double* o, const double* d, const double* lo, const double* hi,
                             const double* origin, int32_t* keep, int32_t* need) {
  if (n < 0 || !o || !d || !lo || !hi || !origin || !keep || !need)
    return fail(TOR_ERR_INVALID_ARGUMENT, "tor_selftest_slab32_host: bad argument");
  std::vector<double> bnd((size_t)n * 8, 0.0);
  for (int64_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) { bnd[8 * i + k] = lo[3 * i + k]; bnd[8 * i + 3 + k] = hi[3 * i + k]; }
  std::vector<float> bnd32;
  float hmin = 0.0f;
  const float bmax = tor::block_bounds_f32(bnd, origin, bnd32, &hmin);
  using tor::f2v;
  for (int64_t i = 0; i < n; ++i) {
    const double* oo = o + 3 * i; const double* dd = d + 3 * i;
    const double a = dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2];
    const tor::RayF32 r = tor::make_ray_f32(oo[0], oo[1], oo[2], dd[0], dd[1], dd[2], a, origin[0], origin[1], origin[2]);
    const tor::BoxRay32 b = tor::make_box_ray32(r, bmax, hmin);
    const float* rec = &bnd32[8 * (size_t)i];
    keep[i] = (int32_t)(tor::slab_bit32(b, (f2v){rec[0], rec[1]}, (f2v){rec[2], rec[3]}, (f2v){rec[4], rec[5]}) | (r.wild & 1u));
    const double ix = 1.0 / dd[0], iy = 1.0 / dd[1], iz = 1.0 / dd[2];
    const double* c = &bnd[8 * (size_t)i];
    const double tx0 = (c[0] - oo[0]) * ix, tx1 = (c[3] - oo[0]) * ix;
    const double ty0 = (c[1] - oo[1]) * iy, ty1 = (c[4] - oo[1]) * iy;
    const double tz0 = (c[2] - oo[2]) * iz, tz1 = (c[5] - oo[2]) * iz;
    const double t_in = std::fmax(std::fmax(std::fmin(tx0, tx1), std::fmin(ty0, ty1)), std::fmax(std::fmin(tz0, tz1), 0.0));
    const double t_out = std::fmin(std::fmin(std::fmax(tx0, tx1), std::fmax(ty0, ty1)), std::fmax(tz0, tz1));
    need[i] = (t_in <= t_out) ? 1 : 0;
  }
  return TOR_OK;
}

// ---- self tests -----------------------------------------------------------------------------

int tor_selftest_math_host(int32_t op, const double* x, const double* y, double* out0, double* out1,
                           int64_t n) {
  if (!x || !out0 || n < 0) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_selftest_math_host: bad argument");
  for (int64_t i = 0; i < n; ++i) {
    double a = 0, b = 0;
    tor::selftest_math_one(op, x[i], y ? y[i] : 0.0, a, b);
    out0[i] = a;
    if (out1) out1[i] = b;
  }
  return TOR_OK;
}

int tor_selftest_math_device(int32_t op, const double* x, const double* y, double* out0, double* out1,
                             int64_t n, int32_t device) {
  if (!x || !out0 || n < 0) return fail(TOR_ERR_INVALID_ARGUMENT, "tor_selftest_math_device: bad argument");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) return fail(TOR_ERR_NO_DEVICE, "no HIP device available");
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  const size_t bytes = (size_t)n * 8;
  DeviceBuffer bx, by, b0, b1;  // released on every path
  struct Release {
    DeviceBuffer *a, *b, *c, *d;
    ~Release() { a->release(); b->release(); c->release(); d->release(); }
  } release{&bx, &by, &b0, &b1};
  HIP_TRY(bx.ensure(bytes + 8));
  HIP_TRY(by.ensure(bytes + 8));
  HIP_TRY(b0.ensure(bytes + 8));
  HIP_TRY(b1.ensure(bytes + 8));
  HIP_TRY(hipMemcpy(bx.ptr, x, bytes, hipMemcpyHostToDevice));
  if (y) HIP_TRY(hipMemcpy(by.ptr, y, bytes, hipMemcpyHostToDevice));
  else HIP_TRY(hipMemset(by.ptr, 0, bytes));
  HIP_TRY(hipMemset(b0.ptr, 0, bytes));
  HIP_TRY(hipMemset(b1.ptr, 0, bytes));
  HIP_TRY(tor::launch_selftest(op, (const double*)bx.ptr, (const double*)by.ptr, (double*)b0.ptr, (double*)b1.ptr, n, nullptr));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out0, b0.ptr, bytes, hipMemcpyDeviceToHost));
  if (out1) HIP_TRY(hipMemcpy(out1, b1.ptr, bytes, hipMemcpyDeviceToHost));
  return TOR_OK;
}

int tor_selftest_rng_host(int32_t mode, uint64_t a, uint64_t b, uint64_t c, uint64_t state_out[4],
                          uint64_t* draws_out, int64_t n) {
  tor::Rng g{0, 0, 0, 0};
  if (mode == 1) tor::seed1(g, a);
  else if (mode == 2) tor::seed2(g, a, b);
  else if (mode == 3) tor::seed3(g, a, b, c);
  else return fail(TOR_ERR_INVALID_ARGUMENT, "tor_selftest_rng_host: mode must be 1, 2 or 3");
  if (state_out) { state_out[0] = g.s0; state_out[1] = g.s1; state_out[2] = g.s2; state_out[3] = g.s3; }
  for (int64_t i = 0; i < n && draws_out; ++i) draws_out[i] = tor::next(g);
  return TOR_OK;
}

}  // extern "C"
