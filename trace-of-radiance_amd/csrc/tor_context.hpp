// tor_context.hpp -- private state behind the opaque TorContext of include/tor_render.h, shared by the
// translation units of the C-ABI layer (tor_api.cpp: single-device entry points; tor_multi.cpp: canvas
// download, multi-device tor_render_opt, RCCL gather).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/tor_render.h"
#include "tor_filter32.hpp"
#include "tor_kernels.hpp"
#include "tor_scene.hpp"

namespace tor {

int fail(int code, const std::string& msg);           // sets tor_last_error(), returns code
int fail_hip(hipError_t e, const char* what);         // maps a HIP error to a TOR_ERR_* status
void set_last_note(const std::string& s);             // tor_last_note(): what the last call chose / fell back to (not an error)
const std::string& last_note();

#define HIP_TRY(expr)                                        \
  do {                                                       \
    hipError_t e__ = (expr);                                 \
    if (e__ != hipSuccess) return ::tor::fail_hip(e__, #expr); \
  } while (0)

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  hipError_t ensure(size_t n) {
    if (n <= bytes) return hipSuccess;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&ptr, n);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
};

// page-locked host staging (D2H lands here at PCIe speed; worker threads move it into the caller's pageable canvas)
struct PinnedBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  hipError_t ensure(size_t n) {
    if (n <= bytes) return hipSuccess;
    if (ptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    bytes = 0;
    hipError_t e = hipHostMalloc(&ptr, n, hipHostMallocDefault);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  void release() {
    if (ptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
};

// device copy of a tor::HostLayout: ONE allocation, one H2D copy; the arrays sit at 256-byte aligned offsets
struct DeviceLayout {
  DeviceBuffer blob;
  const double *stat = nullptr, *mov = nullptr, *movy = nullptr, *segs = nullptr, *cold = nullptr;
  const float* hot32 = nullptr;
  const double* coop_trips = nullptr;  // coop_pixel_kernel: 4 float64 per trip of 64 cold slots
  const double *xhdr = nullptr, *xrec = nullptr, *xpl = nullptr;  // second form of the FMA screen and its stage one (tor_scene.hpp)
  int n_segs = 0;
  int n_sorted = 0;  // cold slots (padded)
  int n_xrec = 0;    // float64 in xrec
  bool has_f32 = false;
  hipError_t put(const HostLayout& lay, const std::vector<double>* cold_override = nullptr);
  void release() { blob.release(); }
};

// device copy of a tor::HostAccel (TOR_ACCEL_BLOCKS): always-list + spatial blocks
struct DeviceAccel {
  DeviceLayout always;  // .cold holds always.cold followed by the spatial objects' cold records
  DeviceBuffer hot, grp, hot32;
  void release() { always.release(); hot.release(); grp.release(); hot32.release(); }
};

struct RcclComm;  // tor_multi.cpp

}  // namespace tor

struct TorContext {
  static constexpr int kRing = 64;
  // 64-bit words per ring slot: [0] work counter, [1..4] statistics, [5] probe counter, [6] split, [7] wave-kernel counter,
  // [8..10] tile schedule, then from kMigWord0 on the chain hand-off's control lines (tor_kernels.hpp kMig*)
  static constexpr int kMigWord0 = 16;
  static constexpr int kSlotWords = kMigWord0 + tor::kMigWords;
  int device = 0;
  int num_cus = 0;
  // ---- scene: a byte copy of the caller's list (cache key + source of the lazily built layouts) ----
  std::vector<unsigned char> scene_bytes;
  int64_t n_objects = 0;
  bool scene_ready = false;
  int64_t n_uploads = 0, n_cache_hits = 0, n_layouts_built = 0;
  // [0] float64 loops only, [1] with the TOR_ACCEL_F32 segments; built on first use
  tor::DeviceLayout flat[2];
  bool flat_built[2] = {false, false};
  tor::F32Options f32;
  bool f32_built = false;
  // TOR_ACCEL_BLOCKS layouts (same two variants); bounds travel per launch (ring)
  tor::HostAccel accel[2];
  tor::DeviceAccel d_accel[2];
  bool accel_built[2] = {false, false};
  tor::DeviceBuffer bnd_ring;
  size_t bnd_slot_bytes = 0;
  // ---- per-launch state ----
  tor::DeviceBuffer counters;                      // kRing x 8 u64: [0] work counter, [1..4] stats, [5] probe counter
  tor::DeviceBuffer tile_order[kRing];  // SEED_PIXEL schedule: the tile order a launch reads
  tor::DeviceBuffer probe_buf;          // ... and what the sort is made from: per-pixel probe counts, per-tile key and work
  float hot_frac = 0.4f;   // a pixel chain is hot (arbiter priority 3) from this share of an average wave's iterations on (TOR_HOT_FRAC; 0 = off)
  int probe_spp = 2;            // samples per pixel of the cost probe
  int64_t last_probe_pixels = 0;
  int prio_shift = 16;  // SEED_PIXEL: arbiter-priority rotation period, log2 shader-clock ticks (tor_kernels.hip; 0 = off; TOR_PRIO_SHIFT)
  tor::DeviceBuffer wave_log;                      // debug: 8 x u64 per wave (only with stats enabled)
  tor::DeviceBuffer cam_ring;                      // kRing x TorCamera
  TorCamera cam_host[kRing];                       // host staging must outlive the asynchronous copies
  std::vector<double> bnd_host[kRing];
  std::vector<float> bnd32_host[kRing];
  std::vector<double> probe_bnd_host[kRing];  // block bounds of the cost probe when it runs with other accel bits than the frame (host staging of an
  std::vector<float> probe_bnd32_host[kRing];  // asynchronous copy, per ring slot like bnd_host)
  bool probe_accel = true;  // the cost probe always walks the culling layout when the scene has one (it only counts queries); (round 5: the TOR_PROBE_ACCEL switch is gone)
  int plane_screen = 1;  // TOR_PLANE: 0 = the wave-uniform test for every object (no plane screen in front), 1 = stage one where it pays (tor_screen.hpp plane_pays), 2 = on every segment with a table
  bool screen = true;       // TOR_SCREEN=0: strict brute-force launches evaluate the reference's unfused discriminant for every object (no FMA screen)
  hipEvent_t ev_start[kRing] = {}, ev_stop[kRing] = {};
  int64_t launches = 0;  // timed integrator launches so far
  bool timing_valid = false;
  int64_t last_samples = 0;
  int64_t last_n_waves = 0;
  int last_slot = 0;
  bool collect_stats = false;
  // ---- host-canvas entry points ----
  tor::DeviceBuffer scratch;   // tor_render_opt's device framebuffer (this device's shard)
  tor::DeviceBuffer slice;     // tor_render_frame_h264's device slice buffer
  tor::DeviceBuffer gather;    // root of a multi-device render / RCCL gather: the ranks' shards, rank-major
  tor::DeviceBuffer frame;     // ... de-interleaved frame (multi-device tor_render_opt)
  tor::PinnedBuffer staging;   // D2H target
  std::vector<hipEvent_t> chunk_events;
  hipEvent_t ev_call[2] = {};    // tor_render_opt: around the launches of one call (tor_last_render_timing out[1])
  void* last_stream = nullptr;   // stream of the last tor_render_device launch (one stream per context while launches overlap)
  bool last_stream_valid = false;
  hipStream_t stream = nullptr;  // own stream of the host-canvas entry points (created on first use)
  tor::RcclComm* comm = nullptr; // tor_comm_init_rank
  // Launch shape, measured on MI355X (profiles/r1_wave_service.txt): waves that share a SIMD get
  // very unequal service (hardware slot 0 ~38 us per bounce iteration, slot 4 0.6-2 ms), so
  // extra waves add little throughput and park work in slow waves.  Per seeding mode:
  //   SAMPLE: 3 workgroups/CU (kernel compiled for <= 168 VGPRs)  -> best throughput
  //   PIXEL : 2 workgroups/CU (<= 256 VGPRs): a pixel is a sequential chain of spp samples, every
  //           wave that holds one must get good service; with an exact acceleration the iteration is short
  //           enough that 3 workgroups/CU win (C2: f32 1768 -> 1900, blocks 1625 -> 1790, both 2133 -> 2350)
  // Overridable for experiments: TOR_WAVES_PER_SIMD (2|3), TOR_BLOCKS_PER_CU.
  int max_blocks_per_cu[2][2] = {{3, 3}, {3, 3}};  // [seeding][any TOR_ACCEL_* bit set]
  int lpt_min_spp = 32;    // SEED_PIXEL: probe + cost-ordered tiles from this many spp on (0 = never)
  int waves_override = 0;  // TOR_WAVES_PER_SIMD (2|3): force a register-budget variant of the kernel
  // SEED_PIXEL frames of at most this many (local) pixels run coop_pixel_kernel (one wave per pixel: the frame is
  // too small to fill the machine with one lane per pixel chain).  TOR_COOP_MAX_PIXELS overrides; 0 = never.
  long long coop_max_pixels = 114688;
  // Split mode for SEED_PIXEL frames between split_min_pixels and split_max_pixels (when the cost probe runs): the
  // tiles that carry split_frac of the probed cost go to the wave-per-pixel kernel on stream2, the lane kernel takes
  // the rest concurrently.  TOR_SPLIT_FRAC (0 = off), TOR_SPLIT_MIN_PIXELS, TOR_SPLIT_MAX_PIXELS.
  float split_frac = -1.0f;  // < 0: automatic (tor_render_device), 0: off
  long long split_min_pixels = 16384, split_max_pixels = 100000000;
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork[kRing] = {}, ev_join[kRing] = {};
  // Chain hand-off (DESIGN 4.7 (HISTORY 4.10); SEED_PIXEL with both exact accelerations on a single-level layout, from lpt_min_spp on):
  // lanes push long pixel chains to server waves inside the same launch.  TOR_MIGRATE=0 restores split mode / the
  // wave-per-pixel kernel.  Knobs: TOR_SRV_FRAC (share of the workgroups that start as servers when the frame can hold a
  // chain above the threshold's floor), TOR_SRV_MIN_FRAC (otherwise), TOR_SRV_PATIENCE_US (a dedicated server without work
  // for that long becomes a lane wave), TOR_PUSH_THETA / TOR_FLOOR_THETA (first value / floor of the adaptive push
  // threshold, x l_avg), TOR_CHAIN_THETA (floor x the frame's mean chain), TOR_TAIL_LANES, TOR_TAIL_REST, TOR_MIG_FLAGS.
  int mig_mode = 1;
  float srv_frac = 0.07f, srv_min_frac = 0.005f, push_theta = 3.0f, chain_theta = 3.5f, floor_theta = 1.33f;
  int srv_patience_us = 8000;
  int key_mode = 1;  // tile sort key of the SEED_PIXEL schedule: certain long chains first, then by the tile's sum (tile_key_kernel; the round-2 key, 0, is kept for the record only)
  int mig_tail_lanes = 8;
  int mig_tail_rest = 256;   // TOR_TAIL_REST
  unsigned mig_flags = (8u << 8) | 2u | 4u;  // TOR_MIG_FLAGS (tor_kernels.hpp KParams::mig_flags)
  tor::DeviceBuffer mig_rec, mig_flag;
  // A waiting server that sees no progress of the frame for this long flags the frame as incomplete and leaves
  // (KParams::mig_stall_ticks; TOR_SRV_STALL_S, 0 = never): the launch assumes all of its workgroups resident, and this is
  // the way out when they are not.  The host-canvas entry points then render the frame again without the hand-off.
  double mig_stall_s = 60.0;
  int64_t n_stalled_frames = 0;  // frames the host-canvas entry points rendered again because the hand-off stalled
  bool last_migrate = false;  // the last launch carried the hand-off (its kMigStalled word means something)
};

namespace tor {

// Options with defaults applied; false when malformed.  `for_drop_in`: NULL options take tor_render()'s
// environment defaults (TOR_DEFAULT_ACCEL, TOR_DEVICES, TOR_GATHER).
bool valid_options(const TorOptions* opt, TorOptions& o, bool for_drop_in);
const std::string& options_why();  // why the last valid_options() call on this thread returned false

// Makes sure the layouts a launch with (accel bits) needs exist on the device (built from scene_bytes).
int ensure_layouts(TorContext* ctx, int accel);

// Copies n_rows compact rows of row_bytes from device memory (on ctx's device, ready on `stream`) into host
// memory: row k lands at dst + rows[k] * row_bytes (rows == nullptr: k * row_bytes).  D2H in chunks into the
// context's pinned staging, worker threads move the chunks on while later chunks are still in flight.  Blocking.
int download_rows(TorContext* ctx, const void* d_src, int64_t n_rows, size_t row_bytes, const int32_t* rows,
                  char* dst, hipStream_t stream);

// Hand-off launches (ctx->last_migrate): handoff_stalled waits for `stream` and reads the launch's kMigStalled word;
// rerender_if_stalled renders the frame again with the hand-off off when it is set (same arguments as tor_render_device).
int handoff_stalled(TorContext* ctx, hipStream_t stream, bool* stalled);
int rerender_if_stalled(TorContext* ctx, const TorCamera* cam, int32_t nrows, int32_t ncols, int32_t spp, float gamma_correction,
                        int64_t max_depth, const TorOptions* o, double* d_pixels, hipStream_t stream);

// default (cached) context of the host-canvas entry points: one per (device, replica)
int default_context(int device, int replica, TorContext** out);

int render_multi_device(TorCanvas* canvas, const TorCamera* cam, TorHittableList world, int64_t max_depth,
                        const TorOptions& o, double timing_ms[5]);

}  // namespace tor
