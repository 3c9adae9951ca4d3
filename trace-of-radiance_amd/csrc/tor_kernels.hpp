// tor_kernels.hpp -- launch interface between the C-ABI layer (tor_api.cpp) and the gfx950
// kernels (tor_kernels.hip).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "tor_device.hpp"
#include "tor_filter32.hpp"

namespace tor {

constexpr int kThreads = 256;  // 4 waves per workgroup
constexpr int kPad = 8;        // hot-record arrays are padded to a multiple of this (= kBlock, the candidate-mask width)
// TOR_ACCEL_BLOCKS: blocks of 8 objects behind ONE culling box (box b = blocks [b F, (b + 1) F)).  A compile-time constant: the
// round-4 experiment ran it as a launch parameter (commit c39354b, profiles/r4_fanout_sweep.txt: 56.4 / 59.0 / 73.0 / 115.3 ms for
// F = 1 / 2 / 4 / 8 on configs[1] with both accelerations) and the run-time loops alone cost the accelerated kernels 2-3 %.
#ifdef TOR_BOX_FANOUT
constexpr int kBoxFanout = TOR_BOX_FANOUT;  // (A/B builds only)
#else
constexpr int kBoxFanout = 1;
#endif

// Device scene (built by tor_scene_upload from the AoS HittableVariant list):
//   stat : static spheres, 4 float64 each   {cx, cy, cz, radius^2}
//   mov  : moving spheres, 8 float64 each   {c0x, c0y, c0z, radius^2, dcx, dcy, dcz, 0}
//          with dc = center1 - center0, grouped by (time0, time1)
//   movy : moving spheres of groups that move along y only, 6 float64 each
//          {c0x, c0y, c0z, radius^2, dcy, 0}
//   segs : 8 float64 per segment {kind (0 static, 1 moving along y only, 2 moving, 3 block bounds), first
//          record, padded count, (first sorted index)/kPad, time0, time1 - time0, 0, 0}
//   hot32: TOR_ACCEL_F32 segments (kinds 5 static, 6 moving along y only, 7 moving; segs[1] = float offset,
//          segs[6] = max |c0 - origin|, segs[7] = max |dc| over the segment): float32 records per PAIR of
//          objects, interleaved for the packed-float32 filter (tor_filter32.hpp), coordinates relative
//          to KParams.org:  {cx cx', cy cy', cz cz', r2 r2', k k'} (+ {dcy dcy'} | + {dcx dcx', dcy dcy', dcz dcz'})
//   cold : 16 float64 per sorted slot {c0 xyz, dc xyz, 1/radius, time0, time1-time0,
//          albedo xyz, fuzz|refraction_index, flags(bit0 moving, bits 8..15 material kind),
//          original index, radius^2}; flags/original index are int64 bit patterns.
struct KParams {
  const double* stat;
  const double* mov;
  const double* movy;
  const double* segs;
  const double* cold;
  const double* xrec;
  int xrec_lds_doubles; // > 0: stage two reads the second-form records from LDS (that many float64 of xrec copied per workgroup); 0: vector loads
  const double* xhdr;   // ARITH 2: one 128-byte header per segment (16 float64 slots; tor_scene.hpp HostLayout::xhdr), everything the object loop reads per segment
  const double* xpl;    // stage one of every xkind >= 10 (the plane screen): {cx, cz} per slot ({c0x, c0z, dcx, dcz} for xkind 13); null = the wave-uniform test for every object (TOR_PLANE=0)
  double plane_gate2;   // stage one runs on a segment when most of the wave's rays expect it to keep less than 1 / gate of the segment (tor_screen.hpp plane_pays; 4 gate^2); 0 = always (TOR_PLANE=2)
  const float* hot32;  // TOR_ACCEL_F32 pair records, or null
  double org[3];       // origin of the float32 coordinates
  const double* bnd;   // TOR_ACCEL_BLOCKS: 8 float64 per block {lo xyz, hi xyz, 0, 0} (segment kind 3), else null
  const double* shot;  // TOR_ACCEL_BLOCKS: 8 float64 per spatial slot {c0 xyz, r^2, dc xyz, time-group id | -1}
  int shot_stride;       // float64 per compact record: 8, or 4 when no spatial object moves
  int shot_lds_doubles;  // > 0: copy that many float64 of shot into LDS per workgroup (fits next to the queues)
  const double* sgrp;  // TOR_ACCEL_BLOCKS: {time0, time1 - time0} per time group
  // TOR_ACCEL_BLOCKS | TOR_ACCEL_F32: float32 pair records of the spatial slots (4 pairs of shot32_stride floats
  // per block), replacing shot/sgrp in the block expansion; null -> float64 expansion
  const float* shot32;
  int shot32_stride;      // 10 | 12 | 16 floats per pair
  int shot32_block_stride;  // floats per block of 4 pairs (4 * shot32_stride + padding)
  int shot32_lds_floats;  // > 0: copy that many floats of shot32 into LDS per workgroup
  float sp_mc0max, sp_dcmax;
  int two_level;          // the culling layout has super boxes (selects the BLOCKS = 2 kernel variants)
  const float* bnd32;     // float32 boxes (8 floats per record, same indices as bnd): {c.x h.x c.y h.y c.z h.z 0 0}, centre - org and half-extent
  float sp_bmax;          // max |box coordinate - org|
  float sp_hmin;          // smallest half-extent of a culling box (tor_filter32.hpp make_box_ray32)
  int bnd32_lds_floats;   // > 0: the block boxes (two-level scenes: the per-lane descent reads them) are staged in LDS too
  double sp_t0, sp_dt;    // the spatial movers' time group
  int spatial_base;    // first cold slot of the spatial blocks: block b owns cold[spatial_base + 8b .. +8)
  int n_super;         // two-level scenes: number of real super boxes (the kind-4 segment is padded to 8; the padding
                       // entries have no block boxes / records behind them and must never be entered)
  int n_segs;
  int nrows, ncols, spp, max_depth;
  double inv_spp, inv_ncols, inv_row_tile;  // 1.0 / spp, 1.0 / ncols, 1.0 / row_tile for udiv_by (tor_device.hpp)
  int shard_index, shard_count, row_tile;
  unsigned chunk;   // SEED_SAMPLE: largest chunk of the guided schedule
  unsigned n_waves; // waves launched
  unsigned n_pixels;  // local pixels (rows of this shard x ncols)
  const unsigned* order;  // SEED_PIXEL: nullable tile order (cost-descending), n_tiles entries
  unsigned* pixel_cost;   // probe launch: per-pixel closest-hit query count (2 samples)
  unsigned long long* sched;  // SEED_PIXEL + order: [0] hot chain length (bounce iterations from which a pixel chain gets arbiter priority 3); else null
  int prio_shift;         // SEED_PIXEL: rotate the waves' arbiter priority every 2^prio_shift shader-clock ticks (0 = off)
  unsigned long long total_work;
  unsigned long long* work_counter;
  double* out;
  unsigned long long* stats;  // nullable: {queries, candidates, wave iterations, samples}
  unsigned long long* wave_log;  // nullable (with stats): per wave {start, end (100 MHz clock), iterations, queries}
  const double* cam_dev;  // 24 float64, TorCamera layout (cameras.nim:15-22)
  // coop_pixel_kernel (one wave per pixel): number of cold slots of the flat layout and that number padded to 64
  int n_cold_slots, coop_slots;
  const unsigned long long* split;  // split mode: number of cost-ordered tiles the wave kernel takes (device word), else null
  const double* coop_trips;  // 4 float64 per trip of 64 slots {kind, time0, time1 - time0, 0} (tor_scene.hpp)
  // SEED_PIXEL chain hand-off (tor_kernels.hip "chain servers", DESIGN 4.7 (HISTORY 4.10)); null = off.  Lanes push the state of a pixel
  // chain at a sample boundary -- {pixel, samples done, RNG state, running sum} -- and whole waves ("servers") continue it,
  // the 64 lanes sharing each closest-hit query: same arithmetic, same pixel, a tenth of the latency per bounce.
  unsigned long long* mig;      // control words, kMig* below (zero before the launch; tile_order_kernel fills the schedule part)
  unsigned long long* mig_rec;  // mig_cap records of 8 x u64 {pixel | samples done << 32, rng s0..s3, sum x y z}
  unsigned* mig_flag;           // mig_cap ready flags (zero before the launch)
  unsigned mig_cap;
  int mig_tail_lanes;           // a wave that has run out of fresh pixels hands its chains over from this many live lanes down
  int mig_tail_rest;            // ... with more live lanes than that: only chains with at least this many bounce iterations to go, and only to idle servers
  unsigned mig_patience;        // 100 MHz ticks a dedicated server waits without a chain before it turns into a lane wave (0: never)
  unsigned long long mig_stall_ticks;  // 100 MHz ticks without any progress of the frame after which a WAITING server gives up (sets mig[kMigStalled], leaves): the escape from a launch whose workgroups are not all resident; 0 = wait for ever
  unsigned mig_flags;           // bit 0: acquire (not relaxed) polling; bit 1: adaptive push threshold; bits 8-15: longest back-off of a waiting server in naps of ~3.4 us; bits 16-31: at most this many waiting servers (0 = no limit)
  int screen;                   // strict launches of brute-force layouts: 1 = conservative FMA screen in the object loop (kernel variant ARITH 2; same canvas), 0 = the reference's unfused discriminant for every object
  int n_boxes;                  // block boxes of a single-level culling layout (padded to kPad), the servers' first trip
};

// Control words of the hand-off, one 128-byte line per access pattern (thousands of waiting servers poll their flags and,
// rarely, kMigLaneWaves; the lanes must never queue behind that traffic):
enum : int {
  kMigHead = 0,        // line 0: tickets taken by servers (ticket i waits for record i)
  kMigTail = 16,       // line 1: records pushed by lanes
  kMigLaneWaves = 32,  // line 2: waves still inside the lane loop (set by tile_order_kernel; a server leaves when this is 0 and its ticket >= tail)
  kMigSrvWgs = 48,     // line 3, constant during the launch (read through the scalar cache): workgroups (blockIdx < this) that are servers from the start
  kMigPush = 49,       //         initial push threshold: a chain is handed over once its projected length exceeds this many bounce iterations
  kMigPushFloor = 50,  //         ... and the value the adaptive threshold never goes below
  kMigServed = 64,     // line 4, diagnostics: chains served, hot pushes, tail pushes
  kMigHotPushes = 65,
  kMigTailPushes = 66,
  kMigT0 = 67,         // 100 MHz wall clock: first wave's start, last wave leaving the lane loop, last hot / tail chain finished
  kMigTLaneEnd = 68,
  kMigTHotDone = 69,
  kMigTTailDone = 70,
  kMigItsHot = 71,     // bounce iterations served for hot / tail chains
  kMigItsTail = 72,
  kMigTCounterDry = 73,  // first wave that found the work counter dry
  kMigConverted = 74,  // dedicated server waves that turned into lane waves
  kMigProgress = 75,   // heartbeat of work IN PROGRESS: a server bumps it every 64th sample of the chain it serves, a lane wave every 4096th bounce iteration -- part of the stall detector's signature, so that a frame whose counters stand still while long chains are walked is not taken for a stalled one
  kMigStalled = 95,    // a waiting server saw no progress of the frame for mig_stall_ticks and left holding a ticket: the canvas is INCOMPLETE (host: tor_api.cpp handoff_stalled)
  kMigPushNow = 80,    // line 5: the ADAPTIVE push threshold (lanes read it every bounce; idle servers lower it, pushers that meet a backlog raise it)
  kMigWords = 96
};

hipError_t launch_integrate(const KParams& p, int seeding, int arith, int waves_per_simd, int blocks,
                            hipStream_t stream);
int integrate_blocks_per_cu(const KParams& p, int seeding, int arith, int waves_per_simd);
size_t coop_lds_bytes(int coop_slots);
int coop_blocks_per_cu(const KParams& p, int arith);  // 0: the objects do not fit LDS
hipError_t launch_coop(const KParams& p, int arith, int blocks, hipStream_t stream);
int integrate_fixed_lds_bytes(int blocks, int f32, int seeding = 0);  // per workgroup: queues, accumulator cache, debug counters (+ cooperative-resolve lists | the camera-ray reservoir)
hipError_t launch_probe(const KParams& p, int blocks, hipStream_t stream);
// schedule of the chain hand-off, computed on the device from the probe's total (tile_order_kernel): l_avg = probed queries x
// lavg_scale = bounce iterations an average lane runs in this frame; dedicated server workgroups = srv_frac x blocks when a
// chain of the frame can reach the threshold's floor (else srv_min_frac); first push threshold = push_theta x l_avg
struct MigSchedule {
  unsigned long long* mig = nullptr;
  float lavg_scale = 0.f, srv_frac = 0.f, srv_min_frac = 0.f, push_theta = 0.f;
  float chain_scale = 0.f, chain_theta = 0.f, floor_theta = 0.f;  // mean chain of the frame = probed queries x chain_scale; the threshold's floor = chain_theta x that
  int blocks = 0, spp = 0, max_depth = 0;
  int key_mode = 0, probe_spp = 2;  // tile sort key (tile_key_kernel): 0 = longest probed pixel, 1 = certain long chains first, then by the tile's sum
};
hipError_t launch_tile_order(const unsigned* pixel_cost, unsigned n_pixels, unsigned* key, unsigned* work, unsigned* order, int n_tiles,
                             float split_frac, unsigned long long* split_out, unsigned long long* lane_counter,
                             float hot_chain, unsigned long long* sched, const MigSchedule& mig, hipStream_t stream);
bool integrate_variant_serves_chains(const KParams& p, int seeding);  // the launch's kernel variant carries the server code
constexpr int kTilePixelsHost = 64;  // == kTilePixels in tor_kernels.hip
constexpr size_t kTileSortScratchBytes = 4096 * (8 + 4 + 4) + 16;  // launch_tile_order's histogram / offsets behind the per-tile arrays (kCostBins = 4096)
hipError_t launch_finalize(double* pixels, long long n_values, double scale, double gamma, hipStream_t stream);
hipError_t launch_quantize(const double* pixels, long long n_values, uint8_t* out, hipStream_t stream);
hipError_t launch_encode_ipcm(const double* pixels, int nrows, int ncols, uint8_t* out, uint8_t* plane_y,
                              uint8_t* plane_cb, uint8_t* plane_cr, hipStream_t stream);
// multi-GPU assembly: gathered = shard_count slots of shard_stride float64, slot k = shard k's rows (compact, in
// increasing row order); frame = nrows x ncols x 3 float64 in image order (tor_shard_rows' mapping, inverted)
hipError_t launch_gather_rows(const double* gathered, double* frame, int nrows, int ncols, int row_tile, int shard_count,
                              long long shard_stride, hipStream_t stream);
// RCCL watchdog test (TOR_FAULT_INJECT=rccl_hang): a kernel that keeps `stream` busy until *flag != 0 (device-visible host
// memory) or max_ticks of the 100 MHz clock have passed
hipError_t launch_spin_until(volatile unsigned* flag, unsigned long long max_ticks, hipStream_t stream);
hipError_t launch_selftest(int op, const double* x, const double* y, double* out0, double* out1, long long n,
                           hipStream_t stream);

// One element of the math self-test (same source compiled for host and device).
TOR_HD void selftest_math_one(int op, double x, double y, double& out0, double& out1) {
  switch (op) {
    case 0: sincos_2pi(x, out0, out1); break;
    case 1: out0 = pow5(x); break;
    case 2: out0 = pow_pos(x, y); break;
    case 3: out0 = __builtin_sqrt(x); break;
    case 4: out0 = x / y; break;
    case 5: out0 = quantize36(x); break;
    case 6: {  // unit_vector((x, y, 0.5)) . (1, 2, 3): exercises reciprocal-multiply normalise
      V3 u = unit_vector(v3(x, y, 0.5));
      out0 = dot(u, v3(1.0, 2.0, 3.0));
      out1 = u.x;
      break;
    }
    default: out0 = x; break;
  }
}

}  // namespace tor
