// tor_scene.hpp -- host-side flattening of the reference's AoS object list
// (hittables_lists.nim:15-46, HittableVariant hittables_variants.nim:50-57) into the device layouts
// described in tor_kernels.hpp, and the optional block-bounds acceleration layout.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/tor_render.h"

namespace tor {

// Brute-force layout of a subset of the objects (tor_kernels.hpp: stat / movy / mov / segs / cold).
struct HostLayout {
  std::vector<double> stat, mov, movy, segs, cold;
  // second form of the strict loop's FMA screen (tor_screen.hpp): per segment {xkind (0: none, 10 static, 11 static with a
  // common c0.y, 12 mover along y with a common c0.y), first float64 of its records in xrec, the common c0.y, 0...}; records
  // {cx, cy, cz, K} | {cx, cz, K', 0} | {cx, cz, K', dcy}: 4 float64 each, padded like the first form's
  std::vector<double> xsegs, xrec;
  // what the kernel reads of segs / xsegs, one 128-byte record per segment (16 float64 slots): int32 {xkind, kind, padded count |
  // padding << 24, first block, first hot record, first float64 in xrec, first float64 in xpl, the plane screen's gate (0 never, 1 always, 2 the wave votes)}, then float64 {time0, time1 -
  // time0, reach, travel, common c0.y, largest radius^2, extent x, extent z} and the segment's first plane record (4 slots)
  std::vector<double> xhdr;
  // stage one in front of it (tor_screen.hpp: the plane screen, kinds 11 / 12 only): {cx, cz} per slot -- 16 bytes, the wave-uniform
  // loop reads nothing else -- at xpl[xsegs[3] + 2 * slot of the segment]; xsegs[4] = the largest radius^2 of the segment
  std::vector<double> xpl;
  std::vector<float> hot32;  // TOR_ACCEL_F32 segments (kinds 5/6/7): packed pair records, see tor_kernels.hpp
  int n_segs = 0;
  size_t n_sorted = 0;  // cold slots (padded)
  // coop_pixel_kernel walks the cold slots 64 at a time ("trips"); 4 float64 per trip {kind, time0, time1 - time0, 0}:
  // kind 0 = no slot of the trip moves, 1 = all move along y only and share (time0, time1), 2 = all move and share
  // (time0, time1), 3 = mixed (per-object handling)
  std::vector<double> coop_trips;
};

// TOR_ACCEL_F32: objects that qualify (tor_filter32.hpp: f32_eligible, and not farther than far_limit from
// the origin) are laid out for the float32 pre-filter, relative to `origin`; the others keep the float64 loop.
struct F32Options {
  double origin[3] = {0.0, 0.0, 0.0};
  double far_limit = 0.0;
};
// Origin = per-axis median of the (start) centres, far_limit = 8 x the median distance from it: robust
// against a few huge or far-away objects (the ground sphere of random_scene sits 1000 units below).
F32Options f32_options_for(const TorHittableVariant* objs, int64_t n);

// Builds the layout for objects `ids` (original indices, kept in this order inside each segment).
// Returns false and sets err for unknown kinds.
bool build_layout(const TorHittableVariant* objs, const std::vector<int64_t>& ids, HostLayout& out, std::string& err,
                  const F32Options* f32 = nullptr);

// Acceleration layout (TOR_ACCEL_BLOCKS): large or irregular objects stay in an "always" brute-force
// layout; the rest are ordered by a top-down surface-area build (tor_scene.cpp build_accel) and cut into blocks of 8 consecutive objects.
// Per render call the blocks get conservative bounding spheres (they depend on the ray-time range).
struct HostAccel {
  bool available = false;
  HostLayout always;            // + one segment of kind 3 (the bounds) appended to always.segs
  std::vector<double> cold;     // always.cold followed by the spatial objects' cold records
  std::vector<double> hot;      // hot_stride float64 per spatial slot: {c0 xyz, r^2, dc xyz, time-group id or -1},
                                // or {c0 xyz, r^2} when none of the spatial objects moves
  int hot_stride = 8;
  std::vector<double> groups;   // {time0, time1 - time0} per time group of the spatial objects
  size_t spatial_base = 0;      // first cold slot of the spatial objects (multiple of 8)
  size_t n_blocks = 0;
  // Box fan-out (round 4 experiment, default 1 = off): a culling box of the wave-uniform loop covers `fanout` consecutive blocks
  // (16 / 32 objects for fanout 2 / 4).  Every lane tests every box (14 float32 instructions each), the pooled resolve filters only
  // the boxes a ray enters (8.5 instructions per object), so fewer, larger boxes looked cheaper on paper; measured, they are not
  // (tor_scene.cpp build_accel).  Blocks stay 8 objects (records, cold slots, masks); box b stands for blocks [b * fanout, (b + 1) * fanout).
  int fanout = 1;
  size_t n_boxes = 0;           // = ceil(n_blocks / fanout)
  bool two_level = false;       // the uniform loop tests boxes around 8 boxes; lanes descend to those
  struct Obj { double c0[3], dc[3], t0, dt, abs_r; bool moving, valid; };
  std::vector<Obj> spatial;     // n_blocks * 8 entries (padding: valid = false)
  // TOR_ACCEL_F32 block expansion: float32 pair records of the spatial slots (tor_filter32.hpp), 4 pairs per
  // block, hot32_stride floats per pair: 10 {cx cx' cy cy' cz cz' r2 r2' k k'} (no mover), 12 (+ dcy dcy':
  // movers along y only), 16 (+ dcx.. dcy.. dcz..).  Built when every spatial object qualifies for the
  // filter and the movers share one (time0, time1); otherwise the lanes expand the float64 records.
  bool sp32 = false;
  std::vector<float> hot32;
  int hot32_stride = 10;
  int hot32_block_stride = 40;  // floats from one block's records to the next (4 pairs + padding against LDS bank conflicts)
  double sp_t0 = 0.0, sp_dt = 1.0;      // the movers' time group
  float sp_mc0max = 0.0f, sp_dcmax = 0.0f;  // max |c0 - origin| and |dc| over the spatial objects (rounded up)
};

void build_accel(const TorHittableVariant* objs, int64_t n, HostAccel& out, const F32Options* f32 = nullptr);

// 8 float64 per block {lo xyz, hi xyz, 0, 0}: a conservative (inflated) axis-aligned box around the
// block's spheres over the ray-time range; padded to a multiple of 8 blocks with never-entered (NaN)
// boxes; then one slack record and the super boxes (one per 8 block boxes, again padded to 8).
// Returns false when the time range is not finite (caller falls back to brute force).
bool compute_block_bounds(const HostAccel& acc, double t_lo, double t_hi, std::vector<double>& bnd);
inline size_t accel_boxes_padded(const HostAccel& acc) { return (acc.n_boxes + 7) / 8 * 8; }  // records of the kind-3 segment

// The same boxes for the float32 slab test (TOR_ACCEL_BLOCKS | TOR_ACCEL_F32): one record of 8 float32 per box,
// same record indices as `bnd`, {c.x, h.x, c.y, h.y, c.z, h.z, 0, 0} relative to `origin` -- centre and half-extent, the
// half-extent rounded up so that the float32 box contains the float64 one (NaN records stay NaN).  Returns max |coordinate| over the
// valid boxes, rounded up; *hmin_out = the smallest half-extent (tor_filter32.hpp: the ray-side inflation is relative to it).
float block_bounds_f32(const std::vector<double>& bnd, const double origin[3], std::vector<float>& bnd32, float* hmin_out);

}  // namespace tor
