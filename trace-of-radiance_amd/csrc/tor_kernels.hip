// tor_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the trace-of-radiance integrator.
//
//   integrate_kernel<SEEDING, ARITH, WAVES_PER_SIMD>
//     render.nim:49-68 (render) + render.nim:21-47 (radiance) + hittables_lists.nim:48-55
//     (closest hit) + spheres.nim:28-49 / moving_spheres.nim:46-67 + materials.nim:21-96.
//
//     Persistent waves with path regeneration: every lane owns one path at a time; a lane whose
//     path ended (sky, absorbed, depth exhausted) is refilled at the top of the next bounce
//     iteration with the next work item (wave ballot + prefix count over a wave-uniform range that
//     is itself pulled from one global counter).  The hot loop -- the brute-force ray x all-objects
//     test, ~87 % of the VALU instructions -- therefore always runs with full waves, whatever the
//     mix of path lengths (1..max_depth).
//       SEEDING 0 (TOR_SEED_PIXEL) : work item = a tile of 64 pixels; a lane runs its pixel's spp
//                        samples in order on the pixel's own stream and sums them in sample order
//                        (bit-faithful to render.nim:59-67).  Tiles come most-expensive-first when
//                        the host ran the probe (SEEDING 2) and tile_order_kernel.
//       SEEDING 1 (TOR_SEED_SAMPLE): work item = pixel-sample, guided chunks; per-sample stream;
//                        radiance is rounded to 2^-36 and accumulated in a per-wave LDS cache, then
//                        with float64 atomics -- every partial sum is exact, so the pixel does not
//                        depend on the schedule.
//       SEEDING 2 (probe)          : SEEDING 1 streams, 2 spp; only counts closest-hit queries per
//                        tile (input of the SEED_PIXEL tile schedule); never touches the canvas.
//
//     Objects are wave-uniform inside the hot loop, so their records come through the scalar data
//     path (s_load into SGPRs, constant-bus operand of the VALU op): no VGPRs, no LDS bandwidth,
//     no per-lane addresses.  Per test the lanes compute only the discriminant of the quadratic
//     (17 / 19 / 23 float64 ops for a static / y-only moving / moving sphere) and a sign-bit
//     filter (one v_bitop3 + one v_alignbit into an 8-object mask); the square root and the two
//     divisions of the reference's `hit` are deferred to a short per-lane pass over the few
//     objects whose discriminant was positive (masks queued in LDS).  Closest hit is order
//     independent (hittables_lists.nim:48-55: strict `<`, ties keep the lowest index) so the
//     deferred pass reproduces the sequential scan exactly.
//
//     TOR_ACCEL_BLOCKS (segment kind 3): the wave-uniform loop tests conservative boxes around
//     spatial blocks of 8 objects instead of the objects; the deferred pass expands, per lane, only
//     the blocks whose box the ray can touch (compact records, staged in LDS when they fit).
//
//     TOR_ACCEL_BLOCKS | TOR_ACCEL_F32 variants resolve their candidates COOPERATIVELY: the wave pools the (ray, block)
//     pairs and (ray, object) survivors of its 64 rays in LDS lists and works through them 64 at a time -- every
//     lane busy, the owners' rays travel over ds_bpermute, closest hits are merged with LDS atomics (section
//     "cooperative resolve" below).
//
//   coop_pixel_kernel  TOR_SEED_PIXEL, one WAVE per pixel chain: whole small frames, or -- split mode -- the most
//                      expensive tiles of a mid-size frame while integrate_kernel renders the rest
//   tile_order_kernel  counting sort of the SEED_PIXEL tiles by probed cost (LPT schedule) + the split point
//   gather_rows_kernel multi-GPU assembly: rank-major row shards -> frame in image order
//   finalize_kernel    canvas.nim:47-54 (draw): pow(sum * 1/spp, 1/gamma)
//   quantize_kernel    io/ppm.nim:15-16
//
// float64 throughout, no FMA contraction (-ffp-contract=off); TOR_ARITH_FUSED uses explicit
// fma() in the discriminant and the moving-sphere centre only.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "tor_device.hpp"
#include "tor_kernels.hpp"
#include "tor_screen.hpp"

namespace tor {

// scalar (constant address space) view of the read-only scene so the compiler emits s_load
typedef const double __attribute__((address_space(4))) * cdptr;
typedef const float __attribute__((address_space(4))) * cfptr;
typedef const double __attribute__((address_space(3))) * ldptr;  // LDS
typedef const float __attribute__((address_space(3))) * lfptr;
typedef const float __attribute__((address_space(1))) * gfptr;
typedef const double __attribute__((address_space(1))) * gdptr;  // global

__device__ __forceinline__ cdptr as_const(const double* p) { return (cdptr)(uintptr_t)p; }

__device__ __forceinline__ int hi32(double x) { return __double2hiint(x); }

__device__ __forceinline__ unsigned long long ballot64(bool p) { return __ballot(p); }

__device__ __forceinline__ unsigned lane_prefix(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

__device__ __forceinline__ unsigned long long bcast_first_u64(unsigned long long v) {
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// One ray/object discriminant, strict or fused.  Returns the sign-bit filter word: negative
// (bit 31 set) iff disc has a clear sign bit (disc >= +0 or NaN+) and (half_b < 0 or c < 0),
// a superset of the objects the reference's hit() can accept (both roots are <= 0 when
// half_b >= 0 and c >= 0).  One v_bitop3_b32: f(a,b,c) = (a|b) & ~c  -> truth table 0x54.
template <int ARITH>
__device__ __forceinline__ int disc_filter(double ox, double oy, double oz, double dx, double dy,
                                           double dz, double a, double cx, double cy, double cz,
                                           double r2) {
  double ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
  double hb, cc, disc;
  if (ARITH != 1) {
    hb = ocx * dx + ocy * dy + ocz * dz;          // spheres.nim:31
    cc = (ocx * ocx + ocy * ocy + ocz * ocz) - r2;  // spheres.nim:32
    disc = hb * hb - a * cc;                      // spheres.nim:33
  } else {
    hb = fma_(ocz, dz, fma_(ocy, dy, ocx * dx));
    cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -r2)));
    disc = fma_(hb, hb, -(a * cc));
  }
#if __has_builtin(__builtin_amdgcn_bitop3_b32)
  return (int)__builtin_amdgcn_bitop3_b32((unsigned)hi32(hb), (unsigned)hi32(cc), (unsigned)hi32(disc), 0x54);
#else
  return (hi32(hb) | hi32(cc)) & ~hi32(disc);
#endif
}

// (ARITH 2, the conservative FMA screen of the strict object loop: tor_screen.hpp)

// candidate-queue entries per lane (LDS, u32): (block << 8) | 8-bit mask.  Brute-force scenes queue one entry per
// 8 objects with a candidate; with TOR_ACCEL_BLOCKS the entries are box masks (8 blocks each) and a lane rarely
// holds more than a handful, so those variants run with half the queue -- the 8 KB per workgroup are what lets
// the block boxes of a 1600-object scene sit in LDS next to its records.  A full queue is not an error: the loop
// resolves what is queued and resumes.
constexpr int queue_cap(int blocks) { return blocks ? 8 : 16; }
constexpr int kBlock = 8;   // objects per queue entry; hot arrays are padded to this (= kPad)
constexpr int kTilePixels = 64;  // SEED_PIXEL work unit: one wave-load of consecutive pixels
constexpr int kAccSlots = 16;  // per-wave LDS pixel-accumulator cache (TOR_SEED_SAMPLE)
static_assert(kBlock == kPad, "hot-record padding must equal the candidate block size");

// LDS per wave: queue (queue_cap*64 u32) + accumulator cache (kAccSlots * (3 f64 + tag)) + debug counters
constexpr int kProfSlots = 16;  // debug counters (u64, stats / wave log only): 5 section sums, trips, begin, last stamp, 4 statistics, 3 wave-log stamps
// cooperative resolve (F32 && BLOCKS variants): pair list and survivor list (64 carried over + 512 new per trip), the
// per-ray closest hit {t bits, (original index, slot)}
constexpr int kCoopList = 576;
// (+ the paths' attenuation: 3 x 64 float64 -- touched once per bounce, so it lives in LDS, not in 6 of the 168 registers)
constexpr int coop_bytes(int blocks) { return (blocks ? 2 : 1) * kCoopList * 4 + 64 * 2 * 8 + 3 * 64 * 8; }  // (no pair list without boxes)
constexpr int wave_lds_bytes(int blocks, int coop = 0) {
  return queue_cap(blocks) * 64 * 4 + kAccSlots * 3 * 8 + kAccSlots * 4 + kProfSlots * 8 + (coop ? coop_bytes(blocks) + 64 : 0);
}
// Which kernel variants resolve cooperatively: TOR_ACCEL_BLOCKS | TOR_ACCEL_F32.  (The code also runs the variants
// without boxes -- `blocks == 0 || f32 != 0` passes every parity test -- but there the candidates are few (1.25-1.43
// per query): a pooled trip costs twice a per-lane trip, the resolve share stays at 5-6 % and the extra LDS and
// registers cost the float64 brute force 2 % (C3 1190 -> 1163 Msamples/s).  Measured, not kept.)
constexpr bool coop_variant(int f32, int blocks) { return blocks != 0 && f32 != 0; }
static_assert(wave_lds_bytes(0) % 16 == 0 && wave_lds_bytes(1) % 16 == 0 && wave_lds_bytes(1, 1) % 16 == 0 && wave_lds_bytes(0, 1) % 16 == 0,
              "keep LDS carve-outs 16-byte aligned");

// The camera (24 float64) is needed once per new path only; read it there instead of keeping
// it in 48 SGPRs across the object loop.  The empty asm makes the pointer opaque per call so
// the loads are not hoisted out of the bounce loop.
__device__ __forceinline__ Camera load_camera(const double* cam_dev) {
  const double* pc = cam_dev;
  asm volatile("" : "+s"(pc));
  cdptr c = as_const(pc);
  Camera cam;
  cam.origin = v3(c[0], c[1], c[2]);
  cam.lower_left_corner = v3(c[3], c[4], c[5]);
  cam.horizontal = v3(c[6], c[7], c[8]);
  cam.vertical = v3(c[9], c[10], c[11]);
  cam.u = v3(c[12], c[13], c[14]);
  cam.v = v3(c[15], c[16], c[17]);
  cam.w = v3(c[18], c[19], c[20]);
  cam.lens_radius = c[21];
  cam.shutter_open = c[22];
  cam.shutter_close = c[23];
  return cam;
}

// Inclusive prefix sum over the 64 lanes of a wave and the wave total: DPP row shifts inside rows of 16 lanes, the
// three row totals through scalar registers.
__device__ __forceinline__ void wave_scan_u32(unsigned v, unsigned& incl, unsigned& total) {
  // __builtin_amdgcn_update_dpp(old, src, dpp_ctrl, row_mask, bank_mask, bound_ctrl): lanes without a source get `old`
  v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 15), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 31);
  const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 47), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
  const unsigned row = (unsigned)(threadIdx.x & 63) >> 4;
  incl = v + (row > 0 ? r0 : 0u) + (row > 1 ? r1 : 0u) + (row > 2 ? r2 : 0u);
  total = r0 + r1 + r2 + r3;
}

// m = (m << 1) | (t >> 31) in one v_alignbit_b32
__device__ __forceinline__ unsigned push_bit(unsigned m, int t) {
  return __builtin_amdgcn_alignbit(m, (unsigned)t, 31);
}

// m = (m << 1) | c with the condition still in its scalar register pair: v_cmp ... + ONE v_addc_co_u32 (m + m + carry-in)
// instead of v_cmp, v_cndmask, v_or (+ a shift every other time) -- the compiler turns `m + m + c` back into those.
__device__ __forceinline__ unsigned push_cond(unsigned m, bool c) {
  const unsigned long long mask = __ballot(c);
  unsigned long long carry_out;
  unsigned r;
  asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(r), "=s"(carry_out) : "v"(m), "s"(mask));
  return r;
}

// chain servers (defined after the kernel): whole waves that continue pixel chains handed over by the lanes
template <int ARITH>
__device__ __forceinline__ bool serve_chains(const KParams& p, bool dedicated);
// which variants carry the hand-off: the reference's streams with both exact accelerations on a single-level layout --
// what tor_render() runs by default
constexpr bool migrate_variant(int seeding, int f32, int blocks) { return seeding == 0 && f32 != 0 && blocks == 1; }

template <int SEEDING, int ARITH, int WAVES_PER_SIMD, int F32, int BLOCKS>
__global__ __launch_bounds__(kThreads, WAVES_PER_SIMD) void integrate_kernel(const KParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  // SEED_PIXEL cooperative variants keep the pixel's running sum in LDS too (touched once per sample; as registers it is
  // six of the 168, and the variant spills inside the bounce loop): it takes the upper half of the queue -- 4 entries are
  // enough there, the box hits live in registers -- plus the SAMPLE accumulator cache this seeding does not use
  constexpr bool kAccInLds = SEEDING == 0 && coop_variant(F32, BLOCKS);
  constexpr int kQLayout = queue_cap(BLOCKS);          // queue entries the LDS layout reserves
  constexpr int kQCap = kAccInLds ? 4 : kQLayout;      // ... and the ones this variant uses
  constexpr int kAccPad = coop_variant(F32, BLOCKS) ? 64 : 0;
  constexpr int kWaveLdsBytes = wave_lds_bytes(BLOCKS, coop_variant(F32, BLOCKS));
  unsigned char* wave_lds = smem_raw + wave * kWaveLdsBytes;
  unsigned* q = reinterpret_cast<unsigned*>(wave_lds) + lane;  // q[k * 64]: k-th entry of this lane
  double* acc_lds = reinterpret_cast<double*>(wave_lds + kQLayout * 64 * 4);          // [kAccSlots][3]
  int* tag_lds = reinterpret_cast<int*>(wave_lds + kQLayout * 64 * 4 + kAccSlots * 24);  // [kAccSlots]
  double* pix_acc = reinterpret_cast<double*>(wave_lds + 4 * 64 * 4) + lane;            // [3][64] (kAccInLds variants; ends kAccPad past the tags)
  unsigned long long* prof_lds = reinterpret_cast<unsigned long long*>(wave_lds + kQLayout * 64 * 4 + kAccSlots * 28 + kAccPad);
  // cooperative resolve state (only carved out in the F32 && BLOCKS variants)
  unsigned char* coop_base = wave_lds + kQLayout * 64 * 4 + kAccSlots * 28 + kAccPad + kProfSlots * 8;
  unsigned long long* coop_t = reinterpret_cast<unsigned long long*>(coop_base);   // [64] closest t so far (bit pattern)
  unsigned long long* coop_w = coop_t + 64;                                        // [64] (original index << 32) | cold slot at that t
  unsigned* coop_surv = reinterpret_cast<unsigned*>(coop_w + 64);                  // [kCoopList] lane | cold slot << 6
  unsigned* coop_pair = coop_surv + kCoopList;                                     // [kCoopList] lane | block << 6 (BLOCKS variants only)
  double* coop_att = reinterpret_cast<double*>(coop_pair + kCoopList) + lane;      // [3][64] the paths' attenuation (coop variants)

  const cdptr stat = as_const(p.stat);
  const cdptr mov = as_const(p.mov);
  const cdptr movy = as_const(p.movy);
  const cdptr segs = as_const(p.segs);
  const double w_div = (double)(p.ncols - 1);  // render.nim:64 divides by float64(ncols-1)
  const double h_div = (double)(p.nrows - 1);
  constexpr bool kProbe = (SEEDING == 2);  // cost probe for the SEED_PIXEL tile schedule

  // TOR_ACCEL_BLOCKS: the block expansion gathers 8 x 64 B per lane and trip with 64 different
  // addresses; when the compact records fit they are staged in LDS once per workgroup.
  const bool staged = BLOCKS && (p.shot_lds_doubles > 0 || p.shot32_lds_floats > 0 || p.bnd32_lds_floats > 0);
  double* stage = reinterpret_cast<double*>(smem_raw + (kThreads / 64) * kWaveLdsBytes);
  const ldptr shot_lds = (ldptr)stage;
  // (a launch stages either the float64 compact records or the float32 pair records)
  float* stage32 = reinterpret_cast<float*>(stage);
  const lfptr shot32_lds = (lfptr)stage32;
  float* stage_b32 = stage32 + p.shot32_lds_floats;  // float32 block boxes of two-level scenes
  const lfptr bnd32_lds = (lfptr)stage_b32;
  if (staged) {
    if (F32) {
      for (int k = threadIdx.x; k < p.shot32_lds_floats; k += kThreads) stage32[k] = p.shot32[k];
      for (int k = threadIdx.x; k < p.bnd32_lds_floats; k += kThreads) stage_b32[k] = p.bnd32[k];
    } else {
      for (int k = threadIdx.x; k < p.shot_lds_doubles; k += kThreads) stage[k] = p.shot[k];
    }
    __syncthreads();
  }

  if (SEEDING == 1) {
    if (lane < kAccSlots) {
      tag_lds[lane] = -1;
      acc_lds[lane * 3 + 0] = 0.0; acc_lds[lane * 3 + 1] = 0.0; acc_lds[lane * 3 + 2] = 0.0;
    }
  }

  // ---- lane state ------------------------------------------------------------------
  bool active = false;     // owns a live path
  bool have_item = false;  // owns a work item whose camera ray has not been generated yet
  V3 o = v3(0, 0, 0), d = v3(0, 0, 1), att = v3(1, 1, 1);
  // attenuation of the lane's path (render.nim:22,35): a register triple, or -- cooperative variants, which are short
  // of registers -- three LDS words per lane
  constexpr bool kAttInLds = coop_variant(F32, BLOCKS);
  auto get_att = [&]() { return kAttInLds ? v3(coop_att[0], coop_att[64], coop_att[128]) : att; };
  auto set_att = [&](V3 v) {
    if (kAttInLds) { coop_att[0] = v.x; coop_att[64] = v.y; coop_att[128] = v.z; }
    else att = v;
  };
  double time = 0.0;
  Rng rng{0, 0, 0, 0};
  int depth = 0;
  int row = 0, col = 0, s = 0;
  int pix = -1;             // index into p.out (local pixel)
  int path_q = 0;           // probe: closest-hit queries of the current path
  V3 acc = v3(0, 0, 0);     // SEED_PIXEL: the pixel's sum, in sample order

  // ---- wave-uniform work range: [w_next, w_end) of the global index space; (cur_pl, cur_s)
  //      is the (local pixel, sample) of w_next --------------------------------------------
  unsigned long long w_next = 0, w_end = 0;
  unsigned cur_pl = 0, cur_s = 0;
  unsigned next_chunk = p.chunk;
  const unsigned prio_slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);  // HW_ID.wave_id: the wave's slot in its SIMD
  unsigned prio_now = 0;
  // SEED_PIXEL tile schedule (see the fetch below): waves of the slow slots skip region A
  bool from_back = SEEDING == 0 && p.sched != nullptr && prio_slot >= (unsigned)p.back_slot;
  const unsigned long long a_end = (SEEDING == 0 && p.sched != nullptr) ? p.sched[0] : p.total_work;
  // a pixel chain is HOT when, extrapolated from its samples so far, it needs more than hot_iters bounce iterations
  const unsigned hot_iters = (SEEDING == 0 && p.sched != nullptr) ? (unsigned)p.sched[2] : 0u;
  bool a_done = from_back;
  unsigned pix_iters = 0;  // bounce iterations the lane has spent on its current pixel
  // arbiter priorities (below): every SEED_PIXEL variant.  (In the cooperative variants the lane's iteration counter was
  // one register too many while the pixel sum still lived in registers: 1-2 % slower then, 1-2 % faster at 100 spp now.)
  constexpr bool kPrio = SEEDING == 0;
  bool exhausted = false;
  // ---- chain hand-off -------------------------------------------------------------------------------
  // A pixel is a sequential chain of spp samples (render.nim:59-67) and a lane needs ~16 us per bounce of it, so a frame
  // cannot end before its longest chain x 16 us -- that, not the machine, bounds small frames, row shards of a multi-GPU
  // job and the glass pixels of any frame.  Lanes therefore hand chains over at a sample boundary (state = pixel, samples
  // done, RNG state, running sum: 64 bytes) to SERVER waves that give all 64 lanes to one chain (serve_chains below):
  //   (a) hot: the chain's projected length (bounce iterations so far / samples so far x spp) exceeds the push threshold;
  //   (b) tail: the wave has run out of fresh pixels and either few of its lanes are still alive or servers sit idle.
  // Servers: the workgroups blockIdx.x < mig[kMigSrvWgs] from the start, and every wave that leaves the lane loop.
  constexpr bool kMigrate = migrate_variant(SEEDING, F32, BLOCKS);
  const bool mig_on = kMigrate && p.mig != nullptr;
  if (mig_on && threadIdx.x == 0 && blockIdx.x == 0) p.mig[kMigT0] = wall_clock64();
  if (mig_on && lane == 0) { prof_lds[0] = ~0ull; prof_lds[1] = 0; }  // the wave's copy of the push threshold, bounce counter
  const bool server_only = mig_on && (unsigned long long)blockIdx.x < ((const unsigned long long __attribute__((address_space(4)))*)(uintptr_t)p.mig)[kMigSrvWgs];
  // statistics (tor_last_stats / wave log) live in LDS and are touched only when they were asked for: the
  // kernel is short of scalar registers, counters that are always live would be paid for on every launch
  const bool stats_on = p.stats != nullptr;
  // debug (wave_log only): shader-clock cycles per section of the bounce iteration and trips of the resolve
  // loop, kept in LDS (lane 0) so that the counters cost no registers when they are off
  const bool prof = p.wave_log != nullptr;
  enum { kSecRefill = 0, kSecLoop, kSecResolve, kSecShade, kSecDeposit, kSecTrips, kSecBegin, kSecMark,
         kStQueries, kStCand, kStIters, kStSamples, kLogStart, kLogExhausted, kLogItersAtExhaustion };
  if ((stats_on || prof) && lane == 0) {
    for (int k = 0; k < kProfSlots; ++k) prof_lds[k] = 0;
    prof_lds[kSecBegin] = prof_lds[kSecMark] = __builtin_readcyclecounter();
    if (prof) prof_lds[kLogStart] = wall_clock64();
  }
#define TOR_SEC(slot)                                             \
  if (prof && lane == 0) {                                        \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    prof_lds[slot] += now_ - prof_lds[kSecMark];                  \
    prof_lds[kSecMark] = now_;                                    \
  }

  // Roles (hand-off): a dedicated server serves first -- and comes back from there only to become a lane wave (nothing to
  // serve: a scene without long chains) or because the frame is over; then the lane loop; then every wave serves until
  // the frame is over.  serve_chains is inlined at BOTH places (two copies of its code, no loop around them): as a real
  // function (noinline) the call's ABI -- reserved scalar registers, stack -- cost the lane loop ~240 more scalar spills
  // (v_readlane / v_writelane) and 5 % on every frame; inlined once inside a role loop its registers stayed live across the
  // lane loop (170 spilled VGPRs).
  bool lane_role = !server_only;
  if constexpr (kMigrate) {
    if (server_only && p.mig_tail_lanes > -2) lane_role = serve_chains<ARITH>(p, true);
  }
  const bool ran_lanes = lane_role;
  for (; lane_role;) {
    // ================= (A) refill lanes that have no live path =========================
    bool need_fetch = !active && !have_item;
    unsigned long long need_mask = ballot64(need_fetch);
    if (need_mask != 0) {
      if (w_next >= w_end && !exhausted) {
        const int leader = (int)__builtin_ctzll(need_mask);
        // SEED_PIXEL: tiles of p.chunk consecutive pixels (neighbouring pixels in one wave keep the
        // rays coherent).  SEED_SAMPLE: guided self-scheduling, the chunk shrinks with the work
        // that is left (waves of one SIMD get very unequal service -- slot 0 runs ~20x faster than
        // slot 4 -- so anything parked in a slow wave becomes the tail of the frame).
        // (whole pixels per grab where the chunk allows it: a pixel's samples then meet in ONE wave's LDS accumulator
        // and reach HBM in one flush instead of one per wave that touched the pixel -- 5.7x less write traffic at
        // 1000 spp; the tail of the guided schedule shrinks below a pixel and splits it again, which is fine)
        unsigned grab = (SEEDING == 0) ? p.chunk : next_chunk;
        if (SEEDING != 0 && grab >= (unsigned)p.spp) grab = grab / (unsigned)p.spp * (unsigned)p.spp;
        unsigned long long base = 0;
        bool retry = false;  // SEED_PIXEL: nothing fetched this time, but the wave is not done
        if (SEEDING == 0) {
          // Two regions of the chain-length-descending tile order (tile_order_kernel): A = [.., a_end) holds the long
          // chains, B = [a_end, total) the cheap end.  Waves in the fast hardware slots work through A and then B;
          // waves in the slow slots take from B only.  Why: the instruction arbiter serves the oldest wave of a SIMD
          // first -- 39 / 64 / 177 us per bounce iteration in wave slots 0 / 1 / 2 with three waves per SIMD -- and a
          // pixel is one sequential chain of spp samples: the longest chains (glass, up to ~36 queries per sample) need about
          // as many iterations as an average wave runs in the whole frame, so they finish in time only in a fast slot;
          // in slot 2 they end the frame alone (measured: counter dry at 172 ms, last wave at 200-225 ms).  Equalising
          // the service with s_setprio makes it worse (64 us for everybody: every long chain is late).  Everybody ends
          // in B, cheapest tiles last.
          if (!a_done) {
            if (lane == leader) base = atomicAdd(p.work_counter, (unsigned long long)kTilePixels);
            base = bcast_first_u64(__shfl(base, leader));
            if (base >= a_end) a_done = true;
          }
          if (a_done) {
            base = p.total_work;
            bool take = p.sched != nullptr;
            if (take && from_back) {
              // a slow wave stops taking pixels once the fast waves have finished region A and arrive in B: even the
              // cheapest pixel (sky, spp iterations) takes it ~10 % of the frame; it drains at the top priority and
              // leaves its issue slots to the others
              unsigned long long front = 0;
              if (lane == leader) front = __hip_atomic_load(p.work_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              front = bcast_first_u64(__shfl(front, leader));
              take = front < a_end;
            }
            if (take) {
              if (lane == leader) base = atomicAdd(p.sched + 1, (unsigned long long)kTilePixels);
              base = bcast_first_u64(__shfl(base, leader));
              if (from_back && base >= p.total_work) {
                // B ran dry while A still has tiles.  The hardware slot is only a hint about the wave's speed -- with
                // other kernels on the device every wave of this launch may sit in a "slow" slot, and then nobody would
                // render A: from here on the wave is a front wave (it asks A at its next fetch).
                from_back = false;
                a_done = false;
                retry = true;
              }
            }
          }
        } else {
          if (lane == leader) base = atomicAdd(p.work_counter, (unsigned long long)grab);
          base = bcast_first_u64(__shfl(base, leader));
        }
        if (base >= p.total_work) {
          exhausted = !retry;
          if (kMigrate && p.mig != nullptr && exhausted && lane == 0) atomicMin(p.mig + kMigTCounterDry, (unsigned long long)wall_clock64());
          if (prof && lane == 0 && exhausted) { prof_lds[kLogExhausted] = wall_clock64(); prof_lds[kLogItersAtExhaustion] = prof_lds[kStIters]; }
        } else {
          w_next = base;
          w_end = (base + grab < p.total_work) ? base + grab : p.total_work;
          if (SEEDING == 0) {
            // work index space = tiles of kTilePixels pixels, optionally in cost order (LPT): the
            // k-th fetch renders tile order[k]
            unsigned tile = (unsigned)(base / kTilePixels);
            if (p.order != nullptr) tile = p.order[tile];
            cur_pl = tile * kTilePixels;
            cur_s = 0;
            const unsigned left_px = p.n_pixels - cur_pl;
            w_end = base + ((left_px < (unsigned)kTilePixels) ? left_px : (unsigned)kTilePixels);
          } else {  // one 64-bit division per chunk, wave-uniform
            unsigned long long pl0 = base / (unsigned)p.spp;
            cur_pl = (unsigned)pl0;
            cur_s = (unsigned)(base - pl0 * (unsigned)p.spp);
            const unsigned long long left = p.total_work - w_end;
            unsigned long long g = left / ((unsigned long long)p.n_waves * 32ull);
            g = (g > p.chunk) ? p.chunk : g;
            next_chunk = (g < 64) ? 64u : (unsigned)g;
          }
        }
      }
      if (w_next < w_end) {
        const unsigned prefix = lane_prefix(need_mask);
        const unsigned avail = (unsigned)(w_end - w_next);
        const unsigned want = (unsigned)__builtin_popcountll(need_mask);
        const unsigned take = (want < avail) ? want : avail;
        const bool got = need_fetch && prefix < take;
        if (got) {
          unsigned pl;
          if (SEEDING == 0) {
            pl = cur_pl + prefix;
            s = 0;
            if (kAccInLds) { pix_acc[0] = 0.0; pix_acc[64] = 0.0; pix_acc[128] = 0.0; }
            else acc = v3(0, 0, 0);
          } else {  // sample and probe
            const unsigned t = cur_s + prefix;
            const unsigned dp = t / (unsigned)p.spp;
            pl = cur_pl + dp;
            s = (int)(t - dp * (unsigned)p.spp);
          }
          const unsigned lrow = pl / (unsigned)p.ncols;
          col = (int)(pl - lrow * (unsigned)p.ncols);
          // local row -> image row (tiles of row_tile rows dealt round-robin to the shards)
          const unsigned tile = lrow / (unsigned)p.row_tile;
          const unsigned within = lrow - tile * (unsigned)p.row_tile;
          row = (int)((tile * (unsigned)p.shard_count + (unsigned)p.shard_index) * (unsigned)p.row_tile + within);
          pix = (int)pl;
          have_item = true;
          if (SEEDING == 0) {
            seed2(rng, (uint64_t)(int64_t)row, (uint64_t)(int64_t)col);  // render.nim:59-60
            if (kPrio) pix_iters = 0;
          }
        }
        w_next += take;
        if (SEEDING == 0) {
          cur_pl += take;
        } else {
          const unsigned t = cur_s + take;
          const unsigned dp = t / (unsigned)p.spp;
          cur_pl += dp;
          cur_s = t - dp * (unsigned)p.spp;
        }
      }
    }
    if (!active && have_item) {
      if (SEEDING != 0) seed3(rng, (uint64_t)row, (uint64_t)col, (uint64_t)s);
      have_item = false;  // the pending sample is consumed by starting its path
      // render.nim:64-66
      const double u = ((double)col + uniform01(rng)) / w_div;
      const double v = ((double)row + uniform01(rng)) / h_div;
      const Camera cam = load_camera(p.cam_dev);
      const Ray r = camera_ray(cam, u, v, rng);
      o = r.origin;
      d = r.direction;
      time = r.time;
      set_att(v3(1.0, 1.0, 1.0));  // render.nim:22
      depth = 0;
      active = true;
    }
    TOR_SEC(kSecRefill)
    const unsigned long long active_mask = ballot64(active);
    if (active_mask == 0) {
      if (exhausted) break;
      continue;
    }
    if (kPrio && p.sched != nullptr) {
      // Arbiter priority of the wave (s_setprio; the levels beat the age order):
      //  3  while one of its lanes works on a HOT pixel -- a chain so long (glass: up to ~36 queries per sample against
      //     a mean of 2.6) that even an average slot would finish it after everybody else.  Judged on the lane's own
      //     record: iterations so far, extrapolated to spp samples, against hot_iters (a share of what an average wave
      //     runs in the whole frame, from the probe's total); a few dozen pixels per frame;
      //  1-2 fast slots, taking turns (level = 1 + (slot + clock phase) mod 2, the phase from the shader clock they
      //     share) so that both get the same service;
      //  0  slow slots (region B only).
      if (active) pix_iters += 1;
      const bool lane_hot = active && hot_iters != 0 && pix_iters >= 64u &&
                            (unsigned long long)pix_iters * (unsigned)p.spp >= (unsigned long long)hot_iters * (unsigned)(s + 1);
      const bool wave_hot = ballot64(lane_hot) != 0;
      unsigned level = 0;
      if (wave_hot || (from_back && exhausted)) {
        level = 3;
      } else if (!from_back) {
        level = 1;
        if (p.prio_shift > 0) {
          const unsigned n_front = ((unsigned)p.back_slot < (unsigned)WAVES_PER_SIMD) ? (unsigned)p.back_slot : (unsigned)WAVES_PER_SIMD;
          const unsigned phase = (unsigned)(__builtin_readcyclecounter() >> p.prio_shift);
          level = 1u + (prio_slot + phase) % (n_front < 2u ? 1u : 2u);
        }
      }
      // (s_setprio is a scalar instruction: it must sit behind scalar branches.  Everything `level` depends on is the same
      // in all lanes, but some of it is assigned under conditions the compiler cannot prove uniform; as a vector value the
      // four s_setprio ended up in exec-masked regions and simply ran one after the other)
      level = (unsigned)__builtin_amdgcn_readfirstlane((int)level);
      if (level != prio_now) {
        prio_now = level;
        if (level == 0) __builtin_amdgcn_s_setprio(0);
        else if (level == 1) __builtin_amdgcn_s_setprio(1);
        else if (level == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
      }
    }
    if (stats_on && lane == 0) {
      prof_lds[kStIters] += 1;
      prof_lds[kStQueries] += (unsigned long long)__builtin_popcountll(active_mask);
    }
    if (kProbe && active) path_q += 1;
    bool ended = false;
    V3 radiance = v3(0.0, 0.0, 0.0);
    // kCoop: TOR_ACCEL_BLOCKS | TOR_ACCEL_F32 resolve their candidates COOPERATIVELY (all 64 lanes, active or not,
    // work through the wave's (ray, block) pairs and (ray, object) survivors: see the resolve section below)
    constexpr bool kCoop = coop_variant(F32, BLOCKS);
    // super boxes (two-level culling layouts, > 96 blocks): the cooperative variants carry that code only as BLOCKS = 2
    // -- compiled into the single-level variants it cost them ~1 % (registers) -- the others always
    constexpr bool kSuper = BLOCKS == 2 || (BLOCKS == 1 && F32 == 0);
    // ARITH 2: the reference's arithmetic (as ARITH 0) behind a conservative FMA screen in the wave-uniform object loop
    constexpr bool kScreen = ARITH == 2;
    {
      // ================= (B) closest hit over all objects ==============================
      // hittables_lists.nim:48-55 with t_min = 0.001, t_max = Inf (render.nim:28)
      // (computed by every lane: a lane without a live path works on stale values that nobody reads)
      const double ox = o.x, oy = o.y, oz = o.z, dx = d.x, dy = d.y, dz = d.z;
      const double a_strict = dx * dx + dy * dy + dz * dz;  // spheres.nim:30
      const double a = (ARITH != 1) ? a_strict : fma_(dz, dz, fma_(dy, dy, dx * dx));
      double best_t = __builtin_inf();
      int best_idx = -1;
      int best_orig = 0x7fffffff;
      double best_f = 0.0;
      // ARITH 2 (conservative FMA screen, screen_filter above): the ray's share of the margins
      double scr_s1 = 0.0, scr_d1 = 0.0, scr_negmu = 0.0, scr_am = 0.0;
      ScreenRay sray{};  // second form of the screen (tor_screen.hpp): the normalised direction, once per query
      if (kScreen) {
        scr_s1 = __builtin_fabs(ox) + __builtin_fabs(oy) + __builtin_fabs(oz);
        scr_d1 = __builtin_fabs(dx) + __builtin_fabs(dy) + __builtin_fabs(dz);
        sray = screen2_ray(ox, oy, oz, dx, dy, dz, a_strict);
      }

      // TOR_ACCEL_F32: the ray in float32, relative to the scene origin (used by segment kinds 5-7 only)
      RayF32 r32{};
      if (F32) r32 = make_ray_f32(ox, oy, oz, dx, dy, dz, a_strict, p.org[0], p.org[1], p.org[2]);

      BoxRay32 b32{};
      const bool boxes32 = F32 && BLOCKS && p.bnd32 != nullptr;
      if (boxes32) b32 = make_box_ray32(r32, p.sp_bmax);

      // TOR_ACCEL_F32 block expansion: the spatial movers share one time group
      SegF32 sp32{};
      double f_sp = 0.0;  // moving_spheres.nim:42 for the spatial movers' (time0, time1)
      if (F32 && BLOCKS && p.shot32 != nullptr) {
        f_sp = (time - p.sp_t0) / p.sp_dt;
        sp32 = make_seg_f32(r32, f_sp, p.sp_mc0max, p.sp_dcmax);
      }
      if (kCoop) {  // this lane's closest hit so far lives in LDS, where the lanes that test its candidates can reach it
        coop_t[lane] = 0x7ff0000000000000ull;  // +inf
        coop_w[lane] = ~0ull;                  // (original index << 32) | cold slot of the object at that t
      }
      // kCoop: the slab tests of a box segment of at most 64 boxes leave their bits in a register pair (bit 63 - k <->
      // box k) instead of the LDS queue
      unsigned long long box_mask = 0;
      int box_kind = 0;       // 3: block boxes, 4: super boxes
      unsigned box_group0 = 0;  // index of the first group of 8 boxes (first box / 8)
      int seg = 0;
      int i = 0;
      // kWords (the ARITH 2 variants: strict brute force behind the FMA screen): the candidate bits of 32 consecutive slots
      // travel in ONE word per lane -- the v_alignbit of every test shifts the running mask, nothing else happens per block of 8 --
      // and word k of a pass over the objects goes to LDS slot k unconditionally: no queue counter, no compaction, no overflow
      // test per block (6 VALU instructions and a ballot per 8 objects before; 3 per 32 now).  A 16-bit summary (one bit per
      // word, set when the word is not empty) lets the resolve pass skip the empty ones.  Scenes of more than 16 words (512
      // slots) take several passes: the `full` mechanism of the queue, wave-uniform here.
      constexpr bool kWords = kScreen;
      // the time fraction of the last moving segment this query walked, for the resolve pass: a candidate of the same
      // (time0, time1 - time0) -- bit patterns -- takes it instead of dividing again (moving_spheres.nim:42: same operands, same quotient)
      double fc_f = 0.0;
      unsigned long long fc_t0 = 0x7ff8dead00000001ull, fc_dt = 0x7ff8dead00000002ull;  // (NaN patterns no object carries)
      unsigned mw = 0;      // the running mask
      int w_base = 0;       // first word of the current pass (words = slots / 32 over the whole sorted list)
      for (;;) {
        unsigned qn = 0;
        bool full = false;
        unsigned nz = 0;    // kWords: bit (w_count - 1 - k) <-> word k of this pass is not empty
        int w_count = 0;    // kWords: words stored in this pass (wave-uniform)
        int gb_next = 0;    // kWords: the block index behind the last block tested (wave-uniform)
        // (seg, i) are the same in every lane that has a live path, but they are updated under `if (active)` inside
        // this loop, which makes them divergent in the compiler's eyes -- and the object records would then come
        // through vector loads instead of the scalar data path (measured: half the speed).  readfirstlane inside
        // the branch restores their uniformity.
        if (active) {
        seg = __builtin_amdgcn_readfirstlane(seg);
        i = __builtin_amdgcn_readfirstlane(i);
        if (kWords) w_base = __builtin_amdgcn_readfirstlane(w_base);
        // end of a block of 8 in the kWords variants: the mask lives on in mw; behind every 4th block of the sorted list the word is stored
#define TOR_WORDS_END_BLOCK()                                                  \
        {                                                                      \
          mw = m;                                                              \
          const int gb_ = seg_block0 + i / kBlock;                             \
          gb_next = gb_ + 1;                                                   \
          if ((gb_ & 3) == 3) {                                                \
            q[(unsigned)w_count * 64] = m;                                     \
            nz = push_cond(nz, m != 0u);                                       \
            w_count += 1;                                                      \
            if (w_count == kQCap) { full = true; i += kBlock; break; }         \
          }                                                                    \
        }
        while (seg < p.n_segs) {
          const int seg_kind = (int)segs[seg * 8 + 0];
          const int seg_begin = (int)segs[seg * 8 + 1];    // first hot record of the segment
          const int seg_count_word = (int)segs[seg * 8 + 2];
          const int seg_count = seg_count_word & 0xffffff;   // padded to kBlock
          const int seg_real = seg_count - (seg_count_word >> 24);  // kinds 0-2: the objects that exist (the last block's tail is padding)
          const int seg_block0 = (int)segs[seg * 8 + 3];   // (first sorted index) / kBlock
          // ARITH 2: segments with second-form records (tor_screen.hpp: the quadratic expanded around the ray, direction
          // normalised per ray -- 8 / 6 / 9 float64 instructions per object, all but one fused multiply-adds against scalar
          // operands); the others keep the first form below
          const int xkind = kScreen ? (int)as_const(p.xsegs)[seg * 8 + 0] : 0;
          if (kScreen && xkind >= 10) {
            const int x_first = (int)as_const(p.xsegs)[seg * 8 + 1];
            double f = 0.0;
            if (xkind == 12) {
              f = (time - segs[seg * 8 + 4]) / segs[seg * 8 + 5];  // moving_spheres.nim:42
              fc_f = f; fc_t0 = double_to_bits(segs[seg * 8 + 4]); fc_dt = double_to_bits(segs[seg * 8 + 5]);
            }
            const ScreenSeg ss = screen2_seg(sray, segs[seg * 8 + 6], segs[seg * 8 + 7], as_const(p.xsegs)[seg * 8 + 2], f);
            if (xkind == 12) {
              cdptr rec = as_const(p.xrec) + ((long)x_first + 6 * (long)i);
              double n0 = rec[0], n1 = rec[1], n2 = rec[2], n3 = rec[3], n4 = rec[4];
              for (; i < seg_count; i += kBlock) {
                unsigned m = kWords ? mw : 0u;
                const int n_live = seg_real - i;  // (wave-uniform)
                if (n_live >= kBlock) {
#pragma unroll
                  for (int j = 0; j < kBlock; ++j) {
                    const double c0 = n0, c1 = n1, c2 = n2, c3 = n3, c4 = n4;
                    n0 = rec[6 * (j + 1) + 0]; n1 = rec[6 * (j + 1) + 1]; n2 = rec[6 * (j + 1) + 2];
                    n3 = rec[6 * (j + 1) + 3]; n4 = rec[6 * (j + 1) + 4];
                    m = push_bit(m, screen2_movy_y(ss, c0, c1, c2, c3, c4));
                  }
                } else {  // the last block: only its real objects
#pragma unroll 1
                  for (int j = 0; j < n_live; ++j) m = push_bit(m, screen2_movy_y(ss, rec[6 * j + 0], rec[6 * j + 1], rec[6 * j + 2], rec[6 * j + 3], rec[6 * j + 4]));
                  m <<= (unsigned)(kBlock - n_live);
                }
                rec += 6 * kBlock;
                if constexpr (kWords) TOR_WORDS_END_BLOCK()
                else {
                q[qn * 64] = ((unsigned)(seg_block0 + i / kBlock) << 8) | m;
                qn += (m != 0) ? 1u : 0u;
                if (ballot64(qn >= (unsigned)kQCap) != 0) { full = true; i += kBlock; break; }
                }
              }
            } else {
              cdptr rec = as_const(p.xrec) + ((long)x_first + 4 * (long)i);
              double n0 = rec[0], n1 = rec[1], n2 = rec[2], n3 = rec[3];
              for (; i < seg_count; i += kBlock) {
                unsigned m = kWords ? mw : 0u;
                const int n_live = seg_real - i;  // (wave-uniform)
                if (n_live >= kBlock) {
                  if (xkind == 10) {
#pragma unroll
                    for (int j = 0; j < kBlock; ++j) {
                      const double c0 = n0, c1 = n1, c2 = n2, c3 = n3;
                      n0 = rec[4 * (j + 1) + 0]; n1 = rec[4 * (j + 1) + 1]; n2 = rec[4 * (j + 1) + 2]; n3 = rec[4 * (j + 1) + 3];
                      m = push_bit(m, screen2_static(ss, c0, c1, c2, c3));
                    }
                  } else {
#pragma unroll
                    for (int j = 0; j < kBlock; ++j) {
                      const double c0 = n0, c1 = n1, c2 = n2;
                      n0 = rec[4 * (j + 1) + 0]; n1 = rec[4 * (j + 1) + 1]; n2 = rec[4 * (j + 1) + 2];
                      m = push_bit(m, screen2_static_y(ss, c0, c1, c2));
                    }
                  }
                } else {  // the last block: only its real objects
#pragma unroll 1
                  for (int j = 0; j < n_live; ++j)
                    m = push_bit(m, xkind == 10 ? screen2_static(ss, rec[4 * j + 0], rec[4 * j + 1], rec[4 * j + 2], rec[4 * j + 3])
                                                : screen2_static_y(ss, rec[4 * j + 0], rec[4 * j + 1], rec[4 * j + 2]));
                  m <<= (unsigned)(kBlock - n_live);
                }
                rec += 4 * kBlock;
                if constexpr (kWords) TOR_WORDS_END_BLOCK()
                else {
                q[qn * 64] = ((unsigned)(seg_block0 + i / kBlock) << 8) | m;
                qn += (m != 0) ? 1u : 0u;
                if (ballot64(qn >= (unsigned)kQCap) != 0) { full = true; i += kBlock; break; }
                }
              }
            }
          } else if (seg_kind == 0) {
            // one base pointer per block, immediate offsets inside it, and the next record is
            // requested one object ahead of its use (s_load latency hides under ~17 VALU ops)
            cdptr rec = stat + 4 * (long)(seg_begin + i);
            double n0 = rec[0], n1 = rec[1], n2 = rec[2], n3 = rec[3];
            if (kScreen) screen_margins(scr_s1 + segs[seg * 8 + 6], scr_d1, a, scr_negmu, scr_am);
            for (; i < seg_count; i += kBlock) {
              unsigned m = kWords ? mw : 0u;
              const int n_live = seg_real - i;  // (wave-uniform)
              if (n_live >= kBlock) {
#pragma unroll
                for (int j = 0; j < kBlock; ++j) {
                  const double c0 = n0, c1 = n1, c2 = n2, c3 = n3;
                  n0 = rec[4 * (j + 1) + 0]; n1 = rec[4 * (j + 1) + 1];   // next object (the arrays carry one
                  n2 = rec[4 * (j + 1) + 2]; n3 = rec[4 * (j + 1) + 3];   // record of slack past the last block)
                  if (kScreen) m = push_bit(m, screen_filter(ox - c0, oy - c1, oz - c2, dx, dy, dz, a, scr_negmu, scr_am, c3));
                  else m = push_bit(m, disc_filter<ARITH>(ox, oy, oz, dx, dy, dz, a, c0, c1, c2, c3));
                }
              } else {
                // the segment's last block: only its real objects (a padding record costs as much as a sphere; the four
                // always-tested spheres of a culling layout are half a block)
#pragma unroll 1
                for (int j = 0; j < n_live; ++j) {
                  const double c0 = rec[4 * j + 0], c1 = rec[4 * j + 1], c2 = rec[4 * j + 2], c3 = rec[4 * j + 3];
                  if (kScreen) m = push_bit(m, screen_filter(ox - c0, oy - c1, oz - c2, dx, dy, dz, a, scr_negmu, scr_am, c3));
                  else m = push_bit(m, disc_filter<ARITH>(ox, oy, oz, dx, dy, dz, a, c0, c1, c2, c3));
                }
                m <<= (unsigned)(kBlock - n_live);
              }
              rec += 4 * kBlock;
              if constexpr (kWords) TOR_WORDS_END_BLOCK()
              else {
              q[qn * 64] = ((unsigned)(seg_block0 + i / kBlock) << 8) | m;
              qn += (m != 0) ? 1u : 0u;
              if (ballot64(qn >= (unsigned)kQCap) != 0) { full = true; i += kBlock; break; }
              }
            }
          } else if (F32 && seg_kind >= 5) {
            // TOR_ACCEL_F32 (tor_filter32.hpp): conservative packed-float32 discriminant, two objects per
            // instruction; keeps a superset of what the float64 sign filter keeps, the deferred pass below
            // re-does the kept objects in float64 exactly as the reference.
            double f64 = 0.0;
            if (seg_kind != 5) f64 = (time - segs[seg * 8 + 4]) / segs[seg * 8 + 5];  // moving_spheres.nim:42
            const SegF32 s32 = make_seg_f32(r32, f64, (float)segs[seg * 8 + 6], (float)segs[seg * 8 + 7]);
            const int stride = (seg_kind == 5) ? 10 : ((seg_kind == 6) ? 12 : 16);
            cfptr rec = (cfptr)(uintptr_t)p.hot32 + ((long)seg_begin + (long)(i / 2) * stride);
            if (seg_kind == 5) {
              for (; i < seg_count; i += kBlock) {
                unsigned m = kWords ? mw : 0u;
#pragma unroll
                for (int j = 0; j < kBlock / 2; ++j) {
                  cfptr r = rec + 10 * j;
                  m = filter_pair32(r32, s32, oc_static32(r32.ox, (f2v){r[0], r[1]}), oc_static32(r32.oy, (f2v){r[2], r[3]}),
                                    oc_static32(r32.oz, (f2v){r[4], r[5]}), (f2v){r[6], r[7]}, (f2v){r[8], r[9]}, m);
                }
                rec += 10 * (kBlock / 2);
                m |= s32.wild;
                if constexpr (kWords) TOR_WORDS_END_BLOCK()
                else {
                q[qn * 64] = ((unsigned)(seg_block0 + i / kBlock) << 8) | m;
                qn += (m != 0) ? 1u : 0u;
                if (ballot64(qn >= (unsigned)kQCap) != 0) { full = true; i += kBlock; break; }
                }
              }
            } else if (seg_kind == 6) {
              for (; i < seg_count; i += kBlock) {
                unsigned m = kWords ? mw : 0u;
#pragma unroll
                for (int j = 0; j < kBlock / 2; ++j) {
                  cfptr r = rec + 12 * j;
                  m = filter_pair32(r32, s32, oc_static32(r32.ox, (f2v){r[0], r[1]}),
                                    oc_moving32(r32.oy, (f2v){r[2], r[3]}, (f2v){r[10], r[11]}, s32.nf),
                                    oc_static32(r32.oz, (f2v){r[4], r[5]}), (f2v){r[6], r[7]}, (f2v){r[8], r[9]}, m);
                }
                rec += 12 * (kBlock / 2);
                m |= s32.wild;
                if constexpr (kWords) TOR_WORDS_END_BLOCK()
                else {
                q[qn * 64] = ((unsigned)(seg_block0 + i / kBlock) << 8) | m;
                qn += (m != 0) ? 1u : 0u;
                if (ballot64(qn >= (unsigned)kQCap) != 0) { full = true; i += kBlock; break; }
                }
              }
            } else {
              for (; i < seg_count; i += kBlock) {
                unsigned m = kWords ? mw : 0u;
#pragma unroll
                for (int j = 0; j < kBlock / 2; ++j) {
                  cfptr r = rec + 16 * j;
                  m = filter_pair32(r32, s32, oc_moving32(r32.ox, (f2v){r[0], r[1]}, (f2v){r[10], r[11]}, s32.nf),
                                    oc_moving32(r32.oy, (f2v){r[2], r[3]}, (f2v){r[12], r[13]}, s32.nf),
                                    oc_moving32(r32.oz, (f2v){r[4], r[5]}, (f2v){r[14], r[15]}, s32.nf),
                                    (f2v){r[6], r[7]}, (f2v){r[8], r[9]}, m);
                }
                rec += 16 * (kBlock / 2);
                m |= s32.wild;
                if constexpr (kWords) TOR_WORDS_END_BLOCK()
                else {
                q[qn * 64] = ((unsigned)(seg_block0 + i / kBlock) << 8) | m;
                qn += (m != 0) ? 1u : 0u;
                if (ballot64(qn >= (unsigned)kQCap) != 0) { full = true; i += kBlock; break; }
                }
              }
            }
          } else if (F32 && BLOCKS && (seg_kind == 3 || seg_kind == 4)) {
            // the same boxes through the float32 slab test (tor_filter32.hpp): 8 floats per record via scalar loads
            cfptr rec = (cfptr)(uintptr_t)p.bnd32 + 8 * (long)(seg_begin + i);
            if (seg_count <= 64) {
              box_kind = seg_kind;
              box_group0 = (unsigned)seg_block0;
              for (; i < seg_count; i += kBlock) {
                unsigned m = kWords ? mw : 0u;
#pragma unroll
                for (int j = 0; j < kBlock; ++j)
                  m = push_cond(m, slab_bit32(b32, (f2v){rec[8 * j + 0], rec[8 * j + 1]}, (f2v){rec[8 * j + 2], rec[8 * j + 3]},
                                            (f2v){rec[8 * j + 4], rec[8 * j + 5]}) != 0u);
                rec += 8 * kBlock;
                unsigned wild = r32.wild;  // (see below: never into the padding super boxes)
                if (seg_kind == 4) {
                  const int n_valid = p.n_super - i;
                  wild &= (n_valid >= kBlock) ? 0xffu : ((n_valid <= 0) ? 0u : ((0xff00u >> n_valid) & 0xffu));
                }
                m |= wild;
                box_mask |= (unsigned long long)m << (56 - i);
              }
            } else
            for (; i < seg_count; i += kBlock) {
              unsigned m = kWords ? mw : 0u;
#pragma unroll
              for (int j = 0; j < kBlock; ++j)
                m = push_cond(m, slab_bit32(b32, (f2v){rec[8 * j + 0], rec[8 * j + 1]}, (f2v){rec[8 * j + 2], rec[8 * j + 3]},
                                          (f2v){rec[8 * j + 4], rec[8 * j + 5]}) != 0u);
              rec += 8 * kBlock;
              // a 'wild' ray (outside the float32 filter's guarded ranges) enters every box -- every REAL box: the
              // padding entries of the super-box segment have no block boxes or records behind them
              unsigned wild = r32.wild;
              if (seg_kind == 4) {
                const int n_valid = p.n_super - i;  // bit (7 - j) <-> record i + j
                wild &= (n_valid >= kBlock) ? 0xffu : ((n_valid <= 0) ? 0u : ((0xff00u >> n_valid) & 0xffu));
              }
              m |= wild;
              q[qn * 64] = ((seg_kind == 3) ? 0x80000000u : 0x40000000u) | ((unsigned)(seg_block0 + i / kBlock) << 8) | m;
              qn += (m != 0) ? 1u : 0u;
              if (ballot64(qn >= (unsigned)kQCap) != 0) { full = true; i += kBlock; break; }
            }
          } else if (BLOCKS && (seg_kind == 3 || seg_kind == 4)) {
            // TOR_ACCEL_BLOCKS: the records are inflated axis-aligned boxes around spatial blocks of 8
            // objects (kind 3, entries flagged with bit 31) or around 8 such blocks (kind 4, bit 30).
            // Slab test; a set bit means 'this lane has to look inside'.  1/d may be +-inf (d = 0): (lo - o) * inf is +-inf, or
            // NaN when lo == o, and v_min/v_max_f64 drop a NaN operand -- the axis then imposes no
            // constraint, which is the conservative answer.
            const double ix = 1.0 / dx, iy = 1.0 / dy, iz = 1.0 / dz;
            cdptr rec = as_const(p.bnd) + 8 * (long)(seg_begin + i);
            for (; i < seg_count; i += kBlock) {
              unsigned m = kWords ? mw : 0u;
#pragma unroll
              for (int j = 0; j < kBlock; ++j) {
                const double tx0 = (rec[8 * j + 0] - ox) * ix, tx1 = (rec[8 * j + 3] - ox) * ix;
                const double ty0 = (rec[8 * j + 1] - oy) * iy, ty1 = (rec[8 * j + 4] - oy) * iy;
                const double tz0 = (rec[8 * j + 2] - oz) * iz, tz1 = (rec[8 * j + 5] - oz) * iz;
                const double t_in = __builtin_fmax(__builtin_fmax(__builtin_fmin(tx0, tx1), __builtin_fmin(ty0, ty1)),
                                                   __builtin_fmax(__builtin_fmin(tz0, tz1), 0.0));
                const double t_out = __builtin_fmin(__builtin_fmin(__builtin_fmax(tx0, tx1), __builtin_fmax(ty0, ty1)),
                                                    __builtin_fmax(tz0, tz1));
                m = (m << 1) | ((t_in <= t_out) ? 1u : 0u);
              }
              rec += 8 * kBlock;
              q[qn * 64] = ((seg_kind == 3) ? 0x80000000u : 0x40000000u) | ((unsigned)(seg_block0 + i / kBlock) << 8) | m;
              qn += (m != 0) ? 1u : 0u;
              if (ballot64(qn >= (unsigned)kQCap) != 0) { full = true; i += kBlock; break; }
            }
          } else {
            // moving_spheres.nim:39-44: f = (time - time0) / (time1 - time0)
            const double t0 = segs[seg * 8 + 4], dt = segs[seg * 8 + 5];
            const double f = (time - t0) / dt;
            fc_f = f; fc_t0 = double_to_bits(t0); fc_dt = double_to_bits(dt);
            if (kScreen) screen_margins(scr_s1 + segs[seg * 8 + 6] + segs[seg * 8 + 7] * __builtin_fabs(f), scr_d1, a, scr_negmu, scr_am);
            const double neg_f = -f;
            if (seg_kind == 1) {
              // every sphere of the segment moves along y only (center1.x == center0.x and
              // center1.z == center0.z): c0 + f*0 == c0 exactly, so x and z need no arithmetic
              cdptr rec = movy + 6 * (long)(seg_begin + i);
              double n0 = rec[0], n1 = rec[1], n2 = rec[2], n3 = rec[3], n4 = rec[4];
              for (; i < seg_count; i += kBlock) {
                unsigned m = kWords ? mw : 0u;
                const int n_live = seg_real - i;  // (wave-uniform)
                auto test_y = [&](double c0x, double c0y, double c0z, double r2, double dcy) {
                  if (kScreen) {
                    // (centre folded into oc: o - (c0 + dc f) as fma(-f, dc, o - c0), one scalar operand per instruction)
                    m = push_bit(m, screen_filter(ox - c0x, fma_(neg_f, dcy, oy - c0y), oz - c0z, dx, dy, dz, a, scr_negmu, scr_am, r2));
                  } else {
                    const double cy = (ARITH != 1) ? c0y + dcy * f : fma_(dcy, f, c0y);
                    m = push_bit(m, disc_filter<ARITH>(ox, oy, oz, dx, dy, dz, a, c0x, cy, c0z, r2));
                  }
                };
                if (n_live >= kBlock) {
#pragma unroll
                  for (int j = 0; j < kBlock; ++j) {
                    const double c0x = n0, c0y = n1, c0z = n2, r2 = n3, dcy = n4;
                    n0 = rec[6 * (j + 1) + 0]; n1 = rec[6 * (j + 1) + 1]; n2 = rec[6 * (j + 1) + 2];
                    n3 = rec[6 * (j + 1) + 3]; n4 = rec[6 * (j + 1) + 4];
                    test_y(c0x, c0y, c0z, r2, dcy);
                  }
                } else {  // the last block: only its real objects
#pragma unroll 1
                  for (int j = 0; j < n_live; ++j) test_y(rec[6 * j + 0], rec[6 * j + 1], rec[6 * j + 2], rec[6 * j + 3], rec[6 * j + 4]);
                  m <<= (unsigned)(kBlock - n_live);
                }
                rec += 6 * kBlock;
                if constexpr (kWords) TOR_WORDS_END_BLOCK()
                else {
                q[qn * 64] = ((unsigned)(seg_block0 + i / kBlock) << 8) | m;
                qn += (m != 0) ? 1u : 0u;
                if (ballot64(qn >= (unsigned)kQCap) != 0) { full = true; i += kBlock; break; }
                }
              }
            } else {
              for (; i < seg_count; i += kBlock) {
                unsigned m = kWords ? mw : 0u;
                const int n_live = seg_real - i;  // (wave-uniform)
#pragma unroll
                for (int j = 0; j < kBlock; ++j) {
                  if (j >= n_live) { m <<= 1; continue; }  // padding of the last block
                  const int k = seg_begin + i + j;
                  const double c0x = mov[8 * k + 0], c0y = mov[8 * k + 1], c0z = mov[8 * k + 2];
                  const double dcx = mov[8 * k + 4], dcy = mov[8 * k + 5], dcz = mov[8 * k + 6];
                  if (kScreen) {
                    m = push_bit(m, screen_filter(fma_(neg_f, dcx, ox - c0x), fma_(neg_f, dcy, oy - c0y), fma_(neg_f, dcz, oz - c0z), dx, dy, dz, a,
                                                  scr_negmu, scr_am, mov[8 * k + 3]));
                  } else {
                    double cx, cy, cz;
                    if (ARITH != 1) {
                      cx = c0x + dcx * f; cy = c0y + dcy * f; cz = c0z + dcz * f;
                    } else {
                      cx = fma_(dcx, f, c0x); cy = fma_(dcy, f, c0y); cz = fma_(dcz, f, c0z);
                    }
                    m = push_bit(m, disc_filter<ARITH>(ox, oy, oz, dx, dy, dz, a, cx, cy, cz, mov[8 * k + 3]));
                  }
                }
                if constexpr (kWords) TOR_WORDS_END_BLOCK()
                else {
                q[qn * 64] = ((unsigned)(seg_block0 + i / kBlock) << 8) | m;
                qn += (m != 0) ? 1u : 0u;
                if (ballot64(qn >= (unsigned)kQCap) != 0) { full = true; i += kBlock; break; }
                }
              }
            }
          }
          if (full) break;
          seg += 1;
          i = 0;
        }
#undef TOR_WORDS_END_BLOCK
        if (kWords) {
          if (!full && (gb_next & 3) != 0) {  // the list ended inside a word: its bits move up into place
            mw <<= (unsigned)(8 * (4 - (gb_next & 3)));
            q[(unsigned)w_count * 64] = mw;
            nz = push_cond(nz, mw != 0u);
            w_count += 1;
          }
        }
        }  // if (active)

        TOR_SEC(kSecLoop)
        if constexpr (kCoop) {
          // ---- cooperative resolve (TOR_ACCEL_BLOCKS | TOR_ACCEL_F32) -------------------------------------------
          // The per-lane version below walks each lane's own boxes: 3.2 block expansions per ray on average but 8
          // for the worst lane of a wave, ~250 instructions per trip, most lanes idle.  Here the wave pools the
          // work: (A) every lane turns its queue entries into (ray, block) PAIRS and direct (ray, object) SURVIVORS
          // in two LDS lists (positions from a DPP prefix sum, the counts stay in scalar registers); (B) whenever
          // 64 pairs are pending, the 64 lanes expand one pair each -- the owner's float32 ray comes over
          // ds_bpermute -- and append the objects the filter keeps to the survivor list; (C) whenever 64 survivors
          // are pending, the 64 lanes run the reference's float64 test (spheres.nim:28-49) for one survivor each --
          // the owner's float64 ray comes over ds_bpermute -- and merge the root into the owner's closest hit in
          // LDS: atomic min on t, ties to the lowest original index (hittables_lists.nim:48-55).
          // Every trip of (B)/(C) has 64 busy lanes; lanes whose own path is dead work for the others.
          const unsigned my_qn = active ? qn : 0u;
          unsigned n_pairs = 0, n_surv = 0;  // pending list entries (wave-uniform)
          // (C) one trip: survivors [base, base + n)
          auto trip_c = [&](unsigned base, unsigned n) {
            const bool mine = (unsigned)lane < n;
            const unsigned e = mine ? coop_surv[base + (unsigned)lane] : (unsigned)lane;
            const int src = (int)(e & 63u);
            const unsigned slot = e >> 6;
            const double sox = __shfl(ox, src), soy = __shfl(oy, src), soz = __shfl(oz, src);
            const double sdx = __shfl(dx, src), sdy = __shfl(dy, src), sdz = __shfl(dz, src);
            const double sa = __shfl(a, src), stime = __shfl(time, src), sfsp = __shfl(f_sp, src);
            bool ok = false;
            double sol = 0.0;
            int orig = 0;
            if (mine) {
              const double* c = p.cold + (size_t)slot * 16;
              double cx = c[0], cy = c[1], cz = c[2];
              const int flags = (int)__double_as_longlong(c[13]);
              if (flags & 1) {
                // the owner already divided for the spatial movers' time group: same operands, same quotient
                const double f = (c[7] == p.sp_t0 && c[8] == p.sp_dt) ? sfsp : (stime - c[7]) / c[8];
                if (ARITH != 1) { cx = cx + c[3] * f; cy = cy + c[4] * f; cz = cz + c[5] * f; }
                else { cx = fma_(c[3], f, cx); cy = fma_(c[4], f, cy); cz = fma_(c[5], f, cz); }
              }
              const double ocx = sox - cx, ocy = soy - cy, ocz = soz - cz;
              double hb, cc, disc;
              if (ARITH != 1) {
                hb = ocx * sdx + ocy * sdy + ocz * sdz;
                cc = (ocx * ocx + ocy * ocy + ocz * ocz) - c[15];
                disc = hb * hb - sa * cc;
              } else {
                hb = fma_(ocz, sdz, fma_(ocy, sdy, ocx * sdx));
                cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -c[15])));
                disc = fma_(hb, hb, -(sa * cc));
              }
              if (disc > 0.0) {  // spheres.nim:35-48
                const double root = __builtin_sqrt(disc);
                sol = (-hb - root) / sa;
                ok = (0.001 < sol) && (sol < __builtin_inf());
                if (!ok) {
                  sol = (-hb + root) / sa;
                  ok = (0.001 < sol) && (sol < __builtin_inf());
                }
              }
              if (ok) orig = (int)__double_as_longlong(c[14]);
            }
            // Merge into the owner's closest hit {t, (original index, slot)}.  t > 0, so its bit pattern orders like
            // the value.  A lane that LOWERS t knows the (index, slot) word belongs to a larger t and resets it;
            // then every lane whose root equals the owner's t (ties: duplicate objects) competes with its
            // (index << 32 | slot), the lowest original index wins.
            const unsigned long long sb = (unsigned long long)__double_as_longlong(sol);
            unsigned long long old_t = 0;
            if (ok) old_t = atomicMin(&coop_t[src], sb);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const bool win = ok && coop_t[src] == sb;
            if (win && old_t > sb) coop_w[src] = ~0ull;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (win) atomicMin(&coop_w[src], ((unsigned long long)(unsigned)orig << 32) | (unsigned long long)slot);
          };
          auto drain_c = [&](bool all) {
            while (n_surv != 0 && (all || n_surv >= 64)) {
              const unsigned n = n_surv < 64 ? n_surv : 64;
              n_surv -= n;
              if (stats_on) atomicAdd(&prof_lds[kStCand], lane == 0 ? (unsigned long long)n : 0ull);
              if (prof && lane == 0) prof_lds[kSecTrips] += 1;
              trip_c(n_surv, n);
            }
          };
          // appends cnt entries per lane (cnt <= 8) to a list: returns this lane's first position, bumps the count
          auto reserve = [&](unsigned cnt, unsigned& total) {
            unsigned incl, sum;
            wave_scan_u32(cnt, incl, sum);
            const unsigned at = total + incl - cnt;
            total += sum;
            return at;
          };
          // (B) one trip: pairs [base, base + n)
          auto trip_b = [&](unsigned base, unsigned n) {
            if constexpr (BLOCKS != 0) {
            const bool mine = (unsigned)lane < n;
            const unsigned e = mine ? coop_pair[base + (unsigned)lane] : (unsigned)lane;
            const int src = (int)(e & 63u);
            const unsigned blk_id = e >> 6;
            RayF32 r;
            r.ox = __shfl(r32.ox, src); r.oy = __shfl(r32.oy, src); r.oz = __shfl(r32.oz, src);
            r.dx = __shfl(r32.dx, src); r.dy = __shfl(r32.dy, src); r.dz = __shfl(r32.dz, src);
            r.gma = __shfl(r32.gma, src); r.g = __shfl(r32.g, src);
            r.ro = 0.f; r.ro2 = 0.f; r.sa = 0.f; r.wild = 0u;
            SegF32 s;
            s.nmbl = splat2(__shfl(sp32.nmbl.x, src)); s.gk = splat2(__shfl(sp32.gk.x, src)); s.nf = splat2(__shfl(sp32.nf.x, src));
            s.wild = (unsigned)__shfl((int)sp32.wild, src);
            unsigned m8 = 0;
            if (mine) {
              auto filter_block = [&](auto blk, auto ST) {
                constexpr int st = decltype(ST)::value;
                // two pairs at a time: with all four in flight the 168-register variants spill inside the bounce loop
                // (and every spill store shows up as HBM write traffic: 20 GB per C2 frame, measured)
#pragma unroll 2
                for (int j = 0; j < kBlock / 2; ++j) {
                  auto rr = blk + st * j;
                  const f2v c0x = {rr[0], rr[1]}, c0y = {rr[2], rr[3]}, c0z = {rr[4], rr[5]};
                  f2v ocx, ocy, ocz;
                  if (st == 16) {
                    ocx = oc_moving32(r.ox, c0x, (f2v){rr[10], rr[11]}, s.nf);
                    ocy = oc_moving32(r.oy, c0y, (f2v){rr[12], rr[13]}, s.nf);
                    ocz = oc_moving32(r.oz, c0z, (f2v){rr[14], rr[15]}, s.nf);
                  } else {
                    ocx = oc_static32(r.ox, c0x);
                    ocy = (st == 12) ? oc_moving32(r.oy, c0y, (f2v){rr[10], rr[11]}, s.nf) : oc_static32(r.oy, c0y);
                    ocz = oc_static32(r.oz, c0z);
                  }
                  m8 = filter_pair32(r, s, ocx, ocy, ocz, (f2v){rr[6], rr[7]}, (f2v){rr[8], rr[9]}, m8);
                }
              };
              using T10 = std::integral_constant<int, 10>;
              using T12 = std::integral_constant<int, 12>;
              using T16 = std::integral_constant<int, 16>;
              const size_t off = (size_t)blk_id * (size_t)p.shot32_block_stride;
              if (p.shot32_lds_floats > 0) {
                if (p.shot32_stride == 12) filter_block(shot32_lds + off, T12{});
                else if (p.shot32_stride == 10) filter_block(shot32_lds + off, T10{});
                else filter_block(shot32_lds + off, T16{});
              } else {
                const gfptr g32 = (gfptr)(uintptr_t)p.shot32;
                if (p.shot32_stride == 12) filter_block(g32 + off, T12{});
                else if (p.shot32_stride == 10) filter_block(g32 + off, T10{});
                else filter_block(g32 + off, T16{});
              }
              m8 |= s.wild;
            }
            // append the kept objects to the survivor list
            unsigned at = reserve((unsigned)__builtin_popcount(m8), n_surv);
            while (m8 != 0) {
              const int bb = 31 - __builtin_clz(m8);
              m8 &= ~(1u << bb);
              coop_surv[at++] = (unsigned)src | (((unsigned)p.spatial_base + blk_id * kBlock + (unsigned)(7 - bb)) << 6);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
          };
          auto drain_b = [&](bool all) {
            if constexpr (BLOCKS != 0)
            while (n_pairs != 0 && (all || n_pairs >= 64)) {
              const unsigned n = n_pairs < 64 ? n_pairs : 64;
              n_pairs -= n;
              if (stats_on) atomicAdd(&prof_lds[kStCand], lane == 0 ? (unsigned long long)n * kBlock : 0ull);
              if (prof && lane == 0) prof_lds[kSecTrips] += 1;
              trip_b(n_pairs, n);
              drain_c(false);
            }
          };
          // (A) queue entries -> pairs / direct survivors.  One 'unit' per lane and trip: a whole entry of object
          // bits or block-box bits, or ONE bit of a super-box entry (its 8 child boxes are slab-tested here) --
          // at most 8 list entries per lane and trip, so the lists (< 64 carried over + 512) cannot overflow.
          unsigned kq = 0, sup_mask = 0, sup_block = 0;
          if constexpr (kSuper) {
            // Two-level scenes (more than 96 blocks): the slab loop left one bit per SUPER box (8 blocks) in box_mask, and
            // the 8 child boxes of every super box a ray entered still have to be slab-tested for that ray.  Pooled like
            // (B) and (C): every round packs up to 64 (ray, super box) units -- a prefix sum over the lanes' bit counts,
            // the first lanes in scan order give as many as fit -- into the free space above the pending pairs, the 64
            // lanes expand one unit each (the owner's box ray over ds_bpermute, the child boxes from LDS) and append
            // the (ray, block) pairs: at most 512 on top of fewer than 64 carried over, the list holds 576.  Rounds =
            // all the wave's units / 64 instead of the worst lane's count (measured on the 1601-object animation
            // frames: 8+ per-lane trips of ~300 instructions before).
            // (box_kind is set under `if (active)`: the test has to be made wave-uniform by hand, the block below is full
            // of cross-lane operations)
            const bool sup_lane = active && box_kind == 4;
            if (ballot64(sup_lane) != 0) {
              unsigned long long sm = sup_lane ? box_mask : 0ull;
              if (sup_lane) box_mask = 0;
              while (ballot64(sm != 0) != 0) {
                const unsigned c = (unsigned)__builtin_popcountll(sm);
                unsigned incl, total;
                wave_scan_u32(c, incl, total);
                const unsigned excl = incl - c;
                unsigned take = excl >= 64u ? 0u : (c < 64u - excl ? c : 64u - excl);
                const unsigned n_round = total < 64u ? total : 64u;
                unsigned pos = n_pairs + excl;
                for (; take != 0; --take) {
                  const int b = __builtin_clzll(sm);  // super box b of the segment
                  sm &= ~(0x8000000000000000ull >> b);
                  coop_pair[pos++] = (unsigned)lane | ((box_group0 * kBlock + (unsigned)b) << 6);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                {
                  const bool mine = (unsigned)lane < n_round;
                  const unsigned e = mine ? coop_pair[n_pairs + (unsigned)lane] : (unsigned)lane;
                  const int src = (int)(e & 63u);
                  const unsigned rec = e >> 6;  // super box `rec`: its block boxes are rec*8 .. rec*8+7
                  BoxRay32 ob;
                  ob.ax = (f2v){__shfl(b32.ax.x, src), __shfl(b32.ax.y, src)};
                  ob.ay = (f2v){__shfl(b32.ay.x, src), __shfl(b32.ay.y, src)};
                  ob.az = (f2v){__shfl(b32.az.x, src), __shfl(b32.az.y, src)};
                  ob.ix = splat2(__shfl(b32.ix.x, src)); ob.iy = splat2(__shfl(b32.iy.x, src)); ob.iz = splat2(__shfl(b32.iz.x, src));
                  const unsigned owild = (unsigned)__shfl((int)r32.wild, src);
                  unsigned mc = 0;
                  if (mine) {
                    auto child32 = [&](auto cb) {
#pragma unroll 4
                      for (int j = 0; j < kBlock; ++j)
                        mc = push_cond(mc, slab_bit32(ob, (f2v){cb[8 * j + 0], cb[8 * j + 1]}, (f2v){cb[8 * j + 2], cb[8 * j + 3]},
                                                    (f2v){cb[8 * j + 4], cb[8 * j + 5]}) != 0u);
                    };
                    if (p.bnd32_lds_floats > 0) child32(bnd32_lds + (size_t)rec * (8 * kBlock));
                    else child32((gfptr)(uintptr_t)p.bnd32 + (size_t)rec * (8 * kBlock));
                    mc |= owild;
                  }
                  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // every lane has read its unit: the space is free again
                  if (prof && lane == 0) prof_lds[kSecTrips] += 1;
                  unsigned at = reserve((unsigned)__builtin_popcount(mc), n_pairs);
                  while (mc != 0) {
                    const int bb = 31 - __builtin_clz(mc);
                    mc &= ~(1u << bb);
                    coop_pair[at++] = (unsigned)src | ((rec * kBlock + (unsigned)(7 - bb)) << 6);
                  }
                  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
                drain_b(false);
                drain_c(false);
              }
            }
          }
          {
            // Fast path (no super-box entries in the wave, everything fits the lists): count the bits of all
            // entries, ONE prefix sum for both lists (the two counts share a word), then write.
            if (!active) box_mask = 0;
            bool has_super = kSuper && box_kind == 4 && box_mask != 0;
            unsigned cnt2 = has_super ? 0u : (unsigned)__builtin_popcountll(box_mask);  // pairs | direct survivors << 16
            for (unsigned k = 0; k < my_qn; ++k) {
              const unsigned e = q[k * 64];
              const unsigned k3 = e >> 30;
              const unsigned bits = (unsigned)__builtin_popcount(e & 0xffu);
              cnt2 += (k3 == 2) ? bits : ((k3 == 0) ? (bits << 16) : 0u);
              has_super = has_super || (kSuper && k3 == 1);
            }
            unsigned incl, total;
            wave_scan_u32(cnt2, incl, total);
            const unsigned tot_p = total & 0xffffu, tot_s = total >> 16;
            if (ballot64(has_super) == 0 && n_pairs + tot_p <= (unsigned)kCoopList && n_surv + tot_s <= (unsigned)kCoopList) {
              unsigned at_p = n_pairs + (incl & 0xffffu) - (cnt2 & 0xffffu);
              unsigned at_s = n_surv + (incl >> 16) - (cnt2 >> 16);
              n_pairs += tot_p;
              n_surv += tot_s;
              while (box_mask != 0) {
                const int b = __builtin_clzll(box_mask);  // box b of the segment
                box_mask &= ~(0x8000000000000000ull >> b);
                coop_pair[at_p++] = (unsigned)lane | ((box_group0 * kBlock + (unsigned)b) << 6);
              }
              for (; kq < my_qn; ++kq) {
                const unsigned e = q[kq * 64];
                const unsigned blk = (e >> 8) & 0x3fffffu;
                unsigned m = e & 0xffu;
                if ((e >> 30) == 2) {
                  while (m != 0) {
                    const int b = 31 - __builtin_clz(m);
                    m &= ~(1u << b);
                    coop_pair[at_p++] = (unsigned)lane | ((blk * kBlock + (unsigned)(7 - b)) << 6);
                  }
                } else {
                  while (m != 0) {
                    const int b = 31 - __builtin_clz(m);
                    m &= ~(1u << b);
                    coop_surv[at_s++] = (unsigned)lane | ((blk * kBlock + (unsigned)(7 - b)) << 6);
                  }
                }
              }
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
          }
          for (;;) {
            unsigned kind = 3, m = 0, blk = 0;  // 3: nothing this trip
            if (kSuper && sup_mask != 0) {
              const int b = 31 - __builtin_clz(sup_mask);
              sup_mask &= ~(1u << b);
              const unsigned rec = sup_block * kBlock + (unsigned)(7 - b);  // super box `rec`: its block boxes are rec*8 .. rec*8+7
              unsigned mc = 0;
              auto child32 = [&](auto cb) {
#pragma unroll 4
                for (int j = 0; j < kBlock; ++j)
                  mc = push_cond(mc, slab_bit32(b32, (f2v){cb[8 * j + 0], cb[8 * j + 1]}, (f2v){cb[8 * j + 2], cb[8 * j + 3]},
                                              (f2v){cb[8 * j + 4], cb[8 * j + 5]}) != 0u);
              };
              if (p.bnd32_lds_floats > 0) child32(bnd32_lds + (size_t)rec * (8 * kBlock));
              else child32((gfptr)(uintptr_t)p.bnd32 + (size_t)rec * (8 * kBlock));
              mc |= r32.wild;
              kind = 2; m = mc; blk = rec;
            } else if (box_mask != 0) {
              const int g = __builtin_clzll(box_mask) >> 3;  // group of 8 boxes
              const unsigned mm = (unsigned)(box_mask >> (56 - 8 * g)) & 0xffu;
              box_mask &= ~(0xffull << (56 - 8 * g));
              if (kSuper && box_kind == 4) { sup_mask = mm; sup_block = box_group0 + (unsigned)g; kind = 4; }
              else { kind = 2; m = mm; blk = box_group0 + (unsigned)g; }
            } else if (kq < my_qn) {
              const unsigned e = q[kq * 64];
              kq += 1;
              const unsigned k3 = e >> 30;  // 0: object mask, 2: block boxes, 1: super boxes
              if (kSuper && k3 == 1) { sup_mask = e & 0xffu; sup_block = (e >> 8) & 0x3fffffu; kind = 4; }  // expanded bit by bit from the next trip on
              else { kind = k3; m = e & 0xffu; blk = (e >> 8) & 0x3fffffu; }
            }
            if (ballot64(kind != 3) == 0) break;
            const unsigned n_bits = (unsigned)__builtin_popcount(m);
            unsigned at_p = reserve(kind == 2 ? n_bits : 0u, n_pairs);
            unsigned at_s = reserve(kind == 0 ? n_bits : 0u, n_surv);
            if (kind != 0 && kind != 2) m = 0;
            while (m != 0) {
              const int b = 31 - __builtin_clz(m);
              m &= ~(1u << b);
              const unsigned id = blk * kBlock + (unsigned)(7 - b);  // block id, or cold slot of a direct candidate
              if (BLOCKS != 0 && kind == 2) coop_pair[at_p++] = (unsigned)lane | (id << 6);
              else coop_surv[at_s++] = (unsigned)lane | (id << 6);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            drain_b(false);
            drain_c(false);
          }
          drain_b(true);
          drain_c(true);
        } else if (active) {
        // ---- resolve the queued candidates exactly as spheres.nim:35-48 does -------------
        // Queue entries are 8-bit masks over 8 consecutive cold slots (direct candidates) or, flagged
        // with bit 31, over 8 block bounds.  A trip of the loop handles one set bit per lane: either
        // one object, or one spatial block of 8 objects (an unrolled discriminant/filter stage over
        // the block's compact records, then exact roots for the few that pass).
        int f_gid = -1;      // time group whose fraction f_val = (time - time0)/(time1 - time0) is cached
        double f_val = 0.0;  // (moving_spheres.nim:42; identical for every member of a time group)
        // exact part of hit(): spheres.nim:35-48 with t_min = 0.001 and the order-independent update
        auto exact_hit = [&](double cx, double cy, double cz, double r2, unsigned idx, double f) {
          const double ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
          double hb, cc, disc;
          if (ARITH != 1) {
            hb = ocx * dx + ocy * dy + ocz * dz;
            cc = (ocx * ocx + ocy * ocy + ocz * ocz) - r2;
            disc = hb * hb - a * cc;
          } else {
            hb = fma_(ocz, dz, fma_(ocy, dy, ocx * dx));
            cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -r2)));
            disc = fma_(hb, hb, -(a * cc));
          }
          if (disc > 0.0) {
            const double root = __builtin_sqrt(disc);
            double sol = (-hb - root) / a;
            bool ok = (0.001 < sol) && (sol < __builtin_inf());
            if (!ok) {
              sol = (-hb + root) / a;
              ok = (0.001 < sol) && (sol < __builtin_inf());
            }
            if (ok) {
              const int orig = (int)__double_as_longlong(p.cold[(size_t)idx * 16 + 14]);
              if (sol < best_t || (sol == best_t && orig < best_orig)) {
                best_t = sol;
                best_idx = (int)idx;
                best_orig = orig;
                best_f = f;
              }
            }
          }
        };
        // centre of a spatial object from its compact record {c0 xyz, r^2, dc xyz, group id | -1}
        // HS = float64 per record: 8, or 4 for an all-static set (no dc / group fields)
        auto spatial_center = [&](auto hrec, auto HS, double& cx, double& cy, double& cz, double& f) {
          constexpr int hs = decltype(HS)::value;
          cx = hrec[0]; cy = hrec[1]; cz = hrec[2];
          f = 0.0;
          const int gid = (hs == 8) ? (int)hrec[7] : -1;
          if (gid >= 0) {
            if (gid != f_gid) {
              f_val = (time - p.sgrp[2 * gid]) / p.sgrp[2 * gid + 1];
              f_gid = gid;
            }
            f = f_val;
            if (ARITH != 1) {
              cx = cx + hrec[4] * f; cy = cy + hrec[5] * f; cz = cz + hrec[6] * f;
            } else {
              cx = fma_(hrec[4], f, cx); cy = fma_(hrec[5], f, cy); cz = fma_(hrec[6], f, cz);
            }
          }
        };
        // exact hit of the object in cold slot `slot` (centre from the cold record)
        auto exact_cold = [&](unsigned slot) {
          const double* c = p.cold + (size_t)slot * 16;
          double cx = c[0], cy = c[1], cz = c[2], f = 0.0;
          const int flags = (int)__double_as_longlong(c[13]);
          if (flags & 1) {
            const double ct0 = c[7], cdt = c[8];
            if (double_to_bits(ct0) == fc_t0 && double_to_bits(cdt) == fc_dt) f = fc_f;
            else f = (time - ct0) / cdt;
            if (ARITH != 1) {
              cx = cx + c[3] * f; cy = cy + c[4] * f; cz = cz + c[5] * f;
            } else {
              cx = fma_(c[3], f, cx); cy = fma_(c[4], f, cy); cz = fma_(c[5], f, cz);
            }
          }
          exact_hit(cx, cy, cz, c[15], slot, f);
        };
        if constexpr (kWords) {
          // words: bit (31 - j) of word k <-> cold slot (w_base + k) * 32 + j
          unsigned cur_mask = 0, cur_base = 0, nzr = nz;
          const int n_words = __builtin_amdgcn_readfirstlane(w_count), word0 = __builtin_amdgcn_readfirstlane(w_base);
          for (;;) {
            if (cur_mask == 0 && nzr != 0) {
              const int b = 31 - __builtin_clz(nzr);
              nzr &= ~(1u << b);
              const unsigned k = (unsigned)(n_words - 1 - b);
              cur_mask = q[k * 64];
              cur_base = ((unsigned)word0 + k) * 32u;
            }
            const bool has = cur_mask != 0;
            if (ballot64(has) == 0) break;
            if (prof && lane == 0) prof_lds[kSecTrips] += 1;
            if (has) {
              const int b = 31 - __builtin_clz(cur_mask);
              cur_mask &= ~(1u << b);
              if (stats_on) atomicAdd(&prof_lds[kStCand], 1ull);
              exact_cold(cur_base + (unsigned)(31 - b));
            }
          }
        } else {
        unsigned kq = 0, cur_mask = 0, cur_block = 0, cur_is_bound = 0;
        for (;;) {
          if (cur_mask == 0 && kq < qn) {
            const unsigned e = q[kq * 64];
            kq += 1;
            cur_is_bound = e >> 30;  // 0: object mask, 2: block boxes, 1: super boxes
            cur_block = (e >> 8) & 0x3fffffu;
            cur_mask = e & 0xffu;
          }
          const bool has = cur_mask != 0;
          if (ballot64(has) == 0) break;
          if (prof && lane == 0) prof_lds[kSecTrips] += 1;
          if (has) {
            const int b = 31 - __builtin_clz(cur_mask);  // bit (7 - j) <-> record j of the group
            cur_mask &= ~(1u << b);
            const unsigned rec = cur_block * kBlock + (unsigned)(7 - b);
            if (BLOCKS && cur_is_bound) {
              // ---- spatial block `blk_id`: filter its 8 objects, then exact roots for the survivors
              auto expand = [&](auto blk, auto HS, unsigned blk_id) {
                constexpr int hs = decltype(HS)::value;
                unsigned m8 = 0;
                auto test = [&](int j) {
                  double cx, cy, cz, f;
                  spatial_center(blk + hs * j, HS, cx, cy, cz, f);
                  m8 = push_bit(m8, disc_filter<ARITH>(ox, oy, oz, dx, dy, dz, a, cx, cy, cz, blk[hs * j + 3]));
                };
                if constexpr (hs == 4) {  // 32-byte static records: all eight in flight
#pragma unroll
                  for (int j = 0; j < kBlock; ++j) test(j);
                } else {  // 64-byte records: four at a time (eight make the 168-register variants spill inside the bounce loop)
#pragma unroll 4
                  for (int j = 0; j < kBlock; ++j) test(j);
                }
                while (m8 != 0) {
                  const int bb = 31 - __builtin_clz(m8);
                  m8 &= ~(1u << bb);
                  const int j = 7 - bb;
                  double cx, cy, cz, f;
                  spatial_center(blk + hs * j, HS, cx, cy, cz, f);
                  exact_hit(cx, cy, cz, blk[hs * j + 3], (unsigned)p.spatial_base + blk_id * kBlock + (unsigned)j, f);
                }
              };
              using S8 = std::integral_constant<int, 8>;
              using S4 = std::integral_constant<int, 4>;
              // (float32 block records belong to the cooperative resolve above; this per-lane path serves F32 = 0)
              auto expand_block = [&](unsigned blk_id) {
                if (stats_on) atomicAdd(&prof_lds[kStCand], (unsigned long long)kBlock);
                if (p.shot_stride == 8) {  // float64 compact records
                  if (staged) expand(shot_lds + (size_t)blk_id * (8 * kBlock), S8{}, blk_id);            // ds_read
                  else expand((gdptr)(uintptr_t)p.shot + (size_t)blk_id * (8 * kBlock), S8{}, blk_id);  // global_load
                } else {
                  if (staged) expand(shot_lds + (size_t)blk_id * (4 * kBlock), S4{}, blk_id);
                  else expand((gdptr)(uintptr_t)p.shot + (size_t)blk_id * (4 * kBlock), S4{}, blk_id);
                }
              };
              if (cur_is_bound == 2) {
                expand_block(rec);
              } else {
                // super box `rec`: slab-test its 8 block boxes, descend into the ones the ray can touch
                unsigned mc = 0;
                {
                const gdptr cb = (gdptr)(uintptr_t)p.bnd + (size_t)rec * (8 * kBlock);
                const double ix = 1.0 / dx, iy = 1.0 / dy, iz = 1.0 / dz;
#pragma unroll
                for (int j = 0; j < kBlock; ++j) {
                  const double tx0 = (cb[8 * j + 0] - ox) * ix, tx1 = (cb[8 * j + 3] - ox) * ix;
                  const double ty0 = (cb[8 * j + 1] - oy) * iy, ty1 = (cb[8 * j + 4] - oy) * iy;
                  const double tz0 = (cb[8 * j + 2] - oz) * iz, tz1 = (cb[8 * j + 5] - oz) * iz;
                  const double t_in = __builtin_fmax(__builtin_fmax(__builtin_fmin(tx0, tx1), __builtin_fmin(ty0, ty1)),
                                                     __builtin_fmax(__builtin_fmin(tz0, tz1), 0.0));
                  const double t_out = __builtin_fmin(__builtin_fmin(__builtin_fmax(tx0, tx1), __builtin_fmax(ty0, ty1)),
                                                      __builtin_fmax(tz0, tz1));
                  mc = (mc << 1) | ((t_in <= t_out) ? 1u : 0u);
                }
                }
                while (mc != 0) {
                  const int cbit = 31 - __builtin_clz(mc);
                  mc &= ~(1u << cbit);
                  expand_block(rec * kBlock + (unsigned)(7 - cbit));
                }
              }
            } else {
              if (stats_on) atomicAdd(&prof_lds[kStCand], 1ull);
              exact_cold(rec);
            }
          }
        }
        }  // (queue entries)
        }  // per-lane resolve
        TOR_SEC(kSecResolve)
        if (ballot64(active && full) == 0) break;
        if (kWords) w_base += kQCap;
      }
      if (kCoop) {  // the closest hit the wave found for this lane's ray
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        best_t = __longlong_as_double((long long)coop_t[lane]);
        best_idx = (best_t < __builtin_inf()) ? (int)(unsigned)coop_w[lane] : -1;
        if (best_idx >= 0) {  // the time fraction of a moving object, as moving_spheres.nim:42 computes it
          const double* c = p.cold + (size_t)best_idx * 16;
          if ((int)__double_as_longlong(c[13]) & 1) best_f = (c[7] == p.sp_t0 && c[8] == p.sp_dt) ? f_sp : (time - c[7]) / c[8];
        }
      }
      if (active) {

      // ================= (C) shade ====================================================
      // unit_vector(direction) is needed by the sky (render.nim:42), Metal (materials.nim:40) and Dielectric
      // (materials.nim:68): the wave would walk through three copies of the square root and the division -- one
      // copy, in front of the branches, serves all of them (same operations on the same ray: same bits)
      const V3 ud_ray = unit_vector(d);
      if (best_idx < 0) {
        radiance = sky_unit(ud_ray, get_att());  // render.nim:41-45
        ended = true;
      } else {
        const double* c = p.cold + (size_t)best_idx * 16;
        const int flags = (int)__double_as_longlong(c[13]);
        V3 center = v3(c[0], c[1], c[2]);
        if (flags & 1) {
          if (ARITH != 1) center = center + v3(c[3], c[4], c[5]) * best_f;
          else center = v3(fma_(c[3], best_f, c[0]), fma_(c[4], best_f, c[1]), fma_(c[5], best_f, c[2]));
        }
        const V3 hp = o + d * best_t;                       // rays.nim:24-25
        const V3 outward = (hp - center) * c[6];            // spheres.nim:43 (c[6] = 1.0/radius)
        const bool front = dot(d, outward) < 0.0;           // core.nim:47-49
        const V3 n = front ? outward : -outward;
        const int mat = (flags >> 8) & 0xff;
        const V3 albedo = v3(c[9], c[10], c[11]);
        if (mat == kLambertian) {  // materials.nim:24-30
          d = n + random_unit_vector(rng);
          o = hp;
          set_att(mul_att(get_att(), albedo));  // render.nim:35
        } else if (mat == kMetal) {  // materials.nim:39-47
          const V3 reflected = reflect(ud_ray, n);
          const V3 nd = reflected + random_in_unit_sphere(rng) * c[12];
          o = hp;
          d = nd;
          time = 0.0;  // rays.nim:19 default
          if (dot(nd, n) > 0.0) {
            set_att(mul_att(get_att(), albedo));
          } else {
            ended = true;  // render.nim:38: absorbed -> black
          }
        } else {  // materials.nim:62-86
          const double eta = front ? c[9] : c[12];  // 1.0 / ri : ri  (the quotient comes from the host, tor_scene.cpp)
          const V3 ud = ud_ray;
          const double dn = dot(-ud, n);
          const double cos_theta = (dn <= 1.0) ? dn : 1.0;
          const double sin_theta = __builtin_sqrt(1.0 - cos_theta * cos_theta);
          V3 nd;
          if (eta * sin_theta > 1.0) {
            nd = reflect(ud, n);
          } else {
            const double reflect_prob = schlick_r0(cos_theta, front ? c[10] : c[11]);
            if (uniform01(rng) < reflect_prob) nd = reflect(ud, n);
            else nd = refract(ud, n, eta);
          }
          o = hp;
          d = nd;
          time = 0.0;
          // (materials.nim:63 attenuates by (1, 1, 1): x * 1.0 == x for every float64, so the product is not formed)
        }
        if (!ended) {
          depth += 1;
          if (depth >= p.max_depth) ended = true;  // render.nim:25,47: loop exhausted -> black
        }
      }

      if (ended) {
        active = false;
        if (stats_on) atomicAdd(&prof_lds[kStSamples], 1ull);
        if (kProbe) {
          atomicAdd(p.pixel_cost + (unsigned)pix, (unsigned)path_q);
          path_q = 0;
        }
        if (SEEDING == 0) {
          if (kAccInLds) acc = v3(pix_acc[0], pix_acc[64], pix_acc[128]);
          acc = acc + radiance;  // render.nim:67
          s += 1;
          if (s >= p.spp) {
            double* out = p.out + (size_t)pix * 3;
            out[0] = acc.x; out[1] = acc.y; out[2] = acc.z;
          } else {
            if (kAccInLds) { pix_acc[0] = acc.x; pix_acc[64] = acc.y; pix_acc[128] = acc.z; }
            have_item = true;  // next sample of the same pixel, same stream
          }
        }
      }
      }  // if (active)
      if constexpr (kMigrate) {
        // ---- hand the chain over at this sample boundary? ----
        // (nothing of this is kept in registers across the bounce iteration -- the variant has none to spare: the control
        // words are re-read through the scalar cache where they are needed)
        unsigned long long* mq = p.mig;
        asm volatile("" : "+s"(mq));
        if (mq != nullptr) {
          const bool boundary = ended && have_item;  // a sample just ended and the pixel has more
          // the adaptive threshold: a per-wave copy in LDS (the debug-counter slots: statistics and hand-off exclude each
          // other), refreshed from memory every 16th bounce -- thousands of waves reading one line every bounce is traffic the
          // waiting servers' polls already compete with
          if (lane == 0) {
            const unsigned long long n = prof_lds[1];
            prof_lds[1] = n + 1;
            if ((n & 15ull) == 0) prof_lds[0] = __hip_atomic_load(mq + kMigPushNow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          const unsigned long long push_at = prof_lds[0];
          const bool hot = boundary && s >= 8 && pix_iters >= 64u &&
                           (unsigned long long)pix_iters * (unsigned)p.spp >= push_at * (unsigned)s;
          bool tail_push = boundary && exhausted && !hot && p.mig_tail_lanes >= 0;
          if (exhausted && p.mig_tail_lanes >= 0) {
            const unsigned live = (unsigned)__builtin_popcountll(ballot64(active || have_item));
            const unsigned long long rest = (unsigned long long)(unsigned)(p.spp - s) * pix_iters / (unsigned)(s > 0 ? s : 1);
            if (live <= (unsigned)p.mig_tail_lanes) {
              // the wave's last lanes: hand over what has a way to go; a short rest is finished here (a server bounce costs ~10
              // lane bounces, and in the tail of a frame the servers are what is scarce)
              tail_push = tail_push && (rest * 4 >= (unsigned long long)p.mig_tail_rest || live <= 1u);
            } else {
              // more live lanes than the tail threshold: only as many chains as servers are waiting for one right now
              unsigned long long hd = 0, tl = 0;
              if (lane == 0) {
                hd = __hip_atomic_load(mq + kMigHead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                tl = __hip_atomic_load(mq + kMigTail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
              hd = bcast_first_u64(hd);
              tl = bcast_first_u64(tl);
              const unsigned idle = hd > tl ? (unsigned)((hd - tl) > 64ull ? 64ull : (hd - tl)) : 0u;  // tickets beyond the last record = waiting servers
              // ... and only chains with a long way to go
              tail_push = tail_push && rest >= (unsigned long long)p.mig_tail_rest;
              const unsigned long long tm = ballot64(tail_push);
              tail_push = tail_push && lane_prefix(tm) < idle;
            }
          }
          if (hot || tail_push) {
            const unsigned long long at = atomicAdd(mq + kMigTail, 1ull);
            if (at < (unsigned long long)p.mig_cap) {
              unsigned long long* r = p.mig_rec + at * 8;
              r[0] = (unsigned long long)(unsigned)pix | ((unsigned long long)(unsigned)s << 32) | (hot ? 1ull << 63 : 0ull);
              r[1] = rng.s0; r[2] = rng.s1; r[3] = rng.s2; r[4] = rng.s3;
              // (the running sum comes back from LDS, where the sample boundary above left it: kept in registers across this
              // block it was spilled -- one 16-byte scratch store per finished sample, 24 GB of HBM writes per configs[2] frame)
              if (kAccInLds) { r[5] = double_to_bits(pix_acc[0]); r[6] = double_to_bits(pix_acc[64]); r[7] = double_to_bits(pix_acc[128]); }
              else { r[5] = double_to_bits(acc.x); r[6] = double_to_bits(acc.y); r[7] = double_to_bits(acc.z); }
              __hip_atomic_store(p.mig_flag + at, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
              have_item = false;  // the lane is free again
              atomicAdd(mq + (hot ? kMigHotPushes : kMigTailPushes), 1ull);
              if (hot && (p.mig_flags & 2u)) {
                // adaptive threshold: a hot chain that finds others waiting in front of it raises the bar by 1/16
                const unsigned long long hd = __hip_atomic_load(mq + kMigHead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (at >= hd) {
                  const unsigned long long now = __hip_atomic_load(mq + kMigPushNow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  __hip_atomic_store(mq + kMigPushNow, now + (now >> 4) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
              }
            }
          }
        }
      }
    }

    TOR_SEC(kSecShade)
    if (SEEDING == 1) {
      // ---- deposit finished samples: exact (2^-36-quantised) float64 sums, any order ------
      // The wave keeps the pixels it is currently filling in a small LDS cache (pixels arrive
      // in increasing order, so a slot is evicted when the pixel is complete bar stragglers);
      // HBM only sees one flush per pixel and wave instead of one atomic per sample.
      unsigned long long ended_mask = ballot64(ended);
      const double qx = quantize36(radiance.x), qy = quantize36(radiance.y), qz = quantize36(radiance.z);
      while (ended_mask != 0) {
        const int src = (int)__builtin_ctzll(ended_mask);
        const int pp = __builtin_amdgcn_readlane(pix, src);
        const bool mine = ended && pix == pp;
        // (hashed: a wave's successive pixels are a multiple of the wave count apart -- 3072 on MI355X -- and would
        // all land in slot pp & 15, so every late sample of the previous pixel evicted the current one)
        const int slot = (pp ^ (pp >> 4) ^ (pp >> 9) ^ (pp >> 14)) & (kAccSlots - 1);
        const int tag = __builtin_amdgcn_readfirstlane(tag_lds[slot]);
        if (tag != pp) {
          if (lane < 3) {
            if (tag >= 0) unsafeAtomicAdd(p.out + (size_t)tag * 3 + lane, acc_lds[slot * 3 + lane]);
            acc_lds[slot * 3 + lane] = 0.0;
          }
          if (lane == 0) tag_lds[slot] = pp;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
        if (mine) {
          unsafeAtomicAdd(&acc_lds[slot * 3 + 0], qx);
          unsafeAtomicAdd(&acc_lds[slot * 3 + 1], qy);
          unsafeAtomicAdd(&acc_lds[slot * 3 + 2], qz);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        ended_mask &= ~ballot64(mine);
      }
    }
    TOR_SEC(kSecDeposit)
  }
#undef TOR_SEC

  if constexpr (kMigrate) {
    if (mig_on && ran_lanes) {
      // every push of this wave happens-before the decrement: a server that reads 0 here has seen every record
      if (lane == 0) {
        atomicMax(p.mig + kMigTLaneEnd, (unsigned long long)wall_clock64());
        __hip_atomic_fetch_add(p.mig + kMigLaneWaves, ~0ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (kPrio && prio_now != 0) __builtin_amdgcn_s_setprio(0);
      // (TOR_TAIL_LANES=-2: debugging, nobody serves -- only valid when nobody pushes)
      if (p.mig_tail_lanes > -2) (void)serve_chains<ARITH>(p, false);
    }
  }

  if (SEEDING == 1) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < kAccSlots * 3) {
      const int slot = lane / 3, ch = lane - slot * 3;
      const int tag = tag_lds[slot];
      if (tag >= 0) unsafeAtomicAdd(p.out + (size_t)tag * 3 + ch, acc_lds[slot * 3 + ch]);
    }
  }
  if (stats_on) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane == 0 && p.wave_log != nullptr) {
      unsigned long long* w = p.wave_log + (size_t)(blockIdx.x * (kThreads / 64) + wave) * 8;
      w[0] = prof_lds[kLogStart]; w[1] = wall_clock64(); w[2] = prof_lds[kStIters];
      w[3] = prof_lds[kStQueries] | ((unsigned long long)__builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4) << 44);  // HW_ID[15:0]
      w[4] = prof_lds[kLogExhausted]; w[5] = (prof_lds[kLogItersAtExhaustion] & 0xffffffffull) | (prof_lds[kSecTrips] << 32);
      // six 21-bit fields in units of 4096 shader cycles: refill+camera, object loop, exact resolve | shade, deposit, total
      const unsigned long long total = __builtin_readcyclecounter() - prof_lds[kSecBegin];
      auto f21 = [](unsigned long long c) { c >>= 12; return c > 0x1fffffull ? 0x1fffffull : c; };
      w[6] = f21(prof_lds[kSecRefill]) | (f21(prof_lds[kSecLoop]) << 21) | (f21(prof_lds[kSecResolve]) << 42);
      w[7] = f21(prof_lds[kSecShade]) | (f21(prof_lds[kSecDeposit]) << 21) | (f21(total) << 42);
    }
    if (lane == 0) {
      atomicAdd(p.stats + 0, prof_lds[kStQueries]);
      atomicAdd(p.stats + 1, prof_lds[kStCand]);
      atomicAdd(p.stats + 2, prof_lds[kStIters]);
      atomicAdd(p.stats + 3, prof_lds[kStSamples]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// coop_pixel_kernel -- TOR_SEED_PIXEL on SMALL frames: one WAVE per pixel.
//
// render.nim:59-67 makes a pixel a sequential chain: its spp samples share one RNG stream, so sample s+1 cannot
// start before sample s has consumed its (data dependent) number of draws.  integrate_kernel gives a lane to each
// pixel; on a small frame (C1: 82 944 pixels) the machine is then mostly idle while every wave waits for its most
// expensive lane -- the frame takes max-chain-length x the latency of one bounce iteration (485 objects x ~20
// float64 ops, ~38 us for a lone wave: 143 ms for C1).  Here the 64 lanes of a wave split the OBJECT loop of ONE
// chain instead: lane L tests objects L, L+64, ... with the reference's float64 arithmetic (spheres.nim:28-49 /
// moving_spheres.nim:39-67 in their own operation order), the wave min-reduces (t, original index) -- closest hit
// is order independent, ties keep the lowest index (hittables_lists.nim:48-55) -- and every lane then runs the
// (wave-uniform) scatter, RNG and sky code on identical values.  A bounce iteration shrinks to ~8 object tests +
// a DPP reduction + ONE material branch; 82 944 chains run in parallel across the machine.  Per-sample work is
// ~60x less lane-efficient outside the object loop, so the host picks this kernel only where the lane-per-pixel
// kernel is latency bound (tor_api.cpp: coop_max_pixels).  Bit-identical canvases (parity tests).
//
// Objects come from the `cold` records, staged once per workgroup as a structure of arrays in LDS (consecutive
// lanes read consecutive float64: conflict free).
// ---------------------------------------------------------------------------------------------
constexpr int kCoopArrays = 10;  // c0 xyz, dc xyz, time0, dt, r^2, flags

// all-lanes minimum of a float64 / int32: DPP inside rows of 16 lanes, then the four row results through SGPRs
__device__ __forceinline__ double wave_min_f64(double v) {
  auto step = [&](int ctrl) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    // __builtin_amdgcn_update_dpp(old, src, dpp_ctrl, row_mask, bank_mask, bound_ctrl)
    unsigned lo2, hi2;
    switch (ctrl) {
      case 0: lo2 = __builtin_amdgcn_update_dpp(lo, lo, 0xb1, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(hi, hi, 0xb1, 0xf, 0xf, false); break;  // quad_perm [1,0,3,2]
      case 1: lo2 = __builtin_amdgcn_update_dpp(lo, lo, 0x4e, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(hi, hi, 0x4e, 0xf, 0xf, false); break;  // quad_perm [2,3,0,1]
      case 2: lo2 = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xf, 0xf, false); break;  // row_half_mirror
      default: lo2 = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xf, 0xf, false); break;  // row_mirror
    }
    const double o = __longlong_as_double((long long)(((unsigned long long)hi2 << 32) | lo2));
    v = __builtin_fmin(v, o);
  };
  step(0); step(1); step(2); step(3);
  auto row = [&](int l) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)b, l), hi = __builtin_amdgcn_readlane((unsigned)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  };
  return __builtin_fmin(__builtin_fmin(row(0), row(16)), __builtin_fmin(row(32), row(48)));
}

__device__ __forceinline__ int wave_min_i32(int v) {
  auto mn = [](int a, int b) { return a < b ? a : b; };
  v = mn(v, (int)__builtin_amdgcn_update_dpp((unsigned)v, (unsigned)v, 0xb1, 0xf, 0xf, false));
  v = mn(v, (int)__builtin_amdgcn_update_dpp((unsigned)v, (unsigned)v, 0x4e, 0xf, 0xf, false));
  v = mn(v, (int)__builtin_amdgcn_update_dpp((unsigned)v, (unsigned)v, 0x141, 0xf, 0xf, false));
  v = mn(v, (int)__builtin_amdgcn_update_dpp((unsigned)v, (unsigned)v, 0x140, 0xf, 0xf, false));
  return mn(mn(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
            mn(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

template <int ARITH>
__global__ __launch_bounds__(kThreads) void coop_pixel_kernel(const KParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* soa = reinterpret_cast<double*>(smem_raw);
  const int n_pad = p.coop_slots;  // multiple of 64, >= cold slots
  const int n_trips = n_pad >> 6;
  const int lane = threadIdx.x & 63;
  // ---- stage the objects: cold record (16 float64) -> per trip of 64 slots, 10 arrays of 64 float64
  //      [trip][array][lane]: every ds_read of the object loop is `lane * 8 + trip base` + an immediate offset ----
  for (int k = threadIdx.x; k < n_pad; k += kThreads) {
    double c[kCoopArrays] = {0, 0, 0, 0, 0, 0, 0, 1.0, -1.0, 0};  // padding: never hit (r^2 = -1 -> discriminant < 0)
    if (k < p.n_cold_slots) {
      const double* r = p.cold + (size_t)k * 16;
      c[0] = r[0]; c[1] = r[1]; c[2] = r[2]; c[3] = r[3]; c[4] = r[4]; c[5] = r[5];
      c[6] = r[7]; c[7] = r[8]; c[8] = r[15]; c[9] = r[13];
    }
    double* dst = soa + (size_t)(k >> 6) * (kCoopArrays * 64) + (k & 63);
#pragma unroll
    for (int a = 0; a < kCoopArrays; ++a) dst[a * 64] = c[a];
  }
  __syncthreads();
  const ldptr L = (ldptr)soa + lane;  // + trip * (kCoopArrays * 64) + array * 64
  // the trip table (tor_scene.cpp) into scalar registers once: 2 bits of kind per trip (scenes of up to 2048 cold
  // slots; larger ones run every trip through the general path), and the movers' (time0, time1 - time0) when all
  // uniform trips share one -- then a query divides for the time fraction exactly once
  unsigned long long kinds = 0;
  bool one_group = true, have_group = false;
  double sg_t0 = 0.0, sg_dt = 1.0;
  {
    const cdptr trips = as_const(p.coop_trips);
    for (int t = 0; t < n_trips; ++t) {
      int kind = (int)trips[4 * t + 0];
      if (t >= 32) kind = 3;
      if (kind == 1 || kind == 2) {
        const double t0 = trips[4 * t + 1], dt = trips[4 * t + 2];
        if (!have_group) { sg_t0 = t0; sg_dt = dt; have_group = true; }
        else if (t0 != sg_t0 || dt != sg_dt) one_group = false;
      }
      if (t < 32) kinds |= (unsigned long long)kind << (2 * t);
    }
    if (!one_group)  // several time groups: the uniform fast paths would need a division per trip -> general path
      for (int t = 0; t < n_trips && t < 32; ++t)
        if (((kinds >> (2 * t)) & 3) != 0) kinds |= 3ull << (2 * t);
  }
  const double w_div = (double)(p.ncols - 1);  // render.nim:64
  const double h_div = (double)(p.nrows - 1);
  const cdptr cold = as_const(p.cold);
  // work items: every pixel of the (shard's) frame, or -- split mode -- the pixels of the first *p.split tiles of the
  // cost-ordered list (the lane kernel renders the others at the same time)
  const unsigned long long n_items =
      p.split != nullptr ? bcast_first_u64(*(const volatile unsigned long long*)p.split) * (unsigned long long)kTilePixels : (unsigned long long)p.n_pixels;

  for (;;) {
    // No `if (lane == 0)` around the fetch or the store below: with both in the loop the compiler threads the two
    // branches together and the other lanes re-enter the loop without lane 0 (they then read their own, stale
    // index through readfirstlane and never leave -- seen on hardware).  All lanes take part instead: the atomic
    // adds 1 for lane 0 and 0 for the others (one aggregated atomic), the store writes 64 identical values.
    unsigned long long pl64 = atomicAdd(p.work_counter, lane == 0 ? 1ull : 0ull);
    pl64 = bcast_first_u64(pl64);
    if (pl64 >= n_items) break;
    unsigned pl = (unsigned)pl64;
    if (p.split != nullptr) {  // the k-th item is pixel (k mod 64) of the (k / 64)-th most expensive tile
      pl = p.order[pl64 >> 6] * (unsigned)kTilePixels + (unsigned)(pl64 & 63);
      if (pl >= p.n_pixels) continue;  // the frame's last tile may be partial
    }
    const unsigned lrow = pl / (unsigned)p.ncols;
    const int col = (int)(pl - lrow * (unsigned)p.ncols);
    const unsigned tile = lrow / (unsigned)p.row_tile;
    const unsigned within = lrow - tile * (unsigned)p.row_tile;
    const int row = (int)((tile * (unsigned)p.shard_count + (unsigned)p.shard_index) * (unsigned)p.row_tile + within);
    Rng rng;
    seed2(rng, (uint64_t)(int64_t)row, (uint64_t)(int64_t)col);  // render.nim:59-60
    V3 acc = v3(0.0, 0.0, 0.0);
    for (int s = 0; s < p.spp; ++s) {
      // render.nim:64-66
      const double u = ((double)col + uniform01(rng)) / w_div;
      const double v = ((double)row + uniform01(rng)) / h_div;
      const Camera cam = load_camera(p.cam_dev);
      const Ray r0 = camera_ray(cam, u, v, rng);
      V3 o = r0.origin, d = r0.direction, att = v3(1.0, 1.0, 1.0);
      double time = r0.time;
      V3 radiance = v3(0.0, 0.0, 0.0);  // absorbed / loop exhausted -> black (render.nim:38,47)
      for (int depth = 0; depth < p.max_depth; ++depth) {
        // ---- closest hit, the object loop split across the lanes (hittables_lists.nim:48-55) ----
        const double ox = o.x, oy = o.y, oz = o.z, dx = d.x, dy = d.y, dz = d.z;
        const double a = (ARITH == 0) ? dx * dx + dy * dy + dz * dz : fma_(dz, dz, fma_(dy, dy, dx * dx));  // spheres.nim:30
        double best_t = __builtin_inf(), best_f = 0.0;
        int best_slot = -1, best_orig = 0x7fffffff;
        // exact test of one object (spheres.nim:28-49) and the order-independent closest-hit update
        auto test_object = [&](int k, double cx, double cy, double cz, double r2, double f) {
          const double ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
          double hb, cc, disc;
          if (ARITH == 0) {
            hb = ocx * dx + ocy * dy + ocz * dz;            // spheres.nim:31
            cc = (ocx * ocx + ocy * ocy + ocz * ocz) - r2;  // spheres.nim:32
            disc = hb * hb - a * cc;                        // spheres.nim:33
          } else {
            hb = fma_(ocz, dz, fma_(ocy, dy, ocx * dx));
            cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -r2)));
            disc = fma_(hb, hb, -(a * cc));
          }
          // both roots are <= 0 when half_b >= 0 and c >= 0: such an object can never be accepted (t_min = 0.001)
          if (disc > 0.0 && (hb < 0.0 || cc < 0.0)) {
            const double root = __builtin_sqrt(disc);  // spheres.nim:35-48
            double sol = (-hb - root) / a;
            bool ok = (0.001 < sol) && (sol < __builtin_inf());
            if (!ok) {
              sol = (-hb + root) / a;
              ok = (0.001 < sol) && (sol < __builtin_inf());
            }
            if (ok) {
              const int orig = (int)__double_as_longlong(p.cold[(size_t)k * 16 + 14]);
              if (sol < best_t || (sol == best_t && orig < best_orig)) { best_t = sol; best_slot = k; best_orig = orig; best_f = f; }
            }
          }
        };
        // One trip = 64 consecutive cold slots, one per lane.  The host classified every trip (tor_scene.cpp): the
        // common cases -- nothing moves / everything moves along y within one (time0, time1) / everything moves
        // within one (time0, time1) -- run without per-object branches, and the time fraction is divided once per
        // (time0, time1) and query (moving_spheres.nim:42: the same operands give the same quotient).
        const double g_t0 = sg_t0, g_dt = sg_dt;
        const double g_f = have_group ? (time - sg_t0) / sg_dt : 0.0;
        for (int t = 0; t < n_trips; ++t) {
          const ldptr T = L + t * (kCoopArrays * 64);
          const int k = t * 64 + lane;
          const int kind = (t < 32) ? (int)((kinds >> (2 * t)) & 3) : 3;
          if (kind == 0) {
            test_object(k, T[0 * 64], T[1 * 64], T[2 * 64], T[8 * 64], 0.0);
            continue;
          }
          if (kind != 3) {
            const double f = g_f;
            double cx = T[0 * 64], cy = T[1 * 64], cz = T[2 * 64];
            if (kind == 1) {  // center1.x == center0.x and center1.z == center0.z: c0 + f * 0 == c0
              cy = (ARITH == 0) ? cy + T[4 * 64] * f : fma_(T[4 * 64], f, cy);
            } else if (ARITH == 0) {
              cx = cx + T[3 * 64] * f; cy = cy + T[4 * 64] * f; cz = cz + T[5 * 64] * f;  // moving_spheres.nim:43
            } else {
              cx = fma_(T[3 * 64], f, cx); cy = fma_(T[4 * 64], f, cy); cz = fma_(T[5 * 64], f, cz);
            }
            test_object(k, cx, cy, cz, T[8 * 64], f);
          } else {  // mixed trip: per-object kind and time group
            double cx = T[0 * 64], cy = T[1 * 64], cz = T[2 * 64];
            double f = 0.0;
            if ((int)__double_as_longlong(T[9 * 64]) & 1) {
              const double t0 = T[6 * 64], dt = T[7 * 64];
              f = (have_group && t0 == g_t0 && dt == g_dt) ? g_f : (time - t0) / dt;
              if (ARITH == 0) { cx = cx + T[3 * 64] * f; cy = cy + T[4 * 64] * f; cz = cz + T[5 * 64] * f; }
              else { cx = fma_(T[3 * 64], f, cx); cy = fma_(T[4 * 64], f, cy); cz = fma_(T[5 * 64], f, cz); }
            }
            test_object(k, cx, cy, cz, T[8 * 64], f);
          }
        }
        const double t_min = wave_min_f64(best_t);
        if (!(t_min < __builtin_inf())) {
          radiance = sky(d, att);  // render.nim:41-45
          break;
        }
        unsigned long long win = ballot64(best_t == t_min);
        if (win & (win - 1)) {  // several lanes at the same t (duplicate objects): the lowest original index wins
          const int o_min = wave_min_i32(best_t == t_min ? best_orig : 0x7fffffff);
          win = ballot64(best_t == t_min && best_orig == o_min);
        }
        const int wl = (int)__builtin_ctzll(win);
        const int slot = __builtin_amdgcn_readlane(best_slot, wl);
        const unsigned long long fb = (unsigned long long)__double_as_longlong(best_f);
        const unsigned hit_f_lo = (unsigned)__builtin_amdgcn_readlane((unsigned)fb, wl);  // (readlane returns int: no sign extension into the high word)
        const unsigned hit_f_hi = (unsigned)__builtin_amdgcn_readlane((unsigned)(fb >> 32), wl);
        const double hit_f = __longlong_as_double((long long)(((unsigned long long)hit_f_hi << 32) | (unsigned long long)hit_f_lo));
        // ---- shade: wave-uniform (every lane holds the same values) ----
        const cdptr c = cold + (size_t)slot * 16;
        const int flags = (int)__double_as_longlong(c[13]);
        V3 center = v3(c[0], c[1], c[2]);
        if (flags & 1) {
          if (ARITH == 0) center = center + v3(c[3], c[4], c[5]) * hit_f;
          else center = v3(fma_(c[3], hit_f, c[0]), fma_(c[4], hit_f, c[1]), fma_(c[5], hit_f, c[2]));
        }
        const V3 hp = o + d * t_min;               // rays.nim:24-25
        const V3 outward = (hp - center) * c[6];   // spheres.nim:43
        const bool front = dot(d, outward) < 0.0;  // core.nim:47-49
        const V3 n = front ? outward : -outward;
        const int mat = (flags >> 8) & 0xff;
        const V3 albedo = v3(c[9], c[10], c[11]);
        bool absorbed = false;
        if (mat == kLambertian) {  // materials.nim:24-30
          d = n + random_unit_vector(rng);
          o = hp;
          att = mul_att(att, albedo);
        } else if (mat == kMetal) {  // materials.nim:39-47
          const V3 reflected = reflect(unit_vector(d), n);
          const V3 nd = reflected + random_in_unit_sphere(rng) * c[12];
          o = hp;
          d = nd;
          time = 0.0;
          if (dot(nd, n) > 0.0) att = mul_att(att, albedo);
          else absorbed = true;
        } else {  // materials.nim:62-86
          const double eta = front ? c[9] : c[12];  // 1.0 / ri : ri
          const V3 ud = unit_vector(d);
          const double dn = dot(-ud, n);
          const double cos_theta = (dn <= 1.0) ? dn : 1.0;
          const double sin_theta = __builtin_sqrt(1.0 - cos_theta * cos_theta);
          V3 nd;
          if (eta * sin_theta > 1.0) {
            nd = reflect(ud, n);
          } else {
            const double reflect_prob = schlick_r0(cos_theta, front ? c[10] : c[11]);
            if (uniform01(rng) < reflect_prob) nd = reflect(ud, n);
            else nd = refract(ud, n, eta);
          }
          o = hp;
          d = nd;
          time = 0.0;
          // (attenuation (1, 1, 1), materials.nim:63: x * 1.0 == x, nothing to do)
        }
        if (absorbed) break;  // render.nim:38
      }
      acc = acc + radiance;  // render.nim:67
    }
    double* out = p.out + (size_t)pl * 3;  // every lane holds the same sum
    out[0] = acc.x; out[1] = acc.y; out[2] = acc.z;
  }
}

// ---------------------------------------------------------------------------------------------
// SrvRng -- the chain servers' generator: a pixel's xoshiro256+ stream (rng.nim:58-74), drawn one output at a time as the
// lanes do, plus a TABLE of what depends on the stream alone.  A server runs ONE chain with all 64 lanes, so
// random_unit_vector (sampling.nim:51-55: two outputs, a square root and the correctly rounded sin / cos -- 280
// double-double instructions, a third of a Lambertian bounce) can be evaluated for 64 stream positions at the price of one:
// at the top of a bounce, when the table is used up, the scalar unit steps a COPY of the (wave-uniform) state 64 times and
// leaves output k in lane k, and every lane evaluates the vector that starts at its position.  A Lambertian scatter then
// reads the entry of the running position (v_readlane) and steps the live state twice.  Same operations on the same outputs
// as the lanes' sequential code: same bits (tests/test_gpu_round3.py: every hand-off test compares the two).
// ---------------------------------------------------------------------------------------------
struct SrvRng {
  Rng st;             // the live state (wave-uniform): every draw steps it, as in the lanes
  double tx, ty, tz;  // lane k: random_unit_vector drawn at position k of the table (outputs k and k + 1)
  unsigned i;         // the live state's position in the table (wave-uniform); >= 63: no entry starts here
};
constexpr unsigned kSrvTableLast = 62;  // the last position whose two outputs are both in the table

// Builds the table for the 64 positions from the live state on.  ONE call site per copy of serve_chains (top of the
// bounce loop -- a Lambertian scatter draws its vector first, so the position it reads is the position seen there).
__device__ __forceinline__ void srv_refill(SrvRng& g) {
  Rng s = g.st;
  unsigned lo = 0, hi = 0;
  const int lane = threadIdx.x & 63;
#pragma unroll 1
  for (int k = 0; k < 64; ++k) {  // scalar unit: 64 steps of a copy of the state
    const uint64_t o = next(s);
    const bool mine = lane == k;  // (this compiler has no writelane builtin: a compare and two selects per step)
    lo = mine ? (unsigned)o : lo;
    hi = mine ? (unsigned)(o >> 32) : hi;
  }
  // position k's second output is output k + 1 (lane 63 has none: kSrvTableLast)
  const unsigned nlo = (unsigned)__shfl_down((int)lo, 1), nhi = (unsigned)__shfl_down((int)hi, 1);
  const V3 t = random_unit_vector_of(((uint64_t)hi << 32) | lo, ((uint64_t)nhi << 32) | nlo);
  g.tx = t.x; g.ty = t.y; g.tz = t.z;
  g.i = 0u;
}

__device__ __forceinline__ uint64_t next(SrvRng& g) {
  g.i += 1u;
  return next(g.st);
}

__device__ __forceinline__ V3 random_unit_vector(SrvRng& g) {
  const uint64_t out_a = next(g.st);  // (the live state moves on either way)
  const uint64_t out_z = next(g.st);
  const unsigned at = g.i;
  g.i += 2u;
  if (at <= kSrvTableLast) {
    const int k = __builtin_amdgcn_readfirstlane((int)at);
    auto pick = [&](double v) {
      const unsigned long long bits = double_to_bits(v);
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, k), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), k);
      return bits_to_double(((unsigned long long)hi << 32) | (unsigned long long)lo);
    };
    return v3(pick(g.tx), pick(g.ty), pick(g.tz));
  }
  return random_unit_vector_of(out_a, out_z);  // (not reached from serve_chains: it refills in front of every bounce that would)
}

// ---------------------------------------------------------------------------------------------
// serve_chains -- the SERVER side of the SEED_PIXEL chain hand-off, inside integrate_kernel<0, A, W, 1, 1>.
//
// A server is a whole wave that continues ONE pixel chain at a time from the state a lane pushed at a sample boundary
// (pixel, samples done, xoshiro256+ state, running sum): the same stream, the same operations, the same pixel -- only the
// closest-hit query is shared by the 64 lanes, as in coop_pixel_kernel, and here through the exact block culling as well:
//   trip 1  lane L slab-tests block box L (and L + 64) in float32 (tor_filter32.hpp: slab_bit32, conservative);
//   trip 2+ the candidates -- the always-tested objects, then the 8 objects of every box the ray can touch -- one per lane
//           through the reference's float64 test (spheres.nim:28-49 / moving_spheres.nim:39-67, own operation order);
//   then    a DPP min-reduction of t, ties to the lowest original index (hittables_lists.nim:48-55), and every lane runs
//           the now wave-uniform scatter / RNG / sky code on identical values.
// Any candidate set that contains the true closest hit gives the lane kernel's result bit for bit (closest hit is order
// independent, every candidate gets the identical float64 test).  A bounce costs a server ~1-2 us instead of ~16 us in a
// lane, at ~7x the instructions per bounce: worth it exactly for the chains that would otherwise end the frame alone.
// Queue protocol: a server takes ticket i (atomic add on mig[kMigHead]) and waits for record i's ready flag; lanes take
// slot i (atomic add on mig[kMigTail]), write the record, release the flag.  A server leaves when no wave is left in the
// lane loop and its ticket lies beyond the last record.
// ---------------------------------------------------------------------------------------------
template <int ARITH>
__device__ __forceinline__ bool serve_chains(const KParams& p, bool dedicated) {
  const int lane = threadIdx.x & 63;
  const double w_div = (double)(p.ncols - 1);  // render.nim:64
  const double h_div = (double)(p.nrows - 1);
  const cdptr cold = as_const(p.cold);
  // this lane's block boxes: records lane and lane + 64 of the float32 boxes (constant for the launch)
  const float nanf_ = __builtin_nanf("");
  f2v bx0 = splat2(nanf_), by0 = bx0, bz0 = bx0, bx1 = bx0, by1 = bx0, bz1 = bx0;
  const bool valid0 = lane < p.n_boxes, valid1 = lane + 64 < p.n_boxes;
  {
    const gfptr b = (gfptr)(uintptr_t)p.bnd32;
    if (valid0) { const gfptr r = b + 8 * lane; bx0 = (f2v){r[0], r[1]}; by0 = (f2v){r[2], r[3]}; bz0 = (f2v){r[4], r[5]}; }
    if (valid1) { const gfptr r = b + 8 * (lane + 64); bx1 = (f2v){r[0], r[1]}; by1 = (f2v){r[2], r[3]}; bz1 = (f2v){r[4], r[5]}; }
  }
  const unsigned n_always = (unsigned)p.spatial_base;
  const unsigned long long cap = (unsigned long long)p.mig_cap;
  const unsigned max_naps = (unsigned)(p.mig_flags >> 8) & 0xffu;
  if (dedicated && p.mig_patience != 0) {
    // A dedicated server takes no ticket before the first chain has been handed over at all: if none has after
    // `mig_patience`, this scene has no long chains (no glass) and the wave becomes a lane wave -- counted into kMigLaneWaves
    // first, and only while that count is not 0 (the frame is still in its lane phase).  Without a ticket it leaves no hole
    // in the queue.
    const unsigned long long t_begin = wall_clock64();
    for (;;) {
      unsigned long long tl = 0, running = 1;
      if (lane == 0) {
        tl = __hip_atomic_load(p.mig + kMigTail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        running = __hip_atomic_load(p.mig + kMigLaneWaves, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      tl = bcast_first_u64(tl);
      running = bcast_first_u64(running);
      if (tl != 0 || running == 0) break;  // chains are coming (or the frame is over): serve
      if (wall_clock64() - t_begin > (unsigned long long)p.mig_patience) {
        unsigned joined = 0;
        if (lane == 0) {
          unsigned long long cur = __hip_atomic_load(p.mig + kMigLaneWaves, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          while (cur != 0 && !__hip_atomic_compare_exchange_strong(p.mig + kMigLaneWaves, &cur, cur + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {}
          joined = cur != 0 ? 1u : 0u;
          if (joined) atomicAdd(p.mig + kMigConverted, 1ull);
        }
        if (__builtin_amdgcn_readfirstlane((int)joined) != 0) return true;
        break;
      }
      __builtin_amdgcn_s_sleep(127);
      __builtin_amdgcn_s_sleep(127);
    }
  }
  for (;;) {
    // ---- take a ticket, wait for its record (or for the end of the frame) ----
    unsigned long long tk = atomicAdd(p.mig + kMigHead, lane == 0 ? 1ull : 0ull);  // (all lanes take part, see coop_pixel_kernel)
    tk = bcast_first_u64(tk);
    bool quit = false;
    // Waiting: poll this ticket's own flag (one lane; the flags of consecutive tickets share a line, waiting servers are
    // spread over many) with relaxed loads and a growing back-off, and look at the end-of-frame words only every 8th poll.
    // An ACQUIRE load here invalidates the CU's vector cache on every poll: with thousands of waves waiting at the end of a
    // frame the lanes that still ran lost theirs every few hundred ns -- measured 25x slower; a compare-and-swap claim
    // (instead of tickets) made every waiter hammer one line whenever a chain was pending: the same.
    unsigned polls = 0, naps = 1;
    // (stall escape, ADVICE r3: the launch assumes that all of its workgroups are resident -- servers wait for lane waves.  If
    // some never start (another process's persistent kernel, a CU mask), the resident ones would wait for ever and the host with
    // them.  A waiting server therefore watches the frame's progress words; when none of them has moved for mig_stall_ticks it
    // flags the frame as incomplete and leaves -- its ticket stays a hole, which is why the host must re-render: tor_api.cpp)
    unsigned long long stall_t0 = wall_clock64(), stall_sig = ~0ull;
    for (;;) {
      unsigned ready = 0;
      if (lane == 0 && tk < cap) {
        if (p.mig_flags & 1) ready = __hip_atomic_load(p.mig_flag + tk, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        else ready = __hip_atomic_load(p.mig_flag + tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (__builtin_amdgcn_readfirstlane((int)ready) != 0) break;
      if ((polls & 7u) == 7u || tk >= cap) {
        unsigned long long running = 1;
        if (lane == 0) running = __hip_atomic_load(p.mig + kMigLaneWaves, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bcast_first_u64(running) == 0) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          unsigned long long tail = 0;
          if (lane == 0) tail = __hip_atomic_load(p.mig + kMigTail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tail = bcast_first_u64(tail);
          if (tail > cap) tail = cap;
          if (tk >= tail) { quit = true; break; }
        }
        if (p.mig_stall_ticks != 0 && (polls & 15u) == 15u) {
          unsigned long long sig = 0;
          if (lane == 0) {
            sig = __hip_atomic_load(p.work_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                  __hip_atomic_load(p.mig + kMigTail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                  __hip_atomic_load(p.mig + kMigServed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                  (__hip_atomic_load(p.mig + kMigLaneWaves, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 40);
          }
          sig = bcast_first_u64(sig);
          const unsigned long long now = wall_clock64();
          // (mig_stall_ticks == 1, a test setting: the first look counts as a stall)
          if (sig != stall_sig && p.mig_stall_ticks != 1ull) { stall_sig = sig; stall_t0 = now; }
          else if (now - stall_t0 > p.mig_stall_ticks || p.mig_stall_ticks == 1ull) {
            if (lane == 0) __hip_atomic_store(p.mig + kMigStalled, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            quit = true;
            break;
          }
        }
      }
      polls += 1;
      if ((p.mig_flags & 2u) && lane == 0) {
        // adaptive threshold: a server with nothing to do lowers the bar by 1/32 (never below the floor)
        const unsigned long long now = __hip_atomic_load(p.mig + kMigPushNow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long fl = __hip_atomic_load(p.mig + kMigPushFloor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long next = now - (now >> 5);
        if (next < fl) next = fl;
        if (next != now) __hip_atomic_store(p.mig + kMigPushNow, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (unsigned k = 0; k < naps; ++k) __builtin_amdgcn_s_sleep(127);
      if (naps < max_naps) naps *= 2;
    }
    if (quit) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the record's words were written before the flag was released
    unsigned long long rec[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) rec[k] = bcast_first_u64(__hip_atomic_load(p.mig_rec + tk * 8 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const unsigned pl = (unsigned)rec[0];
    const int s_begin = (int)((rec[0] >> 32) & 0x7fffffffull);
    const bool was_hot = (rec[0] >> 63) != 0;
    unsigned chain_its = 0;
#ifdef TOR_SERVE_PROF
    // (profiling build only, tools/server_phases.py: shader-clock ticks per phase of a served bounce)
    unsigned long long pf_query = 0, pf_shade[3] = {0, 0, 0}, pf_miss = 0, pf_setup = 0, pf_n[3] = {0, 0, 0}, pf_nmiss = 0, pf_rounds = 0, pf_cand = 0;
    unsigned long long pf_sub[5] = {0, 0, 0, 0, 0};  // (TOR_SERVE_PROF=2: inside the query -- slab tests done, slots known, records in, tests done, winner known)
    const unsigned long long pf_begin = __builtin_readcyclecounter();
#endif
    SrvRng rng;
    rng.st = Rng{rec[1], rec[2], rec[3], rec[4]};
    rng.tx = rng.ty = rng.tz = 0.0;
    rng.i = 64u;  // no table yet: the first bounce builds one
    V3 acc = v3(bits_to_double(rec[5]), bits_to_double(rec[6]), bits_to_double(rec[7]));
    const unsigned lrow = pl / (unsigned)p.ncols;
    const int col = (int)(pl - lrow * (unsigned)p.ncols);
    const unsigned rtile = lrow / (unsigned)p.row_tile;
    const unsigned within = lrow - rtile * (unsigned)p.row_tile;
    const int row = (int)((rtile * (unsigned)p.shard_count + (unsigned)p.shard_index) * (unsigned)p.row_tile + within);
    // arbiter priority while serving: a hot chain is the frame's critical path (level 3); a chain taken over in the tail of the
    // frame shares its SIMD with lane waves that are finishing theirs (mig_flags bit 2: level 1 then)
    if (was_hot || !(p.mig_flags & 4u)) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(1);
    for (int s = s_begin; s < p.spp; ++s) {
#ifdef TOR_SERVE_PROF
      const unsigned long long pf_s0 = __builtin_readcyclecounter();
#endif
      // render.nim:64-66
      const double u = ((double)col + uniform01(rng)) / w_div;
      const double v = ((double)row + uniform01(rng)) / h_div;
      const Camera cam = load_camera(p.cam_dev);
      const Ray r0 = camera_ray(cam, u, v, rng);
      V3 o = r0.origin, d = r0.direction, att = v3(1.0, 1.0, 1.0);
      double time = r0.time;
      V3 radiance = v3(0.0, 0.0, 0.0);  // absorbed / loop exhausted -> black (render.nim:38,47)
#ifdef TOR_SERVE_PROF
      pf_setup += __builtin_readcyclecounter() - pf_s0;
#endif
      for (int depth = 0; depth < p.max_depth; ++depth) {
        if (rng.i > kSrvTableLast) srv_refill(rng);  // (the only call site: see SrvRng)
        // ---- closest hit (hittables_lists.nim:48-55), boxes and candidates split across the lanes ----
        chain_its += 1;
#ifdef TOR_SERVE_PROF
        const unsigned long long pf_q0 = __builtin_readcyclecounter();
#endif
        const double ox = o.x, oy = o.y, oz = o.z, dx = d.x, dy = d.y, dz = d.z;
        const double a_strict = dx * dx + dy * dy + dz * dz;  // spheres.nim:30
        const double a = (ARITH != 1) ? a_strict : fma_(dz, dz, fma_(dy, dy, dx * dx));
        const RayF32 r32 = make_ray_f32(ox, oy, oz, dx, dy, dz, a_strict, p.org[0], p.org[1], p.org[2]);
        const BoxRay32 b32 = make_box_ray32(r32, p.sp_bmax);
        const unsigned wild = r32.wild & 1u;  // a ray outside the float32 test's guarded ranges enters every box
        const unsigned long long m0 = ballot64(valid0 && ((slab_bit32(b32, bx0, by0, bz0) | wild) != 0u));
        const unsigned long long m1 = ballot64(valid1 && ((slab_bit32(b32, bx1, by1, bz1) | wild) != 0u));
        const unsigned c0 = (unsigned)__builtin_popcountll(m0), n_hit = c0 + (unsigned)__builtin_popcountll(m1);
        const unsigned n_cand = n_always + 8u * n_hit;
        // The boxes the ray can touch, compacted: lane r gets the index of the r-th such box -- every lane whose box was hit
        // pushes its box index to the lane of its rank (v_mbcnt + ds_permute; the others push to lane 63, which no rank below
        // 64 hits reaches).  A candidate lane then pulls `its` box with one ds_bpermute instead of walking the set bits
        // (850 of the query's 2800 cycles).  A ray that enters 64 boxes or more (a `wild` one enters all) takes the walk.
        const bool compact = n_hit <= 63u;
        int hit_list = 0;
        if (compact) {
          const bool h0 = ((m0 >> lane) & 1ull) != 0, h1 = ((m1 >> lane) & 1ull) != 0;
          const unsigned r0 = lane_prefix(m0), r1 = c0 + lane_prefix(m1);
          const int la = __builtin_amdgcn_ds_permute((int)((h0 ? r0 : 63u) << 2), lane);
          const int lb = __builtin_amdgcn_ds_permute((int)((h1 ? r1 : 63u) << 2), lane + 64);
          hit_list = ((unsigned)lane < c0) ? la : lb;
        }
#if defined(TOR_SERVE_PROF) && TOR_SERVE_PROF >= 2
        const unsigned long long pf_a = __builtin_readcyclecounter() + (n_cand == 0xffffffffu ? 1ull : 0ull);
        pf_sub[0] += pf_a - pf_q0;
        unsigned long long pf_b = pf_a;
#endif
        const double f_sp = (time - p.sp_t0) / p.sp_dt;  // moving_spheres.nim:42 for the spatial movers' (time0, time1)
        double best_t = __builtin_inf();
        int best_orig = 0x7fffffff, bflags = 0;
        double bcx = 0, bcy = 0, bcz = 0, b6 = 0, b9 = 0, b10 = 0, b11 = 0, b12 = 0;  // this lane's closest hit: centre, 1/radius, material
        for (unsigned base = 0; base < n_cand; base += 64u) {
          const unsigned i = base + (unsigned)lane;
          int slot = -1;
          const unsigned want = (i - n_always) >> 3;  // the (i - n_always)/8-th box the ray touches (garbage in lanes that have none)
          int box = 0;
          if (compact) {
            box = __builtin_amdgcn_ds_bpermute((int)((want & 63u) << 2), hit_list);
          } else {  // a wave-uniform walk over the set bits
            unsigned rank = 0;
            for (unsigned long long m = m0; m != 0; m &= m - 1, ++rank)
              if (rank == want) box = (int)__builtin_ctzll(m);
            for (unsigned long long m = m1; m != 0; m &= m - 1, ++rank)
              if (rank == want) box = 64 + (int)__builtin_ctzll(m);
          }
          if (i < n_always) slot = (int)i;
          else if (i < n_cand) slot = p.spatial_base + 8 * box + (int)((i - n_always) & 7u);
#if defined(TOR_SERVE_PROF) && TOR_SERVE_PROF >= 2
          {
            const unsigned long long pf_t = __builtin_readcyclecounter() + (slot == -77 ? 1ull : 0ull);
            pf_sub[1] += pf_t - pf_b;
            pf_b = pf_t;
          }
#endif
          if (slot >= 0) {
            // The WHOLE record in one batch of loads, used without a branch in between: a server's bounce is a chain of
            // dependent steps and every extra round trip to L2 is paid in full -- so the fields the shading needs (1/radius,
            // material) travel with the fields of the test, and the winner's are broadcast from its lane afterwards instead
            // of being fetched again.
            const double* c = p.cold + (size_t)slot * 16;
            const double k0 = c[0], k1 = c[1], k2 = c[2], k3 = c[3], k4 = c[4], k5 = c[5], k6 = c[6], k7 = c[7], k8 = c[8];
            const double k9 = c[9], k10 = c[10], k11 = c[11], k12 = c[12], k13 = c[13], k14 = c[14], k15 = c[15];
#if defined(TOR_SERVE_PROF) && TOR_SERVE_PROF >= 2
            {
              const unsigned long long pf_t = __builtin_readcyclecounter() + (double_to_bits(k0 + k15 + k13) == 1ull ? 1ull : 0ull);
              pf_sub[2] += pf_t - pf_b;
              pf_b = pf_t;
            }
#endif
            const bool moving = ((int)__double_as_longlong(k13) & 1) != 0;
            double f = f_sp;  // (the spatial movers share one time group: same operands as the division, same quotient)
            if (moving && !(k7 == p.sp_t0 && k8 == p.sp_dt)) f = (time - k7) / k8;
            double mx, my, mz;  // centre of a mover (moving_spheres.nim:43); a static sphere keeps c0 untouched
            if (ARITH != 1) { mx = k0 + k3 * f; my = k1 + k4 * f; mz = k2 + k5 * f; }
            else { mx = fma_(k3, f, k0); my = fma_(k4, f, k1); mz = fma_(k5, f, k2); }
            const double cx = moving ? mx : k0, cy = moving ? my : k1, cz = moving ? mz : k2;
            const double ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
            double hb, cc, disc;
            if (ARITH != 1) {
              hb = ocx * dx + ocy * dy + ocz * dz;             // spheres.nim:31
              cc = (ocx * ocx + ocy * ocy + ocz * ocz) - k15;  // spheres.nim:32
              disc = hb * hb - a * cc;                         // spheres.nim:33
            } else {
              hb = fma_(ocz, dz, fma_(ocy, dy, ocx * dx));
              cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -k15)));
              disc = fma_(hb, hb, -(a * cc));
            }
            // both roots are <= 0 when half_b >= 0 and c >= 0: such an object can never be accepted (t_min = 0.001)
            if (disc > 0.0 && (hb < 0.0 || cc < 0.0)) {
              const double root = __builtin_sqrt(disc);  // spheres.nim:35-48
              double sol = (-hb - root) / a;
              bool ok = (0.001 < sol) && (sol < __builtin_inf());
              if (!ok) {
                sol = (-hb + root) / a;
                ok = (0.001 < sol) && (sol < __builtin_inf());
              }
              if (ok) {
                const int orig = (int)__double_as_longlong(k14);
                if (sol < best_t || (sol == best_t && orig < best_orig)) {
                  best_t = sol; best_orig = orig;
                  bcx = cx; bcy = cy; bcz = cz; b6 = k6; b9 = k9; b10 = k10; b11 = k11; b12 = k12;
                  bflags = (int)__double_as_longlong(k13);
                }
              }
            }
          }
        }
#if defined(TOR_SERVE_PROF) && TOR_SERVE_PROF >= 2
        {
          const unsigned long long pf_t = __builtin_readcyclecounter() + (double_to_bits(best_t) == 1ull ? 1ull : 0ull);
          pf_sub[3] += pf_t - pf_b;
          pf_b = pf_t;
        }
#endif
        const double t_min = wave_min_f64(best_t);
#ifdef TOR_SERVE_PROF
        const unsigned long long pf_q1 = __builtin_readcyclecounter();
        pf_query += pf_q1 - pf_q0;
#if TOR_SERVE_PROF >= 2
        pf_sub[4] += pf_q1 - pf_b;
#endif
        pf_rounds += (n_cand + 63u) / 64u;
        pf_cand += n_cand;
#endif
        if (!(t_min < __builtin_inf())) {
          radiance = sky(d, att);  // render.nim:41-45
#ifdef TOR_SERVE_PROF
          pf_miss += __builtin_readcyclecounter() - pf_q1;
          pf_nmiss += 1;
#endif
          break;
        }
        unsigned long long win = ballot64(best_t == t_min);
        if (win & (win - 1)) {  // several lanes at the same t (duplicate objects): the lowest original index wins
          const int o_min = wave_min_i32(best_t == t_min ? best_orig : 0x7fffffff);
          win = ballot64(best_t == t_min && best_orig == o_min);
        }
        // ---- shade: wave-uniform (every lane holds the same values: the winner's, broadcast from its lane) ----
        const int wl = (int)__builtin_ctzll(win);
        auto bc = [&](double v) {
          const unsigned long long bits = double_to_bits(v);
          const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, wl), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), wl);
          return bits_to_double(((unsigned long long)hi << 32) | (unsigned long long)lo);
        };
        const V3 center = v3(bc(bcx), bc(bcy), bc(bcz));  // (the centre the test computed: moving_spheres.nim:43 evaluated once)
        const double c6 = bc(b6), c9 = bc(b9), c10 = bc(b10), c11 = bc(b11), c12 = bc(b12);
        const int flags = __builtin_amdgcn_readlane(bflags, wl);
        const V3 hp = o + d * t_min;               // rays.nim:24-25
        const V3 outward = (hp - center) * c6;     // spheres.nim:43
        const bool front = dot(d, outward) < 0.0;  // core.nim:47-49
        const V3 n = front ? outward : -outward;
        const int mat = (flags >> 8) & 0xff;
        const V3 albedo = v3(c9, c10, c11);
        bool absorbed = false;
        if (mat == kLambertian) {  // materials.nim:24-30
          d = n + random_unit_vector(rng);
          o = hp;
          att = mul_att(att, albedo);
        } else if (mat == kMetal) {  // materials.nim:39-47
          const V3 reflected = reflect(unit_vector(d), n);
          const V3 nd = reflected + random_in_unit_sphere(rng) * c12;
          o = hp;
          d = nd;
          time = 0.0;
          if (dot(nd, n) > 0.0) att = mul_att(att, albedo);
          else absorbed = true;
        } else {  // materials.nim:62-86
          const double eta = front ? c9 : c12;  // 1.0 / ri : ri
          const V3 ud = unit_vector(d);
          const double dn = dot(-ud, n);
          const double cos_theta = (dn <= 1.0) ? dn : 1.0;
          const double sin_theta = __builtin_sqrt(1.0 - cos_theta * cos_theta);
          V3 nd;
          if (eta * sin_theta > 1.0) {
            nd = reflect(ud, n);
          } else {
            const double reflect_prob = schlick_r0(cos_theta, front ? c10 : c11);
            if (uniform01(rng) < reflect_prob) nd = reflect(ud, n);
            else nd = refract(ud, n, eta);
          }
          o = hp;
          d = nd;
          time = 0.0;
          // (attenuation (1, 1, 1), materials.nim:63: x * 1.0 == x, nothing to do)
        }
#ifdef TOR_SERVE_PROF
        {
          // (o.x is the last value the shading produces: the read waits for it)
          const unsigned long long pf_now = __builtin_readcyclecounter() + (double_to_bits(d.x + d.y + d.z) == 1ull ? 1ull : 0ull);
          pf_shade[mat] += pf_now - pf_q1;
          pf_n[mat] += 1;
        }
#endif
        if (absorbed) break;  // render.nim:38
      }
      acc = acc + radiance;  // render.nim:67
    }
    __builtin_amdgcn_s_setprio(0);
    double* out = p.out + (size_t)pl * 3;  // every lane holds the same sum
    out[0] = acc.x; out[1] = acc.y; out[2] = acc.z;
    if (lane == 0) {
      atomicAdd(p.mig + kMigServed, 1ull);
#ifdef TOR_SERVE_PROF
      atomicAdd(p.mig + 81, pf_query); atomicAdd(p.mig + 82, pf_shade[0]); atomicAdd(p.mig + 83, pf_shade[1]); atomicAdd(p.mig + 84, pf_shade[2]);
      atomicAdd(p.mig + 85, pf_miss); atomicAdd(p.mig + 86, pf_setup); atomicAdd(p.mig + 87, pf_n[0]); atomicAdd(p.mig + 88, pf_n[1]);
      atomicAdd(p.mig + 89, pf_n[2]); atomicAdd(p.mig + 90, pf_nmiss); atomicAdd(p.mig + 91, pf_rounds); atomicAdd(p.mig + 92, pf_cand);
      atomicAdd(p.mig + 93, (unsigned long long)__builtin_readcyclecounter() - pf_begin);
      atomicAdd(p.mig + 94, (unsigned long long)(p.spp - s_begin));
#if TOR_SERVE_PROF >= 2
      atomicAdd(p.mig + 75, pf_sub[0]); atomicAdd(p.mig + 76, pf_sub[1]); atomicAdd(p.mig + 77, pf_sub[2]); atomicAdd(p.mig + 78, pf_sub[3]); atomicAdd(p.mig + 79, pf_sub[4]);
#endif
#endif
      atomicAdd(p.mig + (was_hot ? kMigItsHot : kMigItsTail), (unsigned long long)chain_its);
      atomicMax(p.mig + (was_hot ? kMigTHotDone : kMigTTailDone), (unsigned long long)wall_clock64());
    }
  }
  return false;
}

// Tile schedule for SEED_PIXEL: counting sort of the tiles by probed cost, most expensive first
// (longest-processing-time-first: a pixel is a sequential chain of spp samples, so the expensive
// chains must start at t = 0).  One workgroup; the order of equal-cost tiles is irrelevant (the
// schedule never changes a pixel's value).
constexpr int kCostBins = 4096;

// The probe counts closest-hit queries per PIXEL (2 samples).  A tile's sort key is led by its longest pixel chain --
// lanes pull pixels one by one, so the chain, not the tile's sum, is what has to start early (a tile on the rim of a
// glass sphere has a few 27-bounce pixels among sky: by its sum it would start mid-frame and its chains would end the
// frame) -- with the tile's sum as the tie-breaker; the sum itself is kept for the work accounting of the cuts.
__global__ __launch_bounds__(256) void tile_key_kernel(const unsigned* pixel_cost, unsigned n_pixels, int n_tiles, unsigned* key,
                                                        unsigned* work, unsigned key_mode, unsigned probe_spp, unsigned* ghist,
                                                        unsigned long long* gwork) {
  const int tile = (int)(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64);
  if (tile >= n_tiles) return;
  const unsigned pl = (unsigned)tile * kTilePixels + (threadIdx.x & 63);
  const unsigned c = pl < n_pixels ? pixel_cost[pl] : 0u;
  unsigned mx = c, sum = c;
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned om = (unsigned)__shfl_xor((int)mx, off), os = (unsigned)__shfl_xor((int)sum, off);
    mx = om > mx ? om : mx;
    sum += os;
  }
  if ((threadIdx.x & 63) == 0) {
    if (key_mode == 0) {  // round-2 key: the tile's longest probed pixel, the sum breaks ties
      const unsigned m = mx < 127u ? mx : 127u;        // 2 samples x max_depth 50 <= 100 (more only with a deeper max_depth)
      const unsigned t = (sum >> 8) < 31u ? (sum >> 8) : 31u;
      key[tile] = (m << 5) | t;
    } else {
      // Two classes.  A pixel whose probe samples were ALL deep (>= 28 queries per sample on average: inside glass) is a
      // long chain for certain -- one deep path among ordinary ones is not, half of all tiles hold one -- and its tile must
      // start first: upper half of the key space, by that pixel's count.  Every other tile is ordered by its SUM: 64 pixels
      // x the probe's samples predict the mean cost of the tile's chains well, and the frame should end on the cheapest
      // chains (the tail of a frame is as long as the chains that are started last).
      const unsigned per2 = 2u * mx / probe_spp, sum2 = 2u * sum / probe_spp;  // normalised to 2 probe samples
      if (per2 >= 56u) key[tile] = 2048u + ((per2 < 127u ? per2 : 127u) << 4) + ((sum2 >> 6) < 15u ? (sum2 >> 6) : 15u);
      else key[tile] = sum2 < 2047u ? sum2 : 2047u;
    }
    work[tile] = sum > 0 ? sum : 1u;
    // histogram of the counting sort (tile_order_kernel scans it, tile_scatter_kernel places the tiles): per key, tiles and probed work
    const unsigned b = key[tile] < (unsigned)kCostBins ? key[tile] : (unsigned)kCostBins - 1;
    atomicAdd(ghist + b, 1u);
    atomicAdd(gwork + b, (unsigned long long)(sum > 0 ? sum : 1u));
  }
}

// Counting sort of the tiles by descending key (longest-processing-time-first: a pixel is a sequential chain of spp
// samples, so the long chains must start at t = 0).  One workgroup; the order of equal-key tiles is irrelevant (the
// schedule never changes a pixel's value).  Two cuts of the sorted list, both by probed work:
//  * split_frac > 0: the first K tiles carry split_frac of the work (the longest chains: they go to coop_pixel_kernel,
//    one wave per pixel); the lane kernel's counter is started at tile K, K is written for the wave kernel to read;
//  * tail_frac: the last tiles, carrying tail_frac of the lane kernel's work, form region B of the lane kernel's
//    schedule: sched[0] = first index of B, sched[1] = B's work counter (started there);
//  * sched[2] = hot_chain x the probed total: the chain length (bounce iterations) from which a pixel is HOT (priority 3).
__global__ __launch_bounds__(1024) void tile_order_kernel(const unsigned* ghist, const unsigned long long* gwork, unsigned* goffs, int n_tiles,
                                                          float split_frac, unsigned long long* split_out,
                                                          unsigned long long* lane_counter, float tail_frac, float hot_chain,
                                                          unsigned long long* sched, const MigSchedule mig) {
  // (one workgroup, but only over the 4096 bins: the per-tile passes on either side -- histogram in tile_key_kernel, placement
  // in tile_scatter_kernel -- run on the whole machine; round 2 did all three here in 0.49 ms at 1080p)
  __shared__ unsigned hist[kCostBins];
  __shared__ unsigned offs[kCostBins];
  __shared__ unsigned long long bin_work[kCostBins];
  for (int i = threadIdx.x; i < kCostBins; i += blockDim.x) { hist[i] = ghist[i]; bin_work[i] = gwork[i]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned run = 0;
    unsigned long long total = 0;
    for (int b = kCostBins - 1; b >= 0; --b) {  // descending key
      offs[b] = run;
      run += hist[b];
      total += bin_work[b];
    }
    // first k tiles of the order that carry `target` work (tiles of one bin count with the bin's mean)
    auto tiles_for = [&](unsigned long long target) {
      unsigned long long cum = 0;
      unsigned k = 0;
      for (int b = kCostBins - 1; b >= 0 && cum < target; --b) {
        if (hist[b] == 0) continue;
        const unsigned long long mean = (bin_work[b] + hist[b] - 1) / hist[b];
        const unsigned long long want = (target - cum + mean - 1) / mean;
        const unsigned take = want < hist[b] ? (unsigned)want : hist[b];
        k += take;
        cum += mean * take;
      }
      return k;
    };
    unsigned k_split = 0;
    unsigned long long lane_work = total;
    if (split_out != nullptr) {
      const unsigned long long target = (unsigned long long)((double)split_frac * (double)total);
      k_split = tiles_for(target);
      *split_out = k_split;
      *lane_counter = (unsigned long long)k_split * kTilePixels;
      lane_work = total > target ? total - target : 0;
    }
    if (sched != nullptr) {
      const unsigned long long front = total - (unsigned long long)((double)tail_frac * (double)lane_work);
      unsigned k_tail = tiles_for(front);
      if (k_tail > (unsigned)n_tiles) k_tail = (unsigned)n_tiles;
      // Region A must not be empty: a slow-slot wave takes from B only while the front waves are still inside A, and with
      // an empty A (tail_frac >= 1, or a split that takes all of it) no wave would ever fetch B when every wave of the
      // launch sits in a slow slot -- tiles never rendered.  Then there is no region B: everything is A, and the
      // slow-slot waves turn into front waves at their first fetch (integrate_kernel: "B ran dry").
      if (k_tail <= k_split) k_tail = (unsigned)n_tiles;
      sched[0] = (unsigned long long)k_tail * kTilePixels;
      sched[1] = (unsigned long long)k_tail * kTilePixels;
      // hot chains: hot_chain x the probed total, scaled by the host to bounce iterations of the frame
      sched[2] = hot_chain > 0.0f ? (unsigned long long)(hot_chain * (float)total) + 1ull : 0ull;
    }
    if (mig.mig != nullptr) {
      // Chain hand-off (integrate_kernel / serve_chains).  l_avg = bounce iterations an average lane runs in this frame.
      // The longest chains (glass: ~34 queries per sample whatever the frame) are a fixed number of iterations, so the
      // smaller l_avg -- a small frame, a row shard of a multi-GPU job -- the larger their share of the frame time.
      // A chain is handed over once its projected length exceeds the (adaptive) threshold: push_theta x l_avg at first.
      const float l_avg = (float)total * mig.lavg_scale;
      // push threshold: push_theta x l_avg, but never below chain_theta x the frame's MEAN chain -- on a frame with fewer
      // pixels than lanes l_avg says nothing about how long a chain is
      const float mean_chain = (float)total * mig.chain_scale;
      float push = mig.push_theta * l_avg;
      if (push < mig.chain_theta * mean_chain) push = mig.chain_theta * mean_chain;
      if (push < 64.0f) push = 64.0f;
      // the adaptive threshold starts at `push` and moves between it ... and the length from which a chain cannot finish in a lane
      // before the frame does (floor_theta x l_avg; same floor from the mean chain as above)
      float fl = mig.floor_theta * l_avg;
      if (fl < mig.chain_theta * mean_chain) fl = mig.chain_theta * mean_chain;
      if (fl < 64.0f) fl = 64.0f;
      if (fl > push) fl = push;
      // Dedicated servers.  The share of a frame's work that sits in chains above the threshold is a property of the scene
      // (glass: ~1 %), and serving it costs 64 x that share x (server bounce / lane bounce = ~3 us / 16 us) of the machine
      // whatever the frame size: srv_frac of the workgroups start as servers whenever a chain of this frame CAN reach the
      // threshold at all (a sample has at most max_depth queries; the longest chains of a scene with glass run at ~0.7 of
      // that).  Servers that find nothing to do turn into lane waves after `mig_patience` (serve_chains), so a scene without
      // long chains pays ~patience x srv_frac once.
      float frac = mig.srv_frac;
      if ((float)mig.spp * 0.7f * (float)mig.max_depth < fl) frac = mig.srv_min_frac;
      int n_srv = (int)(frac * (float)mig.blocks + 0.999f);
      // waves that never get a tile (frames with fewer tiles than waves) are servers from the start anyway
      const int free_wgs = (mig.blocks * (kThreads / 64) - n_tiles) / (kThreads / 64);
      if (free_wgs > 0) n_srv = n_srv > free_wgs ? n_srv - free_wgs : 0;
      if (n_srv > mig.blocks - 1) n_srv = mig.blocks - 1;
      if (n_srv < 0) n_srv = 0;
      mig.mig[kMigTCounterDry] = ~0ull;
      mig.mig[kMigSrvWgs] = (unsigned long long)n_srv;
      mig.mig[kMigLaneWaves] = (unsigned long long)(mig.blocks - n_srv) * (kThreads / 64);
      mig.mig[kMigPush] = (unsigned long long)push;
      mig.mig[kMigPushFloor] = (unsigned long long)fl;
      mig.mig[kMigPushNow] = (unsigned long long)push;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kCostBins; i += blockDim.x) goffs[i] = offs[i];
}

// placement pass of the counting sort: tile i goes to the next free position of its key's range
__global__ __launch_bounds__(256) void tile_scatter_kernel(const unsigned* key, unsigned* goffs, unsigned* order, int n_tiles) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n_tiles) return;
  const unsigned b = key[i] < (unsigned)kCostBins ? key[i] : (unsigned)kCostBins - 1;
  order[atomicAdd(goffs + b, 1u)] = (unsigned)i;
}

// canvas.nim:47-54
__global__ __launch_bounds__(256) void finalize_kernel(double* pixels, long long n_values, double scale,
                                                        double gamma) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_values) pixels[i] = pow_pos(scale * pixels[i], gamma);
}

// io/ppm.nim:15-16 ; safe_math.nim:10-14
__global__ __launch_bounds__(256) void quantize_kernel(const double* pixels, long long n_values, uint8_t* out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_values) {
    double c = pixels[i];
    double cl = (c < 0.0) ? 0.0 : ((c > 0.999) ? 0.999 : c);
    out[i] = (uint8_t)(int)(256 * cl);
  }
}

// ---------------------------------------------------------------------------------------------
// Per-frame output stage of the animation driver (trace_of_radiance_animation.nim:186-196), fused:
//   Canvas -> RGB8        io/rgb.nim:17-31   uint8(256 * clamp(c, 0, 0.999)), top scanline first.
//                         (rgb.nim:29-31 indexes canvas[nrows - i, j], one row past the end for
//                         i = 0; the intended flip canvas[nrows - 1 - i, j] is implemented here.)
//   RGB8 -> Y'CbCr 4:2:0  io/color_conversions.nim:180-252, BT.601 fixed point:
//                         kr,kg,kb = 77,150,29 (>>8); y_scale = 110 (>>7), y_min = 16; fb = 127, fr = 160 (>>8)
//   planes -> I_PCM slice io/h264.nim:189-259: slice header, per macroblock [0x0d 0x00 except the
//                         first] + 256 Y + 64 Cb + 64 Cr raw bytes, stop byte 0x80.
// One workgroup per 16x16 macroblock, one thread per pixel.  Integer and byte work, HBM-bound:
// 24 B read and 1.5 B written per pixel.
__global__ __launch_bounds__(256) void encode_ipcm_kernel(const double* pixels, int nrows, int ncols, uint8_t* out,
                                                          uint8_t* plane_y, uint8_t* plane_cb, uint8_t* plane_cr) {
  __shared__ short s_u[256], s_v[256];
  const int mb_cols = (ncols + 15) >> 4;
  const int mb = blockIdx.x;
  const int mi = mb / mb_cols, mj = mb - mi * mb_cols;
  const int x = threadIdx.x >> 4, y = threadIdx.x & 15;      // row, column inside the macroblock
  const int vr = mi * 16 + x, vc = mj * 16 + y;              // video row (0 = top), column
  // a size that is not a multiple of 16: the last macroblock row / column is padded by edge replication (the SPS crops it)
  const int sr = vr < nrows ? vr : nrows - 1, sc = vc < ncols ? vc : ncols - 1;
  const bool inside = vr < nrows && vc < ncols;
  const double* px = pixels + ((size_t)(nrows - 1 - sr) * ncols + sc) * 3;
  int rgb[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double v = px[c];
    const double cl = (v < 0.0) ? 0.0 : ((v > 0.999) ? 0.999 : v);  // safe_math.nim:10-14
    rgb[c] = (int)(uint8_t)(int)(256 * cl);
  }
  const int tY = (77 * rgb[0] + 150 * rgb[1] + 29 * rgb[2]) >> 8;     // color_conversions.nim:218-220
  const uint8_t Y = (uint8_t)(((tY * 110) >> 7) + 16);                // :223
  s_u[threadIdx.x] = (short)(rgb[2] - tY);                            // :221
  s_v[threadIdx.x] = (short)(rgb[0] - tY);                            // :222
  const size_t data = 9 + (size_t)mb * 386;                           // first payload byte of this macroblock
  out[data + threadIdx.x] = Y;
  if (plane_y && inside) plane_y[(size_t)vr * ncols + vc] = Y;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cx = threadIdx.x >> 3, cy = threadIdx.x & 7;
    const int b0 = (2 * cx) * 16 + 2 * cy;
    const int tU = s_u[b0] + s_u[b0 + 1] + s_u[b0 + 16] + s_u[b0 + 17];
    const int tV = s_v[b0] + s_v[b0 + 1] + s_v[b0 + 16] + s_v[b0 + 17];
    const uint8_t U = (uint8_t)((((tU >> 2) * 127) >> 8) + 128);      // :249
    const uint8_t V = (uint8_t)((((tV >> 2) * 160) >> 8) + 128);      // :250
    out[data + 256 + threadIdx.x] = U;
    out[data + 320 + threadIdx.x] = V;
    const size_t cpos = (size_t)(mi * 8 + cx) * (ncols >> 1) + (mj * 8 + cy);
    const bool cinside = mi * 8 + cx < (nrows >> 1) && mj * 8 + cy < (ncols >> 1);
    if (plane_cb && cinside) plane_cb[cpos] = U;
    if (plane_cr && cinside) plane_cr[cpos] = V;
  }
  if (threadIdx.x == 0) {
    if (mb == 0) {  // h264.nim:38: constant slice header (start code, IDR slice NAL, I_PCM first macroblock)
      const uint8_t hdr[9] = {0x00, 0x00, 0x00, 0x01, 0x05, 0x88, 0x84, 0x21, 0xa0};
      for (int k = 0; k < 9; ++k) out[k] = hdr[k];
    } else {        // h264.nim:39,191-192: mb_type I_PCM for every further macroblock
      out[data - 2] = 0x0d;
      out[data - 1] = 0x00;
    }
    if (mb == (int)gridDim.x - 1) out[data + 384] = 0x80;  // h264.nim:40,259: slice stop bit
  }
}

// Multi-GPU assembly: the shards arrive rank-major and compact; put every row at its image position.
// One thread per float64 value; HBM-bound copy (48 B per pixel).
__global__ __launch_bounds__(256) void gather_rows_kernel(const double* gathered, double* frame, int nrows, int ncols,
                                                          int row_tile, int shard_count, long long shard_stride) {
  const long long row_values = (long long)ncols * 3;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)nrows * row_values) return;
  const int row = (int)(i / row_values);
  const long long within_row = i - (long long)row * row_values;
  const int tile = row / row_tile;
  const int shard = tile % shard_count;
  const int local_row = (tile / shard_count) * row_tile + (row - tile * row_tile);
  frame[i] = gathered[(long long)shard * shard_stride + (long long)local_row * row_values + within_row];
}

// keeps its stream busy until the host sets *flag (or max_ticks pass): the stand-in for a collective that never completes
__global__ void spin_until_kernel(volatile unsigned* flag, unsigned long long max_ticks) {
  const unsigned long long t0 = wall_clock64();
  while (*flag == 0u && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(127);
}

__global__ void selftest_kernel(int op, const double* x, const double* y, double* out0, double* out1,
                                long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  selftest_math_one(op, x[i], y ? y[i] : 0.0, out0[i], out1 ? out1[i] : out0[i]);
}

// ---------------------------------------------------------------------------------------
// host-side launchers (called from tor_api.cpp)
// ---------------------------------------------------------------------------------------
// variant table: [seeding 0|1][arith 0|1][W 2|3][f32 0|1][blocks 0|1|2 (2: two-level layouts, cooperative variants only)].  The block-expansion code (an unrolled
// 8-object stage per lane) is what makes the 168-register variants spill; launches without TOR_ACCEL_BLOCKS
// use kernels compiled without it (no scratch traffic at all).
typedef void (*IntegrateFn)(const KParams);
static IntegrateFn integrate_variant(int seeding, int arith, int w, int f32, int blocks) {
#define TOR_V(S, A, W, F, B) if (seeding == S && arith == A && w == W && f32 == F && blocks == B) return integrate_kernel<S, A, W, F, B>;
#define TOR_V4(S, A, W) TOR_V(S, A, W, 0, 0) TOR_V(S, A, W, 0, 1) TOR_V(S, A, W, 1, 0) TOR_V(S, A, W, 1, 1) TOR_V(S, A, W, 1, 2)
  TOR_V4(0, 0, 2) TOR_V4(0, 1, 2) TOR_V4(1, 0, 2) TOR_V4(1, 1, 2)
  TOR_V4(0, 0, 3) TOR_V4(0, 1, 3) TOR_V4(1, 0, 3) TOR_V4(1, 1, 3)
  TOR_V4(2, 0, 3)   // cost probe of the SEED_PIXEL tile schedule
  // arith 2: the reference's arithmetic behind the conservative FMA screen (brute-force layouts only)
  TOR_V(0, 2, 2, 0, 0) TOR_V(0, 2, 3, 0, 0) TOR_V(1, 2, 2, 0, 0) TOR_V(1, 2, 3, 0, 0) TOR_V(2, 2, 3, 0, 0)
#undef TOR_V4
#undef TOR_V
  return nullptr;
}

static int clamp_w(int waves_per_simd) {
  // register budget follows the launch shape: 2 workgroups/CU -> 256 VGPRs, 3 (or more) -> 168
  return waves_per_simd <= 2 ? 2 : 3;
}

static int wants_f32(const KParams& p) { return (p.hot32 != nullptr || p.shot32 != nullptr) ? 1 : 0; }
static int wants_blocks(const KParams& p) { return p.bnd != nullptr ? ((p.two_level != 0 && wants_f32(p) != 0) ? 2 : 1) : 0; }
static size_t dynamic_lds(const KParams& p) {
  return (size_t)wave_lds_bytes(wants_blocks(p), coop_variant(wants_f32(p), wants_blocks(p))) * (kThreads / 64) + (size_t)p.shot_lds_doubles * 8 + (size_t)p.shot32_lds_floats * 4 +
         (size_t)p.bnd32_lds_floats * 4;
}

// arith as the caller asked (0 strict, 1 fused) -> the kernel variant: strict launches of the brute-force layouts run behind
// the conservative FMA screen (variant 2: the same canvas bit for bit) unless the context turned it off (KParams::screen)
static int arith_variant(const KParams& p, int arith) {
  return (arith == 0 && p.screen != 0 && wants_f32(p) == 0 && wants_blocks(p) == 0) ? 2 : arith;
}

hipError_t launch_probe(const KParams& p, int blocks, hipStream_t stream) {
  IntegrateFn fn = integrate_variant(2, arith_variant(p, 0), 3, wants_f32(p), wants_blocks(p));
  if (!fn) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kThreads), dynamic_lds(p), stream, p);
  return hipGetLastError();
}

hipError_t launch_tile_order(const unsigned* pixel_cost, unsigned n_pixels, unsigned* key, unsigned* work, unsigned* order, int n_tiles,
                             float split_frac, unsigned long long* split_out, unsigned long long* lane_counter, float tail_frac,
                             float hot_chain, unsigned long long* sched, const MigSchedule& mig, hipStream_t stream) {
  // sort scratch behind the per-tile arrays (tor_api.cpp sizes the buffer): histogram (tiles, work per key), running offsets
  unsigned long long* gwork = (unsigned long long*)(((uintptr_t)(work + n_tiles) + 7) & ~(uintptr_t)7);
  unsigned* ghist = (unsigned*)(gwork + kCostBins);
  unsigned* goffs = ghist + kCostBins;
  hipError_t e = hipMemsetAsync(gwork, 0, (size_t)kCostBins * (8 + 4), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(tile_key_kernel, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, stream, pixel_cost, n_pixels, n_tiles, key, work,
                     (unsigned)mig.key_mode, (unsigned)(mig.probe_spp > 0 ? mig.probe_spp : 2), ghist, gwork);
  hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, stream, (const unsigned*)ghist, (const unsigned long long*)gwork, goffs, n_tiles,
                     split_frac, split_out, lane_counter, tail_frac, hot_chain, sched, mig);
  hipLaunchKernelGGL(tile_scatter_kernel, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, stream, (const unsigned*)key, goffs, order, n_tiles);
  return hipGetLastError();
}

bool integrate_variant_serves_chains(const KParams& p, int seeding) {
  return migrate_variant(seeding, wants_f32(p), wants_blocks(p)) && p.bnd32 != nullptr && p.shot32 != nullptr;
}

hipError_t launch_integrate(const KParams& p, int seeding, int arith, int waves_per_simd, int blocks,
                            hipStream_t stream) {
  IntegrateFn fn = integrate_variant(seeding, arith_variant(p, arith), clamp_w(waves_per_simd), wants_f32(p), wants_blocks(p));
  if (!fn) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kThreads), dynamic_lds(p), stream, p);
  return hipGetLastError();
}

int integrate_blocks_per_cu(const KParams& p, int seeding, int arith, int waves_per_simd) {
  IntegrateFn fn = integrate_variant(seeding, arith_variant(p, arith), clamp_w(waves_per_simd), wants_f32(p), wants_blocks(p));
  int n = 0;
  if (!fn || hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, kThreads, dynamic_lds(p)) != hipSuccess || n < 1) n = 1;
  return n;
}

// One wave per pixel (TOR_SEED_PIXEL, small frames).  Returns hipErrorInvalidValue when the objects do not fit LDS.
size_t coop_lds_bytes(int coop_slots) { return (size_t)kCoopArrays * 8 * (size_t)coop_slots; }

int coop_blocks_per_cu(const KParams& p, int arith) {
  int n = 0;
  auto fn = arith ? coop_pixel_kernel<1> : coop_pixel_kernel<0>;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, kThreads, coop_lds_bytes(p.coop_slots)) != hipSuccess) n = 0;
  return n;
}

hipError_t launch_coop(const KParams& p, int arith, int blocks, hipStream_t stream) {
  auto fn = arith ? coop_pixel_kernel<1> : coop_pixel_kernel<0>;
  const size_t lds = coop_lds_bytes(p.coop_slots);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kThreads), lds, stream, p);
  return hipGetLastError();
}

int integrate_fixed_lds_bytes(int blocks, int coop) { return wave_lds_bytes(blocks, coop) * (kThreads / 64); }

hipError_t launch_finalize(double* pixels, long long n_values, double scale, double gamma, hipStream_t stream) {
  if (n_values <= 0) return hipSuccess;
  unsigned blocks = (unsigned)((n_values + 255) / 256);
  hipLaunchKernelGGL(finalize_kernel, dim3(blocks), dim3(256), 0, stream, pixels, n_values, scale, gamma);
  return hipGetLastError();
}

hipError_t launch_quantize(const double* pixels, long long n_values, uint8_t* out, hipStream_t stream) {
  if (n_values <= 0) return hipSuccess;
  unsigned blocks = (unsigned)((n_values + 255) / 256);
  hipLaunchKernelGGL(quantize_kernel, dim3(blocks), dim3(256), 0, stream, pixels, n_values, out);
  return hipGetLastError();
}

hipError_t launch_encode_ipcm(const double* pixels, int nrows, int ncols, uint8_t* out, uint8_t* plane_y,
                              uint8_t* plane_cb, uint8_t* plane_cr, hipStream_t stream) {
  const int n_mb = ((nrows + 15) >> 4) * ((ncols + 15) >> 4);
  if (n_mb <= 0) return hipSuccess;
  hipLaunchKernelGGL(encode_ipcm_kernel, dim3((unsigned)n_mb), dim3(256), 0, stream, pixels, nrows, ncols, out, plane_y,
                     plane_cb, plane_cr);
  return hipGetLastError();
}

hipError_t launch_gather_rows(const double* gathered, double* frame, int nrows, int ncols, int row_tile, int shard_count,
                              long long shard_stride, hipStream_t stream) {
  const long long n = (long long)nrows * ncols * 3;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, gathered, frame, nrows, ncols,
                     row_tile, shard_count, shard_stride);
  return hipGetLastError();
}

hipError_t launch_spin_until(volatile unsigned* flag, unsigned long long max_ticks, hipStream_t stream) {
  hipLaunchKernelGGL(spin_until_kernel, dim3(1), dim3(64), 0, stream, flag, max_ticks);
  return hipGetLastError();
}

hipError_t launch_selftest(int op, const double* x, const double* y, double* out0, double* out1, long long n,
                           hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(selftest_kernel, dim3(blocks), dim3(256), 0, stream, op, x, y, out0, out1, n);
  return hipGetLastError();
}

}  // namespace tor
