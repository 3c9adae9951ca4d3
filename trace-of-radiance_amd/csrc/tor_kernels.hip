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
//                      expensive tiles of a mid-size frame while integrate_kernel renders the rest (DESIGN 4.7 (HISTORY 4.7-4.8)
//   tile_order_kernel  counting sort of the SEED_PIXEL tiles by probed cost (LPT schedule) + the split point
//   gather_rows_kernel multi-GPU assembly: rank-major row shards -> frame in image order
//   finalize_kernel    canvas.nim:47-54 (draw): pow(sum * 1/spp, 1/gamma)
//   quantize_kernel    io/ppm.nim:15-16
//
// float64 throughout, no FMA contraction (-ffp-contract=off): the only fused operations are the explicit fma() of the
// conservative screens (tor_screen.hpp, tor_filter32.hpp), whose candidates are all re-tested with the reference's operations.
//
// Layout of the source (round 4; the pieces are TEXTUAL includes -- one translation unit, the same code the single file
// compiled to, checked by comparing the generated ISA):
//   tor_kernels.hip                      shared device helpers, the skeleton of integrate_kernel (lane state, roles, the bounce
//                                        loop, the arbiter priorities, the per-query set-up), the host-side launchers
//   kernel/integrate_refill.inc          (A) work distribution + camera ray
//   kernel/integrate_loop_*.inc          (B) the wave-uniform object loop, one file per segment family:
//                                        plane (ARITH 2, common-height segments: the plane screen for every object, then the
//                                        second form per lane on what it keeps -- DESIGN 4.2), screen2 (ARITH 2 second-form
//                                        records, wave-uniform -- 4.12), f64_static, f64_movers, f32 (TOR_ACCEL_F32),
//                                        boxes32 / boxes64 (TOR_ACCEL_BLOCKS)
//   kernel/integrate_resolve_lane.inc    exact float64 tests of the candidates, per lane
//   kernel/integrate_resolve_coop.inc    ... pooled over the wave (TOR_ACCEL_BLOCKS | TOR_ACCEL_F32)
//   kernel/integrate_shade.inc           (C) sky / scatter / end of a sample
//   kernel/integrate_handoff_push.inc    SEED_PIXEL chain hand-off, lane side
//   kernel/integrate_deposit.inc         SEED_SAMPLE accumulation
//   kernel/chain_servers.hpp             serve_chains (hand-off, server side), kernel/coop_pixel.hpp (one wave per pixel),
//   kernel/tile_sort.hpp                 SEED_PIXEL tile schedule, kernel/output_stage.hpp (finalize, quantize, I_PCM, gather, ...)
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

// One ray/object discriminant (the reference's unfused operations).  Returns the sign-bit filter word: negative
// (bit 31 set) iff disc has a clear sign bit (disc >= +0 or NaN+) and (half_b < 0 or c < 0),
// a superset of the objects the reference's hit() can accept (both roots are <= 0 when
// half_b >= 0 and c >= 0).  One v_bitop3_b32: f(a,b,c) = (a|b) & ~c  -> truth table 0x54.
__device__ __forceinline__ int disc_filter(double ox, double oy, double oz, double dx, double dy,
                                           double dz, double a, double cx, double cy, double cz,
                                           double r2) {
  double ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
  const double hb = ocx * dx + ocy * dy + ocz * dz;          // spheres.nim:31
  const double cc = (ocx * ocx + ocy * ocy + ocz * ocz) - r2;  // spheres.nim:32
  const double disc = hb * hb - a * cc;                      // spheres.nim:33
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
#ifdef TOR_FINE_PROBE
// tools/fine_probe.py's build (make fine): eight more cycle counters inside the object loop and the pooled resolve of the cooperative
// variants; they leave the kernel through the wave log's wall-clock words (which that build therefore does not report)
constexpr int kProfSlots = 24;
#else
constexpr int kProfSlots = 16;  // debug counters (u64, stats / wave log only): 5 section sums, trips, begin, last stamp, 4 statistics, 3 wave-log stamps
#endif
// cooperative resolve (F32 && BLOCKS variants): pair list and survivor list (64 carried over + 512 new per trip), the
// per-ray closest hit {t bits, (original index, slot)}
constexpr int kCoopList = 576;
// (+ the paths' attenuation: 3 x 64 float64 -- touched once per bounce, so it lives in LDS, not in 6 of the 168 registers)
constexpr int coop_bytes(int blocks) { return (blocks ? 2 : 1) * kCoopList * 4 + 64 * 2 * 8 + 3 * 64 * 8; }  // (no pair list without boxes)
// Camera-ray reservoir (TOR_SEED_SAMPLE, float64 brute force: kernel/integrate_refill.inc): 64 started samples per wave --
// generator state after the camera ray (4 x u64), lens offsets, film coordinates, time (5 x f64), pixel as a byte offset from
// the batch's first pixel (+ that pixel, 16-byte padded)
constexpr bool reservoir_variant(int seeding, int f32, int blocks) { return seeding == 1 && f32 == 0 && blocks == 0; }
constexpr int kResBytes = 9 * 64 * 8 + 64 + 16;
constexpr int wave_lds_bytes(int blocks, int coop = 0, int res = 0) {
  return queue_cap(blocks) * 64 * 4 + kAccSlots * 3 * 8 + kAccSlots * 4 + kProfSlots * 8 + (coop ? coop_bytes(blocks) + 64 : 0) + (res ? kResBytes : 0);
}
// Which kernel variants resolve cooperatively: TOR_ACCEL_BLOCKS | TOR_ACCEL_F32.  (The code also runs the variants
// without boxes -- `blocks == 0 || f32 != 0` passes every parity test -- but there the candidates are few (1.25-1.43
// per query): a pooled trip costs twice a per-lane trip, the resolve share stays at 5-6 % and the extra LDS and
// registers cost the float64 brute force 2 % (C3 1190 -> 1163 Msamples/s).  Measured, not kept.)
constexpr bool coop_variant(int f32, int blocks) { return blocks != 0 && f32 != 0; }
static_assert(wave_lds_bytes(0) % 16 == 0 && wave_lds_bytes(1) % 16 == 0 && wave_lds_bytes(1, 1) % 16 == 0 && wave_lds_bytes(0, 1) % 16 == 0 && kResBytes % 16 == 0,
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
  constexpr bool kRes = reservoir_variant(SEEDING, F32, BLOCKS);
  constexpr int kWaveLdsBytes = wave_lds_bytes(BLOCKS, coop_variant(F32, BLOCKS), kRes);
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
  // camera-ray reservoir (kRes variants: no cooperative state, so it starts where that would)
  unsigned long long* res_rng = reinterpret_cast<unsigned long long*>(coop_base);  // [4][64] generator state after the camera ray
  double* res_f = reinterpret_cast<double*>(coop_base + 4 * 64 * 8);               // [5][64] lens offset x, y; film s, t; time
  unsigned char* res_dp = coop_base + 9 * 64 * 8;                                  // [64] pixel - first pixel of the batch
  unsigned* res_pl0 = reinterpret_cast<unsigned*>(coop_base + 9 * 64 * 8 + 64);    // first pixel of the batch

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
      // block boxes of two-level scenes, TRANSPOSED on the way in: child j of super box s at ((j * n_super + s) * 6 .. + 6) -- 24 bytes,
      // read as three ds_read_b64.  In the table's own order (s * 64 + j * 8 floats) the 64 lanes of a pooled expansion -- 64
      // different super boxes, the same child j -- all start in ONE bank: 58 % of the LDS cycles of the 1601-sphere animation
      // frames were bank conflicts (round 6: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, 28 % on random_scene, 17 % unstaged)
      const int n_sup_lds = p.bnd32_lds_floats / (8 * kBlock);
      for (int k = threadIdx.x; k < p.bnd32_lds_floats; k += kThreads) {
        const int c = k & 7, b = k >> 3;
        if (c < 6) stage_b32[((b & 7) * n_sup_lds + (b >> 3)) * 6 + c] = p.bnd32[k];
      }
    } else {
      for (int k = threadIdx.x; k < p.shot_lds_doubles; k += kThreads) stage[k] = p.shot[k];
    }
    __syncthreads();
  }

  // ARITH 2: the second-form records of the screen for stage two of the plane-screened segments (kernel/integrate_loop_plane.inc)
  const bool xstaged = ARITH == 2 && p.xrec_lds_doubles > 0;
  if (ARITH == 2 && xstaged) {
    for (int k = threadIdx.x; k < p.xrec_lds_doubles; k += kThreads) stage[k] = p.xrec[k];
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
  unsigned res_head = 0, res_cnt = 0;  // kRes: the camera-ray reservoir's next entry and fill (kernel/integrate_refill.inc)
  const unsigned prio_slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);  // HW_ID.wave_id: the wave's slot in its SIMD
  unsigned prio_now = 0;
  // a pixel chain is HOT when, extrapolated from its samples so far, it needs more than hot_iters bounce iterations
  const unsigned hot_iters = (SEEDING == 0 && p.sched != nullptr) ? (unsigned)p.sched[0] : 0u;
  unsigned pix_iters = 0;  // bounce iterations the lane has spent on its current pixel
  // arbiter priorities (below): every SEED_PIXEL variant.  (In the cooperative variants the lane's iteration counter was
  // one register too many while the pixel sum still lived in registers: 1-2 % slower then, 1-2 % faster at 100 spp now.)
  constexpr bool kPrio = SEEDING == 0;
  bool exhausted = false;
  // ---- chain hand-off (DESIGN 4.7 (HISTORY 4.10)) -------------------------------------------------------------------------------
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
         kStQueries, kStCand, kStIters, kStSamples, kLogStart, kLogExhausted, kLogItersAtExhaustion, kSecStage2,
         // (TOR_FINE_PROBE builds only) the mark and seven interval sums: ray set-up, box loop, direct float32 segments, segment headers /
         // loop control, list building (A), block expansion (B), exact tests (C)
         kFineMark, kFineSetup, kFineBoxes, kFineF32, kFineHdr, kFineA, kFineB, kFineC };
  // objects tested inside expanded blocks (TOR_ACCEL_BLOCKS; part of kStCand): shares the slot of stage two's cycle counter, which
  // only the ARITH 2 variants -- brute-force layouts, no blocks -- use
  constexpr int kStBlock = kSecStage2;
  static_assert(kSecStage2 < kProfSlots, "debug counters");
#ifdef TOR_FINE_PROBE
  static_assert(kFineC < kProfSlots, "debug counters");
  int fine_state = kFineA;
#define TOR_FINE(slot)                                            \
  if (prof && lane == 0) {                                        \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    prof_lds[slot] += now_ - prof_lds[kFineMark];                 \
    prof_lds[kFineMark] = now_;                                   \
  }
#else
#define TOR_FINE(slot)
#endif
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
#include "kernel/integrate_refill.inc"
    TOR_SEC(kSecRefill)
    TOR_FINE(kFineMark)
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
      //  1-2 everybody else, taking turns (level = 1 + (slot + clock phase) mod 2, the phase from the shader clock the slots of
      //     a SIMD share) so that they get the same service.
      if (active) pix_iters += 1;
      const bool lane_hot = active && hot_iters != 0 && pix_iters >= 64u &&
                            (unsigned long long)pix_iters * (unsigned)p.spp >= (unsigned long long)hot_iters * (unsigned)(s + 1);
      const bool wave_hot = ballot64(lane_hot) != 0;
      unsigned level = 1;
      if (wave_hot) {
        level = 3;
      } else if (p.prio_shift > 0) {
        const unsigned phase = (unsigned)(__builtin_readcyclecounter() >> p.prio_shift);
        level = 1u + (prio_slot + phase) % (WAVES_PER_SIMD < 2 ? 1u : 2u);
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
      const double a = a_strict;
      double best_t = __builtin_inf();
      int best_idx = -1;
      int best_orig = 0x7fffffff;
      double best_f = 0.0;
      // ARITH 2 (conservative FMA screen, screen_filter above): the ray's share of the margins
      double scr_s1 = 0.0, scr_d1 = 0.0, scr_negmu = 0.0, scr_am = 0.0;
      ScreenRay sray{};  // second form of the screen (tor_screen.hpp): the normalised direction, once per query
      PlaneRay pray{0.0, 0.0, 0.0, true};  // ... and the ground track's normal for the plane screen in front of it (launches with KParams.xpl)
      if (kScreen) {
        scr_s1 = __builtin_fabs(ox) + __builtin_fabs(oy) + __builtin_fabs(oz);
        scr_d1 = __builtin_fabs(dx) + __builtin_fabs(dy) + __builtin_fabs(dz);
        sray = screen2_ray(ox, oy, oz, dx, dy, dz, a_strict);
        if (p.xpl != nullptr) pray = plane_ray(sray);
      }

      // TOR_ACCEL_F32: the ray in float32, relative to the scene origin (used by segment kinds 5-7 only)
      RayF32 r32{};
      if (F32) r32 = make_ray_f32(ox, oy, oz, dx, dy, dz, a_strict, p.org[0], p.org[1], p.org[2]);

      BoxRay32 b32{};
      const bool boxes32 = F32 && BLOCKS && p.bnd32 != nullptr;
      if (boxes32) b32 = make_box_ray32(r32, p.sp_bmax, p.sp_hmin);

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
      TOR_FINE(kFineSetup)
      for (;;) {
        unsigned qn = 0;
        bool full = false;
        unsigned nz = 0;    // kWords: bit (w_count - 1 - k) <-> word k of this pass is not empty
        int w_count = 0;    // kWords: words stored in this pass (wave-uniform)
        int gb_next = 0;    // kWords: the block index behind the last block tested (wave-uniform)
        bool plane_ran = false;  // kWords: stage two of a plane-screened segment cleared bits of this pass's words (wave-uniform)
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
          // ---- the segment's header.  ARITH 2 (round 5): ONE 128-byte record per segment (KParams.xhdr, tor_scene.hpp) read up front
          // -- two s_load_dwordx16 issued together, one round trip through the scalar cache -- with the integers stored as integers.
          // Before, a plane-screened segment paid three DEPENDENT round trips (kind -> offsets, margins, gate inputs -> first record)
          // and a v_cvt_i32_f64 + v_readfirstlane per integer field; under the scalar traffic of twelve waves per CU walking
          // the plane table a dependent round trip is ~0.3 us of a 30 us bounce iteration (measured: one more of them per query,
          // for the gate of a 4-object segment, cost 2 %).  The header also carries the segment's FIRST plane record, so the loop's
          // first request (record 1 on) overlaps the per-segment set-up instead of following it.
          int seg_kind, seg_begin, seg_count_word, seg_block0, xkind = 0, h_xfirst = 0, h_plfirst = 0, h_gate = 0;
          double h_t0 = 0.0, h_dt = 1.0, h_reach = 0.0, h_travel = 0.0, h_y = 0.0, h_rmax2 = 0.0, h_sx = 0.0, h_sz = 0.0;
          double h_r0 = 0.0, h_r1 = 0.0, h_r2 = 0.0, h_r3 = 0.0;
          if constexpr (kScreen) {
            // (written as two s_load_dwordx16 by hand: left to itself the compiler sinks every field's load to its first use -- a
            // load per branch again -- and then carries some of the integers in vector registers, which turns the object loop's
            // record addresses, and with them its scalar loads, into per-lane vector loads: measured, 1.6 x slower)
            typedef int __attribute__((ext_vector_type(16))) i16v;
            const double* hp = p.xhdr + (size_t)seg * 16;
            i16v ha, hb;
            asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(ha), "=&s"(hb) : "s"(hp) : "memory");
            auto f64_of = [](int lo, int hi_) { return __hiloint2double(hi_, lo); };
            xkind = ha[0]; seg_kind = ha[1]; seg_count_word = ha[2]; seg_block0 = ha[3]; seg_begin = ha[4]; h_xfirst = ha[5]; h_plfirst = ha[6]; h_gate = ha[7];
            h_t0 = f64_of(ha[8], ha[9]); h_dt = f64_of(ha[10], ha[11]); h_reach = f64_of(ha[12], ha[13]); h_travel = f64_of(ha[14], ha[15]);
            h_y = f64_of(hb[0], hb[1]); h_rmax2 = f64_of(hb[2], hb[3]); h_sx = f64_of(hb[4], hb[5]); h_sz = f64_of(hb[6], hb[7]);
            h_r0 = f64_of(hb[8], hb[9]); h_r1 = f64_of(hb[10], hb[11]); h_r2 = f64_of(hb[12], hb[13]); h_r3 = f64_of(hb[14], hb[15]);
          } else {
            seg_kind = (int)segs[seg * 8 + 0];
            seg_begin = (int)segs[seg * 8 + 1];    // first hot record of the segment
            seg_count_word = (int)segs[seg * 8 + 2];
            seg_block0 = (int)segs[seg * 8 + 3];   // (first sorted index) / kBlock
          }
          const int seg_count = seg_count_word & 0xffffff;   // padded to kBlock
          const int seg_real = seg_count - (seg_count_word >> 24);  // kinds 0-2: the objects that exist (the last block's tail is padding)
          // (what the segment loops read of the header: from the record above in the ARITH 2 variants, from segs[] in the others --
          // kScreen is a compile-time constant, the dead arm emits nothing)
#define SEG_T0 (kScreen ? h_t0 : segs[seg * 8 + 4])
#define SEG_DT (kScreen ? h_dt : segs[seg * 8 + 5])
#define SEG_REACH (kScreen ? h_reach : segs[seg * 8 + 6])
#define SEG_TRAVEL (kScreen ? h_travel : segs[seg * 8 + 7])
          // ARITH 2: segments with second-form records (xkind 10 / 11 / 12; tor_screen.hpp: the quadratic expanded around the ray,
          // direction normalised per ray -- 8 / 6 / 9 float64 instructions per object, all but one fused multiply-adds against scalar
          // operands); the others (13 / 14: movers in general position; 0: degenerate time groups) keep the first form below
          // stage one (the plane screen, kernel/integrate_loop_plane.inc) in front of the segment's wave-uniform test: when the
          // launch carries the table and the band around the ground track is thin against the segment for most of the wave's rays
          // (tor_screen.hpp plane_pays; the same vote on every pass of a query: the rays do not change)
          // (the host settles the clear cases per segment -- h_gate 1: the band is thin whatever the track's direction, 0: never
          // (a segment of a few huge spheres), 2: it depends on the direction: the wave votes; random_scene has no segment of the
          // third kind.  A vote costs ~35 vector instructions with the scalar-register spills around it.)
          bool use_plane = false;
          if (kScreen && xkind >= 10 && p.xpl != nullptr) {
            use_plane = h_gate == 1 || p.plane_gate2 <= 0.0;
            if (h_gate == 2 && p.plane_gate2 > 0.0) {
              const bool pays = plane_pays(pray, p.plane_gate2, h_rmax2, h_sx, h_sz);
              use_plane = 2 * __builtin_popcountll(ballot64(pays)) > __builtin_popcountll(active_mask);
            }
          }
          if (kScreen && use_plane) {
#include "kernel/integrate_loop_plane.inc"
          } else if (kScreen && xkind >= 10 && xkind <= 12) {
#include "kernel/integrate_loop_screen2.inc"
          } else if (seg_kind == 0) {
#include "kernel/integrate_loop_f64_static.inc"
          } else if (F32 && seg_kind >= 5) {
#include "kernel/integrate_loop_f32.inc"
          } else if (F32 && BLOCKS && (seg_kind == 3 || seg_kind == 4)) {
#include "kernel/integrate_loop_boxes32.inc"
          } else if (BLOCKS && (seg_kind == 3 || seg_kind == 4)) {
#include "kernel/integrate_loop_boxes64.inc"
          } else {
#include "kernel/integrate_loop_f64_movers.inc"
          }
          if (full) break;
          seg += 1;
          i = 0;
        }
#undef SEG_T0
#undef SEG_DT
#undef SEG_REACH
#undef SEG_TRAVEL
#undef TOR_WORDS_END_BLOCK
        if (kWords) {
          if (!full && (gb_next & 3) != 0) {  // the list ended inside a word: its bits move up into place
            mw <<= (unsigned)(8 * (4 - (gb_next & 3)));
            q[(unsigned)w_count * 64] = mw;
            nz = push_cond(nz, mw != 0u);
            w_count += 1;
          }
          if (plane_ran) {  // stage two emptied words: the summary again, from the words as they are now
            // (all kQCap slots are read, back to back -- one LDS latency instead of one per word -- and the slots this pass did
            // not write, stale words of an earlier query, are shifted out)
            unsigned all = 0;
#pragma unroll
            for (int k2 = 0; k2 < kQCap; ++k2) all = push_cond(all, q[(unsigned)k2 * 64] != 0u);
            nz = all >> (unsigned)(kQCap - w_count);
          }
        }
        }  // if (active)

        TOR_SEC(kSecLoop)
        TOR_FINE(kFineHdr)
        if constexpr (kCoop) {
#include "kernel/integrate_resolve_coop.inc"
        } else if (active) {
#include "kernel/integrate_resolve_lane.inc"
        }  // per-lane resolve
        TOR_SEC(kSecResolve)
        TOR_FINE(kFineA)
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

#include "kernel/integrate_shade.inc"
      }  // if (active)
      if constexpr (kMigrate) {
#include "kernel/integrate_handoff_push.inc"
      }
    }

    TOR_SEC(kSecShade)
    if (SEEDING == 1) {
#include "kernel/integrate_deposit.inc"
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
      w[0] = prof_lds[kLogStart]; w[1] = wall_clock64();
      w[3] = prof_lds[kStQueries] | ((unsigned long long)__builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4) << 44);  // HW_ID[15:0]
      w[4] = prof_lds[kLogExhausted]; w[5] = (prof_lds[kLogItersAtExhaustion] & 0xffffffffull) | (prof_lds[kSecTrips] << 32);
      // six 21-bit fields in units of 4096 shader cycles: refill+camera, object loop, exact resolve | shade, deposit, total
      const unsigned long long total = __builtin_readcyclecounter() - prof_lds[kSecBegin];
      auto f21 = [](unsigned long long c) { c >>= 12; return c > 0x1fffffull ? 0x1fffffull : c; };
      w[2] = (prof_lds[kStIters] & 0xffffffffffull) | (ARITH == 2 ? f21(prof_lds[kSecStage2]) << 43 : 0ull);  // bounce iterations | stage two of the plane-screened segments (part of the object loop's field)
      w[6] = f21(prof_lds[kSecRefill]) | (f21(prof_lds[kSecLoop]) << 21) | (f21(prof_lds[kSecResolve]) << 42);
      w[7] = f21(prof_lds[kSecShade]) | (f21(prof_lds[kSecDeposit]) << 21) | (f21(total) << 42);
#ifdef TOR_FINE_PROBE
      auto f32u = [](unsigned long long c) { c >>= 8; return c > 0xffffffffull ? 0xffffffffull : c; };  // units of 256 shader cycles
      w[0] = f32u(prof_lds[kFineSetup]) | (f32u(prof_lds[kFineBoxes]) << 32);
      w[1] = f32u(prof_lds[kFineF32]) | (f32u(prof_lds[kFineHdr]) << 32);
      w[4] = f32u(prof_lds[kFineA]) | (f32u(prof_lds[kFineB]) << 32);
      w[5] = f32u(prof_lds[kFineC]) | (prof_lds[kSecTrips] << 32);
#endif
    }
    if (lane == 0) {
      atomicAdd(p.stats + 0, prof_lds[kStQueries]);
      atomicAdd(p.stats + 1, prof_lds[kStCand]);
      atomicAdd(p.stats + 2, prof_lds[kStIters]);
      atomicAdd(p.stats + 3, prof_lds[kStSamples]);
      if (ARITH != 2) atomicAdd(p.stats + 10, prof_lds[kStBlock]);
    }
  }
}

#include "kernel/coop_pixel.hpp"
#include "kernel/chain_servers.hpp"
#include "kernel/tile_sort.hpp"
#include "kernel/output_stage.hpp"
// ---------------------------------------------------------------------------------------
// host-side launchers (called from tor_api.cpp)
// ---------------------------------------------------------------------------------------
// variant table: [seeding 0|1|2 (2: the cost probe)][arith 0 | 2 (2: behind the FMA screen, brute-force layouts only)][W 2|3][f32 0|1]
// [blocks 0|1|2 (2: two-level layouts, cooperative variants only)].  The block-expansion code (an unrolled 8-object stage per lane)
// is what makes the 168-register variants spill; launches without TOR_ACCEL_BLOCKS use kernels compiled without it (no scratch
// traffic at all).  (Round 5: the 20 TOR_ARITH_FUSED instantiations -- `arith 1`, not the reference's rounding -- are gone.)
typedef void (*IntegrateFn)(const KParams);
static IntegrateFn integrate_variant(int seeding, int arith, int w, int f32, int blocks) {
#define TOR_V(S, A, W, F, B) if (seeding == S && arith == A && w == W && f32 == F && blocks == B) return integrate_kernel<S, A, W, F, B>;
#define TOR_V4(S, A, W) TOR_V(S, A, W, 0, 0) TOR_V(S, A, W, 0, 1) TOR_V(S, A, W, 1, 0) TOR_V(S, A, W, 1, 1) TOR_V(S, A, W, 1, 2)
  TOR_V4(0, 0, 2) TOR_V4(1, 0, 2)
  TOR_V4(0, 0, 3) TOR_V4(1, 0, 3)
  TOR_V4(2, 0, 3)   // cost probe of the SEED_PIXEL tile schedule
  // arith 2: the reference's arithmetic behind the conservative FMA screen (brute-force layouts only)
  TOR_V(0, 2, 2, 0, 0) TOR_V(0, 2, 3, 0, 0) TOR_V(1, 2, 2, 0, 0) TOR_V(1, 2, 3, 0, 0) TOR_V(2, 2, 3, 0, 0)
  // (round 5: a 128-register build <1, 2, 4, 0, 0> for a 4th workgroup per CU now runs 2.4 x SLOWER -- 1097 against 2613 Msamples/s
  // at configs[1]: stage two's per-lane state spills inside the loops; not built)
  // (a 128-register build of <1, 2, W, 0, 0> for a 4th workgroup per CU was measured in round 4: 1951 against 1961 Msamples/s at
  // configs[2] for 3 workgroups -- tools/wpc_sweep.py, profiles/r4_wpc_sweep.txt -- and is not built)
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
static size_t dynamic_lds(const KParams& p, int seeding) {
  return (size_t)wave_lds_bytes(wants_blocks(p), coop_variant(wants_f32(p), wants_blocks(p)), reservoir_variant(seeding, wants_f32(p), wants_blocks(p))) * (kThreads / 64) + (size_t)p.shot_lds_doubles * 8 + (size_t)p.shot32_lds_floats * 4 +
         (size_t)p.bnd32_lds_floats * 4 +
         // (the second-form table of stage two: only the ARITH 2 variants -- brute-force layouts behind the screen -- stage it)
         ((p.screen != 0 && wants_f32(p) == 0 && wants_blocks(p) == 0) ? (size_t)p.xrec_lds_doubles * 8 : (size_t)0);
}

// the kernel variant of a launch: the brute-force layouts run behind the conservative FMA screen (variant 2: the same canvas
// bit for bit) unless the context turned it off (KParams::screen); `arith` is TOR_ARITH_STRICT (0), the only arithmetic there is
static int arith_variant(const KParams& p, int arith) {
  return (arith == 0 && p.screen != 0 && wants_f32(p) == 0 && wants_blocks(p) == 0) ? 2 : 0;
}

hipError_t launch_probe(const KParams& p, int blocks, hipStream_t stream) {
  IntegrateFn fn = integrate_variant(2, arith_variant(p, 0), 3, wants_f32(p), wants_blocks(p));
  if (!fn) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kThreads), dynamic_lds(p, 2), stream, p);
  return hipGetLastError();
}

hipError_t launch_tile_order(const unsigned* pixel_cost, unsigned n_pixels, unsigned* key, unsigned* work, unsigned* order, int n_tiles,
                             float split_frac, unsigned long long* split_out, unsigned long long* lane_counter,
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
                     split_frac, split_out, lane_counter, hot_chain, sched, mig);
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
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kThreads), dynamic_lds(p, seeding), stream, p);
  return hipGetLastError();
}

int integrate_blocks_per_cu(const KParams& p, int seeding, int arith, int waves_per_simd) {
  IntegrateFn fn = integrate_variant(seeding, arith_variant(p, arith), clamp_w(waves_per_simd), wants_f32(p), wants_blocks(p));
  int n = 0;
  if (!fn || hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, kThreads, dynamic_lds(p, seeding)) != hipSuccess || n < 1) n = 1;
  return n;
}

// One wave per pixel (TOR_SEED_PIXEL, small frames).  Returns hipErrorInvalidValue when the objects do not fit LDS.
size_t coop_lds_bytes(int coop_slots) { return (size_t)kCoopArrays * 8 * (size_t)coop_slots; }

int coop_blocks_per_cu(const KParams& p, int arith) {
  int n = 0;
  (void)arith;
  auto fn = coop_pixel_kernel<0>;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, kThreads, coop_lds_bytes(p.coop_slots)) != hipSuccess) n = 0;
  return n;
}

hipError_t launch_coop(const KParams& p, int arith, int blocks, hipStream_t stream) {
  (void)arith;
  auto fn = coop_pixel_kernel<0>;
  const size_t lds = coop_lds_bytes(p.coop_slots);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kThreads), lds, stream, p);
  return hipGetLastError();
}

int integrate_fixed_lds_bytes(int blocks, int f32, int seeding) {
  return wave_lds_bytes(blocks, coop_variant(f32, blocks), reservoir_variant(seeding, f32, blocks)) * (kThreads / 64);
}

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
