// kernel/chain_servers.hpp -- SrvRng and serve_chains: the SERVER side of the SEED_PIXEL chain hand-off (textually included by
// tor_kernels.hip, inside namespace tor, behind integrate_kernel which declares serve_chains).
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
// serve_chains -- the SERVER side of the SEED_PIXEL chain hand-off (DESIGN 4.7 (HISTORY 4.10)), inside integrate_kernel<0, A, W, 1, 1>.
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
                  __hip_atomic_load(p.mig + kMigProgress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
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
        const double a = a_strict;
        const RayF32 r32 = make_ray_f32(ox, oy, oz, dx, dy, dz, a_strict, p.org[0], p.org[1], p.org[2]);
        const BoxRay32 b32 = make_box_ray32(r32, p.sp_bmax, p.sp_hmin);
        const unsigned wild = r32.wild & 1u;  // a ray outside the float32 test's guarded ranges enters every box
        const unsigned long long m0 = ballot64(valid0 && ((slab_bit32(b32, bx0, by0, bz0) | wild) != 0u));
        const unsigned long long m1 = ballot64(valid1 && ((slab_bit32(b32, bx1, by1, bz1) | wild) != 0u));
        const unsigned c0 = (unsigned)__builtin_popcountll(m0), n_hit = c0 + (unsigned)__builtin_popcountll(m1);
        const unsigned per_box = 8u * (unsigned)kBoxFanout;  // a box stands for box_fanout blocks of 8 consecutive cold slots
        const unsigned n_cand = n_always + per_box * n_hit;
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
          const unsigned want = (i - n_always) / per_box;  // the box the ray touches that this candidate belongs to (garbage in lanes that have none)
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
          else if (i < n_cand) slot = p.spatial_base + (int)per_box * box + (int)((i - n_always) % per_box);
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
            mx = k0 + k3 * f; my = k1 + k4 * f; mz = k2 + k5 * f;
            const double cx = moving ? mx : k0, cy = moving ? my : k1, cz = moving ? mz : k2;
            const double ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
            const double hb = ocx * dx + ocy * dy + ocz * dz;             // spheres.nim:31
            const double cc = (ocx * ocx + ocy * ocy + ocz * ocz) - k15;  // spheres.nim:32
            const double disc = hb * hb - a * cc;                         // spheres.nim:33
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
      // heartbeat for the stall detector, every 64th sample of the chain: at the end of a frame every wave of the machine serves, and
      // one atomic per sample from each of them on one address cost the tail 100 ms of a 520 ms frame (measured, round 5); a
      // finished chain moves kMigServed anyway
      if (p.mig_stall_ticks != 0 && lane == 0 && (s & 63) == 63) __hip_atomic_fetch_add(p.mig + kMigProgress, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

