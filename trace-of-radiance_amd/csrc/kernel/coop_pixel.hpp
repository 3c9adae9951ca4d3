// kernel/coop_pixel.hpp -- coop_pixel_kernel (textually included by tor_kernels.hip, inside namespace tor).
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
        const double a = dx * dx + dy * dy + dz * dz;  // spheres.nim:30
        double best_t = __builtin_inf(), best_f = 0.0;
        int best_slot = -1, best_orig = 0x7fffffff;
        // exact test of one object (spheres.nim:28-49) and the order-independent closest-hit update
        auto test_object = [&](int k, double cx, double cy, double cz, double r2, double f) {
          const double ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
          const double hb = ocx * dx + ocy * dy + ocz * dz;            // spheres.nim:31
          const double cc = (ocx * ocx + ocy * ocy + ocz * ocz) - r2;  // spheres.nim:32
          const double disc = hb * hb - a * cc;                        // spheres.nim:33
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
              cy = cy + T[4 * 64] * f;
            } else {
              cx = cx + T[3 * 64] * f; cy = cy + T[4 * 64] * f; cz = cz + T[5 * 64] * f;  // moving_spheres.nim:43
            }
            test_object(k, cx, cy, cz, T[8 * 64], f);
          } else {  // mixed trip: per-object kind and time group
            double cx = T[0 * 64], cy = T[1 * 64], cz = T[2 * 64];
            double f = 0.0;
            if ((int)__double_as_longlong(T[9 * 64]) & 1) {
              const double t0 = T[6 * 64], dt = T[7 * 64];
              f = (have_group && t0 == g_t0 && dt == g_dt) ? g_f : (time - t0) / dt;
              cx = cx + T[3 * 64] * f; cy = cy + T[4 * 64] * f; cz = cz + T[5 * 64] * f;
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
          center = center + v3(c[3], c[4], c[5]) * hit_f;
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

