// tor_filter32.hpp -- TOR_ACCEL_F32: a conservative float32 pre-filter in front of the float64 sphere test.
//
// The reference decides ray x sphere with the float64 discriminant  D = hb*hb - a*c  (spheres.nim:30-33,
// moving_spheres.nim:46-51) and accepts a root only if it is > 0.001, which needs  D > 0  and  (hb < 0 or
// c < 0).  The hot loop only has to find the few objects for which that CAN hold; the exact float64 hit
// (roots, closest-hit update) is recomputed for those anyway.  So the loop may run in any arithmetic as long
// as it never drops an object the float64 test would keep.  This header does it in packed float32
// (v_pk_fma_f32 / v_pk_add_f32: two objects per instruction, object records through the scalar data path),
// with the rounding error of the float32 evaluation bounded a priori and folded into the comparison.
//
// Error bound (u = 2^-24, all operations round to nearest, no overflow/underflow -- the caller guards the
// ranges; hats are float32 values, capitals the real-number values of the float64 inputs):
//   coordinates relative to an origin P near the scene:  o^ = fl(o - P), c0^ = fl(c0 - P), dc^ = fl(dc), f^ = fl(f)
//   oc^ = fl(o^ - c0^)  [static]   or   fl(fl(o^ - c0^) - f^ dc^)  [moving, one fma]
//   Lambda := 2|o-P| + 3.1 Mc + |OC|,  Mc := |c0-P| + |f| |dc|         (Euclidean norms)
//   |oc^ - OC|           <= 1.001 u Lambda                              (input roundings + 2 operation roundings)
//   |hb^ - HB|           <= 5.02 u |d| Lambda                           (3 products, 3 roundings)
//   |c^  - C |           <= u (2.01 |OC| Lambda + 3.02 |OC|^2 + 4.02 r^2) + u^2 Lambda^2
// The filter evaluates, per object,
//   hb'' = hb^ - mbl          mbl >= 5.1 u |d| (3|o-P| + 4.1 max Mc)   =>  HB < 0  implies  hb'' < 0 and |hb''| >= |HB|
//   y    = m - a^ c^          m   >= 1.002 u A (9.81 c^ + 15.84 r^2 + 7.25 Mc^2 + 3.02 |o-P|^2)
//                             evaluated as  c^ (g - a^) + (g k + gK),  g = 10.5 u a^  (two fmas)
//   D''  = hb''^2 + y
// and keeps the object iff  D'' >= 0  and  (hb'' < 0  or  y >= 0)   [sign bits only].
//   * C < 0           =>  c^ < E_c <= m / a^  =>  y >= 0, and then D'' >= 0 as a sum of non-negatives;
//   * HB < 0, D > 0   =>  hb''^2 + y >= HB^2 - A C + (m - error terms) >= D > 0.
// (2.01 |OC| Lambda <= 4.02 |OC|^2 + 0.2513 Lambda^2 and Lambda^2 <= 3 (4 |o-P|^2 + 9.61 Mc^2 + |OC|^2) turn the
// bound into the linear form above; |OC|^2 = C + r^2.)  The constants below carry a further 5-8 % of slack,
// which also covers the roundings of the margin arithmetic itself and the 2^-50-relative gap between the
// float64-computed D, HB, C and their real values.  For |o-P|, |c-P| ~ 15 and r = 0.2 (random_scene) the
// margins widen a sphere by < 1 %: ~2 % more candidates for rays that come from elsewhere, plus the sphere a
// scattered ray starts on (its c ~ 0: the float64 sign test resolves it, float32 cannot) -- measured 1.43
// candidates per query instead of 1.25.
//
// The second half of the file is the float32 slab test for the culling boxes of TOR_ACCEL_BLOCKS (slab_bit32).
//
// tests/test_filter32.py drives filter_one() / slab_bit32 (host build of this header) with random and
// adversarial (tangent, origin on the surface, far-away, tiny direction; box corners, edges, faces, axis-parallel
// rays) pairs and the scene layouts, and checks they never drop what the float64 test keeps; the GPU parity tests
// and tools/fuzz_accel.py then compare whole canvases bit for bit.
#pragma once

#include "tor_math.hpp"

namespace tor {

typedef float f2v __attribute__((ext_vector_type(2)));

TOR_HD f2v splat2(float x) { return (f2v){x, x}; }
TOR_HD f2v fma2(f2v a, f2v b, f2v c) { return __builtin_elementwise_fma(a, b, c); }

constexpr float kU32 = 0x1p-24f;
constexpr float kF32Gain = 10.5f * kU32;  // >= 1.002 * 9.81 u, + slack
constexpr float kF32Mbl = 5.5f * kU32;    // >= 5.1 u, + slack
constexpr float kF32KRo = 0.32f;          // >= 3.02 / 9.81
constexpr float kF32KDc = 1.5f;           // >= 2 * 7.25 / 9.81   (Mc^2 <= 2 |c0-P|^2 + 2 f^2 |dc|^2)
constexpr double kF32KR2 = 1.65;          // >= 15.84 / 9.81      (host: per-object constant k)
constexpr double kF32KMc = 0.75;          // >= 7.25 / 9.81
constexpr float kF32Lim = 0x1p20f;        // |c0-P|, |dc|, |f| admitted by the guards
constexpr float kF32PadR2 = -1e30f;       // r^2 of a padding record: c^ ~ 1e30, y < 0, D'' < 0

// per query
struct RayF32 {
  float ox, oy, oz, dx, dy, dz;  // o - P and d, rounded to float32
  float gma, g;                  // g - a and g = margin gain * a
  float ro, ro2, sa;             // |o-P|, |o-P|^2, sqrt(a)
  unsigned wild;                 // 0xff: ranges not guaranteed -> the lane keeps every object
};

TOR_HD RayF32 make_ray_f32(double ox, double oy, double oz, double dx, double dy, double dz, double a,
                           double px, double py, double pz) {
  RayF32 r;
  r.ox = (float)(ox - px); r.oy = (float)(oy - py); r.oz = (float)(oz - pz);
  r.dx = (float)dx; r.dy = (float)dy; r.dz = (float)dz;
  const float fa = (float)a;
  r.g = kF32Gain * fa;
  r.gma = r.g - fa;
  r.ro2 = __builtin_fmaf(r.oz, r.oz, __builtin_fmaf(r.oy, r.oy, r.ox * r.ox));
  r.ro = __builtin_sqrtf(r.ro2);
  r.sa = __builtin_sqrtf(fa);
  const bool sane = (fa >= 0x1p-40f) && (fa <= 0x1p40f) && (r.ro2 <= 0x1p40f);  // false for NaN
  r.wild = sane ? 0u : 0xffu;
  return r;
}

// per query and segment (a segment = objects sharing (time0, time1); f = (time - time0)/(time1 - time0),
// 0 for static spheres; mc0max / dcmax = max |c0-P| / max |dc| over the segment, rounded up)
struct SegF32 {
  f2v nmbl, gk, nf;  // -mbl, g * K (the per-lane part of the margin), -f
  unsigned wild;
};

TOR_HD SegF32 make_seg_f32(const RayF32& r, double f64, float mc0max, float dcmax) {
  SegF32 s;
  const float f = (float)f64;
  const float af = __builtin_fabsf(f);
  s.wild = (af <= kF32Lim) ? r.wild : 0xffu;  // NaN -> wild
  const float mcl = mc0max + af * dcmax;
  s.nmbl = splat2(-(kF32Mbl * r.sa * (3.0f * r.ro + 4.2f * mcl)));
  s.gk = splat2(r.g * (kF32KRo * r.ro2 + kF32KDc * ((f * f) * (dcmax * dcmax))));
  s.nf = splat2(-f);
  return s;
}

// o - c for one axis of two objects: static, or moving along that axis (centre c0 + f dc)
TOR_HD f2v oc_static32(float o, f2v c0) { return splat2(o) - c0; }
TOR_HD f2v oc_moving32(float o, f2v c0, f2v dc, f2v nf) { return fma2(nf, dc, splat2(o) - c0); }

// Two objects (the halves of the vectors) against one ray, given o - c per axis; pushes their keep-bits into
// m (first object ends up in the higher bit, as the float64 loops do).
// k = 1.65 r^2 + 0.75 (1|2) |c0-P|^2 per object.  12 packed + 4 integer instructions.
TOR_HD unsigned filter_pair32(const RayF32& r, const SegF32& s, f2v ocx, f2v ocy, f2v ocz, f2v r2, f2v k,
                              unsigned m) {
  const f2v hb = fma2(ocz, splat2(r.dz), fma2(ocy, splat2(r.dy), fma2(ocx, splat2(r.dx), s.nmbl)));
  const f2v cc = fma2(ocz, ocz, fma2(ocy, ocy, fma2(ocx, ocx, -r2)));
  const f2v y = fma2(cc, splat2(r.gma), fma2(splat2(r.g), k, s.gk));
  const f2v dd = fma2(hb, hb, y);
  // keep = (sign(hb) | ~sign(y)) & ~sign(D''): one v_bitop3_b32 (truth table 0x51), bit 31.
  // (Elements are copied to scalars first: __builtin_bit_cast applied directly to `v.y` reads element 0.)
  const float hb0 = hb.x, hb1 = hb.y, y0 = y.x, y1 = y.y, dd0 = dd.x, dd1 = dd.y;
  const unsigned bh0 = __builtin_bit_cast(unsigned, hb0), bh1 = __builtin_bit_cast(unsigned, hb1);
  const unsigned by0 = __builtin_bit_cast(unsigned, y0), by1 = __builtin_bit_cast(unsigned, y1);
  const unsigned bd0 = __builtin_bit_cast(unsigned, dd0), bd1 = __builtin_bit_cast(unsigned, dd1);
#if defined(__HIP_DEVICE_COMPILE__) && __has_builtin(__builtin_amdgcn_bitop3_b32)
  m = __builtin_amdgcn_alignbit(m, __builtin_amdgcn_bitop3_b32(bh0, by0, bd0, 0x51), 31);
  m = __builtin_amdgcn_alignbit(m, __builtin_amdgcn_bitop3_b32(bh1, by1, bd1, 0x51), 31);
#else
  m = (m << 1) | (((bh0 | ~by0) & ~bd0) >> 31);
  m = (m << 1) | (((bh1 | ~by1) & ~bd1) >> 31);
#endif
  return m;
}

// ---- float32 slab test for the culling boxes (TOR_ACCEL_BLOCKS | TOR_ACCEL_F32) ---------------------------------
// Box records hold {centre, half-extent} per axis relative to P: c = fl32 of the float64 box's midpoint, h rounded UP so that
// [c - h, c + h] contains the float64 box (which is itself inflated by 1e-6 relative around the swept spheres).  Round 4: the
// centre form.  The entry / exit parameters of an axis are
//     u = c (1/d^) + (-o^ (1/d^))           one v_fma_f32 (2 cycles; the second product is formed once per ray)
//     (t_near, t_far) = u -+ h |1/d^|       one v_pk_fma_f32 against the ray's (-H, +H)
// with NO per-lane min / max to sort the two planes by the sign of d (the {lo, hi} form needed a packed fma + v_min + v_max:
// 8 cycles per axis instead of 6; 28 instead of 34 cycles per box).
// Errors, all expressed as the displacement of a box plane (a relative error delta of t = (b - o)/d moves the plane by
// |delta| |b - o| <= |delta| (|o-P| + max|box|)), u = 2^-24:  o^ = fl(o - P): u |o-P|;  d^ and 1/d^: 2 u (|o-P| + max|box|);
// the per-ray product -o^ (1/d^): u |o-P|;  the rounding of u: u (|o-P| + max|box|);  the rounding of t and of H: u (|o-P| + 2 max|box|).
// Together < 8 u (|o-P| + max|box|) =: e.  The test runs on every box inflated by e on every side, and since every half-extent is at
// least h_min (host: the smallest one over the valid boxes), scaling the ray's H by (1 + e / h_min) inflates every box by >= e
// without an instruction per box: (h + e) H <= h H (1 + e / h_min).  At random_scene's scale e / h_min ~ 1e-5; a camera a hundred
// scene sizes away still only widens the boxes by a per mille.  The test itself is the plain `t_in <= t_out`.  NaN (0 * inf on an
// axis the ray is parallel to while the origin sits in the slab's mid-plane, or a NaN padding box) is dropped by min / max exactly
// as in the float64 test: that axis stops constraining, a padding box is never entered.
struct BoxRay32 {
  float ix, iy, iz;  // 1/d^ per axis (0 for an axis the ray is parallel to)
  float ax, ay, az;  // -o^ / d^ (0 for such an axis)
  f2v hx, hy, hz;    // (-H, +H), H = |1/d^| (1 + e / h_min) rounded up (+inf for such an axis)
};

TOR_HD BoxRay32 make_box_ray32(const RayF32& r, float bmax, float hmin) {
  BoxRay32 b;
  const float e = 8.0f * kU32 * (r.ro + bmax);
  const float grow = (1.0f + e / hmin) * (1.0f + 4.0f * kU32);  // (h_min = 0: inf -> every box is entered)
  // An axis the ray is (nearly) parallel to -- |1/d^| = inf or so large that the products could overflow -- must not
  // constrain: it gets 1/d^ := 0, addend 0 and H := +inf, i.e. t = (-inf, +inf) for h > 0 (and NaN, dropped, for h = 0).
  const float inf = __builtin_inff();
  auto axis = [&](float o, float d, float& i, float& a, f2v& h) {
    const float id = 1.0f / d;
    const bool open = !(__builtin_fabsf(id) <= 0x1p100f);  // inf, NaN or huge
    i = open ? 0.0f : id;
    a = open ? 0.0f : -o * id;
    const float H = open ? inf : __builtin_fabsf(id) * grow;
    h = (f2v){-H, H};
  };
  axis(r.ox, r.dx, b.ix, b.ax, b.hx);
  axis(r.oy, r.dy, b.iy, b.ay, b.hy);
  axis(r.oz, r.dz, b.iz, b.az, b.hz);
  return b;
}

// 1: the ray may touch the box {bx = (c.x, h.x), by, bz}; 0: it cannot.
TOR_HD unsigned slab_bit32(const BoxRay32& b, f2v bx, f2v by, f2v bz) {
  const float cx = bx.x, hx = bx.y, cy = by.x, hy = by.y, cz = bz.x, hz = bz.y;
  const float ux = __builtin_fmaf(cx, b.ix, b.ax), uy = __builtin_fmaf(cy, b.iy, b.ay), uz = __builtin_fmaf(cz, b.iz, b.az);
  const f2v tx = fma2(splat2(hx), b.hx, splat2(ux)), ty = fma2(splat2(hy), b.hy, splat2(uy)), tz = fma2(splat2(hz), b.hz, splat2(uz));
  const float tx0 = tx.x, tx1 = tx.y, ty0 = ty.x, ty1 = ty.y, tz0 = tz.x, tz1 = tz.y;
  const float t_in = __builtin_fmaxf(__builtin_fmaxf(tx0, ty0), __builtin_fmaxf(tz0, 0.0f));
  const float t_out = __builtin_fminf(__builtin_fminf(tx1, ty1), tz1);
  return (t_in <= t_out) ? 1u : 0u;
}

// Host-side per-object constant k (rounded up) and eligibility; used by tor_scene.cpp and the self test.
inline bool f32_eligible(double mc0, double dcn, double r2) {
  return std::isfinite(mc0) && std::isfinite(dcn) && std::isfinite(r2) && mc0 <= (double)kF32Lim * 0.5 &&
         dcn <= (double)kF32Lim * 0.5 && r2 >= 0x1p-40 && r2 <= 0x1p40;
}
inline float f32_round_up(double x) {
  float f = (float)x;
  if ((double)f < x) f = std::nextafterf(f, INFINITY);
  return f;
}
inline float f32_object_k(double mc0, double r2, bool moving) {
  return f32_round_up(1.001 * (kF32KR2 * r2 + kF32KMc * (moving ? 2.0 : 1.0) * mc0 * mc0));
}

// One ray against one object through the same code path as the kernel (both vector halves carry the object).
// Returns the keep bit.
inline unsigned filter_one(const double o[3], const double d[3], const double c0[3], const double dc[3],
                           bool moving, double f64, double r2, const double P[3]) {
  const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  const RayF32 r = make_ray_f32(o[0], o[1], o[2], d[0], d[1], d[2], a, P[0], P[1], P[2]);
  const double q[3] = {c0[0] - P[0], c0[1] - P[1], c0[2] - P[2]};
  const double mc0 = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  const double dcn = moving ? std::sqrt(dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2]) : 0.0;
  if (!f32_eligible(mc0, dcn, r2)) return 1u;  // such an object stays on the float64 path
  const SegF32 s = make_seg_f32(r, moving ? f64 : 0.0, f32_round_up(mc0), f32_round_up(dcn));
  const f2v cx = splat2((float)q[0]), cy = splat2((float)q[1]), cz = splat2((float)q[2]);
  f2v ocx, ocy, ocz;
  if (moving) {
    ocx = oc_moving32(r.ox, cx, splat2((float)dc[0]), s.nf);
    ocy = oc_moving32(r.oy, cy, splat2((float)dc[1]), s.nf);
    ocz = oc_moving32(r.oz, cz, splat2((float)dc[2]), s.nf);
  } else {
    ocx = oc_static32(r.ox, cx); ocy = oc_static32(r.oy, cy); ocz = oc_static32(r.oz, cz);
  }
  unsigned m = filter_pair32(r, s, ocx, ocy, ocz, splat2((float)r2), splat2(f32_object_k(mc0, r2, moving)), 0u);
  return ((m | s.wild) & 1u);
}

}  // namespace tor
