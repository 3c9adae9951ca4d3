// tor_screen.hpp -- the conservative FMA screen of the strict float64 object loop (integrate_kernel variants ARITH 2).
// One source for the device (tor_kernels.hip) and the host self test (tor_selftest_screen_host, tests/test_screen.py).
#pragma once
#include "tor_math.hpp"

namespace tor {

// ARITH 2 -- the reference's results behind a conservative FMA SCREEN.  The wave-uniform object loop only decides which
// objects become CANDIDATES; every candidate is re-tested by the deferred pass with the reference's own operations in
// the reference's order (exact_hit in tor_kernels.hip: spheres.nim:29-48 unfused), and a candidate that fails there costs nothing but
// time.  So the loop may use any test that never drops an object the reference's test keeps.  This one evaluates the same
// quadratic with FMAs -- 11 / 12 / 14 float64 instructions for a static / y-only moving / moving sphere instead of
// 17 / 19 / 23 -- and leans it towards "keep" by more than the two evaluations can differ:
//     hb'  = fma(ocz, dz, fma(ocy, dy, fma(ocx, dx, -mu)))            mu = 2^-48 B D1
//     cc'  = fma(ocz, ocz, fma(ocy, ocy, fma(ocx, ocx, -r^2)))
//     t'   = fma(-a, cc', a M)                                        M  = 2^-45 B^2
//     disc'= fma(hb', hb', t')
//     keep = (hb' < 0 or t' >= 0) and disc' >= 0          (sign bits, one v_bitop3)
// with B >= |o| + |c0| + |dc||f| + |r| (|o|_1 from the ray, the rest a per-segment constant from the host: segs[6], segs[7])
// and D1 = |dx| + |dy| + |dz|.  With u = 2^-53, H = oc.d, C = |oc|^2 - r^2, D = H^2 - a C in exact arithmetic on the
// float64 inputs: either evaluation has |hb - H| <= 6.3 u B D1, |cc - C| <= 19 u B^2, |disc - D| <= 48 u B^2 a (the
// centres differ by <= 3.1 u B per component between c0 + dc*f rounded twice and the folded fma; |oc| <= B,
// |H| <= B |d|, D1 <= sqrt(3) |d|).  The reference's hit() can accept a root only if disc_ref > 0 and (hb_ref < 0 or
// cc_ref < 0) -- both roots are <= 0 otherwise.  With mu = 32 u B D1 and M = 256 u B^2:
//   hb_ref < 0  =>  H < 6.3 u B D1 < mu / 2  =>  hb' < 0, and (H - mu)^2 >= H^2, so
//                   disc' >= D + a M - 48 u B^2 a > (256 - 48 - 48) u B^2 a > 0;
//   cc_ref < 0  =>  C < 19 u B^2  =>  cc' < 38 u B^2 < M  =>  t' > 0, and
//                   disc' >= (H - mu)^2 - a C + a M - 48 u B^2 a > (256 - 19 - 48) u B^2 a > 0.
// Nothing the reference accepts is dropped.  Overflowing margins (|o| beyond 1e150) keep everything; a NaN is dropped
// by both tests or re-tested exactly; a = 0 (d = 0) keeps everything and the exact test rejects it (disc = 0).
// (TOR_SCREEN_MUTATE: the mutation check of tools/mutation_check.sh -- every margin of the three screens set to zero; the host
// tests of tests/test_screen.py must then FAIL, which shows that they sit on the decision boundary.  Never defined in a product build.)
#if defined(TOR_SCREEN_MUTATE) && TOR_SCREEN_MUTATE == 1
#define TOR_MARGIN(x) 0.0
#else
#define TOR_MARGIN(x) (x)
#endif
TOR_HD void screen_margins(double B, double d1, double a, double& negmu, double& am) {
  negmu = TOR_MARGIN(-(B * d1) * 0x1p-48);
  am = TOR_MARGIN(a * ((B * B) * 0x1p-45));
}
// returns a word whose SIGN BIT is the decision (set = keep)
TOR_HD int screen_filter(double ocx, double ocy, double ocz, double dx, double dy, double dz, double a, double negmu,
                         double am, double r2) {
  const double hb = fma_(ocz, dz, fma_(ocy, dy, fma_(ocx, dx, negmu)));
  const double cc = fma_(ocz, ocz, fma_(ocy, ocy, fma_(ocx, ocx, -r2)));
  const double t = fma_(-a, cc, am);
  const double disc = fma_(hb, hb, t);
  const unsigned h = (unsigned)(double_to_bits(hb) >> 32), y = (unsigned)(double_to_bits(t) >> 32), z = (unsigned)(double_to_bits(disc) >> 32);
  // f(x, y, z) = (x | ~y) & ~z on the sign bits: one v_bitop3_b32, truth table 0x51
#if defined(__HIP_DEVICE_COMPILE__) && __has_builtin(__builtin_amdgcn_bitop3_b32)
  return (int)__builtin_amdgcn_bitop3_b32(h, y, z, 0x51);
#else
  return (int)((h | ~y) & ~z);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Screen, second form (round 4): the same conservative decision with the quadratic EXPANDED around the ray, so that the
// per-object work is nothing but fused multiply-adds against the object's record -- no o - c subtractions -- and with the
// direction NORMALISED once per ray, so that `a` drops out of the per-object work:
//     H~ = oc . d~           d~ = d / |d| (float64: sqrt and division correctly rounded, then one product per component)
//     hb'' = (o . d~ - mu~) - c . d~                                        3 fma against {cx, cy, cz}
//     t''  = (M~ - |o|^2) + 2 o . c - K          K = |c|^2 - r^2 (host)    3 fma + 1 add
//     disc''= fma(hb'', hb'', t'')               keep = (hb'' < 0 or t'' >= 0) and disc'' >= 0   (one v_bitop3, 0x51)
// = 8 float64 instructions for a static sphere (11 in the first form).  When the objects of a segment share c0.y bit for bit
// (spheres resting on a plane: 481 of random_scene's 485) the y terms are per-ray constants too:
//     static, uniform y :  hb'' = P - cx d~x - cz d~z ;  t'' = T + 2 ox cx + 2 oz cz - K'          K' = cx^2 + cz^2 - r^2     6 instructions
//     mover along y     :  c.y(f) = Y + f dcy, oc.y = (oy - Y) - f dcy:
//                          hb'' = P - cx d~x - cz d~z - (f d~y) dcy
//                          t''  = T + 2 ox cx + 2 oz cz + (2 f (oy - Y)) dcy - f^2 dcy^2 - K'                                9 instructions (12)
//   with P = ox d~x + oz d~z + (oy - Y) d~y - mu~ and T = M~ - ox^2 - oz^2 - (oy - Y)^2 per ray and segment.
//
// Why it is conservative.  u = 2^-53.  Exact quantities on the float64 inputs (c^ = c0 + f dc exact, oc^ = o - c^):
// H = oc^ . d, A = |d|^2, C = |oc^|^2 - r^2, D = H^2 - A C; normalised H~ = H / sqrt(A), D~ = D / A = H~^2 - C.  From the first
// form's analysis of the REFERENCE's evaluation: it can accept a root only if disc_ref > 0 and (hb_ref < 0 or cc_ref < 0), and
//   hb_ref < 0 => H < 6.3 u B D1 => H~ < 6.3 sqrt(3) u B < 11 u B;   cc_ref < 0 => C < 19 u B^2;   disc_ref > 0 => D~ > -48 u B^2.
// This form computes hb'' = H~ - mu~ + e_h and t'' = M~ - C + e_t, and the sign of fma(hb'', hb'', t'') is the sign of the exact
// hb''^2 + t''.  Error budgets (every bracket (|o_i| + |c_i| + |f dc_i|) <= B, sum |d~_i| <= sqrt(3)(1 + 4u)):
//   e_h: d~ carries a relative error <= 3.1 u per component (|d|^2 as the reference sums it: 2.5 u, sqrt, reciprocal, product);
//        oy - Y, f d~y one rounding each; at most 7 fused operations in the two chains (per-ray part, per-object part):
//        |e_h| <= (3.1 + 1 + 7) u sqrt(3) B + 7 u mu~ < 19.3 u B;
//   e_t: K, K' (host: 3 roundings), f^2, 2 f (oy - Y), g + nf2 dcy (<= 2 roundings each), at most 9 fused operations over terms whose
//        magnitudes sum to <= B^2 + M~:  |e_t| < 16 u B^2.
// With mu~ = 64 u B = 2^-47 B and M~ = 256 u B^2 = 2^-45 B^2:
//   hb_ref < 0 (and D~ > -48 u B^2):  hb'' < 11 u B + 19.3 u B - 64 u B < 0; with m = mu~ - e_h >= 44.7 u B >= 2 H~,
//        hb''^2 = (H~ - m)^2 >= H~^2, so hb''^2 + t'' >= D~ + M~ - |e_t| > (256 - 48 - 16) u B^2 > 0:  kept;
//   cc_ref < 0:  t'' > (256 - 19 - 16) u B^2 > 0, hence disc'' > 0:  kept.
// B >= |o|_1 + |c0| + |r| + |dc||f| as in the first form (segs[6], segs[7] from the host, |o|_1 from the ray).
// Rays outside the ranges this argument needs -- |d|^2 outside [2^-900, 2^900] (the normalisation would lose bits or
// overflow), B^2 not finite -- are WILD: they keep every object (T = +inf, d~ = 0, 2o = 0) and the exact test decides.
// A non-finite time fraction f gives a non-finite centre in the reference, which can never be hit; dropping is correct then.
// Checked from this source on the host (tor_selftest_screen2_host, tests/test_screen.py) and on the GPU (every strict
// brute-force parity test runs through it).
struct ScreenRay {     // per ray and closest-hit query
  double dnx, dny, dnz;  // d~
  double s1;             // |o|_1
  double ox, oy, oz;
  bool wild;
};
// ONE sign per test.  keep = (hb'' < 0 or t'' >= 0) and disc'' >= 0 is the sign of a single value:
//     q = t'' + max(-hb'', 0)^2        hb'' >= 0: q = t'' (and t'' >= 0 implies disc'' = hb''^2 + t'' >= 0);  hb'' < 0: q = disc''
// and max(., 0) is the CLAMP modifier of the instruction that ends the hb chain -- free -- provided the value stays below the clamp's
// upper bound 1.  So the segment's per-ray factors carry a power of two sigma <= 1 / (2 B) (|hb''| <= B (1 + 2^-40): the clamp never
// saturates): the hb chain is evaluated as Nh = clamp(sigma (c.d~ - P)) = sigma max(-hb'', 0), the t chain as -sigma^2 t'' (the `- K`
// becomes one more fma, same count), and q' = fma(-Nh, Nh, -sigma^2 t'') = -sigma^2 q.  Scaling by powers of two is exact, so the
// decision is the three-sign decision above bit for bit except where q is exactly 0, which the proof never needs (every object
// it keeps has q > 0 strictly).  The v_bitop3_b32 is gone: 7 / 9 / 10 instructions per common-height static / static / mover
// instead of 8 / 10 / 11, one of them (the v_alignbit into the mask) not float64.
struct ScreenSeg {     // per ray and segment (s = sigma)
  double hx, hy, hz;     // s d~ (0 for a wild ray)
  double Pn;             // s (mu~ - o.d~)            kinds 11 / 12: with oy - Y for oy
  double a2x, a2y, a2z;  // -s^2 2 o (0 for a wild ray)
  double Tn;             // -s^2 (M~ - |o|^2); -inf for a wild ray: everything is kept
  double ks;             // s^2
  double fdy, gn, f2n;   // kind 12: s f d~y, -s^2 2 f (oy - Y), s^2 f^2
};
TOR_HD ScreenRay screen2_ray(double ox, double oy, double oz, double dx, double dy, double dz, double a_strict) {
  ScreenRay r;
  const double inv = 1.0 / __builtin_sqrt(a_strict);
  r.dnx = dx * inv; r.dny = dy * inv; r.dnz = dz * inv;
  r.s1 = __builtin_fabs(ox) + __builtin_fabs(oy) + __builtin_fabs(oz);
  r.ox = ox; r.oy = oy; r.oz = oz;
  r.wild = !(a_strict >= 0x1p-900 && a_strict <= 0x1p900);
  return r;
}
// clamp(x) to [0, 1] with NaN -> 0: what the hardware's clamp modifier does (compute kernels run with DX10_CLAMP = 1)
TOR_HD double clamp01(double x) { return x > 0.0 ? (x < 1.0 ? x : 1.0) : 0.0; }
// fma(a, b, c) clamped to [0, 1]; a is wave-uniform (a scalar register on the device)
TOR_HD double fma_clamp_s(double a_uniform, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_fma_f64 %0, %1, %2, %3 clamp" : "=v"(r) : "s"(a_uniform), "v"(b), "v"(c));
  return r;
#else
  return clamp01(fma_(a_uniform, b, c));
#endif
}
// ... the same with every operand in a vector register (stage two of the plane-screened segments: per-lane records)
TOR_HD double fma_clamp_v(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_fma_f64 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
#else
  return clamp01(fma_(a, b, c));
#endif
}
// reach, travel: segs[6], segs[7]; y_rel: 0 for the general static form (kind 10), else the segment's common c0.y (kinds 11, 12);
// f: the segment's time fraction (kind 12), else 0
TOR_HD ScreenSeg screen2_seg(const ScreenRay& r, double reach, double travel, double y_rel, double f) {
  ScreenSeg s;
  const double B = r.s1 + reach + travel * __builtin_fabs(f);
  const double mu = TOR_MARGIN(B * 0x1p-47);
  const double M = TOR_MARGIN((B * B) * 0x1p-45);
  // (f * f must stay finite too: f2n = sg2 f^2 below meets dcy = 0 -- a static that rests in a segment of movers along y -- as
  // inf x 0 = NaN otherwise, and the NaN's clear sign bit would DROP a sphere the reference hits.  |f| > 1.3e154 with
  // travel |f| still below the B^2 limit needs travel < 1: random_scene's movers, shutter_close ~ 1e155.  ADVICE r5.)
  const bool wild = r.wild || !((B * B) * 0x1p-45 < __builtin_inf()) || !(f * f < __builtin_inf());
  // sigma = 2^-(e + 2) for B in [2^e, 2^(e+1)): sigma B < 1/2.  (B = 0 or denormal: 2^1020; harmless, every product below is 0 or tiny)
  uint64_t e = (double_to_bits(B) >> 52) & 0x7ffu;
  if (e < 3u) e = 3u;
  if (e > 0x7fbu) e = 0x7fbu;
  const double sg = bits_to_double((uint64_t)(0x7fcu - e) << 52), sg2 = sg * sg;
  const double oy = r.oy - y_rel;
  const double dnx = wild ? 0.0 : r.dnx, dny = wild ? 0.0 : r.dny, dnz = wild ? 0.0 : r.dnz;
  s.hx = sg * dnx; s.hy = sg * dny; s.hz = sg * dnz;
  const double P = fma_(oy, dny, fma_(r.oz, dnz, r.ox * dnx)) - mu;
  const double Q = fma_(oy, oy, fma_(r.oz, r.oz, r.ox * r.ox));
  s.Pn = wild ? 0.0 : -(sg * P);
  s.Tn = wild ? -__builtin_inf() : -(sg2 * (M - Q));
  const double n2 = wild ? 0.0 : -(sg2 + sg2);
  s.a2x = n2 * r.ox; s.a2y = n2 * oy; s.a2z = n2 * r.oz;
  // (a wild ray keeps EVERYTHING: Tn = -inf and every factor 0 -- ks included, with K clamped to a finite value by the host
  // (screen2_K), so that no 0 x inf = NaN of unspecified sign can come out of the chain: ADVICE r4 -- a camera inside a sphere
  // 1e155 from the origin is hit by the reference, |c|^2 overflows, and the `keeps everything` promise must hold for it)
  s.ks = wild ? 0.0 : sg2;
  const double fw = wild ? 0.0 : f;   // (a non-finite f makes B, hence the ray, wild: no inf x 0 in the factors below)
  s.fdy = fw * s.hy;
  s.gn = fw * s.a2y;
  s.f2n = sg2 * (fw * fw);
  return s;
}
// the per-object tests; the returned word's SIGN BIT is the decision (set = keep), as screen_filter's
TOR_HD int screen2_word(double nh, double tn) {
  const double q = fma_(-nh, nh, tn);
  return (int)(unsigned)(double_to_bits(q) >> 32);
}
// kind 10: record {cx, cy, cz, K = |c|^2 - r^2}
// (kLane: the record is the lane's own -- vector registers -- instead of wave-uniform)
template <bool kLane = false>
TOR_HD int screen2_static(const ScreenSeg& s, double cx, double cy, double cz, double K) {
  const double nh = kLane ? fma_clamp_v(cx, s.hx, fma_(cy, s.hy, fma_(cz, s.hz, s.Pn))) : fma_clamp_s(cx, s.hx, fma_(cy, s.hy, fma_(cz, s.hz, s.Pn)));
  const double tn = fma_(s.ks, K, fma_(s.a2x, cx, fma_(s.a2y, cy, fma_(s.a2z, cz, s.Tn))));
  return screen2_word(nh, tn);
}
// kind 11: record {cx, cz, K' = cx^2 + cz^2 - r^2}
template <bool kLane = false>
TOR_HD int screen2_static_y(const ScreenSeg& s, double cx, double cz, double K) {
  const double nh = kLane ? fma_clamp_v(cx, s.hx, fma_(cz, s.hz, s.Pn)) : fma_clamp_s(cx, s.hx, fma_(cz, s.hz, s.Pn));
  const double tn = fma_(s.ks, K, fma_(s.a2x, cx, fma_(s.a2z, cz, s.Tn)));
  return screen2_word(nh, tn);
}
// kind 12: record {cx, cz, K', dcy} -- 32 bytes like the static records (the whole second-form table of random_scene is then
// 15.5 KB and fits the 16 KB scalar cache; with dcy^2 as a fifth field it was 21.7 KB and 4.5 % of the scalar loads missed).
// 2 f (oy - Y) dcy - f^2 dcy^2 = (g + nf2 dcy) dcy: the same two instructions as with a stored dcy^2.
template <bool kLane = false>
TOR_HD int screen2_movy_y(const ScreenSeg& s, double cx, double cz, double K, double dcy) {
  const double nh = kLane ? fma_clamp_v(dcy, s.fdy, fma_(cx, s.hx, fma_(cz, s.hz, s.Pn))) : fma_clamp_s(dcy, s.fdy, fma_(cx, s.hx, fma_(cz, s.hz, s.Pn)));
  const double w = fma_(s.f2n, dcy, s.gn);
  const double tn = fma_(s.ks, K, fma_(w, dcy, fma_(s.a2x, cx, fma_(s.a2z, cz, s.Tn))));
  return screen2_word(nh, tn);
}
// the host's side of the records (tor_scene.cpp build_layout; the self test)
// (clamped to the finite range: +-inf would meet the wild ray's zero factors as NaN; a NaN K -- a NaN centre or radius, which the
// reference can never hit -- becomes -max (fmax skips the NaN): kept by the screen, rejected by the exact test)
TOR_HD double screen2_clampK(double k) { return __builtin_fmin(__builtin_fmax(k, -0x1.fffffffffffffp1023), 0x1.fffffffffffffp1023); }
TOR_HD double screen2_K(double cx, double cy, double cz, double r2) { return screen2_clampK(((cx * cx + cy * cy) + cz * cz) - r2); }
TOR_HD double screen2_Ky(double cx, double cz, double r2) { return screen2_clampK((cx * cx + cz * cz) - r2); }

// ---------------------------------------------------------------------------------------------------------------------
// STAGE ONE in front of the second form (round 4, kinds 11 and 12): the PLANE screen.  A sphere can only be hit by a ray whose
// LINE passes within r of its centre, and the distance from the centre to the line is at least the distance from the centre to
// any plane that contains the line.  For the vertical plane through the ray that distance does not involve y at all:
//     s = n . (c_xz - o_xz)          n = (-d~z, d~x)   (normal of the ray's ground track, NOT normalised: |n|^2 = w2 = d~x^2 + d~z^2)
//     keep  <=>  s^2 < (R^2 + M~) w2      R = the largest |radius| of the segment, M~ = 2^-45 B^2 as above
// = 2 fused multiply-adds against {cx, cz} + 1 for the sign + the v_alignbit: 4 instructions per object -- for a static sphere
// and for a mover along y alike (its cx, cz do not move) -- instead of 7 / 10, and no square root or division per ray.  It keeps
// the spheres in a band of width 2 R along the ray's ground track: 9.2 of random_scene's 481 small spheres for an average query
// (the second form: 0.6).  Those go through the second form PER LANE (kernel/integrate_loop_plane.inc: stage two), and what
// survives that is exactly what the second form alone would have kept, because stage one never drops what the second form's
// PROOF needs kept:
//   the reference accepts a root only if disc_ref > 0, which implies D~ = r^2 - dist(c^, line)^2 > -48 u B^2 (above), so with the
//   exact unit normal n^ of the exact ground track  (n^ . (c_xz - o_xz))^2 <= dist^2 < r^2 + 48 u B^2.  Computed: the components of
//   n carry the relative error of d~ (3.1 u each, so n = n~ + delta with |delta| <= 3.1 u |n~|, n~ the perpendicular of the exactly
//   normalised direction, |n~| = w); the constant n . o_xz (a product and an fma) and the two fused operations of the chain round
//   four times, each by <= u x a magnitude <= 2 sqrt(2) w B:  |s' - n~ . (c_xz - o_xz)| <= (3.1 + 9.9) u w B = 13 u w B, and with
//   |n~ . (..)| = w dist_2D <= w B:   s'^2 < w^2 (r^2 + (48 + 27) u B^2).   w2 as computed is >= w^2 (1 - 8.2 u) (the components'
//   3.1 u twice, the sum's own rounding), the threshold's sum and product round twice more:
//   s'^2 < w^2 (r^2 + 75 u B^2) < (R^2 (1 + 2^-40) + 256 u B^2) w2' (1 - 2 u)   since 2^-40 >> 11 u and 256 (1 - 11 u) > 75.
// c_xz is exact here: statics, and movers whose c1 - c0 has x = z = 0 exactly (group_moves_along_y_only).
// A ray whose ground track has no direction (w2 < 2^-200: vertical), whose threshold leaves the normal range, or that is wild
// keeps everything.
//
// Round 5 -- the test never reads y, so it serves EVERY float64 segment, not only the common-height ones it was built for:
//   xkind 10 (statics at any height) and 14 (movers along y at any height): the same {cx, cz} table, the same 4 instructions;
//   xkind 13 (movers whose centre travels in x or z): c^_xz = c0_xz + f dc_xz is evaluated inside the chain,
//       s' = nx c0x + nz c0z + (f nx) dcx + (f nz) dcz + c0       record {c0x, c0z, dcx, dcz}: 5 float64 instructions + the v_alignbit
//     against the exact c^ = c0 + f dc that the first form's analysis of the reference is stated for (the reference's own centre,
//     c0 + dc*f rounded twice, is within 3.1 u B of it per component, which the 48 u B^2 above already contains).  Two more
//     roundings in the chain and one each in f nx, f nz (|f nx dcx| <= w |f| |dc| <= w B):  |s' - n~ . (c^_xz - o_xz)| <=
//     (3.1 + 6 x 2.83 + 2) u w B < 22.1 u w B, so s'^2 < w^2 (r^2 + (48 + 45) u B^2) -- still far inside the 256 u B^2 of the
//     threshold (256 (1 - 11 u) > 93).  B >= |o|_1 + |c0| + |r| + |dc| |f| as everywhere (segs[6], segs[7]).
// Stage two then runs, per lane, the test the wave-uniform loop would have run on the segment: the second form for 10 / 11 / 12,
// the first form (screen_filter, centre folded into o - c) for 13 / 14 -- both proofs start from disc_ref > 0, which is all stage
// one relies on.
struct PlaneRay {   // per ray and closest-hit query
  double nx, nz;    // (-d~z, d~x)
  double w2;        // nx^2 + nz^2
  bool all;         // no ground track or a wild ray: every object is kept
};
struct PlaneSeg {   // per ray and segment
  double nx, nz;    // normal of the ground track (0 when everything is kept)
  double c0;        // -(n . o_xz)
  double negthr;    // -(R^2 (1 + 2^-40) + M~) w2; -inf: everything is kept
  double fnx, fnz;  // f n (xkind 13); 0 when everything is kept
};
TOR_HD PlaneRay plane_ray(const ScreenRay& r) {
  PlaneRay pr;
  pr.nx = -r.dnz;
  pr.nz = r.dnx;
  pr.w2 = fma_(r.dnz, r.dnz, r.dnx * r.dnx);
  pr.all = r.wild || !(pr.w2 >= 0x1p-200);
  return pr;
}
TOR_HD PlaneSeg plane_seg(const ScreenRay& r, const PlaneRay& pr, double reach, double travel, double f, double rmax2) {
  PlaneSeg s;
  const double B = r.s1 + reach + travel * __builtin_fabs(f);
#if defined(TOR_SCREEN_MUTATE)   // (level 2 of the mutation check: only stage one's margins)
  const double M = 0.0;
  const double thr = rmax2 * pr.w2;
#else
  const double M = (B * B) * 0x1p-45;
  const double thr = (fma_(rmax2, 0x1p-40, rmax2) + M) * pr.w2;
#endif
  const bool all = pr.all || !(thr >= 0x1p-900 && thr < __builtin_inf());
  s.nx = all ? 0.0 : pr.nx;
  s.nz = all ? 0.0 : pr.nz;
  s.c0 = all ? 0.0 : -fma_(s.nx, r.ox, s.nz * r.oz);
  s.negthr = all ? -__builtin_inf() : -thr;
  s.fnx = all ? 0.0 : f * s.nx;   // (a non-finite f: B and the threshold are not finite either -> `all`)
  s.fnz = all ? 0.0 : f * s.nz;
  return s;
}
// the returned word's SIGN BIT is the decision (set = keep)
TOR_HD int plane_word(const PlaneSeg& s, double cx, double cz) {
  const double v = fma_(s.nx, cx, fma_(s.nz, cz, s.c0));
  const double q = fma_(v, v, s.negthr);
  return (int)(unsigned)(double_to_bits(q) >> 32);
}
// xkind 13: record {c0x, c0z, dcx, dcz}
TOR_HD int plane_word_mov(const PlaneSeg& s, double c0x, double c0z, double dcx, double dcz) {
  const double v = fma_(s.nx, c0x, fma_(s.nz, c0z, fma_(s.fnx, dcx, fma_(s.fnz, dcz, s.c0))));
  const double q = fma_(v, v, s.negthr);
  return (int)(unsigned)(double_to_bits(q) >> 32);
}
// Does stage one PAY on this segment for this ray?  It leaves the objects in a band of half-width R around the ground track to a
// per-lane stage that costs ~8 times a wave-uniform test per object, so it only pays when the band is thin against the segment:
// a wall of spheres seen edge-on (every centre on the ray's own ground track) would send everything through both stages.  The
// centres' bounding box in xz has extents (Sx, Sz) (host: xsegs[5], xsegs[6]); its width across the track is at most
// E = (|nx| Sx + |nz| Sz) / |n|, and the share of a uniformly filled box inside the band about 2 R / E.  The plane runs when that
// estimate is below 1 / kPlaneGate; the wave takes the majority's vote (the loop is wave-uniform).  A performance decision only:
// either way every object meets a conservative test and every candidate the exact one.
// (gate2 = 4 kPlaneGate^2 travels in KParams; 0 = no gate -- TOR_PLANE=2, the yardstick of the gate itself)
constexpr double kPlaneGate = 16.0;
TOR_HD bool plane_pays(const PlaneRay& pr, double gate2, double rmax2, double sx, double sz) {
  const double e = fma_(__builtin_fabs(pr.nx), sx, __builtin_fabs(pr.nz) * sz);
  return !pr.all && (gate2 * rmax2) * pr.w2 < e * e;
}

}  // namespace tor
