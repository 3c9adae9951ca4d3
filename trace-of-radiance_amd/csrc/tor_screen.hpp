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
TOR_HD void screen_margins(double B, double d1, double a, double& negmu, double& am) {
  negmu = -(B * d1) * 0x1p-48;
  am = a * ((B * B) * 0x1p-45);
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
struct ScreenSeg {     // per ray and segment
  double dnx, dny, dnz;  // d~ (0 for a wild ray)
  double o2x, o2y, o2z;  // 2 o (0 for a wild ray)
  double P, T;           // kind 10: o.d~ - mu~ and M~ - |o|^2;  kinds 11 / 12: the same with oy - Y for oy
  double nfdy, g, nf2;   // kind 12: -f d~y, 2 f (oy - Y), -f^2
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
// reach, travel: segs[6], segs[7]; y_rel: 0 for the general static form (kind 10), else the segment's common c0.y (kinds 11, 12);
// f: the segment's time fraction (kind 12), else 0
TOR_HD ScreenSeg screen2_seg(const ScreenRay& r, double reach, double travel, double y_rel, double f) {
  ScreenSeg s;
  const double B = r.s1 + reach + travel * __builtin_fabs(f);
  const double mu = B * 0x1p-47;
  const double M = (B * B) * 0x1p-45;
  const bool wild = r.wild || !(M < __builtin_inf());
  const double oy = r.oy - y_rel;
  s.dnx = wild ? 0.0 : r.dnx; s.dny = wild ? 0.0 : r.dny; s.dnz = wild ? 0.0 : r.dnz;
  s.o2x = wild ? 0.0 : r.ox + r.ox; s.o2y = wild ? 0.0 : oy + oy; s.o2z = wild ? 0.0 : r.oz + r.oz;
  const double P = fma_(oy, s.dny, fma_(r.oz, s.dnz, r.ox * s.dnx)) - mu;
  const double Q = fma_(oy, oy, fma_(r.oz, r.oz, r.ox * r.ox));
  s.P = wild ? -1.0 : P;
  s.T = wild ? __builtin_inf() : M - Q;
  s.nfdy = -(f * s.dny);
  s.g = f * s.o2y;
  s.nf2 = wild ? 0.0 : -(f * f);
  return s;
}
// the three per-object tests; the returned word's SIGN BIT is the decision (set = keep), as screen_filter's
TOR_HD int screen2_sign(double hb, double t) {
  const double disc = fma_(hb, hb, t);
  const unsigned h = (unsigned)(double_to_bits(hb) >> 32), y = (unsigned)(double_to_bits(t) >> 32), z = (unsigned)(double_to_bits(disc) >> 32);
#if defined(__HIP_DEVICE_COMPILE__) && __has_builtin(__builtin_amdgcn_bitop3_b32)
  return (int)__builtin_amdgcn_bitop3_b32(h, y, z, 0x51);
#else
  return (int)((h | ~y) & ~z);
#endif
}
// kind 10: record {cx, cy, cz, K = |c|^2 - r^2}
TOR_HD int screen2_static(const ScreenSeg& s, double cx, double cy, double cz, double K) {
  const double hb = fma_(-cx, s.dnx, fma_(-cy, s.dny, fma_(-cz, s.dnz, s.P)));
  const double t = fma_(s.o2x, cx, fma_(s.o2y, cy, fma_(s.o2z, cz, s.T))) - K;
  return screen2_sign(hb, t);
}
// kind 11: record {cx, cz, K' = cx^2 + cz^2 - r^2}
TOR_HD int screen2_static_y(const ScreenSeg& s, double cx, double cz, double K) {
  const double hb = fma_(-cx, s.dnx, fma_(-cz, s.dnz, s.P));
  const double t = fma_(s.o2x, cx, fma_(s.o2z, cz, s.T)) - K;
  return screen2_sign(hb, t);
}
// kind 12: record {cx, cz, K', dcy} -- 32 bytes like the static records (the whole second-form table of random_scene is then
// 15.5 KB and fits the 16 KB scalar cache; with dcy^2 as a fifth field it was 21.7 KB and 4.5 % of the scalar loads missed).
// 2 f (oy - Y) dcy - f^2 dcy^2 = (g + nf2 dcy) dcy: the same two instructions as with a stored dcy^2.
TOR_HD int screen2_movy_y(const ScreenSeg& s, double cx, double cz, double K, double dcy) {
  const double hb = fma_(s.nfdy, dcy, fma_(-cx, s.dnx, fma_(-cz, s.dnz, s.P)));
  const double w = fma_(s.nf2, dcy, s.g);
  const double t = fma_(w, dcy, fma_(s.o2x, cx, fma_(s.o2z, cz, s.T))) - K;
  return screen2_sign(hb, t);
}
// the host's side of the records (tor_scene.cpp build_layout; the self test)
TOR_HD double screen2_K(double cx, double cy, double cz, double r2) { return ((cx * cx + cy * cy) + cz * cz) - r2; }
TOR_HD double screen2_Ky(double cx, double cz, double r2) { return (cx * cx + cz * cz) - r2; }

}  // namespace tor
