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

}  // namespace tor
