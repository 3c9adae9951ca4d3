/*
 * oracle/tor_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the hot path of mratsim/trace-of-radiance:
 *   render() -> radiance() -> HittableList.hit -> Material.scatter
 * written from the reference's behaviour, each function citing the reference
 * file:line it follows (paths relative to /root/reference/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (trace-of-radiance_amd/) never links, imports or calls it.
 *
 * PARITY PIN: with seeding=PIXEL, math=LIBM, arith=STRICT this oracle reproduces the
 * reference's own rendered image media/book2_motion_blur.png (384x216, 100 spp,
 * depth 50, seed 0xFACADE) bit-for-bit through the io/ppm.nim quantiser
 * (tests/test_oracle_golden.py; fixture tests/golden/book2_motion_blur.png).
 * The reference itself (Nim + Weave) cannot be built in this image: no nim/nimble,
 * Weave is an un-vendored, un-pinned nimble dependency (trace_of_radiance.nimble:9-10).
 *
 * Modes (all selectable per call through OracleOptions):
 *   seeding  PIXEL  : rng.seed(row,col) once per pixel, spp samples share the stream
 *                     (render.nim:59-67)  -- the reference's behaviour.
 *            SAMPLE : rng re-seeded per (row,col,sample) (counter-based stream; the
 *                     extension BASELINE.json's north_star asks for; defined in
 *                     oracle_seed3 below and mirrored by the HIP kernel).
 *   math     LIBM     : sin/cos/pow from the host libm, as the Nim build does.
 *            PORTABLE : sin/cos/pow from the double-double routines below, which use
 *                       only IEEE +,-,*,/,fma and are therefore bit-identical on the
 *                       CPU and on gfx950 (the HIP kernel carries the same algorithm).
 *   arith    STRICT : reference operation order, no FMA contraction (README.md:82:
 *                     the reference build is SSE2, no FMA). Compile with
 *                     -ffp-contract=off.
 *            FUSED  : same formulas with explicit fma() in the ray/sphere quadratic
 *                     (documented throughput variant; NOT the reference's rounding).
 *   accum    SEQUENTIAL : pixel += radiance in sample order (render.nim:67).
 *            QUANTIZED  : each sample's radiance is rounded to a multiple of 2^-36
 *                         before being added; every partial sum is then exactly
 *                         representable, so the sum is order-independent (what the
 *                         GPU's per-sample mode uses to stay deterministic).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* Types (oracle-private layouts; the product's ABI structs live in include/) */
/* ------------------------------------------------------------------------- */

typedef struct { double x, y, z; } V3;

typedef struct { uint64_t s0, s1, s2, s3; } Rng;

enum { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2 }; /* core.nim:25-27 */
enum { OBJ_SPHERE = 0, OBJ_MOVING_SPHERE = 1 };                 /* hittables_variants.nim:53-54 */

/* Flat object record used by the oracle (16 doubles).  kind/material kind are stored
 * as doubles so the whole scene is one float64 array that numpy can hold. */
typedef struct {
  double kind;          /* 0 sphere, 1 moving sphere */
  double c0x, c0y, c0z; /* center (sphere) / center0 */
  double c1x, c1y, c1z; /* center1 (moving sphere), == center0 for a sphere */
  double t0, t1;        /* time0,time1 (moving sphere) */
  double radius;
  double mat;           /* 0 lambertian, 1 metal, 2 dielectric */
  double ax, ay, az;    /* albedo */
  double fuzz;          /* metal */
  double ri;            /* dielectric refraction index */
} Obj;

typedef struct {
  V3 origin, lower_left_corner, horizontal, vertical, u, v, w; /* cameras.nim:15-22 */
  double lens_radius, shutter_open, shutter_close;
} Cam;

typedef struct { V3 origin, direction; double time; } Ray; /* rays.nim:14-17 */

typedef struct {
  int32_t seeding;  /* 0 PIXEL, 1 SAMPLE */
  int32_t math;     /* 0 LIBM, 1 PORTABLE */
  int32_t arith;    /* 0 STRICT, 1 FUSED */
  int32_t accum;    /* 0 SEQUENTIAL, 1 QUANTIZED */
  int32_t row_begin, row_end; /* render rows [row_begin,row_end) ; other rows untouched */
  int32_t threads;  /* 0 = OpenMP default */
  int32_t collect_stats;
  int32_t row_step; /* render every row_step-th row starting at row_begin (0/1 = all) */
  int32_t col_block; /* > 0: the parallel loop runs over (row, block of col_block columns) tiles instead of whole rows --
                        same pixels (every pixel owns its stream, render.nim:59-60), better balance for a few rows */
} OracleOptions;

typedef struct {
  uint64_t hit_calls;     /* world.hit() invocations (closest-hit queries) */
  uint64_t object_tests;  /* ray/object tests = hit_calls * n_objects */
  uint64_t rng_draws;     /* next() calls */
  uint64_t scatter_lambertian, scatter_metal, scatter_dielectric;
  uint64_t depth_exhausted, absorbed;
  uint64_t depth_hist[64]; /* hist[k] = samples that made k hit calls (k<=63) */
} OracleStats;

/* ------------------------------------------------------------------------- */
/* Portable math: double-double kernels (only +,-,*,/,fma,compare)            */
/* ------------------------------------------------------------------------- */

typedef struct { double hi, lo; } dd;

static inline dd two_sum(double a, double b) {
  double s = a + b, bb = s - a;
  dd r = { s, (a - (s - bb)) + (b - bb) };
  return r;
}
static inline dd fast_two_sum(double a, double b) { /* |a| >= |b| or a == 0 */
  double s = a + b;
  dd r = { s, b - (s - a) };
  return r;
}
static inline dd two_prod(double a, double b) {
  double p = a * b;
  dd r = { p, fma(a, b, -p) };
  return r;
}
static inline dd dd_add(dd a, dd b) {
  dd s = two_sum(a.hi, b.hi);
  dd t = two_sum(a.lo, b.lo);
  double c = s.lo + t.hi;
  dd v = fast_two_sum(s.hi, c);
  double w = t.lo + v.lo;
  return fast_two_sum(v.hi, w);
}
static inline dd dd_add_d(dd a, double b) {
  dd s = two_sum(a.hi, b);
  double c = s.lo + a.lo;
  return fast_two_sum(s.hi, c);
}
static inline dd dd_mul(dd a, dd b) {
  dd p = two_prod(a.hi, b.hi);
  double e = fma(a.hi, b.lo, fma(a.lo, b.hi, p.lo));
  return fast_two_sum(p.hi, e);
}
static inline dd dd_mul_d(dd a, double b) {
  dd p = two_prod(a.hi, b);
  double e = fma(a.lo, b, p.lo);
  return fast_two_sum(p.hi, e);
}
static inline dd dd_div(dd a, dd b) { /* a / b, ~2^-100 relative */
  double q1 = a.hi / b.hi;
  dd r = dd_add(a, dd_mul_d(b, -q1));
  double q2 = r.hi / b.hi;
  r = dd_add(r, dd_mul_d(b, -q2));
  double q3 = r.hi / b.hi;
  dd q = fast_two_sum(q1, q2);
  return dd_add_d(q, q3);
}

/* (-1)^k/(2k+1)!, k=1..10  (tools/gen_dd_constants.py) */
static const dd SIN_S[10] = {
  { -0x1.5555555555555p-3, -0x1.5555555555555p-57 },
  { 0x1.1111111111111p-7, 0x1.1111111111111p-63 },
  { -0x1.a01a01a01a01ap-13, -0x1.a01a01a01a01ap-73 },
  { 0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73 },
  { -0x1.ae64567f544e4p-26, 0x1.c062e06d1f209p-80 },
  { 0x1.6124613a86d09p-33, 0x1.f28e0cc748ebep-87 },
  { -0x1.ae7f3e733b81fp-41, -0x1.1d8656b0ee8cbp-97 },
  { 0x1.952c77030ad4ap-49, 0x1.ac981465ddc6cp-103 },
  { -0x1.2f49b46814157p-57, -0x1.2650f61dbdcb4p-112 },
  { 0x1.71b8ef6dcf572p-66, -0x1.d043ae40c4647p-120 },
};
/* (-1)^k/(2k)!, k=1..11 */
static const dd COS_C[11] = {
  { -0x1.0000000000000p-1, 0x0.0p+0 },
  { 0x1.5555555555555p-5, 0x1.5555555555555p-59 },
  { -0x1.6c16c16c16c17p-10, 0x1.f49f49f49f49fp-65 },
  { 0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76 },
  { -0x1.27e4fb7789f5cp-22, -0x1.cbbc05b4fa99ap-76 },
  { 0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83 },
  { -0x1.93974a8c07c9dp-37, -0x1.05d6f8a2efd1fp-92 },
  { 0x1.ae7f3e733b81fp-45, 0x1.1d8656b0ee8cbp-101 },
  { -0x1.6827863b97d97p-53, -0x1.eec01221a8b0bp-107 },
  { 0x1.e542ba4020225p-62, 0x1.ea72b4afe3c2fp-120 },
  { -0x1.0ce396db7f853p-70, 0x1.aebcdbd20331cp-124 },
};
/* 1/(2k+1), k=0..21 */
static const dd INV_ODD[22] = {
  { 0x1.0000000000000p+0, 0x0.0p+0 },
  { 0x1.5555555555555p-2, 0x1.5555555555555p-56 },
  { 0x1.999999999999ap-3, -0x1.999999999999ap-57 },
  { 0x1.2492492492492p-3, 0x1.2492492492492p-57 },
  { 0x1.c71c71c71c71cp-4, 0x1.c71c71c71c71cp-58 },
  { 0x1.745d1745d1746p-4, -0x1.745d1745d1746p-59 },
  { 0x1.3b13b13b13b14p-4, -0x1.3b13b13b13b14p-58 },
  { 0x1.1111111111111p-4, 0x1.1111111111111p-60 },
  { 0x1.e1e1e1e1e1e1ep-5, 0x1.e1e1e1e1e1e1ep-61 },
  { 0x1.af286bca1af28p-5, 0x1.af286bca1af28p-59 },
  { 0x1.8618618618618p-5, 0x1.8618618618618p-59 },
  { 0x1.642c8590b2164p-5, 0x1.642c8590b2164p-60 },
  { 0x1.47ae147ae147bp-5, -0x1.eb851eb851eb8p-61 },
  { 0x1.2f684bda12f68p-5, 0x1.2f684bda12f68p-59 },
  { 0x1.1a7b9611a7b96p-5, 0x1.1a7b9611a7b96p-61 },
  { 0x1.0842108421084p-5, 0x1.0842108421084p-60 },
  { 0x1.f07c1f07c1f08p-6, -0x1.f07c1f07c1f08p-61 },
  { 0x1.d41d41d41d41dp-6, 0x1.0750750750750p-60 },
  { 0x1.bacf914c1bad0p-6, -0x1.bacf914c1bad0p-60 },
  { 0x1.a41a41a41a41ap-6, 0x1.0690690690690p-60 },
  { 0x1.8f9c18f9c18fap-6, -0x1.f3831f3831f38p-61 },
  { 0x1.7d05f417d05f4p-6, 0x1.7d05f417d05f4p-62 },
};
/* 1/k!, k=0..16 */
static const dd INV_FACT[17] = {
  { 0x1.0000000000000p+0, 0x0.0p+0 },
  { 0x1.0000000000000p+0, 0x0.0p+0 },
  { 0x1.0000000000000p-1, 0x0.0p+0 },
  { 0x1.5555555555555p-3, 0x1.5555555555555p-57 },
  { 0x1.5555555555555p-5, 0x1.5555555555555p-59 },
  { 0x1.1111111111111p-7, 0x1.1111111111111p-63 },
  { 0x1.6c16c16c16c17p-10, -0x1.f49f49f49f49fp-65 },
  { 0x1.a01a01a01a01ap-13, 0x1.a01a01a01a01ap-73 },
  { 0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76 },
  { 0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73 },
  { 0x1.27e4fb7789f5cp-22, 0x1.cbbc05b4fa99ap-76 },
  { 0x1.ae64567f544e4p-26, -0x1.c062e06d1f209p-80 },
  { 0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83 },
  { 0x1.6124613a86d09p-33, 0x1.f28e0cc748ebep-87 },
  { 0x1.93974a8c07c9dp-37, 0x1.05d6f8a2efd1fp-92 },
  { 0x1.ae7f3e733b81fp-41, 0x1.1d8656b0ee8cbp-97 },
  { 0x1.ae7f3e733b81fp-45, 0x1.1d8656b0ee8cbp-101 },
};
static const dd DD_LN2 = { 0x1.62e42fefa39efp-1, 0x1.abc9e3b39803fp-56 };
#define TWO_OVER_PI 0x1.45f306dc9c883p-1
#define INV_LN2 0x1.71547652b82fep+0
/* pi/2 = P1+P2+P3+P4, first three carry 33 bits so k*Pi is exact for small k */
#define PIO2_1 0x1.921fb54400000p+0
#define PIO2_2 0x1.0b4611a600000p-34
#define PIO2_3 0x1.3198a2e000000p-69
#define PIO2_4 0x1.b839a252049c1p-104

/* sin and cos of a in [0, 8): the argument range the path produces is
 * [0, 2*pi) (sampling.nim:52).  Both results are the double nearest to the exact
 * value except with probability ~2^-17 per call. */
static void port_sincos_slow_dd(double a, double* s_out, double* c_out, double* lo_out);
static void port_sincos_slow(double a, double* s_out, double* c_out) { port_sincos_slow_dd(a, s_out, c_out, NULL); }
/* (lo_out, tests only: the low words of the two double-double results, {sin, cos}) */
static void port_sincos_slow_dd(double a, double* s_out, double* c_out, double* lo_out) {
  if (lo_out) lo_out[0] = lo_out[1] = 0.0;
  if (!(a >= 0.0 && a < 8.0)) { *s_out = sin(a); *c_out = cos(a); return; } /* off-path */
  int k = (int)(a * TWO_OVER_PI + 0.5);
  double fk = (double)k;
  /* r = a - k*pi/2 as a double-double */
  double r0 = a - fk * PIO2_1; /* exact (Sterbenz) for k>=1, trivially for k==0 */
  dd r = two_sum(r0, -(fk * PIO2_2));
  r = dd_add_d(r, -(fk * PIO2_3));
  r = dd_add_d(r, -(fk * PIO2_4));
  dd z = dd_mul(r, r);
  /* sin(r) = r*(1 + z*(S1 + z*(S2 + z*(S3 + z*Q)))), Q in double */
  double zh = z.hi;
  double q = SIN_S[9].hi;
  for (int i = 8; i >= 3; --i) q = fma(q, zh, SIN_S[i].hi);
  dd t = dd_add(SIN_S[2], dd_mul_d(z, q));
  t = dd_add(SIN_S[1], dd_mul(z, t));
  t = dd_add(SIN_S[0], dd_mul(z, t));
  t = dd_add_d(dd_mul(z, t), 1.0);
  dd sn = dd_mul(r, t);
  /* cos(r) = 1 + z*(C1 + z*(C2 + z*(C3 + z*(C4 + z*Q)))) */
  double p = COS_C[10].hi;
  for (int i = 9; i >= 4; --i) p = fma(p, zh, COS_C[i].hi);
  dd u = dd_add(COS_C[3], dd_mul_d(z, p));
  u = dd_add(COS_C[2], dd_mul(z, u));
  u = dd_add(COS_C[1], dd_mul(z, u));
  u = dd_add(COS_C[0], dd_mul(z, u));
  dd cs = dd_add_d(dd_mul(z, u), 1.0);
  double S = sn.hi, C = cs.hi;
  switch (k & 3) {
    case 0: *s_out = S;  *c_out = C;  break;
    case 1: *s_out = C;  *c_out = -S; break;
    case 2: *s_out = -S; *c_out = -C; break;
    default: *s_out = -C; *c_out = S; break;
  }
  if (lo_out) {
    switch (k & 3) {
      case 0: lo_out[0] = sn.lo;  lo_out[1] = cs.lo;  break;
      case 1: lo_out[0] = cs.lo;  lo_out[1] = -sn.lo; break;
      case 2: lo_out[0] = -sn.lo; lo_out[1] = -cs.lo; break;
      default: lo_out[0] = -cs.lo; lo_out[1] = sn.lo; break;
    }
  }
}

/* ---- sincos: fast path (round 5).  The double-double evaluation above costs ~290 float64 operations per call and the GPU runs it
 * on every bounce iteration of every wave (sampling.nim:51-55 through materials.nim:24-30).  The fast path gets the SAME doubles
 * -- both are the correctly rounded value -- from ~85 operations and PROVES it per call (Ziv's rounding test); a call whose test
 * fails (~1 in 300) takes the double-double path.  csrc/tor_math.hpp carries the same operations in the same order.
 *   a = j h + r, h = pi/128, |r| <= h/2 (+ one rounding of a * 128/pi); r = rh + rl to ~2^-125 (H1, H2 short: j * H1 and j * H2 are
 *     exact for j <= 256, a - j * H1 is exact by Sterbenz's lemma);
 *   sin r - r = rh z ps(z), cos r - 1 = -z/2 + ce, z = rh^2 (Taylor to r^9 / r^8);
 *   sin a = S (1 + (cos r - 1)) + C (r + (sin r - r)) with (S, C) = sin, cos(j h) from a table of double-doubles: the large terms
 *     S.hi, C.hi rh, S.hi (-z/2) enter exactly (two_prod, fast_two_sum), the rest in one correction of relative size < 2^-14;
 *     cos a likewise with (C, -S).
 *   Rounding test: the result is h + corr, and if h + (corr - e) and h + (corr + e) round to the same double for an e that
 *     bounds |h + corr - exact|, that double is the correctly rounded value.  e = 2^-63 |h| (round 6; 2^-64 before), four times
 *     the bound below; it costs a fall-back rate of ~1 in 300 instead of 1 in 600.
 *   Error budget of h + corr (sin side; cos is the same with the roles of S and C swapped), relative to |h|, rho = |r| <=
 *     0.012273; the worst case is a zero crossing of the result (S ~ 0: |h| ~ |C| rho), elsewhere |h| >= 0.012 and every
 *     term is smaller by rho / |h|; u = 2^-53:
 *       reduction   r = rh + rl up to 256 x the tail of pi/128 behind H3 (2^-128) + the rounding of rl (u^2 rho): at j = 0
 *                   a = r exactly, at j = 128, 256 |r| >= 2^-53 (pi is not a double), elsewhere |h| >= 0.012  -> < 2^-75
 *       table       S, C to 2^-106 relative; sin(pi) = 0 and cos(pi / 2) = 0 are exact entries                  -> < 2^-99
 *       truncation  sin r - r after r^9: rho^10 / 11!                                                            -> 2^-88.7
 *                   cos r - 1 after r^8: rho^10 / 10! (times |S| / |h| <= 2)                                     -> 2^-84.3
 *       sr          = (rh z) ps: roundings of z, of the three fma of ps, of rh z, of the product and the four rounded
 *                   coefficients: 4.1 u |sr|, |sr| <= rho^3 / 6                                                  -> 2^-66.24
 *       sr uses rh  d(sin r - r)/dr rl = (cos r - 1) rl, |rl| <= u rho: u rho^2 / 2                              -> 2^-66.70
 *       t           = fl(rl + sr): u |t|                                                                         -> 2^-68.3
 *       ce          ~3 u |ce|, |ce| <= rho^4 / 24 + u rho^2 = 2^-30, times |S| / |h| <= 2                        -> 2^-80
 *       corr        the sum of the four exact low parts and Sl (1 + hz): each <= u max(|S|, |h|), three additions -> 2^-102
 *                   fma(Cl, rh, .), fma(Sh, ce, .): u (2^-30 |S| + 2^-51 |h|)                                    -> 2^-82
 *                   fma(Ch, t, .): u (rho^3 / 6 + ...)                                                           -> 2^-68.3
 *       (second-order terms: u^2)                                                                                -> 2^-100
 *     sum < 2^-64.9 |h|   (the worst measured over 1.95 M arguments incl. the table's nodes, the half-interval boundaries and
 *     the neighbourhoods of pi / 2, pi, 3 pi / 2, 2 pi: 2^-65.6 -- tests/test_oracle_math.py checks EVERY argument against the
 *     double-double evaluation, whose own error, < 2^-73, is checked against mpmath). */
static const double SINCOS_TAB[257 * 4] = {
#include "tor_sincos_table.inc"
};
#define SC_INV_H 0x1.45f306dc9c883p+5
#define SC_H1 0x1.921fb54000000p-6
#define SC_H2 0x1.10b4611a62600p-36
#define SC_H3 0x1.98a2e03707345p-83
#define SC_EPS 0x1p-63
static inline int sc_ziv(double h, double corr, double* out) {
  const double e = fabs(h) * SC_EPS;
  const double r1 = h + (corr + e), r2 = h + (corr - e);
  *out = r1;
  return r1 == r2;
}
/* returns 1 when both results passed the rounding test */
static int port_sincos_fast_parts(double a, double* s_out, double* c_out, double* parts) {
  if (!(a >= 0.0 && a < 6.2890625)) return 0;   /* j <= 256 */
  const int j = (int)(a * SC_INV_H + 0.5);
  const double fj = (double)j;
  const double r0 = fma(-fj, SC_H1, a);         /* exact */
  const dd r = two_sum(r0, -(fj * SC_H2));      /* fj * H2 exact */
  const double rh = r.hi, rl = fma(-fj, SC_H3, r.lo);
  const double z = rh * rh, zl = fma(rh, rh, -z);
  const double ps = fma(fma(fma(0x1.71de3a556c734p-19, z, -0x1.a01a01a01a01ap-13), z, 0x1.1111111111111p-7), z, -0x1.5555555555555p-3);
  const double sr = (rh * z) * ps;                                  /* sin r - r */
  const double pc = fma(fma(0x1.a01a01a01a01ap-16, z, -0x1.6c16c16c16c17p-10), z, 0x1.5555555555555p-5);
  const double hz = -0.5 * z;                                       /* exact */
  const double ce = fma(z * z, pc, fma(-0.5, zl, -(rh * rl)));      /* cos r - 1 - hz */
  const double t = rl + sr;
  const double* T = SINCOS_TAB + 4 * j;
  const double Sh = T[0], Sl = T[1], Ch = T[2], Cl = T[3];
  int ok;
  {
    const dd p = two_prod(Ch, rh), q = two_prod(Sh, hz);
    const dd u = fast_two_sum(Sh, p.hi);
    const dd v = fast_two_sum(u.hi, q.hi);
    double corr = ((u.lo + v.lo) + (p.lo + q.lo)) + fma(Sl, hz, Sl);
    corr = fma(Cl, rh, corr);
    corr = fma(Sh, ce, corr);
    corr = fma(Ch, t, corr);
    ok = sc_ziv(v.hi, corr, s_out);
    if (parts) { parts[0] = v.hi; parts[1] = corr; }
  }
  {
    const dd p = two_prod(Sh, rh), q = two_prod(Ch, hz);
    const dd u = fast_two_sum(Ch, -p.hi);
    const dd v = fast_two_sum(u.hi, q.hi);
    double corr = ((u.lo + v.lo) + (q.lo - p.lo)) + fma(Cl, hz, Cl);
    corr = fma(-Sl, rh, corr);
    corr = fma(Ch, ce, corr);
    corr = fma(-Sh, t, corr);
    ok &= sc_ziv(v.hi, corr, c_out);
    if (parts) { parts[2] = v.hi; parts[3] = corr; }
  }
  return ok;
}
static int port_sincos_fast(double a, double* s_out, double* c_out) { return port_sincos_fast_parts(a, s_out, c_out, NULL); }
static void port_sincos(double a, double* s_out, double* c_out) {
  if (!port_sincos_fast(a, s_out, c_out)) port_sincos_slow(a, s_out, c_out);
}
/* MATH_LIBM: sampling.nim:54 calls sin(a) and cos(a); gcc -O2 merges two such calls in one function into ONE sincos() -- it does
 * for the C that Nim emits, and it did for this file when tests/golden/small_canvases.npz was generated -- and glibc's sincos()
 * differs from its sin() / cos() by an ulp now and then.  Whether gcc merged depended on what else the function inlined (round 5:
 * a larger port_sincos turned the merge off and moved one sample of the fixtures by an ulp), so the call is spelled out. */
extern void sincos(double, double*, double*);

/* x^5, correctly rounded w.h.p. (what a correctly rounded pow(x,5) returns). */
static double port_pow5(double x) {
  dd x2 = two_prod(x, x);
  dd x4 = dd_mul(x2, x2);
  dd x5 = dd_mul_d(x4, x);
  return x5.hi;
}

static inline double bits_to_double(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static inline uint64_t double_to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

/* pow(x, y) for finite x >= 0 and finite y > 0 (the gamma curve, canvas.nim:52-54).
 * Other arguments fall back to libm (never produced by the path). */
static double port_pow(double x, double y) {
  if (!(x >= 0.0) || !(y > 0.0) || x > 0x1p+1000 || y > 0x1p+20) return pow(x, y);
  if (x == 0.0) return 0.0;
  int e = 0;
  if (x < 0x1p-1022) { x *= 0x1p+54; e = -54; } /* subnormal */
  uint64_t b = double_to_bits(x);
  e += (int)((b >> 52) & 0x7ff) - 1023;
  double m = bits_to_double((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL); /* [1,2) */
  if (m > 0x1.6a09e667f3bcdp+0) { m *= 0.5; e += 1; } /* m in (sqrt(1/2), sqrt(2)] */
  /* log(m) = 2*atanh(s), s = (m-1)/(m+1) */
  dd num = { m - 1.0, 0.0 };
  dd den = two_sum(m, 1.0);
  dd s = dd_div(num, den);
  dd s2 = dd_mul(s, s);
  dd acc = INV_ODD[21];
  for (int i = 20; i >= 0; --i) acc = dd_add(INV_ODD[i], dd_mul(s2, acc));
  dd lg = dd_mul(s, acc);
  lg.hi *= 2.0; lg.lo *= 2.0;
  dd L = dd_add(dd_mul_d(DD_LN2, (double)e), lg);
  dd P = dd_mul_d(L, y);
  /* exp(P) = 2^n * exp(r), r = P - n*ln2 */
  double fn = floor(P.hi * INV_LN2 + 0.5);
  dd r = dd_add(P, dd_mul_d(DD_LN2, -fn));
  r.hi *= 0.125; r.lo *= 0.125;
  dd ex = INV_FACT[16];
  for (int i = 15; i >= 0; --i) ex = dd_add(INV_FACT[i], dd_mul(r, ex));
  ex = dd_mul(ex, ex); ex = dd_mul(ex, ex); ex = dd_mul(ex, ex);
  int n = (int)fn;
  double res = ex.hi;
  /* scale by 2^n in two exact steps (n is within [-1100, 1100] here) */
  int n1 = n / 2, n2 = n - n1;
  res *= bits_to_double((uint64_t)(n1 + 1023) << 52);
  res *= bits_to_double((uint64_t)(n2 + 1023) << 52);
  return res;
}

/* ------------------------------------------------------------------------- */
/* RNG  (support/rng.nim)                                                     */
/* ------------------------------------------------------------------------- */

/* rng.nim:31-36 -- note BOTH multipliers are 0xbf58476d1ce4e5b9 (reference quirk) */
static inline uint64_t splitmix64(uint64_t* state) {
  *state += 0x9e3779b97f4a7c15ULL;
  uint64_t r = *state;
  r = (r ^ (r >> 30)) * 0xbf58476d1ce4e5b9ULL;
  r = (r ^ (r >> 27)) * 0xbf58476d1ce4e5b9ULL;
  return r ^ (r >> 31);
}
/* rng.nim:21-29 */
static inline uint64_t pair_u64(uint64_t x, uint64_t y) { return (x << 32) ^ y; }

/* rng.nim:38-44 */
static void rng_seed1(Rng* g, uint64_t x) {
  uint64_t sm = x;
  g->s0 = splitmix64(&sm); g->s1 = splitmix64(&sm);
  g->s2 = splitmix64(&sm); g->s3 = splitmix64(&sm);
}
/* rng.nim:46-53 */
static void rng_seed2(Rng* g, uint64_t x, uint64_t y) { rng_seed1(g, pair_u64(x, y)); }

/* Per-sample stream (NOT in the reference; the counter-based extension):
 *   sm = pair(row,col); h = splitMix64(sm); sm' = h xor sample; s0..s3 = splitMix64(sm') x4 */
static void oracle_seed3(Rng* g, uint64_t row, uint64_t col, uint64_t sample) {
  uint64_t sm = pair_u64(row, col);
  uint64_t h = splitmix64(&sm);
  rng_seed1(g, h ^ sample);
}

/* rng.nim:58-74 xoshiro256+ */
static inline uint64_t rng_next(Rng* g, OracleStats* st) {
  uint64_t res = g->s0 + g->s3;
  uint64_t t = g->s1 << 17;
  g->s2 ^= g->s0; g->s3 ^= g->s1; g->s1 ^= g->s2; g->s0 ^= g->s3;
  g->s2 ^= t;
  g->s3 = (g->s3 << 45) | (g->s3 >> 19);
  if (st) st->rng_draws++;
  return res;
}
/* rng.nim:129-133 */
static inline double uniform01(Rng* g, OracleStats* st) {
  uint64_t m = rng_next(g, st) >> 12;
  return bits_to_double(m | 0x3ff0000000000000ULL) - 1.0;
}
/* rng.nim:135-143 */
static inline double uniform_max(Rng* g, double max_excl, OracleStats* st) {
  return uniform01(g, st) * max_excl;
}
/* rng.nim:116-127 ; Nim's max(x,y) is `if y <= x: x else: y` */
static inline double uniform_range(Rng* g, double lo, double hi, OracleStats* st) {
  double d = uniform01(g, st);
  double v = d * (hi - lo) + lo;
  return (v <= lo) ? lo : v;
}

/* ------------------------------------------------------------------------- */
/* Vec3 (primitives/vec3s.nim, point3s.nim, colors.nim)                        */
/* ------------------------------------------------------------------------- */

static inline V3 v3(double x, double y, double z) { V3 r = { x, y, z }; return r; }
static inline V3 vadd(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 vsub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 vneg(V3 a) { return v3(-a.x, -a.y, -a.z); }
static inline V3 vscale(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); } /* vec3s.nim:86-91 */
static inline V3 vdiv(V3 a, double s) { return vscale(a, 1.0 / s); }              /* vec3s.nim:93-94 */
static inline double vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* :96-98 */
static inline double vlen2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }      /* :19-20 */
static inline V3 vcross(V3 u, V3 v) {                                               /* :100-104 */
  return v3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
static inline V3 vunit(V3 a) { return vdiv(a, sqrt(vlen2(a))); } /* :106-107 */

/* ------------------------------------------------------------------------- */
/* Sampling (sampling.nim)                                                     */
/* ------------------------------------------------------------------------- */

/* sampling.nim:64-68 */
static V3 random_in_unit_disk(Rng* g, OracleStats* st) {
  for (;;) {
    double x = uniform_range(g, -1.0, 1.0, st);
    double y = uniform_range(g, -1.0, 1.0, st);
    V3 p = v3(x, y, 0.0);
    if (vlen2(p) < 1.0) return p;
  }
}
/* sampling.nim:45-49 */
static V3 random_in_unit_sphere(Rng* g, OracleStats* st) {
  for (;;) {
    double x = uniform_range(g, -1.0, 1.0, st);
    double y = uniform_range(g, -1.0, 1.0, st);
    double z = uniform_range(g, -1.0, 1.0, st);
    V3 p = v3(x, y, z);
    if (vlen2(p) < 1.0) return p;
  }
}
/* sampling.nim:51-55 ; 2*PI is Nim's float64 constant 2*3.141592653589793 */
static V3 random_unit_vector(Rng* g, int math_mode, OracleStats* st) {
  double a = uniform_max(g, 2.0 * 3.141592653589793, st);
  double z = uniform_range(g, -1.0, 1.0, st);
  double r = sqrt(1.0 - z * z);
  double s, c;
  if (math_mode == 0) sincos(a, &s, &c); else port_sincos(a, &s, &c);
  return v3(r * c, r * s, z);
}

/* ------------------------------------------------------------------------- */
/* Camera (physics/cameras.nim)                                                */
/* ------------------------------------------------------------------------- */

/* cameras.nim:24-45 (host side; uses libm tan like the reference) */
EXPORT void oracle_camera(const double look_from[3], const double look_at[3], const double vup[3],
                          double vfov_degrees, double aspect_ratio, double aperture,
                          double focus_distance, double shutter_open, double shutter_close,
                          double out24[24]) {
  Cam c;
  double theta = vfov_degrees * (3.141592653589793 / 180.0); /* std/math degToRad */
  double h = tan(theta / 2.0);
  double viewport_height = 2.0 * h;
  double viewport_width = aspect_ratio * viewport_height;
  V3 lf = v3(look_from[0], look_from[1], look_from[2]);
  V3 la = v3(look_at[0], look_at[1], look_at[2]);
  V3 up = v3(vup[0], vup[1], vup[2]);
  c.w = vunit(vsub(lf, la));
  c.u = vunit(vcross(up, c.w));
  c.v = vcross(c.w, c.u);
  c.origin = lf;
  c.horizontal = vscale(c.u, focus_distance * viewport_width);
  c.vertical = vscale(c.v, focus_distance * viewport_height);
  c.lower_left_corner = vsub(vsub(vsub(c.origin, vdiv(c.horizontal, 2.0)), vdiv(c.vertical, 2.0)),
                             vscale(c.w, focus_distance));
  c.lens_radius = aperture / 2.0;
  c.shutter_open = shutter_open;
  c.shutter_close = shutter_close;
  memcpy(out24, &c, sizeof(Cam));
}

/* cameras.nim:47-57 */
static Ray camera_ray(const Cam* c, double s, double t, Rng* g, OracleStats* st) {
  V3 rd = vscale(random_in_unit_disk(g, st), c->lens_radius);
  V3 offset = vadd(vscale(c->u, rd.x), vscale(c->v, rd.y));
  Ray r;
  r.origin = vadd(c->origin, offset);
  r.direction = vsub(vsub(vadd(vadd(c->lower_left_corner, vscale(c->horizontal, s)),
                                vscale(c->vertical, t)), c->origin), offset);
  r.time = uniform_range(g, c->shutter_open, c->shutter_close, st);
  return r;
}

/* ------------------------------------------------------------------------- */
/* Hittables (physics/hittables/*.nim) and materials (physics/materials.nim)   */
/* ------------------------------------------------------------------------- */

typedef struct { V3 p, normal; const Obj* obj; double t; int front_face; } HitRecord; /* core.nim:30-36 */

/* moving_spheres.nim:39-44 */
static inline V3 obj_center(const Obj* o, double time, int arith) {
  V3 c0 = v3(o->c0x, o->c0y, o->c0z);
  if (o->kind == OBJ_SPHERE) return c0;
  V3 c1 = v3(o->c1x, o->c1y, o->c1z);
  V3 dc = vsub(c1, c0);
  double f = (time - o->t0) / (o->t1 - o->t0);
  if (arith == 0) return vadd(c0, vscale(dc, f));
  return v3(fma(dc.x, f, c0.x), fma(dc.y, f, c0.y), fma(dc.z, f, c0.z)); /* FUSED variant */
}

/* spheres.nim:28-49 and moving_spheres.nim:46-67 (identical but for the centre) */
static int obj_hit(const Obj* o, const Ray* r, double t_min, double t_max, int arith, HitRecord* rec) {
  V3 center = obj_center(o, r->time, arith);
  V3 oc = vsub(r->origin, center);
  double a, half_b, c, disc;
  if (arith == 0) {
    a = vlen2(r->direction);
    half_b = vdot(oc, r->direction);
    c = vlen2(oc) - o->radius * o->radius;
    disc = half_b * half_b - a * c;
  } else { /* FUSED variant: same formulas, explicit fma */
    V3 d = r->direction;
    a = fma(d.z, d.z, fma(d.y, d.y, d.x * d.x));
    half_b = fma(oc.z, d.z, fma(oc.y, d.y, oc.x * d.x));
    c = fma(oc.z, oc.z, fma(oc.y, oc.y, fma(oc.x, oc.x, -(o->radius * o->radius))));
    disc = fma(half_b, half_b, -(a * c));
  }
  if (disc > 0) {
    double root = sqrt(disc);
    double sol = (-half_b - root) / a;
    if (!(t_min < sol && sol < t_max)) {
      sol = (-half_b + root) / a;
      if (!(t_min < sol && sol < t_max)) return 0;
    }
    rec->t = sol;
    rec->p = vadd(r->origin, vscale(r->direction, sol)); /* rays.nim:24-25 */
    V3 outward = vdiv(vsub(rec->p, center), o->radius);
    rec->front_face = vdot(r->direction, outward) < 0;   /* core.nim:47-49 */
    rec->normal = rec->front_face ? outward : vneg(outward);
    rec->obj = o;
    return 1;
  }
  return 0;
}

/* hittables_lists.nim:48-55 */
static int world_hit(const Obj* objs, int64_t n, const Ray* r, double t_min, double t_max,
                     int arith, HitRecord* rec, OracleStats* st) {
  double closest = t_max;
  int any = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (obj_hit(&objs[i], r, t_min, closest, arith, rec)) { closest = rec->t; any = 1; }
  }
  if (st) { st->hit_calls++; st->object_tests += (uint64_t)n; }
  return any;
}

/* rays.nim:27-28 */
static inline V3 reflect(V3 u, V3 n) { return vsub(u, vscale(n, 2.0 * vdot(u, n))); }
/* rays.nim:30-37 */
static inline V3 refract(V3 uv, V3 n, double eta) {
  double cos_theta = vdot(vneg(uv), n);
  V3 par = vscale(vadd(uv, vscale(n, cos_theta)), eta);
  V3 perp = vscale(n, -sqrt(1.0 - vlen2(par)));
  return vadd(par, perp);
}
/* materials.nim:55-60 */
static inline double schlick(double cosine, double ri, int math_mode) {
  double r0 = (1.0 - ri) / (1.0 + ri);
  r0 *= r0;
  double p5 = (math_mode == 0) ? pow(1.0 - cosine, 5.0) : port_pow5(1.0 - cosine);
  return r0 + (1.0 - r0) * p5;
}

/* materials.nim:24-30 / 39-47 / 62-86, dispatch :91-96 */
static int scatter(const Ray* r_in, const HitRecord* rec, Rng* g, int math_mode,
                   V3* attenuation, Ray* scattered, OracleStats* st) {
  const Obj* o = rec->obj;
  int mat = (int)o->mat;
  if (mat == MAT_LAMBERTIAN) {
    if (st) st->scatter_lambertian++;
    V3 dir = vadd(rec->normal, random_unit_vector(g, math_mode, st));
    scattered->origin = rec->p; scattered->direction = dir; scattered->time = r_in->time;
    *attenuation = v3(o->ax, o->ay, o->az);
    return 1;
  } else if (mat == MAT_METAL) {
    if (st) st->scatter_metal++;
    V3 reflected = reflect(vunit(r_in->direction), rec->normal);
    V3 dir = vadd(reflected, vscale(random_in_unit_sphere(g, st), o->fuzz));
    scattered->origin = rec->p; scattered->direction = dir; scattered->time = 0.0; /* rays.nim:19 */
    if (vdot(dir, rec->normal) > 0) { *attenuation = v3(o->ax, o->ay, o->az); return 1; }
    return 0;
  } else {
    if (st) st->scatter_dielectric++;
    *attenuation = v3(1.0, 1.0, 1.0);
    double eta = rec->front_face ? 1.0 / o->ri : o->ri;
    V3 ud = vunit(r_in->direction);
    double d = vdot(vneg(ud), rec->normal);
    double cos_theta = (d <= 1.0) ? d : 1.0; /* min(d, 1.0): Nim `if x <= y: x else: y` */
    double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
    scattered->origin = rec->p; scattered->time = 0.0;
    if (eta * sin_theta > 1.0) { scattered->direction = reflect(ud, rec->normal); return 1; }
    double reflect_prob = schlick(cos_theta, eta, math_mode);
    if (uniform01(g, st) < reflect_prob) { scattered->direction = reflect(ud, rec->normal); return 1; }
    scattered->direction = refract(ud, rec->normal, eta);
    return 1;
  }
}

/* render.nim:21-47 */
static V3 radiance(Ray ray, const Obj* objs, int64_t n, int max_depth, Rng* g,
                   int math_mode, int arith, OracleStats* st) {
  V3 att = v3(1.0, 1.0, 1.0);
  int calls = 0;
  for (int depth = 0; depth < max_depth; ++depth) {
    HitRecord rec;
    calls++;
    if (world_hit(objs, n, &ray, 0.001, INFINITY, arith, &rec, st)) {
      V3 matt; Ray sc;
      if (scatter(&ray, &rec, g, math_mode, &matt, &sc, st)) {
        att.x *= matt.x; att.y *= matt.y; att.z *= matt.z;
        ray = sc;
        continue;
      }
      if (st) { st->absorbed++; st->depth_hist[calls < 63 ? calls : 63]++; }
      return v3(0, 0, 0);
    }
    V3 ud = vunit(ray.direction);
    double t = 0.5 * ud.y + 1.0;  /* sic: render.nim:42 */
    V3 res = vadd(vscale(v3(1, 1, 1), 1.0 - t), vscale(v3(0.5, 0.7, 1.0), t));
    res.x *= att.x; res.y *= att.y; res.z *= att.z;
    if (st) st->depth_hist[calls < 63 ? calls : 63]++;
    return res;
  }
  if (st) { st->depth_exhausted++; st->depth_hist[calls < 63 ? calls : 63]++; }
  return v3(0, 0, 0);
}

/* Round x to the nearest multiple of 2^-36 (ties to even); exact for |x| < 2^15.
 * Distance between the two accumulations this oracle offers (ACCUM_QUANTIZED: what TOR_SEED_SAMPLE sums on the GPU, exact in any
 * order; ACCUM_SEQUENTIAL: render.nim:67's plain float64 sum): every sample moves by at most 2^-37, so the sum of spp samples by
 * at most spp * 2^-37 before canvas.nim:49's 1/spp -- 2^-37 = 7.3e-12 per channel after it, whatever spp -- plus the sequential
 * sum's own roundings (spp * 2^-53 * |sum|).  Measured on a row of BASELINE configs[3] (3840x2160, 4096 spp) against the
 * PNG-pinned LIBM / SEQUENTIAL mode: 1.3e-11 (tests/test_gpu_round5.py); the stated tolerance is 1e-5. */
static inline double quantize36(double x) {
  volatile double m = 98304.0; /* 1.5 * 2^16 : ulp = 2^-36 */
  volatile double t = x + m;
  return t - m;
}

static void stats_add(OracleStats* dst, const OracleStats* src) {
  uint64_t* d = (uint64_t*)dst; const uint64_t* s = (const uint64_t*)src;
  for (size_t i = 0; i < sizeof(OracleStats) / 8; ++i) d[i] += s[i];
}

/* render.nim:49-68 + canvas.nim:47-54 (draw).  pixels: nrows*ncols*3 doubles, row 0 = bottom. */
EXPORT int oracle_render(double* pixels, int32_t nrows, int32_t ncols, int32_t spp, float gamma_correction,
                         const double cam24[24], const double* objs16, int64_t n_objects,
                         int32_t max_depth, const OracleOptions* opt, OracleStats* stats_out) {
  OracleOptions o; memset(&o, 0, sizeof o);
  o.row_end = nrows;
  if (opt) o = *opt;
  if (o.row_end <= 0 || o.row_end > nrows) o.row_end = nrows;
  if (o.row_step < 1) o.row_step = 1;
  Cam cam; memcpy(&cam, cam24, sizeof cam);
  const Obj* objs = (const Obj*)objs16;
  if (stats_out) memset(stats_out, 0, sizeof *stats_out);
#ifdef _OPENMP
  if (o.threads > 0) omp_set_num_threads(o.threads);
#endif
#pragma omp parallel
  {
    OracleStats local; memset(&local, 0, sizeof local);
    OracleStats* st = (o.collect_stats && stats_out) ? &local : NULL;
    const int32_t cb = (o.col_block > 0 && o.col_block < ncols) ? o.col_block : ncols;
    const int32_t n_cb = (ncols + cb - 1) / cb;
    const int64_t n_sel = (o.row_end > o.row_begin) ? ((int64_t)(o.row_end - o.row_begin) + o.row_step - 1) / o.row_step : 0;
#pragma omp for schedule(dynamic, 1)
    for (int64_t tile = 0; tile < n_sel * n_cb; ++tile) {
      const int32_t row = o.row_begin + (int32_t)(tile / n_cb) * o.row_step;
      const int32_t col0 = (int32_t)(tile % n_cb) * cb;
      const int32_t col1 = (col0 + cb < ncols) ? col0 + cb : ncols;
      for (int32_t col = col0; col < col1; ++col) {
        Rng g;
        if (o.seeding == 0) rng_seed2(&g, (uint64_t)(int64_t)row, (uint64_t)(int64_t)col);
        V3 pixel = v3(0, 0, 0);
        for (int32_t s = 0; s < spp; ++s) {
          if (o.seeding == 1) oracle_seed3(&g, (uint64_t)row, (uint64_t)col, (uint64_t)s);
          double u = ((double)col + uniform01(&g, st)) / (double)(ncols - 1);
          double v = ((double)row + uniform01(&g, st)) / (double)(nrows - 1);
          Ray r = camera_ray(&cam, u, v, &g, st);
          V3 c = radiance(r, objs, n_objects, max_depth, &g, o.math, o.arith, st);
          if (o.accum == 1) { c.x = quantize36(c.x); c.y = quantize36(c.y); c.z = quantize36(c.z); }
          pixel = vadd(pixel, c);
        }
        /* canvas.nim:47-54 ; gamma_correction is a float32 field (canvas.nim:28) */
        double scale = 1.0 / (double)spp;
        double gamma = 1.0 / (double)gamma_correction;
        double* px = pixels + ((size_t)row * (size_t)ncols + (size_t)col) * 3;
        if (o.math == 0) {
          px[0] = pow(scale * pixel.x, gamma);
          px[1] = pow(scale * pixel.y, gamma);
          px[2] = pow(scale * pixel.z, gamma);
        } else {
          px[0] = port_pow(scale * pixel.x, gamma);
          px[1] = port_pow(scale * pixel.y, gamma);
          px[2] = port_pow(scale * pixel.z, gamma);
        }
      }
    }
    if (st) {
#pragma omp critical
      stats_add(stats_out, &local);
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* random_scene (scenes.nim:13-50)                                             */
/* ------------------------------------------------------------------------- */

static void obj_sphere(Obj* o, V3 c, double radius) {
  memset(o, 0, sizeof *o);
  o->kind = OBJ_SPHERE; o->c0x = c.x; o->c0y = c.y; o->c0z = c.z;
  o->c1x = c.x; o->c1y = c.y; o->c1z = c.z; o->t0 = 0; o->t1 = 1; o->radius = radius;
}
static void mat_lambertian(Obj* o, V3 albedo) { o->mat = MAT_LAMBERTIAN; o->ax = albedo.x; o->ay = albedo.y; o->az = albedo.z; }
static void mat_metal(Obj* o, V3 albedo, double fuzz) { /* materials.nim:35-37 */
  o->mat = MAT_METAL; o->ax = albedo.x; o->ay = albedo.y; o->az = albedo.z; o->fuzz = (fuzz <= 1.0) ? fuzz : 1.0;
}
static void mat_dielectric(Obj* o, double ri) { o->mat = MAT_DIELECTRIC; o->ri = ri; }

/* Returns the number of objects written (<= cap); *draws_out = RNG draws consumed. */
EXPORT int64_t oracle_random_scene(uint64_t seed, double* objs16, int64_t cap, uint64_t* draws_out) {
  Rng g; rng_seed1(&g, seed);
  OracleStats st; memset(&st, 0, sizeof st);
  Obj* objs = (Obj*)objs16;
  int64_t n = 0;
#define PUSH() do { if (n >= cap) return -1; } while (0)
  PUSH(); obj_sphere(&objs[n], v3(0, -1000, 0), 1000); mat_lambertian(&objs[n], v3(0.5, 0.5, 0.5)); n++;
  for (int a = -11; a < 11; ++a) {
    for (int b = -11; b < 11; ++b) {
      double cx = (double)a + 0.9 * uniform01(&g, &st);
      double cz = (double)b + 0.9 * uniform01(&g, &st);
      V3 center = v3(cx, 0.2, cz);
      if (sqrt(vlen2(vsub(center, v3(4, 0.2, 0)))) > 0.9) {
        double choose_mat = uniform01(&g, &st);
        PUSH();
        if (choose_mat < 0.8) {
          double a0 = uniform01(&g, &st), a1 = uniform01(&g, &st), a2 = uniform01(&g, &st);
          double b0 = uniform01(&g, &st), b1 = uniform01(&g, &st), b2 = uniform01(&g, &st);
          V3 albedo = v3(a0 * b0, a1 * b1, a2 * b2);
          V3 center2 = vadd(center, v3(0, uniform_max(&g, 0.5, &st), 0));
          Obj* o = &objs[n];
          memset(o, 0, sizeof *o);
          o->kind = OBJ_MOVING_SPHERE;
          o->c0x = center.x; o->c0y = center.y; o->c0z = center.z;
          o->c1x = center2.x; o->c1y = center2.y; o->c1z = center2.z;
          o->t0 = 0.0; o->t1 = 1.0; o->radius = 0.2;
          mat_lambertian(o, albedo);
        } else if (choose_mat < 0.95) {
          double r0 = uniform_range(&g, 0.5, 1, &st), r1 = uniform_range(&g, 0.5, 1, &st), r2 = uniform_range(&g, 0.5, 1, &st);
          double fuzz = uniform_max(&g, 0.5, &st);
          obj_sphere(&objs[n], center, 0.2); mat_metal(&objs[n], v3(r0, r1, r2), fuzz);
        } else {
          obj_sphere(&objs[n], center, 0.2); mat_dielectric(&objs[n], 1.5);
        }
        n++;
      }
    }
  }
  PUSH(); obj_sphere(&objs[n], v3(0, 1, 0), 1.0); mat_dielectric(&objs[n], 1.5); n++;
  PUSH(); obj_sphere(&objs[n], v3(-4, 1, 0), 1.0); mat_lambertian(&objs[n], v3(0.4, 0.2, 0.1)); n++;
  PUSH(); obj_sphere(&objs[n], v3(4, 1, 0), 1.0); mat_metal(&objs[n], v3(0.7, 0.6, 0.5), 0.0); n++;
#undef PUSH
  if (draws_out) *draws_out = st.rng_draws;
  return n;
}

/* ------------------------------------------------------------------------- */
/* Small exported probes for known-answer tests                                */
/* ------------------------------------------------------------------------- */

EXPORT void oracle_rng_seed1(uint64_t x, uint64_t state[4]) { Rng g; rng_seed1(&g, x); memcpy(state, &g, 32); }
EXPORT void oracle_rng_seed2(uint64_t x, uint64_t y, uint64_t state[4]) { Rng g; rng_seed2(&g, x, y); memcpy(state, &g, 32); }
EXPORT void oracle_rng_seed3(uint64_t r, uint64_t c, uint64_t s, uint64_t state[4]) { Rng g; oracle_seed3(&g, r, c, s); memcpy(state, &g, 32); }
EXPORT uint64_t oracle_rng_next(uint64_t state[4]) { Rng g; memcpy(&g, state, 32); uint64_t r = rng_next(&g, NULL); memcpy(state, &g, 32); return r; }
EXPORT double oracle_rng_uniform01(uint64_t state[4]) { Rng g; memcpy(&g, state, 32); double r = uniform01(&g, NULL); memcpy(state, &g, 32); return r; }
EXPORT double oracle_rng_uniform_range(uint64_t state[4], double lo, double hi) { Rng g; memcpy(&g, state, 32); double r = uniform_range(&g, lo, hi, NULL); memcpy(state, &g, 32); return r; }
EXPORT void oracle_port_sincos(const double* a, double* s, double* c, int64_t n) { for (int64_t i = 0; i < n; ++i) port_sincos(a[i], &s[i], &c[i]); }
/* the double-double path alone, and the fast path alone with its verdict (tests) */
EXPORT void oracle_port_sincos_slow(const double* a, double* s, double* c, int64_t n) { for (int64_t i = 0; i < n; ++i) port_sincos_slow(a[i], &s[i], &c[i]); }
/* the double-double results of the slow path: lo[2 i], lo[2 i + 1] = the low words of sin, cos */
EXPORT void oracle_port_sincos_slow_dd(const double* a, double* s, double* c, double* lo, int64_t n) { for (int64_t i = 0; i < n; ++i) port_sincos_slow_dd(a[i], &s[i], &c[i], lo + 2 * i); }
EXPORT void oracle_port_sincos_fast(const double* a, double* s, double* c, int32_t* ok, double* parts, int64_t n) {
  for (int64_t i = 0; i < n; ++i) ok[i] = port_sincos_fast_parts(a[i], &s[i], &c[i], parts ? parts + 4 * i : NULL);
}
EXPORT void oracle_port_pow5(const double* x, double* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = port_pow5(x[i]); }
EXPORT void oracle_port_pow(const double* x, double e, double* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = port_pow(x[i], e); }
EXPORT void oracle_libm_sincos(const double* a, double* s, double* c, int64_t n) { for (int64_t i = 0; i < n; ++i) { s[i] = sin(a[i]); c[i] = cos(a[i]); } }
EXPORT void oracle_libm_pow(const double* x, double e, double* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = pow(x[i], e); }
EXPORT double oracle_quantize36(double x) { return quantize36(x); }

/* io/ppm.nim:15-16 quantiser: int(256 * clamp(c, 0.0, 0.999)); rows bottom-up like :20.
 * out: nrows*ncols*3 bytes, first row = TOP scanline (row nrows-1 of the canvas). */
EXPORT void oracle_quantize_ppm(const double* pixels, int32_t nrows, int32_t ncols, uint8_t* out) {
  for (int32_t i = nrows - 1, k = 0; i >= 0; --i, ++k)
    for (int32_t j = 0; j < ncols; ++j)
      for (int ch = 0; ch < 3; ++ch) {
        double c = pixels[((size_t)i * ncols + j) * 3 + ch];
        double cl = c < 0.0 ? 0.0 : (c > 0.999 ? 0.999 : c); /* safe_math.nim:10-14 */
        out[((size_t)k * ncols + j) * 3 + ch] = (uint8_t)(int)(256 * cl);
      }
}

EXPORT int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* Animated scene (scenes_animated.nim) -- oracle restatement                  */
/* ------------------------------------------------------------------------- */
/* PARITY UNPINNED for this generator: the reference holds no rendered frame of it, and HEAD
 * does not compile `rng.random(float32)` (scenes_animated.nim:122,136,148; no such overload in
 * sampling.nim:18-25).  Read here as float32(uniform(float64)) with float32 arithmetic. */

typedef struct {
  double velocity, pos_y, coef_restitution, x, z, radius; /* scenes_animated.nim:37-49 */
  double mat, ax, ay, az, fuzz, ri;
} OBall;

typedef struct {
  int32_t nrows, ncols;
  float dt, t_min, t_max, t;
  double look_from_angle;
  int64_t n;
  int skipped;
  OBall balls[1700];
} OAnim;

static double o_velocity_draw(Rng* g) { /* :122 `Velocity(10.0 + (4 * rng.random(float32) - 2.0))` */
  float u = (float)uniform01(g, NULL);
  float v = 10.0f + (4.0f * u - 2.0f);
  return (double)v;
}

EXPORT void* oracle_animation_create(uint64_t seed, int32_t height, int32_t width, float dt, float t_min, float t_max) {
  OAnim* a = (OAnim*)calloc(1, sizeof(OAnim));
  if (!a) return NULL;
  a->nrows = height; a->ncols = width; a->dt = dt; a->t_min = t_min; a->t_max = t_max;
  a->t = 0.0f;
  a->look_from_angle = 2 * 3.141592653589793; /* :104 */
  Rng g; rng_seed1(&g, seed);
  for (int ia = -20; ia < 20; ++ia)
    for (int ib = -20; ib < 20; ++ib) {
      double cx = (double)ia + 0.9 * uniform01(&g, NULL);
      double cz = (double)ib + 0.9 * uniform01(&g, NULL);
      V3 center = v3(cx, 0.2, cz);
      if (sqrt(vlen2(vsub(center, v3(4, 0.2, 0)))) > 0.9) {
        double choose = uniform01(&g, NULL);
        OBall* b = &a->balls[a->n++];
        if (choose < 0.65) { /* :117-129 */
          double a0 = uniform01(&g, NULL), a1 = uniform01(&g, NULL), a2 = uniform01(&g, NULL);
          double b0 = uniform01(&g, NULL), b1 = uniform01(&g, NULL), b2 = uniform01(&g, NULL);
          b->coef_restitution = 0.6; b->velocity = o_velocity_draw(&g);
          b->mat = MAT_LAMBERTIAN; b->ax = a0 * b0; b->ay = a1 * b1; b->az = a2 * b2;
        } else if (choose < 0.95) { /* :130-143 */
          double r0 = uniform_range(&g, 0.5, 1, NULL), r1 = uniform_range(&g, 0.5, 1, NULL), r2 = uniform_range(&g, 0.5, 1, NULL);
          double fuzz = uniform_max(&g, 0.5, NULL);
          b->coef_restitution = 0.5; b->velocity = o_velocity_draw(&g);
          b->mat = MAT_METAL; b->ax = r0; b->ay = r1; b->az = r2; b->fuzz = (fuzz <= 1.0) ? fuzz : 1.0;
        } else { /* :145-154 */
          b->coef_restitution = 0.5; b->velocity = o_velocity_draw(&g);
          b->mat = MAT_DIELECTRIC; b->ri = 1.5;
        }
        b->x = center.x; b->pos_y = center.y; b->z = center.z; b->radius = 0.2;
      }
    }
  return a;
}

EXPORT void oracle_animation_destroy(void* h) { free(h); }
EXPORT int64_t oracle_animation_object_count(void* h) { return ((OAnim*)h)->n + 4; }

static void o_anim_step(OAnim* a) { /* :156-174 */
  a->look_from_angle -= 2.0 * 3.141592653589793 / 1200.0;
  a->t += a->dt;
  double dt64 = (double)a->dt;
  for (int64_t i = 0; i < a->n; ++i) {
    OBall* b = &a->balls[i];
    if (b->velocity < 0.0 && b->pos_y < 0.2) b->velocity = -b->coef_restitution * b->velocity;
    else b->velocity -= 9.80665 * dt64;
    b->pos_y += b->velocity * dt64;
  }
}

static void o_static(Obj* o, double x, double y, double z, double r) { obj_sphere(o, v3(x, y, z), r); }

/* iterator scenes -- :176-225.  Returns 1 (frame produced), 0 (finished). */
EXPORT int oracle_animation_next(void* h, int32_t skip, double cam24[24], double* objs16, int64_t cap, int64_t* n_out, float* t_out) {
  OAnim* a = (OAnim*)h;
  if (!a->skipped) { while (a->t < a->t_min) o_anim_step(a); a->skipped = 1; }
  if (!(a->t < a->t_max)) return 0;
  if (cap < a->n + 4) return -1;
  double aspect = (double)a->ncols / (double)a->nrows;
  double r = sqrt(200.0);
  double lf[3] = { r * cos(a->look_from_angle), 2.0, r * sin(a->look_from_angle) };
  double la[3] = { 4, 1, 0 }, up[3] = { 0, 1, 0 };
  oracle_camera(lf, la, up, 20.0, aspect, 0.1, 10.0, 0.0, 0.0, cam24);
  Obj* objs = (Obj*)objs16;
  int64_t k = 0;
  o_static(&objs[k], 0, -1000, 0, 1000); mat_lambertian(&objs[k], v3(0.5, 0.5, 0.5)); k++;
  for (int64_t i = 0; i < a->n; ++i) {
    OBall* b = &a->balls[i];
    o_static(&objs[k], b->x, b->pos_y, b->z, b->radius);
    objs[k].mat = b->mat; objs[k].ax = b->ax; objs[k].ay = b->ay; objs[k].az = b->az; objs[k].fuzz = b->fuzz; objs[k].ri = b->ri;
    k++;
  }
  o_static(&objs[k], 0, 1, 0, 1.0); mat_dielectric(&objs[k], 1.5); k++;
  o_static(&objs[k], -4, 1, 0, 1.0); mat_lambertian(&objs[k], v3(0.4, 0.2, 0.1)); k++;
  o_static(&objs[k], 4, 1, 0, 1.0); mat_metal(&objs[k], v3(0.7, 0.6, 0.5), 0.0); k++;
  *n_out = k;
  if (t_out) *t_out = a->t;
  for (int32_t i = 0; i < skip; ++i) o_anim_step(a);
  return 1;
}

/* ------------------------------------------------------------------------- */
/* Video output stage -- io/rgb.nim, io/color_conversions.nim, io/h264.nim     */
/* ------------------------------------------------------------------------- */
/* PARITY UNPINNED: the reference holds no encoded frame, and color_conversions.nim's own self-test
 * needs yuv_rgb.c which is not in the tree (:329-331).  Restated from the sources; the tests add
 * known answers of the BT.601 limited-range matrix (white -> 235/128/128, black -> 16/128/128). */

/* io/rgb.nim:17-31 -> out[(i*ncols + j)*3 + c], i = 0 is the TOP scanline.  rgb.nim:29-31 reads
 * canvas[nrows - i, j] (row nrows for i = 0: out of bounds); the intended flip nrows-1-i is used. */
EXPORT void oracle_to_rgb_raw(const double* pixels, int32_t nrows, int32_t ncols, uint8_t* out) {
  for (int32_t i = nrows - 1; i >= 0; --i)
    for (int32_t j = 0; j < ncols; ++j)
      for (int ch = 0; ch < 3; ++ch) {
        double c = pixels[((size_t)(nrows - 1 - i) * ncols + j) * 3 + ch];
        double cl = c < 0.0 ? 0.0 : (c > 0.999 ? 0.999 : c);
        out[((size_t)i * ncols + j) * 3 + ch] = (uint8_t)(256 * cl);
      }
}

static uint8_t to_fixed_u8(double x, int precision) { return (uint8_t)(x * (double)(1 << precision) + 0.5); } /* :107-108 */

/* io/color_conversions.nim:180-252 with the BT601 coefficients of :110-120,176-178 */
EXPORT void oracle_rgb_to_ycbcr420(int32_t width, int32_t height, const uint8_t* rgb, uint8_t* Y, uint8_t* U, uint8_t* V) {
  const double kr_f = 0.299, kb_f = 0.114, ymin = 16.0, ymax = 235.0, cbcr = 240.0 - 16.0;
  const uint8_t kr = to_fixed_u8(kr_f, 8), kb = to_fixed_u8(kb_f, 8);
  const uint8_t kg = (uint8_t)(256 - kr - kb);
  const uint8_t fb = to_fixed_u8((cbcr / 255.0) / (2.0 * (1.0 - kb_f)), 8);
  const uint8_t fr = to_fixed_u8((cbcr / 255.0) / (2.0 * (1.0 - kr_f)), 8);
  const uint8_t y_scale = to_fixed_u8((ymax - ymin) / 255.0, 7);
  const uint8_t y_min = (uint8_t)ymin;
  const int32_t cstride = (width + 1) / 2;
  for (int32_t ii = 0; ii < height; ii += 2)
    for (int32_t jj = 0; jj < width; jj += 2) {
      int16_t tU = 0, tV = 0;
      for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj) {
          const uint8_t* p = rgb + ((size_t)(ii + di) * width + (jj + dj)) * 3;
          uint16_t tY = (uint16_t)(((uint16_t)kr * p[0] + (uint16_t)kg * p[1] + (uint16_t)kb * p[2]) >> 8);
          tU = (int16_t)(tU + ((int16_t)p[2] - (int16_t)tY));
          tV = (int16_t)(tV + ((int16_t)p[0] - (int16_t)tY));
          Y[(size_t)(ii + di) * width + (jj + dj)] = (uint8_t)((uint8_t)(((uint16_t)(tY * y_scale)) >> 7) + y_min);
        }
      U[(size_t)(ii >> 1) * cstride + (jj >> 1)] = (uint8_t)((int16_t)(((int16_t)((tU >> 2) * (int16_t)fb)) >> 8) + 128);
      V[(size_t)(ii >> 1) * cstride + (jj >> 1)] = (uint8_t)((int16_t)(((int16_t)((tV >> 2) * (int16_t)fr)) >> 8) + 128);
    }
}

/* io/h264.nim: BitBuffer (:51-81) */
typedef struct { int shift; uint32_t cache; uint8_t* buf; int cursor; } OBits;
static void ob_store(OBits* bb) { uint32_t c = bb->cache; bb->buf[bb->cursor] = (uint8_t)(c >> 24); bb->buf[bb->cursor + 1] = (uint8_t)(c >> 16); bb->buf[bb->cursor + 2] = (uint8_t)(c >> 8); bb->buf[bb->cursor + 3] = (uint8_t)c; }
static void ob_put(OBits* bb, int n, uint32_t val) {
  bb->shift -= n;
  if (bb->shift < 0) {
    bb->cache |= (val >> -bb->shift);
    ob_store(bb);
    bb->cursor += 4;
    bb->shift += 32;
    bb->cache = 0;
  }
  if (bb->shift < 32) bb->cache |= (val << bb->shift);
}
static void ob_golomb(OBits* bb, uint32_t val) {
  int size = 1;
  uint32_t t = val + 1;
  while ((t >>= 1) != 0) ++size;
  ob_put(bb, 2 * size - 1, val + 1);
}

/* H264Encoder.init (:169-176): SPS (:90-142) then the constant PPS (:37). Returns the byte count. */
EXPORT int oracle_h264_stream_header(int32_t width, int32_t height, uint8_t* out) {
  uint8_t sps[64];
  memset(sps, 0, sizeof sps);
  OBits bb = { 0, 0, sps, 0 };
  sps[0] = 0; sps[1] = 0; sps[2] = 0; sps[3] = 1;
  bb.shift = 32; bb.cursor = 4;
  ob_put(&bb, 1, 0); ob_put(&bb, 2, 3); ob_put(&bb, 5, 7);
  ob_put(&bb, 8, 66);
  ob_put(&bb, 1, 0); ob_put(&bb, 1, 0); ob_put(&bb, 1, 0); ob_put(&bb, 1, 0);
  ob_put(&bb, 4, 0);
  ob_put(&bb, 8, 10);
  ob_golomb(&bb, 0); ob_golomb(&bb, 0); ob_golomb(&bb, 0); ob_golomb(&bb, 0);
  ob_golomb(&bb, 0);
  ob_put(&bb, 1, 0);
  ob_golomb(&bb, (uint32_t)(((width + 15) >> 4) - 1));
  ob_golomb(&bb, (uint32_t)(((height + 15) >> 4) - 1));
  ob_put(&bb, 1, 1); ob_put(&bb, 1, 0); ob_put(&bb, 1, 0);
  ob_put(&bb, 1, 0); ob_put(&bb, 1, 1);
  ob_store(&bb); bb.cursor += 4;                 /* flush */
  int n = bb.cursor - bb.shift / 8;
  memcpy(out, sps, (size_t)n);
  static const uint8_t PPS[8] = { 0x00, 0x00, 0x00, 0x01, 0x68, 0xce, 0x38, 0x80 };
  memcpy(out + n, PPS, 8);
  return n + 8;
}

/* flushFrame (:249-259) + encodeMacroblock (:189-202). Returns the byte count. */
EXPORT int64_t oracle_h264_flush_frame(int32_t width, int32_t height, const uint8_t* Y, const uint8_t* Cb, const uint8_t* Cr, uint8_t* out) {
  static const uint8_t SliceHeader[9] = { 0x00, 0x00, 0x00, 0x01, 0x05, 0x88, 0x84, 0x21, 0xa0 };
  int64_t k = 0;
  memcpy(out, SliceHeader, 9); k = 9;
  for (int i = 0; i < height / 16; ++i)
    for (int j = 0; j < width / 16; ++j) {
      if (!(i == 0 && j == 0)) { out[k++] = 0x0d; out[k++] = 0x00; }
      for (int x = i * 16; x < (i + 1) * 16; ++x)
        for (int y = j * 16; y < (j + 1) * 16; ++y) out[k++] = Y[(size_t)x * width + y];
      for (int x = i * 8; x < (i + 1) * 8; ++x)
        for (int y = j * 8; y < (j + 1) * 8; ++y) out[k++] = Cb[(size_t)x * (width >> 1) + y];
      for (int x = i * 8; x < (i + 1) * 8; ++x)
        for (int y = j * 8; y < (j + 1) * 8; ++y) out[k++] = Cr[(size_t)x * (width >> 1) + y];
    }
  out[k++] = 0x80;
  return k;
}
