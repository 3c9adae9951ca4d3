// tor_device.hpp -- per-lane building blocks of the gfx950 integrator: Vec3 algebra, the
// xoshiro256+ stream, samplers, the thin-lens camera ray and the three material scatters.
//
// Every function restates the arithmetic of the reference operation for operation (the
// file:line it follows is cited next to it; paths relative to the reference root) because
// the path is chaotic: one different rounding can flip a later hit/miss.  float64 only,
// no FMA contraction (build with -ffp-contract=off), divisions and square roots are the
// IEEE correctly rounded ones.
#pragma once

#include <cstdint>

#include "tor_math.hpp"

namespace tor {

// floor(n / d) for a 32-bit n and a divisor d >= 1 whose float64 reciprocal inv = 1.0 / d (correctly rounded, from the host) is at
// hand: (n + 0.5) * inv, truncated -- 4 instructions instead of the ~22 of a 32-bit division by a run-time divisor.  Exact: with
// q = floor(n / d) the quotient (n + 0.5) / d lies in [q + 0.5 / d, q + 1 - 0.5 / d]; n + 0.5 is exact in float64, inv and the
// product carry a relative error below 2^-52 together, i.e. an absolute one below (2^32 / d + 1) 2^-52 < 0.5 / d for every d < 2^32.
TOR_HD unsigned udiv_by(unsigned n, double inv) { return (unsigned)(((double)n + 0.5) * inv); }

struct V3 {
  double x, y, z;
};

// primitives/vec3s.nim:63-107, point3s.nim, colors.nim
TOR_HD V3 v3(double x, double y, double z) { return V3{x, y, z}; }
TOR_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
TOR_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
TOR_HD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
TOR_HD V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }  // vec3s.nim:86-91
TOR_HD V3 div_s(V3 a, double s) { return a * (1.0 / s); }                      // vec3s.nim:93-94
TOR_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }    // vec3s.nim:96-98
TOR_HD double len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }         // vec3s.nim:19-20
TOR_HD V3 unit_vector(V3 a) { return div_s(a, __builtin_sqrt(len2(a))); }      // vec3s.nim:106-107
TOR_HD V3 mul_att(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }  // colors.nim:48-66

// ---------------------------------------------------------------------------------------
// RNG -- support/rng.nim
// ---------------------------------------------------------------------------------------
struct Rng {
  uint64_t s0, s1, s2, s3;
};

// rng.nim:31-36.  Both mixing multipliers are 0xbf58476d1ce4e5b9 in the reference.
TOR_HD uint64_t splitmix64(uint64_t& state) {
  state += 0x9e3779b97f4a7c15ULL;
  uint64_t r = state;
  r = (r ^ (r >> 30)) * 0xbf58476d1ce4e5b9ULL;
  r = (r ^ (r >> 27)) * 0xbf58476d1ce4e5b9ULL;
  return r ^ (r >> 31);
}
// rng.nim:38-44
TOR_HD void seed1(Rng& g, uint64_t x) {
  uint64_t sm = x;
  g.s0 = splitmix64(sm);
  g.s1 = splitmix64(sm);
  g.s2 = splitmix64(sm);
  g.s3 = splitmix64(sm);
}
// rng.nim:21-29,46-53 : seed(row, col), row in the high word (render.nim:60)
TOR_HD void seed2(Rng& g, uint64_t x, uint64_t y) { seed1(g, (x << 32) ^ y); }
// Counter-based per-sample stream (TOR_SEED_SAMPLE; not in the reference):
//   sm = pair(row,col); h = splitMix64(sm); seed(h xor sample)
TOR_HD void seed3(Rng& g, uint64_t row, uint64_t col, uint64_t sample) {
  uint64_t sm = (row << 32) ^ col;
  uint64_t h = splitmix64(sm);
  seed1(g, h ^ sample);
}
// rng.nim:58-74 xoshiro256+
TOR_HD uint64_t next(Rng& g) {
  uint64_t res = g.s0 + g.s3;
  uint64_t t = g.s1 << 17;
  g.s2 ^= g.s0;
  g.s3 ^= g.s1;
  g.s1 ^= g.s2;
  g.s0 ^= g.s3;
  g.s2 ^= t;
  g.s3 = (g.s3 << 45) | (g.s3 >> 19);
  return res;
}
// The samplers below take the generator as a template parameter G: `Rng` (a lane's or the host's own xoshiro256+ state), or the
// chain servers' block generator (tor_kernels.hip: SrvRng -- the same stream, 64 outputs at a time, one lane per output);
// all they need is `uint64_t next(G&)`.
// rng.nim:129-133, from one 64-bit output
TOR_HD double uniform01_of(uint64_t out) { return bits_to_double((out >> 12) | 0x3ff0000000000000ULL) - 1.0; }
template <class G>
TOR_HD double uniform01(G& g) { return uniform01_of(next(g)); }
// rng.nim:135-143
TOR_HD double uniform_max_of(uint64_t out, double max_excl) { return uniform01_of(out) * max_excl; }
template <class G>
TOR_HD double uniform_max(G& g, double max_excl) { return uniform_max_of(next(g), max_excl); }
// rng.nim:116-127 (Nim's max(x, y) is `if y <= x: x else: y`)
TOR_HD double uniform_range_of(uint64_t out, double lo, double hi) {
  double d = uniform01_of(out);
  double v = d * (hi - lo) + lo;
  return (v <= lo) ? lo : v;
}
template <class G>
TOR_HD double uniform_range(G& g, double lo, double hi) { return uniform_range_of(next(g), lo, hi); }
// uniform_range(g, -1.0, 1.0) without the comparison: d is in [0, 1), so v = d * 2.0 + -1.0 >= -1.0 = lo, and `v <= lo` can only
// hold with v == lo, where the reference's max returns lo -- the same bits.  (Three of these per trip of the rejection loops, which
// a wave runs until its unluckiest lane accepts.)
TOR_HD double uniform_pm1_of(uint64_t out) { return uniform01_of(out) * (1.0 - -1.0) + -1.0; }
template <class G>
TOR_HD double uniform_pm1(G& g) { return uniform_pm1_of(next(g)); }

// ---------------------------------------------------------------------------------------
// Samplers -- sampling.nim
// ---------------------------------------------------------------------------------------
// sampling.nim:64-68
template <class G>
TOR_HD V3 random_in_unit_disk(G& g) {
  for (;;) {
    double x = uniform_pm1(g);
    double y = uniform_pm1(g);
    if (x * x + y * y + 0.0 * 0.0 < 1.0) return V3{x, y, 0.0};
  }
}
// sampling.nim:45-49
template <class G>
TOR_HD V3 random_in_unit_sphere(G& g) {
  for (;;) {
    double x = uniform_pm1(g);
    double y = uniform_pm1(g);
    double z = uniform_pm1(g);
    V3 p{x, y, z};
    if (len2(p) < 1.0) return p;
  }
}
// sampling.nim:51-55, from the two outputs it draws
TOR_HD V3 random_unit_vector_of(uint64_t out_a, uint64_t out_z) {
  double a = uniform_max_of(out_a, 2.0 * 3.141592653589793);
  double z = uniform_pm1_of(out_z);
  double r = __builtin_sqrt(1.0 - z * z);
  double s, c;
  sincos_2pi(a, s, c);
  return V3{r * c, r * s, z};
}
TOR_HD V3 random_unit_vector(Rng& g) {
  double a = uniform_max(g, 2.0 * 3.141592653589793);
  double z = uniform_pm1(g);
  double r = __builtin_sqrt(1.0 - z * z);
  double s, c;
  sincos_2pi(a, s, c);
  return V3{r * c, r * s, z};
}

// ---------------------------------------------------------------------------------------
// Rays and camera -- primitives/rays.nim, physics/cameras.nim
// ---------------------------------------------------------------------------------------
struct Ray {
  V3 origin, direction;
  double time;
};

struct Camera {  // cameras.nim:15-22, same field order as TorCamera
  V3 origin, lower_left_corner, horizontal, vertical, u, v, w;
  double lens_radius, shutter_open, shutter_close;
};

// cameras.nim:47-57
template <class G>
TOR_HD Ray camera_ray(const Camera& c, double s, double t, G& g) {
  V3 rd = random_in_unit_disk(g) * c.lens_radius;
  V3 offset = c.u * rd.x + c.v * rd.y;
  Ray r;
  r.origin = c.origin + offset;
  r.direction = c.lower_left_corner + c.horizontal * s + c.vertical * t - c.origin - offset;
  r.time = uniform_range(g, c.shutter_open, c.shutter_close);
  return r;
}

// rays.nim:27-28
TOR_HD V3 reflect(V3 u, V3 n) { return u - n * (2.0 * dot(u, n)); }
// rays.nim:30-37
TOR_HD V3 refract(V3 uv, V3 n, double eta) {
  double cos_theta = dot(-uv, n);
  V3 par = (uv + n * cos_theta) * eta;
  V3 perp = n * (-__builtin_sqrt(1.0 - len2(par)));
  return par + perp;
}
// materials.nim:55-60
TOR_HD double schlick(double cosine, double ri) {
  double r0 = (1.0 - ri) / (1.0 + ri);
  r0 *= r0;
  return r0 + (1.0 - r0) * pow5(1.0 - cosine);
}

// schlick() with r0 = ((1 - eta)/(1 + eta))^2 taken from the object's record (the host evaluates exactly these
// operations once per object and side: tor_scene.cpp fill_material)
TOR_HD double schlick_r0(double cosine, double r0) { return r0 + (1.0 - r0) * pow5(1.0 - cosine); }

enum : int { kLambertian = 0, kMetal = 1, kDielectric = 2 };  // core.nim:25-27

// render.nim:41-45: the sky gradient on a miss (note `0.5*y + 1.0`, as the reference has it)
TOR_HD V3 sky_unit(V3 ud, V3 att) {  // ud = unit_vector(direction), computed by the caller
  double t = 0.5 * ud.y + 1.0;
  V3 res = v3(1.0, 1.0, 1.0) * (1.0 - t) + v3(0.5, 0.7, 1.0) * t;
  return mul_att(res, att);
}
TOR_HD V3 sky(V3 direction, V3 att) {
  V3 ud = unit_vector(direction);
  double t = 0.5 * ud.y + 1.0;
  V3 res = v3(1.0, 1.0, 1.0) * (1.0 - t) + v3(0.5, 0.7, 1.0) * t;
  return mul_att(res, att);
}

// Round to the nearest multiple of 2^-36 (ties to even); exact for |x| < 2^15.  Sums of
// such values below 2^17 are exact in float64, hence independent of summation order.
TOR_HD double quantize36(double x) {
  double t = x + 98304.0;  // 1.5 * 2^16 : ulp(t) = 2^-36
  return t - 98304.0;
}

}  // namespace tor
