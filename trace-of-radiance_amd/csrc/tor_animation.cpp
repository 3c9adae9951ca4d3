// tor_animation.cpp -- host-side mirror of the reference's animated scene
// (trace_of_radiance/scenes_animated.nim): the 40x40 bouncing-spheres generator
// (random_moving_spheres, :90-154), the physics/camera step (:156-174) and the `scenes`
// iterator (:176-225) that feeds one (Camera, Scene) per frame to render()
// (trace_of_radiance_animation.nim:84-97,173-196).  Host code: it produces the inputs of the
// integrator; the per-frame scenes are all-static spheres, the camera shutter is [0,0].
//
// One point of the reference is not compilable at HEAD: `rng.random(float32)`
// (scenes_animated.nim:122,136,148) has no overload in sampling.nim:18-25.  This mirror (and the
// oracle, identically) reads it as float32(rng.uniform(float64)) followed by float32 arithmetic
// (`10.0 + (4 * u - 2.0)` with float32 operands, then widened to the float64 Velocity).
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/tor_render.h"
#include "tor_device.hpp"

namespace {

struct BouncingSphere {   // scenes_animated.nim:37-49
  double velocity, pos_y; // mutable
  double coef_restitution;
  double x, z, radius;
  TorMaterial material;
};

TorMaterial mat_lambertian(tor::V3 a) {
  TorMaterial m; std::memset(&m, 0, sizeof m);
  m.kind = TOR_LAMBERTIAN; m.u.lambertian.albedo = TorVec3{a.x, a.y, a.z};
  return m;
}
TorMaterial mat_metal(tor::V3 a, double fuzz) {
  TorMaterial m; std::memset(&m, 0, sizeof m);
  m.kind = TOR_METAL; m.u.metal.albedo = TorVec3{a.x, a.y, a.z}; m.u.metal.fuzz = (fuzz <= 1.0) ? fuzz : 1.0;
  return m;
}
TorMaterial mat_dielectric(double ri) {
  TorMaterial m; std::memset(&m, 0, sizeof m);
  m.kind = TOR_DIELECTRIC; m.u.dielectric.refraction_index = ri;
  return m;
}
TorHittableVariant static_sphere(double x, double y, double z, double radius, const TorMaterial& mat) {
  TorHittableVariant h; std::memset(&h, 0, sizeof h);
  h.kind = TOR_SPHERE;
  h.u.sphere.center = TorVec3{x, y, z};
  h.u.sphere.radius = radius;
  h.u.sphere.material = mat;
  return h;
}

constexpr double kSmallRadius = 0.2;  // scenes_animated.nim:66
constexpr double kG = 9.80665;        // :67

}  // namespace

struct TorAnimation {  // scenes_animated.nim:51-63
  int32_t nrows, ncols;
  float dt, t_min, t_max;
  float t;
  double look_from_angle;
  std::vector<BouncingSphere> spheres;
  bool skipped = false;
};

namespace {

double velocity_draw(tor::Rng& rng) {  // `Velocity(10.0 + (4 * rng.random(float32) - 2.0))`
  const float u = (float)tor::uniform01(rng);
  const float v = 10.0f + (4.0f * u - 2.0f);
  return (double)v;
}

void step(TorAnimation& a) {  // :156-174
  a.look_from_angle -= 2.0 * 3.141592653589793 / 1200.0;  // stepCamera
  a.t += a.dt;                                            // float32 accumulation
  const double dt64 = (double)a.dt;
  for (auto& s : a.spheres) {
    if (s.velocity < 0.0 && s.pos_y < kSmallRadius) {
      s.velocity = -s.coef_restitution * s.velocity;  // bounce
    } else {
      s.velocity -= kG * dt64;
    }
    s.pos_y += s.velocity * dt64;
  }
}

}  // namespace

extern "C" {

// random_moving_spheres -- scenes_animated.nim:90-154 (rng seeded like
// trace_of_radiance_animation.nim:62-63: worldRNG.seed 0xFACADE)
int tor_animation_create(uint64_t seed, int32_t height, int32_t width, float dt, float t_min, float t_max,
                         TorAnimation** out) {
  if (!out || height < 2 || width < 2) return TOR_ERR_INVALID_ARGUMENT;
  TorAnimation* a = new (std::nothrow) TorAnimation();
  if (!a) return TOR_ERR_OUT_OF_MEMORY;
  a->nrows = height; a->ncols = width;
  a->dt = dt; a->t_min = t_min; a->t_max = t_max;
  a->t = 0.0f;
  a->look_from_angle = 2 * 3.141592653589793;
  tor::Rng rng{0, 0, 0, 0};
  tor::seed1(rng, seed);
  for (int ia = -20; ia < 20; ++ia) {
    for (int ib = -20; ib < 20; ++ib) {
      const double cx = (double)ia + 0.9 * tor::uniform01(rng);
      const double cz = (double)ib + 0.9 * tor::uniform01(rng);
      const tor::V3 center{cx, kSmallRadius, cz};
      if (std::sqrt(tor::len2(center - tor::V3{4, kSmallRadius, 0})) > 0.9) {
        const double choose_mat = tor::uniform01(rng);
        BouncingSphere s;
        std::memset(&s, 0, sizeof s);
        if (choose_mat < 0.65) {
          const double a0 = tor::uniform01(rng), a1 = tor::uniform01(rng), a2 = tor::uniform01(rng);
          const double b0 = tor::uniform01(rng), b1 = tor::uniform01(rng), b2 = tor::uniform01(rng);
          s.coef_restitution = 0.6;
          s.velocity = velocity_draw(rng);
          s.material = mat_lambertian(tor::V3{a0 * b0, a1 * b1, a2 * b2});
        } else if (choose_mat < 0.95) {
          const double r0 = tor::uniform_range(rng, 0.5, 1), r1 = tor::uniform_range(rng, 0.5, 1),
                       r2 = tor::uniform_range(rng, 0.5, 1);
          const double fuzz = tor::uniform_max(rng, 0.5);
          s.coef_restitution = 0.5;
          s.velocity = velocity_draw(rng);
          s.material = mat_metal(tor::V3{r0, r1, r2}, fuzz);
        } else {
          s.coef_restitution = 0.5;
          s.velocity = velocity_draw(rng);
          s.material = mat_dielectric(1.5);
        }
        s.x = center.x; s.pos_y = center.y; s.z = center.z; s.radius = kSmallRadius;
        a->spheres.push_back(s);
      }
    }
  }
  *out = a;
  return TOR_OK;
}

void tor_animation_destroy(TorAnimation* a) { delete a; }

int64_t tor_animation_object_count(const TorAnimation* a) { return a ? (int64_t)a->spheres.size() + 4 : 0; }

// One turn of `iterator scenes(anim, skip)` -- scenes_animated.nim:176-225.  Returns 1 and fills
// (cam, objects[0..*n_out)) when a frame is due, 0 when anim.t has reached t_max, < 0 on error.
int tor_animation_next(TorAnimation* a, int32_t skip, TorCamera* cam, TorHittableVariant* objects, int64_t cap,
                       int64_t* n_out, float* t_out) {
  if (!a || !cam || !objects || !n_out) return TOR_ERR_INVALID_ARGUMENT;
  if (!a->skipped) {
    while (a->t < a->t_min) step(*a);
    a->skipped = true;
  }
  if (!(a->t < a->t_max)) return 0;
  const int64_t n = (int64_t)a->spheres.size() + 4;
  if (cap < n) return TOR_ERR_INVALID_ARGUMENT;
  const double aspect_ratio = (double)a->ncols / (double)a->nrows;  // int / int -> float64 in Nim
  const double r = std::sqrt(200.0);
  const TorVec3 look_from{r * std::cos(a->look_from_angle), 2.0, r * std::sin(a->look_from_angle)};
  const TorVec3 look_at{4, 1, 0}, vup{0, 1, 0};
  int rc = tor_camera_init(cam, &look_from, &look_at, &vup, 20.0, aspect_ratio, 0.1, 10.0, 0.0, 0.0);
  if (rc != TOR_OK) return rc;
  int64_t k = 0;
  objects[k++] = static_sphere(0, -1000, 0, 1000, mat_lambertian(tor::V3{0.5, 0.5, 0.5}));
  for (const auto& s : a->spheres) objects[k++] = static_sphere(s.x, s.pos_y, s.z, s.radius, s.material);
  objects[k++] = static_sphere(0, 1, 0, 1.0, mat_dielectric(1.5));
  objects[k++] = static_sphere(-4, 1, 0, 1.0, mat_lambertian(tor::V3{0.4, 0.2, 0.1}));
  objects[k++] = static_sphere(4, 1, 0, 1.0, mat_metal(tor::V3{0.7, 0.6, 0.5}, 0.0));
  *n_out = k;
  if (t_out) *t_out = a->t;
  for (int32_t i = 0; i < skip; ++i) step(*a);
  return 1;
}

}  // extern "C"
