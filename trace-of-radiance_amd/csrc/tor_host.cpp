// tor_host.cpp -- host-side mirrors of the reference code on either side of the integrator:
// the camera constructor (physics/cameras.nim:24-45), the scene generator
// (scenes.nim:13-50) and the PPM quantiser (io/ppm.nim:14-27).  They produce the inputs the
// reference's main() feeds render() (trace_of_radiance.nim:26-57) and consume its output.
#include <cmath>
#include <cstring>

#include "../../include/tor_render.h"
#include "tor_device.hpp"

namespace {

using tor::V3;

TorVec3 to_abi(V3 v) { return TorVec3{v.x, v.y, v.z}; }
V3 from_abi(const TorVec3& v) { return V3{v.x, v.y, v.z}; }

V3 cross(V3 u, V3 v) {  // vec3s.nim:100-104
  return V3{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
}

TorMaterial lambertian(V3 albedo) {  // materials.nim:21-22
  TorMaterial m;
  std::memset(&m, 0, sizeof m);
  m.kind = TOR_LAMBERTIAN;
  m.u.lambertian.albedo = to_abi(albedo);
  return m;
}
TorMaterial metal(V3 albedo, double fuzz) {  // materials.nim:35-37
  TorMaterial m;
  std::memset(&m, 0, sizeof m);
  m.kind = TOR_METAL;
  m.u.metal.albedo = to_abi(albedo);
  m.u.metal.fuzz = (fuzz <= 1.0) ? fuzz : 1.0;  // min(fuzz, 1)
  return m;
}
TorMaterial dielectric(double ri) {  // materials.nim:52-53
  TorMaterial m;
  std::memset(&m, 0, sizeof m);
  m.kind = TOR_DIELECTRIC;
  m.u.dielectric.refraction_index = ri;
  return m;
}
TorHittableVariant sphere(V3 center, double radius, const TorMaterial& mat) {  // spheres.nim:20-23
  TorHittableVariant h;
  std::memset(&h, 0, sizeof h);
  h.kind = TOR_SPHERE;
  h.u.sphere.center = to_abi(center);
  h.u.sphere.radius = radius;
  h.u.sphere.material = mat;
  return h;
}
TorHittableVariant moving_sphere(V3 c0, double t0, V3 c1, double t1, double radius,
                                 const TorMaterial& mat) {  // moving_spheres.nim:22-32
  TorHittableVariant h;
  std::memset(&h, 0, sizeof h);
  h.kind = TOR_MOVING_SPHERE;
  h.u.moving_sphere.center0 = to_abi(c0);
  h.u.moving_sphere.center1 = to_abi(c1);
  h.u.moving_sphere.time0 = t0;
  h.u.moving_sphere.time1 = t1;
  h.u.moving_sphere.radius = radius;
  h.u.moving_sphere.material = mat;
  return h;
}

}  // namespace

extern "C" {

// physics/cameras.nim:24-45
int tor_camera_init(TorCamera* out, const TorVec3* look_from, const TorVec3* look_at, const TorVec3* view_up,
                    double vertical_fov_degrees, double aspect_ratio, double aperture,
                    double focus_distance, double shutter_open, double shutter_close) {
  if (!out || !look_from || !look_at || !view_up) return TOR_ERR_INVALID_ARGUMENT;
  const double theta = vertical_fov_degrees * (3.141592653589793 / 180.0);  // std/math degToRad
  const double h = std::tan(theta / 2.0);
  const double viewport_height = 2.0 * h;
  const double viewport_width = aspect_ratio * viewport_height;
  const V3 lf = from_abi(*look_from), la = from_abi(*look_at), vup = from_abi(*view_up);
  const V3 w = tor::unit_vector(lf - la);
  const V3 u = tor::unit_vector(cross(vup, w));
  const V3 v = cross(w, u);
  const V3 horizontal = u * (focus_distance * viewport_width);
  const V3 vertical = v * (focus_distance * viewport_height);
  const V3 llc = lf - tor::div_s(horizontal, 2.0) - tor::div_s(vertical, 2.0) - w * focus_distance;
  out->origin = to_abi(lf);
  out->lower_left_corner = to_abi(llc);
  out->horizontal = to_abi(horizontal);
  out->vertical = to_abi(vertical);
  out->u = to_abi(u);
  out->v = to_abi(v);
  out->w = to_abi(w);
  out->lens_radius = aperture / 2.0;
  out->shutter_open = shutter_open;
  out->shutter_close = shutter_close;
  return TOR_OK;
}

// scenes.nim:13-50 with `worldRNG.seed seed` (trace_of_radiance.nim:34-35)
int64_t tor_random_scene(uint64_t seed, TorHittableVariant* out, int64_t cap) {
  if (!out || cap < 0) return TOR_ERR_INVALID_ARGUMENT;
  tor::Rng rng{0, 0, 0, 0};
  tor::seed1(rng, seed);
  int64_t n = 0;
  auto push = [&](const TorHittableVariant& h) -> bool {
    if (n >= cap) return false;
    out[n++] = h;
    return true;
  };
  if (!push(sphere(V3{0, -1000, 0}, 1000, lambertian(V3{0.5, 0.5, 0.5})))) return TOR_ERR_INVALID_ARGUMENT;
  for (int a = -11; a < 11; ++a) {
    for (int b = -11; b < 11; ++b) {
      const double cx = (double)a + 0.9 * tor::uniform01(rng);
      const double cz = (double)b + 0.9 * tor::uniform01(rng);
      const V3 center{cx, 0.2, cz};
      if (std::sqrt(tor::len2(center - V3{4, 0.2, 0})) > 0.9) {
        const double choose_mat = tor::uniform01(rng);
        bool ok;
        if (choose_mat < 0.8) {
          const double a0 = tor::uniform01(rng), a1 = tor::uniform01(rng), a2 = tor::uniform01(rng);
          const double b0 = tor::uniform01(rng), b1 = tor::uniform01(rng), b2 = tor::uniform01(rng);
          const V3 albedo{a0 * b0, a1 * b1, a2 * b2};
          const V3 center2 = center + V3{0, tor::uniform_max(rng, 0.5), 0};
          ok = push(moving_sphere(center, 0.0, center2, 1.0, 0.2, lambertian(albedo)));
        } else if (choose_mat < 0.95) {
          const double r0 = tor::uniform_range(rng, 0.5, 1), r1 = tor::uniform_range(rng, 0.5, 1),
                       r2 = tor::uniform_range(rng, 0.5, 1);
          const double fuzz = tor::uniform_max(rng, 0.5);
          ok = push(sphere(center, 0.2, metal(V3{r0, r1, r2}, fuzz)));
        } else {
          ok = push(sphere(center, 0.2, dielectric(1.5)));
        }
        if (!ok) return TOR_ERR_INVALID_ARGUMENT;
      }
    }
  }
  if (!push(sphere(V3{0, 1, 0}, 1.0, dielectric(1.5)))) return TOR_ERR_INVALID_ARGUMENT;
  if (!push(sphere(V3{-4, 1, 0}, 1.0, lambertian(V3{0.4, 0.2, 0.1})))) return TOR_ERR_INVALID_ARGUMENT;
  if (!push(sphere(V3{4, 1, 0}, 1.0, metal(V3{0.7, 0.6, 0.5}, 0.0)))) return TOR_ERR_INVALID_ARGUMENT;
  return n;
}

// io/ppm.nim:14-27: rows from nrows-1 down to 0, int(256 * clamp(c, 0.0, 0.999))
int tor_canvas_to_rgb8(const TorCanvas* canvas, uint8_t* out) {
  if (!canvas || !canvas->pixels || !out) return TOR_ERR_INVALID_ARGUMENT;
  const int32_t nrows = canvas->nrows, ncols = canvas->ncols;
  size_t k = 0;
  for (int32_t i = nrows - 1; i >= 0; --i)
    for (int32_t j = 0; j < ncols; ++j) {
      const TorVec3& px = canvas->pixels[(size_t)i * ncols + j];
      const double ch[3] = {px.x, px.y, px.z};
      for (int c = 0; c < 3; ++c) {
        const double cl = ch[c] < 0.0 ? 0.0 : (ch[c] > 0.999 ? 0.999 : ch[c]);  // safe_math.nim:10-14
        out[k++] = (uint8_t)(int)(256 * cl);
      }
    }
  return TOR_OK;
}

}  // extern "C"
