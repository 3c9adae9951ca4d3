// tor_scene.cpp -- see tor_scene.hpp.
#include "tor_scene.hpp"
#include "tor_knobs.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "tor_filter32.hpp"
#include "tor_kernels.hpp"
#include "tor_screen.hpp"

namespace tor {

namespace {

inline double i64_as_double(int64_t v) {
  double d;
  std::memcpy(&d, &v, 8);
  return d;
}

size_t padded(size_t c) { return (c + kPad - 1) / kPad * kPad; }

// cold record (16 float64), see tor_kernels.hpp
bool fill_material(double* c, const TorMaterial& m, int moving) {
  int64_t flags = (moving ? 1 : 0) | ((int64_t)m.kind << 8);
  c[13] = i64_as_double(flags);
  switch (m.kind) {
    case TOR_LAMBERTIAN:
      c[9] = m.u.lambertian.albedo.x; c[10] = m.u.lambertian.albedo.y; c[11] = m.u.lambertian.albedo.z;
      return true;
    case TOR_METAL:
      c[9] = m.u.metal.albedo.x; c[10] = m.u.metal.albedo.y; c[11] = m.u.metal.albedo.z;
      c[12] = m.u.metal.fuzz;
      return true;
    case TOR_DIELECTRIC: {
      const double ri = m.u.dielectric.refraction_index;
      c[12] = ri;
      // per-object constants of Dielectric.scatter, evaluated here with the reference's own operations
      // (materials.nim:55-60,66: `1.0 / ri`, r0 = (1 - eta)/(1 + eta), r0 * r0); the albedo slots are free
      // (materials.nim:64: attenuation = (1, 1, 1))
      const double inv = 1.0 / ri;
      double r0f = (1.0 - inv) / (1.0 + inv);
      double r0b = (1.0 - ri) / (1.0 + ri);
      c[9] = inv;
      c[10] = r0f * r0f;   // front face: eta = 1/ri
      c[11] = r0b * r0b;   // back face:  eta = ri
      return true;
    }
    default:
      return false;
  }
}

bool fill_cold(double* c, const TorHittableVariant& h, int64_t orig) {
  if (h.kind == TOR_SPHERE) {
    const TorSphere& s = h.u.sphere;
    c[0] = s.center.x; c[1] = s.center.y; c[2] = s.center.z;
    c[6] = 1.0 / s.radius;  // vec3s.nim:93-94: `/ radius` is `* (1.0 / radius)`
    c[14] = i64_as_double(orig);
    c[15] = s.radius * s.radius;  // spheres.nim:32 `self.radius*self.radius`
    return fill_material(c, s.material, 0);
  }
  const TorMovingSphere& s = h.u.moving_sphere;
  c[0] = s.center0.x; c[1] = s.center0.y; c[2] = s.center0.z;
  c[3] = s.center1.x - s.center0.x; c[4] = s.center1.y - s.center0.y; c[5] = s.center1.z - s.center0.z;  // moving_spheres.nim:43
  c[6] = 1.0 / s.radius;
  c[7] = s.time0; c[8] = s.time1 - s.time0;  // moving_spheres.nim:42
  c[14] = i64_as_double(orig);
  c[15] = s.radius * s.radius;
  return fill_material(c, s.material, 1);
}

}  // namespace

// AoS -> SoA.  Objects are partitioned into one static segment and one segment per distinct
// (time0, time1) pair; closest-hit is order independent (hittables_lists.nim:48-55; ties are broken
// by the original index carried in the cold record), so the reordering is exact.  With `f32`, the
// qualifying objects get a second set of segments (kinds 5/6/7) for the float32 pre-filter.
namespace {

struct TimeGroup {
  uint64_t k0, k1;
  std::vector<int64_t> ids;
};

void add_to_group(std::vector<TimeGroup>& groups, const TorMovingSphere& s, int64_t i) {
  uint64_t k0, k1;
  std::memcpy(&k0, &s.time0, 8);
  std::memcpy(&k1, &s.time1, 8);
  for (auto& g : groups)
    if (g.k0 == k0 && g.k1 == k1) {
      g.ids.push_back(i);
      return;
    }
  groups.push_back({k0, k1, {i}});
}

bool group_moves_along_y_only(const TorHittableVariant* objs, const TimeGroup& g) {
  // every member has center1.x == center0.x and center1.z == center0.z: c0 + f*(c1-c0) leaves x and z untouched, exactly
  for (int64_t idx : g.ids) {
    const TorMovingSphere& s = objs[idx].u.moving_sphere;
    if (!(s.center1.x - s.center0.x == 0.0 && s.center1.z - s.center0.z == 0.0)) return false;
  }
  return true;
}

double norm3(double x, double y, double z) { return std::sqrt(x * x + y * y + z * z); }

}  // namespace

F32Options f32_options_for(const TorHittableVariant* objs, int64_t n) {
  F32Options o;
  std::vector<double> ax[3];
  for (int64_t i = 0; i < n; ++i) {
    const TorVec3* c = objs[i].kind == TOR_SPHERE ? &objs[i].u.sphere.center
                       : (objs[i].kind == TOR_MOVING_SPHERE ? &objs[i].u.moving_sphere.center0 : nullptr);
    if (!c || !std::isfinite(c->x) || !std::isfinite(c->y) || !std::isfinite(c->z)) continue;
    ax[0].push_back(c->x); ax[1].push_back(c->y); ax[2].push_back(c->z);
  }
  if (ax[0].empty()) return o;
  const size_t mid = ax[0].size() / 2;
  for (int a = 0; a < 3; ++a) {
    std::vector<double> t = ax[a];
    std::nth_element(t.begin(), t.begin() + mid, t.end());
    o.origin[a] = t[mid];
  }
  std::vector<double> dist(ax[0].size());
  for (size_t k = 0; k < dist.size(); ++k)
    dist[k] = norm3(ax[0][k] - o.origin[0], ax[1][k] - o.origin[1], ax[2][k] - o.origin[2]);
  std::nth_element(dist.begin(), dist.begin() + mid, dist.end());
  o.far_limit = 8.0 * dist[mid];
  if (!(o.far_limit > 0.0) || !std::isfinite(o.far_limit)) o.far_limit = 0.0;
  return o;
}

bool build_layout(const TorHittableVariant* objs, const std::vector<int64_t>& ids, HostLayout& out, std::string& err,
                  const F32Options* f32) {
  std::vector<int64_t> statics, statics32;
  std::vector<TimeGroup> groups, groups32;
  auto dist0 = [&](const TorVec3& c) {
    return norm3(c.x - f32->origin[0], c.y - f32->origin[1], c.z - f32->origin[2]);
  };
  for (int64_t i : ids) {
    const TorHittableVariant& h = objs[i];
    if (h.kind == TOR_SPHERE) {
      const TorSphere& s = h.u.sphere;
      const bool q = f32 && f32_eligible(dist0(s.center), 0.0, s.radius * s.radius) && dist0(s.center) <= f32->far_limit;
      (q ? statics32 : statics).push_back(i);
    } else if (h.kind == TOR_MOVING_SPHERE) {
      const TorMovingSphere& s = h.u.moving_sphere;
      const double dt = s.time1 - s.time0;
      const bool q = f32 && std::isfinite(s.time0) && std::isfinite(dt) && dt != 0.0 &&
                     f32_eligible(dist0(s.center0), norm3(s.center1.x - s.center0.x, s.center1.y - s.center0.y, s.center1.z - s.center0.z),
                                  s.radius * s.radius) &&
                     dist0(s.center0) <= f32->far_limit;
      add_to_group(q ? groups32 : groups, s, i);
    } else {
      err = "unknown HittableVariant kind";
      return false;
    }
  }
  // ---- the float64 segments.  Statics and every (time0, time1) group of movers are cut once more by c0.y: members that share
  // it bit for bit (spheres resting on a plane) form segments of their own when there are at least kPad of them -- the second
  // form of the strict loop's FMA screen hoists their y terms out of the per-object work (tor_screen.hpp: kinds 11, 12).  The
  // order of the objects is free (closest hit is order independent, ties go by the original index in the cold record).
  // Round 5: every segment of finite parameters carries a plane table (stage one of the screen, tor_screen.hpp), whose band is as
  // wide as the segment's LARGEST sphere -- so spheres much larger than the segment's typical member (r^2 > 16 x the median: the
  // ground and the three big spheres of random_scene) are set apart first.
  struct Seg64 { int kind; std::vector<int64_t> ids; double t0, dt; int xkind; double y; };
  std::vector<Seg64> segs64;
  auto y_of = [&](int64_t i) { return objs[i].kind == TOR_SPHERE ? objs[i].u.sphere.center.y : objs[i].u.moving_sphere.center0.y; };
  auto r2_of = [&](int64_t i) {
    const double r = objs[i].kind == TOR_SPHERE ? objs[i].u.sphere.radius : objs[i].u.moving_sphere.radius;
    return r * r;
  };
  auto cut_by_y = [&](const std::vector<int64_t>& all_members, int kind, double t0, double dt, int xkind_rest, int xkind_uniform) {
    std::vector<int64_t> members, big;
    if (all_members.size() >= 2 * (size_t)kPad) {
      std::vector<double> r2s;
      for (int64_t i : all_members) r2s.push_back(r2_of(i));
      std::nth_element(r2s.begin(), r2s.begin() + r2s.size() / 2, r2s.end());
      const double limit = 16.0 * r2s[r2s.size() / 2];
      for (int64_t i : all_members) (r2_of(i) > limit ? big : members).push_back(i);   // (a NaN radius stays with the members: fmax skips it below)
    } else {
      members = all_members;
    }
    const size_t big_at = segs64.size();   // (the leftovers below join the big spheres' segment when both are small: see there)
    if (!big.empty()) segs64.push_back({kind, big, t0, dt, xkind_rest, 0.0});
    if (members.empty()) return;
    std::vector<std::pair<uint64_t, std::vector<int64_t>>> by_y;  // in order of first appearance
    for (int64_t i : members) {
      const double y = y_of(i);
      uint64_t b;
      std::memcpy(&b, &y, 8);
      bool found = false;
      for (auto& g : by_y)
        if (g.first == b) { g.second.push_back(i); found = true; break; }
      if (!found) by_y.push_back({b, {i}});
      if (by_y.size() > 64) break;  // (a scene without common heights: do not search for them object by object)
    }
    std::vector<int64_t> rest;
    std::vector<Seg64> uniform;
    if (by_y.size() <= 64 && xkind_uniform != 0) {
      for (auto& g : by_y)
        if (g.second.size() >= (size_t)kPad && std::isfinite(y_of(g.second[0]))) uniform.push_back({kind, g.second, t0, dt, xkind_uniform, y_of(g.second[0])});
      for (int64_t i : members) {
        const double y = y_of(i);
        bool in_uniform = false;
        for (const Seg64& u : uniform) {
          uint64_t a, b;
          std::memcpy(&a, &y, 8); std::memcpy(&b, &u.y, 8);
          if (a == b) { in_uniform = true; break; }
        }
        if (!in_uniform) rest.push_back(i);
      }
    } else {
      rest = members;
    }
    // a handful of leftovers and a handful of big spheres: ONE segment (fewer than 48 objects never run stage one -- the gate in
    // xhdr -- so the band's width, the reason for setting the big ones apart, does not matter; a segment's set-up does)
    if (!rest.empty() && !big.empty() && rest.size() + big.size() < 48) {
      segs64[big_at].ids.insert(segs64[big_at].ids.end(), rest.begin(), rest.end());
      rest.clear();
    }
    if (!rest.empty()) segs64.push_back({kind, rest, t0, dt, xkind_rest, 0.0});
    for (Seg64& u : uniform) segs64.push_back(std::move(u));
  };
  if (!statics.empty()) cut_by_y(statics, 0, 0.0, 1.0, 10, 11);
  for (size_t gi = 0; gi < groups.size(); ++gi) {
    const TorMovingSphere& first = objs[groups[gi].ids[0]].u.moving_sphere;
    const double t0 = first.time0, dt = first.time1 - first.time0;
    // xkind of a segment of movers: 12 along y with a common c0.y (second form), 14 along y at any height, 13 in general position
    // (13 / 14: stage one in front of the FIRST form); a group whose time fraction can never be finite keeps the first form
    // alone (xkind 0): the reference cannot hit such a centre, no stage is worth building for it.  A time group is cut into its
    // members that move along y only (c1.x == c0.x and c1.z == c0.z: c0 + f * 0 == c0 exactly) and the others (round 5: one
    // mover in general position no longer takes the whole group's y-only members off the cheaper records).
    const bool sane = std::isfinite(t0) && std::isfinite(dt) && dt != 0.0;
    TimeGroup gy{groups[gi].k0, groups[gi].k1, {}}, gg{groups[gi].k0, groups[gi].k1, {}};
    for (int64_t idx : groups[gi].ids) {
      const TorMovingSphere& s = objs[idx].u.moving_sphere;
      ((s.center1.x - s.center0.x == 0.0 && s.center1.z - s.center0.z == 0.0) ? gy : gg).ids.push_back(idx);
    }
    if (!gy.ids.empty()) cut_by_y(gy.ids, 1, t0, dt, sane ? 14 : 0, sane ? 12 : 0);
    if (!gg.ids.empty()) cut_by_y(gg.ids, 2, t0, dt, sane ? 13 : 0, 0);
  }
  // Statics that rest at the common height of a segment of movers along y JOIN that segment as movers that do not move
  // (dc = 0: c0 + f * 0 == c0 exactly for every finite f; the cold record keeps them static, so the exact test never reads f).
  // One segment instead of two: the per-segment set-up of the screened loop (margins, plane and second-form constants, a vote)
  // is worth ~80 objects of stage one -- random_scene's 86 resting statics and 395 resting movers become one segment of 481.
  // Only for radii of the same class (the plane screen's band is the largest radius of the segment).
  for (size_t a = 0; a < segs64.size(); ++a) {
    if (!(segs64[a].kind == 0 && segs64[a].xkind == 11)) continue;
    for (size_t b = 0; b < segs64.size(); ++b) {
      Seg64& mv = segs64[b];
      if (!(mv.kind == 1 && mv.xkind == 12)) continue;
      uint64_t ya, yb;
      std::memcpy(&ya, &segs64[a].y, 8); std::memcpy(&yb, &mv.y, 8);
      if (ya != yb) continue;
      double ra = 0.0, rb = 0.0;
      for (int64_t i : segs64[a].ids) ra = std::fmax(ra, r2_of(i));
      for (int64_t i : mv.ids) rb = std::fmax(rb, r2_of(i));
      if (!(ra <= 4.0 * rb && rb <= 4.0 * ra)) continue;
      mv.ids.insert(mv.ids.end(), segs64[a].ids.begin(), segs64[a].ids.end());
      segs64.erase(segs64.begin() + (long)a);
      --a;
      break;
    }
  }
  // The largest segment first: the screened loop works through the sorted list in words of 32 slots, and a segment that begins on
  // a word boundary runs whole words from its first object (kernel/integrate_loop_plane.inc: a word costs 21 scalar instructions,
  // the same 32 slots as four blocks of 8 cost ~230).  random_scene: 481 resting spheres, then the ground and the three big ones
  // -- 15 words and 2 blocks where the order of appearance (big spheres first) gave 14 words and 6 blocks.
  std::stable_sort(segs64.begin(), segs64.end(), [](const Seg64& a, const Seg64& b) { return a.ids.size() > b.ids.size(); });
  // (a member of a mover segment as a mover: a static sphere that joined one is c0 == c1)
  auto as_mover = [&](int64_t idx) {
    if (objs[idx].kind == TOR_MOVING_SPHERE) return objs[idx].u.moving_sphere;
    TorMovingSphere m{};
    m.center0 = m.center1 = objs[idx].u.sphere.center;
    m.radius = objs[idx].u.sphere.radius;
    m.material = objs[idx].u.sphere.material;
    return m;
  };
  // float64 per slot of the second-stage records (xrec) and of the plane table (xpl), by xkind
  auto xs_of = [](int xkind) { return xkind == 0 ? 0 : (xkind >= 13 ? 8 : 4); };
  auto pw_of = [](int xkind) { return xkind == 0 ? 0 : (xkind == 13 ? 4 : 2); };
  // Slots per segment: padded to blocks of 8; a plane-screened segment whose tail would be three or four blocks is padded to the
  // next WORD instead (32 slots: the tail runs as one whole word of stage one -- 131 vector + 21 scalar instructions -- where three
  // blocks cost 96 + ~120, and every segment behind it starts on a word boundary: kernel/integrate_loop_plane.inc), unless
  // that would push the sorted list into one more 512-slot pass.  random_scene is not affected (481 = 15 words + 1).
  auto slots_of = [&](const Seg64& sg, bool words) {
    size_t c = padded(sg.ids.size());
    if (words && sg.xkind >= 10 && sg.ids.size() >= 48 && c % 32 > 16) c = (c + 31) / 32 * 32;
    return c;
  };
  bool pad_words = true;
  {
    size_t t0 = 0, t1 = 0;
    for (const Seg64& sg : segs64) { t0 += slots_of(sg, false); t1 += slots_of(sg, true); }
    if ((t1 + 511) / 512 > (t0 + 511) / 512) pad_words = false;
  }
  size_t n_stat_p = 0, n_mov_p = 0, n_movy_p = 0, n_xrec = 0, n_xpl = 0;
  for (const Seg64& sg : segs64) {
    const size_t cp = slots_of(sg, pad_words);
    (sg.kind == 0 ? n_stat_p : (sg.kind == 1 ? n_movy_p : n_mov_p)) += cp;
    n_xrec += cp * (size_t)xs_of(sg.xkind);
    n_xpl += cp * (size_t)pw_of(sg.xkind);
  }
  std::vector<char> yonly32(groups32.size(), 0);
  size_t n32_slots = padded(statics32.size()), n32_floats = padded(statics32.size()) / 2 * 10;
  for (size_t gi = 0; gi < groups32.size(); ++gi) {
    const bool y = group_moves_along_y_only(objs, groups32[gi]);
    yonly32[gi] = y ? 1 : 0;
    n32_slots += padded(groups32[gi].ids.size());
    n32_floats += padded(groups32[gi].ids.size()) / 2 * (y ? 12 : 16);
  }
  out.n_sorted = n_stat_p + n_mov_p + n_movy_p + n32_slots;
  // one record of slack behind every hot array: the object loop requests record k+1 while it
  // works on record k
  out.stat.assign(4 * n_stat_p + 8, 0.0);
  out.mov.assign(8 * n_mov_p + 8, 0.0);
  out.movy.assign(6 * n_movy_p + 8, 0.0);
  out.xrec.assign(n_xrec + 8, 0.0);
  out.xpl.assign(n_xpl + 8, 1e300);  // padding: far from every ground track (a ray that keeps everything keeps it too: the second form drops it)
  out.hot32.assign(n32_floats + 32, 0.0f);
  out.cold.assign(16 * out.n_sorted + 16, 0.0);
  out.segs.clear();
  out.xsegs.clear();
  // padding record: centre 0, radius^2 = -1  => discriminant <= -|d|^2 < 0, never a candidate
  for (size_t k = 0; k < n_stat_p; ++k) out.stat[4 * k + 3] = -1.0;
  for (size_t k = 0; k < n_mov_p; ++k) out.mov[8 * k + 3] = -1.0;
  for (size_t k = 0; k < n_movy_p; ++k) out.movy[6 * k + 3] = -1.0;
  for (size_t k = 0; k < out.n_sorted; ++k) out.cold[16 * k + 15] = -1.0;

  // segs[6], segs[7] of the float64 segments (kinds 0-2): what the conservative FMA screen of the strict object loop needs to
  // know about the segment's members (tor_kernels.hip: screen_filter) -- an upper bound of |c0| + |r| and of |c1 - c0|.  A
  // non-finite bound turns into infinite margins there: everything is kept and the exact test decides.
  auto up = [](double x) { return x * (1.0 + 0x1p-40); };
  size_t sorted = 0, stat_rec = 0, mov_rec = 0, movy_rec = 0, x_off = 0, pl_off = 0;
  for (const Seg64& sg : segs64) {
    const size_t cnt_p = slots_of(sg, pad_words);
    double reach = 0.0, travel = 0.0, rmax2 = 0.0;
    // bounding box of the centres' ground projection (+ the movers' travel): what plane_pays (tor_screen.hpp) weighs the band against
    double xlo = INFINITY, xhi = -INFINITY, zlo = INFINITY, zhi = -INFINITY, dcx_max = 0.0, dcz_max = 0.0;
    auto grow = [&](double x, double z) {
      if (std::isfinite(x)) { xlo = std::fmin(xlo, x); xhi = std::fmax(xhi, x); }
      if (std::isfinite(z)) { zlo = std::fmin(zlo, z); zhi = std::fmax(zhi, z); }
    };
    for (int64_t idx : sg.ids) {
      if (sg.kind == 0) {
        const TorSphere& s = objs[idx].u.sphere;
        reach = std::fmax(reach, norm3(s.center.x, s.center.y, s.center.z) + std::fabs(s.radius));
        rmax2 = std::fmax(rmax2, s.radius * s.radius);
        grow(s.center.x, s.center.z);
      } else {
        const TorMovingSphere s = as_mover(idx);
        reach = std::fmax(reach, norm3(s.center0.x, s.center0.y, s.center0.z) + std::fabs(s.radius));
        rmax2 = std::fmax(rmax2, s.radius * s.radius);
        travel = std::fmax(travel, norm3(s.center1.x - s.center0.x, s.center1.y - s.center0.y, s.center1.z - s.center0.z));
        grow(s.center0.x, s.center0.z);
        dcx_max = std::fmax(dcx_max, std::fabs(s.center1.x - s.center0.x));
        dcz_max = std::fmax(dcz_max, std::fabs(s.center1.z - s.center0.z));
      }
    }
    const double ext_x = xhi >= xlo ? (xhi - xlo) + dcx_max : 0.0, ext_z = zhi >= zlo ? (zhi - zlo) + dcz_max : 0.0;
    const size_t first_rec = sg.kind == 0 ? stat_rec : (sg.kind == 1 ? movy_rec : mov_rec);
    // (segs[2] of kinds 0-2: padded count | padding records << 24 -- the kernel tests only the real objects of the last block)
    out.segs.insert(out.segs.end(), {(double)sg.kind, (double)first_rec, (double)(cnt_p | ((cnt_p - sg.ids.size()) << 24)), (double)(sorted / kPad),
                                     sg.kind == 0 ? 0.0 : sg.t0, sg.kind == 0 ? 0.0 : sg.dt, up(reach), up(travel)});
    // per-lane stage-two records of the screen (tor_screen.hpp): {xkind, first float64 of the records, common c0.y}
    // ... and of its stage one: first float64 of the segment's plane table in xpl, largest radius^2 (a NaN radius: fmax skipped
    // it -- and the reference can never hit that sphere), extents of the centres' ground projection
    out.xsegs.insert(out.xsegs.end(), {(double)sg.xkind, (double)x_off, sg.y, (double)pl_off, rmax2, ext_x, ext_z, 0.0});
    const size_t xs = (size_t)xs_of(sg.xkind), pw = (size_t)pw_of(sg.xkind);
    for (size_t k = 0; k < cnt_p && xs != 0; ++k) {  // padding: never a candidate (second form: t'' = T - 1e300 < 0, disc'' < 0; first form: r^2 = -1), except for a wild ray, which the exact test rejects
      double* x = &out.xrec[x_off + xs * k];
      if (xs == 4) x[sg.xkind == 10 ? 3 : 2] = 1e300;
      else x[3] = -1.0;
    }
    for (size_t k = 0; k < cnt_p && pw == 4; ++k) {  // (the table's fill value 1e300 is a centre far from every ground track; a mover's dc slots must not carry it)
      double* pl = &out.xpl[pl_off + 4 * k];
      pl[2] = 0.0; pl[3] = 0.0;
    }
    for (size_t k = 0; k < sg.ids.size(); ++k) {
      const TorHittableVariant& hv = objs[sg.ids[k]];
      if (sg.kind == 0) {
        const TorSphere& s = hv.u.sphere;
        double* m = &out.stat[4 * (stat_rec + k)];
        m[0] = s.center.x; m[1] = s.center.y; m[2] = s.center.z;
        m[3] = s.radius * s.radius;
        double* x = &out.xrec[x_off + 4 * k];
        if (sg.xkind == 10) { x[0] = s.center.x; x[1] = s.center.y; x[2] = s.center.z; x[3] = screen2_K(s.center.x, s.center.y, s.center.z, s.radius * s.radius); }
        else { x[0] = s.center.x; x[1] = s.center.z; x[2] = screen2_Ky(s.center.x, s.center.z, s.radius * s.radius); x[3] = 0.0; }
        out.xpl[pl_off + 2 * k] = s.center.x; out.xpl[pl_off + 2 * k + 1] = s.center.z;
      } else {
        const TorMovingSphere s = as_mover(sg.ids[k]);
        const double dcx = s.center1.x - s.center0.x, dcy = s.center1.y - s.center0.y, dcz = s.center1.z - s.center0.z;
        if (sg.kind == 1) {
          double* m = &out.movy[6 * (movy_rec + k)];
          m[0] = s.center0.x; m[1] = s.center0.y; m[2] = s.center0.z;
          m[3] = s.radius * s.radius;
          m[4] = dcy;
          if (sg.xkind == 12) {
            double* x = &out.xrec[x_off + 4 * k];
            x[0] = s.center0.x; x[1] = s.center0.z; x[2] = screen2_Ky(s.center0.x, s.center0.z, s.radius * s.radius);
            x[3] = dcy;
          }
          if (sg.xkind != 0) { out.xpl[pl_off + 2 * k] = s.center0.x; out.xpl[pl_off + 2 * k + 1] = s.center0.z; }
        } else {
          double* m = &out.mov[8 * (mov_rec + k)];
          m[0] = s.center0.x; m[1] = s.center0.y; m[2] = s.center0.z;
          m[3] = s.radius * s.radius;
          m[4] = dcx; m[5] = dcy; m[6] = dcz;
          if (sg.xkind == 13) {
            double* pl = &out.xpl[pl_off + 4 * k];
            pl[0] = s.center0.x; pl[1] = s.center0.z; pl[2] = dcx; pl[3] = dcz;
          }
        }
        if (sg.xkind >= 13) {  // stage two of 13 / 14 runs the first form per lane: the mover record, 64 bytes for either kind
          double* x = &out.xrec[x_off + 8 * k];
          x[0] = s.center0.x; x[1] = s.center0.y; x[2] = s.center0.z; x[3] = s.radius * s.radius;
          x[4] = dcx; x[5] = dcy; x[6] = dcz; x[7] = 0.0;
        }
      }
      if (!fill_cold(&out.cold[16 * (sorted + k)], hv, sg.ids[k])) { err = "unknown Material kind"; return false; }
    }
    (sg.kind == 0 ? stat_rec : (sg.kind == 1 ? movy_rec : mov_rec)) += cnt_p;
    x_off += cnt_p * xs;
    pl_off += cnt_p * pw;
    sorted += cnt_p;
  }

  // ---- float32 pre-filter segments: pair records {cx cx', cy cy', cz cz', r2 r2', k k' [, dcy dcy' | dcx.. dcy.. dcz..]}
  size_t f_off = 0;  // float offset into hot32
  auto emit32 = [&](const std::vector<int64_t>& members, int kind, double t0, double dt) -> bool {
    const int stride = kind == 5 ? 10 : (kind == 6 ? 12 : 16);
    const size_t cnt_p = padded(members.size());
    double mc0max = 0.0, dcmax = 0.0;
    for (size_t k = 0; k < cnt_p; ++k) {
      float* rec = &out.hot32[f_off + (k / 2) * (size_t)stride];
      const int h = (int)(k & 1);
      if (k >= members.size()) {  // padding: never kept (tor_filter32.hpp)
        rec[6 + h] = kF32PadR2;
        continue;
      }
      const TorHittableVariant& hv = objs[members[k]];
      const bool moving = hv.kind == TOR_MOVING_SPHERE;
      const TorVec3& c0 = moving ? hv.u.moving_sphere.center0 : hv.u.sphere.center;
      const double r = moving ? hv.u.moving_sphere.radius : hv.u.sphere.radius;
      const double q[3] = {c0.x - f32->origin[0], c0.y - f32->origin[1], c0.z - f32->origin[2]};
      const double mc0 = norm3(q[0], q[1], q[2]);
      rec[0 + h] = (float)q[0]; rec[2 + h] = (float)q[1]; rec[4 + h] = (float)q[2];
      rec[6 + h] = (float)(r * r);
      rec[8 + h] = f32_object_k(mc0, r * r, moving);
      mc0max = std::max(mc0max, mc0);
      if (moving) {
        const TorMovingSphere& s = hv.u.moving_sphere;
        const double dc[3] = {s.center1.x - s.center0.x, s.center1.y - s.center0.y, s.center1.z - s.center0.z};
        dcmax = std::max(dcmax, norm3(dc[0], dc[1], dc[2]));
        if (kind == 6) {
          rec[10 + h] = (float)dc[1];
        } else {
          rec[10 + h] = (float)dc[0]; rec[12 + h] = (float)dc[1]; rec[14 + h] = (float)dc[2];
        }
      }
      if (!fill_cold(&out.cold[16 * (sorted + k)], hv, members[k])) { err = "unknown Material kind"; return false; }
    }
    out.segs.insert(out.segs.end(), {(double)kind, (double)f_off, (double)cnt_p, (double)(sorted / kPad), t0, dt,
                                     (double)f32_round_up(mc0max * 1.000001), (double)f32_round_up(dcmax * 1.000001)});
    out.xsegs.insert(out.xsegs.end(), 8, 0.0);  // (the two tables stay parallel)
    f_off += cnt_p / 2 * (size_t)stride;
    sorted += cnt_p;
    return true;
  };
  if (!statics32.empty() && !emit32(statics32, 5, 0.0, 1.0)) return false;
  for (size_t gi = 0; gi < groups32.size(); ++gi) {
    const TorMovingSphere& first = objs[groups32[gi].ids[0]].u.moving_sphere;
    if (!emit32(groups32[gi].ids, yonly32[gi] ? 6 : 7, first.time0, first.time1 - first.time0)) return false;
  }
  out.n_segs = (int)(out.segs.size() / 8);
  if (out.segs.empty()) out.segs.assign(8, 0.0);
  if (out.xsegs.empty()) out.xsegs.assign(8, 0.0);
  // ---- xhdr: what the ARITH 2 object loop reads per segment, as ONE 128-byte record (tor_scene.hpp): the integer fields of segs /
  // xsegs as 32-bit integers (the loop's scalar unit has no float64 -> int conversion), the float64 fields, the first plane record
  out.xhdr.assign(16 * (size_t)std::max(out.n_segs, 1) + 16, 0.0);
  for (int s = 0; s < out.n_segs; ++s) {
    const double* sg = &out.segs[8 * (size_t)s];
    const double* xs = &out.xsegs[8 * (size_t)s];
    double* h = &out.xhdr[16 * (size_t)s];
    // the plane screen's gate, settled here where it does not depend on the ray (tor_screen.hpp plane_pays: the band 2 R against the
    // width E of the centres' box across the ground track, E between min(Sx, Sz) and hypot(Sx, Sz)): 1 always, 0 never, 2 the wave votes
    // ... and never on a segment of fewer than 48 objects: two stages' set-up costs more than 3 instructions per object save
    int32_t gate = 0;
    const int n_real = ((int)sg[2] & 0xffffff) - ((int)sg[2] >> 24);
    if ((int)xs[0] >= 10 && n_real >= 48) {
      const double band = 2.0 * kPlaneGate * std::sqrt(xs[4]);
      const double e_min = std::fmin(xs[5], xs[6]), e_max = std::hypot(xs[5], xs[6]);
      gate = band < e_min ? 1 : (band < e_max ? 2 : 0);     // (a NaN anywhere: never)
    }
    const int32_t ints[8] = {(int32_t)xs[0], (int32_t)sg[0], (int32_t)sg[2], (int32_t)sg[3], (int32_t)sg[1], (int32_t)xs[1], (int32_t)xs[3], gate};
    std::memcpy(h, ints, sizeof(ints));
    h[4] = sg[4]; h[5] = sg[5]; h[6] = sg[6]; h[7] = sg[7];
    h[8] = xs[2]; h[9] = xs[4]; h[10] = xs[5]; h[11] = xs[6];
    const int xkind = (int)xs[0];
    if (xkind >= 10) {
      const size_t pw = xkind == 13 ? 4 : 2;
      for (size_t k = 0; k < pw; ++k) h[12 + k] = out.xpl[(size_t)xs[3] + k];
    }
  }
  // trip table of the one-wave-per-pixel kernel
  const size_t n_trips = (out.n_sorted + 63) / 64;
  out.coop_trips.assign(4 * n_trips + 4, 0.0);
  for (size_t t = 0; t < n_trips; ++t) {
    bool any_mover = false, any_static = false, y_only = true, same_group = true, sane = true;
    uint64_t g0 = 0, g1 = 0;
    double t0 = 0.0, dt = 1.0;
    for (size_t k = 64 * t; k < 64 * (t + 1); ++k) {
      if (k >= out.n_sorted) { any_static = true; continue; }  // beyond the list: staged as never-hit static records
      const double* cr = &out.cold[16 * k];
      int64_t flags;
      std::memcpy(&flags, &cr[13], 8);
      if (!(flags & 1)) { any_static = true; continue; }
      uint64_t b0, b1;
      std::memcpy(&b0, &cr[7], 8); std::memcpy(&b1, &cr[8], 8);
      if (!any_mover) { g0 = b0; g1 = b1; t0 = cr[7]; dt = cr[8]; }
      else if (b0 != g0 || b1 != g1) same_group = false;
      any_mover = true;
      if (!(cr[3] == 0.0 && cr[5] == 0.0)) y_only = false;
      if (!std::isfinite(cr[7]) || !std::isfinite(cr[8]) || cr[8] == 0.0) sane = false;
    }
    double kind = 3.0;
    if (!any_mover) kind = 0.0;
    else if (!any_static && same_group && sane) kind = y_only ? 1.0 : 2.0;
    out.coop_trips[4 * t + 0] = kind;
    out.coop_trips[4 * t + 1] = t0;
    out.coop_trips[4 * t + 2] = dt;
  }
  return true;
}

namespace {

// scenes of more blocks than this get a second level of boxes (measured, round 2: 61 blocks are faster flat, 200 with two levels)
constexpr size_t kTwoLevelMinBlocks = 96;

uint32_t spread10(uint32_t v) {  // 10 bits -> every third bit
  v &= 0x3ff;
  v = (v | (v << 16)) & 0x030000FF;
  v = (v | (v << 8)) & 0x0300F00F;
  v = (v | (v << 4)) & 0x030C30C3;
  v = (v | (v << 2)) & 0x09249249;
  return v;
}

bool finite3(const TorVec3& v) { return std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z); }

}  // namespace

void build_accel(const TorHittableVariant* objs, int64_t n, HostAccel& out, const F32Options* f32) {
  out = HostAccel{};
  if (n < 64) return;  // not worth a second level
  // radius of the typical object
  std::vector<double> radii;
  radii.reserve((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    const double r = objs[i].kind == TOR_SPHERE ? objs[i].u.sphere.radius : objs[i].u.moving_sphere.radius;
    radii.push_back(std::fabs(r));
  }
  std::vector<double> tmp = radii;
  std::nth_element(tmp.begin(), tmp.begin() + tmp.size() / 2, tmp.end());
  const double median_r = tmp[tmp.size() / 2];
  std::vector<int64_t> always, spatial;
  for (int64_t i = 0; i < n; ++i) {
    bool ok = std::isfinite(radii[(size_t)i]) && radii[(size_t)i] <= 2.5 * median_r;
    if (objs[i].kind == TOR_SPHERE) {
      ok = ok && finite3(objs[i].u.sphere.center);
    } else if (objs[i].kind == TOR_MOVING_SPHERE) {
      const TorMovingSphere& s = objs[i].u.moving_sphere;
      const double dt = s.time1 - s.time0;
      ok = ok && finite3(s.center0) && finite3(s.center1) && std::isfinite(s.time0) && std::isfinite(dt) && dt != 0.0;
    } else {
      return;  // unknown kind: the brute-force upload reports it
    }
    (ok ? spatial : always).push_back(i);
  }
  if (spatial.size() < 32) return;
  std::string err;
  if (!build_layout(objs, always, out.always, err, f32)) return;
  // Order of the spatial objects (closest hit is order independent -- hittables_lists.nim:48-55 -- so the order is free, and
  // everything downstream only assumes "blocks = 8 consecutive objects, super boxes = 8 consecutive blocks").
  // Round 6: a top-down build by the surface-area heuristic instead of the Morton curve of rounds 1-5.  Every node sorts its
  // objects along each axis, prices every split position that keeps the children whole -- multiples of 8 objects, of 64 (one super
  // box) above that when the scene will get a second level -- by area(left) x count(left) + area(right) x count(right) over
  // the hulls of the spheres' whole motion, and takes the cheapest.  Blocks cut from a Morton curve straddle the curve's
  // jumps: on random_scene the summed area of the 61 block boxes falls from 2802 to 1467 and a ray enters ~0.6 x the blocks, on
  // the 1601-sphere animation frames 23 034 -> 11 259 and ~0.5 x (both levels); the expansion work per query falls with it.
  // TOR_ACCEL_ORDER=morton keeps the old order (A/B; same canvas either way).
  std::vector<std::pair<uint32_t, int64_t>> keyed;
  keyed.reserve(spatial.size());
  const char* order_knob = knob("TOR_ACCEL_ORDER");
  const bool use_morton = order_knob && std::strcmp(order_knob, "morton") == 0;
  if (use_morton) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    auto c0 = [&](int64_t i, int a) {
      const TorVec3& c = objs[i].kind == TOR_SPHERE ? objs[i].u.sphere.center : objs[i].u.moving_sphere.center0;
      return a == 0 ? c.x : (a == 1 ? c.y : c.z);
    };
    for (int64_t i : spatial)
      for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], c0(i, a)); hi[a] = std::max(hi[a], c0(i, a)); }
    double ext = 1e-300;
    for (int a = 0; a < 3; ++a) ext = std::max(ext, hi[a] - lo[a]);
    for (int64_t i : spatial) {
      uint32_t q[3];
      for (int a = 0; a < 3; ++a) q[a] = (uint32_t)std::min(1023.0, std::max(0.0, (c0(i, a) - lo[a]) / ext * 1023.0));
      keyed.push_back({spread10(q[0]) | (spread10(q[2]) << 1) | (spread10(q[1]) << 2), i});
    }
    std::stable_sort(keyed.begin(), keyed.end());
  } else {
    const size_t ns = spatial.size();
    const bool second_level = ((ns + kPad - 1) / kPad + (size_t)kBoxFanout - 1) / (size_t)kBoxFanout > kTwoLevelMinBlocks;
    struct Hull { double lo[3], hi[3], mid[3]; };
    std::vector<Hull> hull(ns);
    for (size_t k = 0; k < ns; ++k) {
      const TorHittableVariant& ob = objs[spatial[k]];
      const bool mv = ob.kind == TOR_MOVING_SPHERE;
      const TorVec3& a0 = mv ? ob.u.moving_sphere.center0 : ob.u.sphere.center;
      const TorVec3& a1 = mv ? ob.u.moving_sphere.center1 : ob.u.sphere.center;
      const double r = radii[(size_t)spatial[k]];
      const double p0[3] = {a0.x, a0.y, a0.z}, p1[3] = {a1.x, a1.y, a1.z};
      for (int a = 0; a < 3; ++a) {
        hull[k].lo[a] = std::min(p0[a], p1[a]) - r;
        hull[k].hi[a] = std::max(p0[a], p1[a]) + r;
        hull[k].mid[a] = 0.5 * (p0[a] + p1[a]);
      }
    }
    std::vector<uint32_t> ord(ns), best_ord, cand;
    for (size_t k = 0; k < ns; ++k) ord[k] = (uint32_t)k;
    std::vector<double> suffix;  // area of the hull of cand[pos..n)
    auto half_area = [](const double lo[3], const double hi[3]) {
      const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
      return dx * dy + dy * dz + dx * dz;
    };
    // (explicit stack: a degenerate scene must not recurse 10^5 deep)
    std::vector<std::pair<size_t, size_t>> todo{{0, ns}};
    while (!todo.empty()) {
      const auto [b, e] = todo.back();
      todo.pop_back();
      const size_t n_here = e - b;
      if (n_here <= (size_t)kPad) continue;
      // (a culling box stands for kBoxFanout blocks -- 1 in every shipped build --, a super box for 8 boxes)
      const size_t box_objs = (size_t)kPad * (size_t)kBoxFanout;
      const size_t unit = (second_level && n_here > box_objs * kPad) ? box_objs * kPad : (n_here > box_objs ? box_objs : (size_t)kPad);
      double best_cost = INFINITY;
      size_t best_pos = (n_here / 2 + unit - 1) / unit * unit;  // fall-back (non-finite areas): the median, on x
      if (best_pos >= n_here) best_pos = (n_here - 1) / unit * unit;
      bool have = false;
      for (int ax = 0; ax < 3; ++ax) {
        cand.assign(ord.begin() + (long)b, ord.begin() + (long)e);
        std::stable_sort(cand.begin(), cand.end(), [&](uint32_t x, uint32_t y) { return hull[x].mid[ax] < hull[y].mid[ax]; });
        suffix.assign(n_here + 1, 0.0);
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t k = n_here; k-- > 0;) {
          for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], hull[cand[k]].lo[a]); hi[a] = std::max(hi[a], hull[cand[k]].hi[a]); }
          suffix[k] = half_area(lo, hi);
        }
        for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
        for (size_t k = 0; k + 1 < n_here; ++k) {
          for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], hull[cand[k]].lo[a]); hi[a] = std::max(hi[a], hull[cand[k]].hi[a]); }
          const size_t pos = k + 1;
          if (pos % unit != 0) continue;
          const double cost = half_area(lo, hi) * (double)pos + suffix[pos] * (double)(n_here - pos);
          // (ties -- many identical spheres: every split costs the same -- go to the most balanced split: the build stays
          // n log^2 n instead of peeling one block per node)
          auto off_middle = [&](size_t q_) { return q_ > n_here / 2 ? q_ - n_here / 2 : n_here / 2 - q_; };
          if (cost < best_cost || (have && cost == best_cost && off_middle(pos) < off_middle(best_pos))) {
            best_cost = cost; best_pos = pos; best_ord = cand; have = true;
          }
        }
        if (!have && ax == 0) best_ord = cand;
      }
      std::copy(best_ord.begin(), best_ord.end(), ord.begin() + (long)b);
      todo.push_back({b + best_pos, e});
      todo.push_back({b, b + best_pos});
    }
    for (size_t k = 0; k < ns; ++k) keyed.push_back({(uint32_t)k, spatial[ord[k]]});
  }
  out.n_blocks = (keyed.size() + kPad - 1) / kPad;
  // box fan-out (tor_scene.hpp, tor_kernels.hpp kBoxFanout): 1.  Measured as a launch parameter in round 4 (commit c39354b,
  // profiles/r4_fanout_sweep.txt): halving the box tests does NOT pay -- random_scene 56.4 / 59.0 / 73.0 / 115.3 ms and the
  // 1601-object animation frame 38.3 / 40.7 / 50.7 / 108.1 ms for fan-out 1 / 2 / 4 / 8 (both accelerations, per-sample streams): a
  // ray that skims the ground enters the union of two neighbouring blocks almost as often as it enters either, so the expansion
  // work nearly doubles.  The host code stays general; the kernels take the constant.
  out.fanout = kBoxFanout;
  out.n_boxes = (out.n_blocks + (size_t)out.fanout - 1) / (size_t)out.fanout;
  out.spatial_base = out.always.n_sorted;
  out.cold = out.always.cold;
  // cold slots for every block behind the (padded) box segment, all never-hit until filled
  const size_t n_bnd_slots = (out.n_boxes + kPad - 1) / kPad * kPad * (size_t)out.fanout;
  out.cold.resize(16 * (out.spatial_base + n_bnd_slots * kPad) + 16, 0.0);
  out.spatial.assign(out.n_blocks * kPad, HostAccel::Obj{});
  for (size_t k = 0; k < n_bnd_slots * kPad; ++k) out.cold[16 * (out.spatial_base + k) + 15] = -1.0;
  out.hot.assign(8 * n_bnd_slots * kPad + 8, 0.0);
  out.groups.clear();
  for (size_t k = 0; k < n_bnd_slots * kPad; ++k) { out.hot[8 * k + 3] = -1.0; out.hot[8 * k + 7] = -1.0; }
  for (size_t k = 0; k < keyed.size(); ++k) {
    const int64_t i = keyed[k].second;
    double* c = &out.cold[16 * (out.spatial_base + k)];
    std::fill(c, c + 16, 0.0);
    if (!fill_cold(c, objs[i], i)) { out = HostAccel{}; return; }
    // compact record the block expansion reads: {c0 xyz, r^2, dc xyz, time-group id (-1: static)};
    // re-packed to {c0 xyz, r^2} below when no spatial object moves
    double* hrec = &out.hot[8 * k];
    hrec[0] = c[0]; hrec[1] = c[1]; hrec[2] = c[2]; hrec[3] = c[15];
    hrec[4] = c[3]; hrec[5] = c[4]; hrec[6] = c[5];
    hrec[7] = -1.0;
    if (objs[i].kind == TOR_MOVING_SPHERE) {
      size_t g = 0;
      for (; g < out.groups.size() / 2; ++g) {
        uint64_t a0, a1, b0, b1;
        std::memcpy(&a0, &out.groups[2 * g], 8); std::memcpy(&a1, &out.groups[2 * g + 1], 8);
        std::memcpy(&b0, &c[7], 8); std::memcpy(&b1, &c[8], 8);
        if (a0 == b0 && a1 == b1) break;
      }
      if (g == out.groups.size() / 2) { out.groups.push_back(c[7]); out.groups.push_back(c[8]); }
      hrec[7] = (double)g;
    }
    HostAccel::Obj& o = out.spatial[k];
    o.valid = true;
    o.moving = objs[i].kind == TOR_MOVING_SPHERE;
    for (int a = 0; a < 3; ++a) { o.c0[a] = c[a]; o.dc[a] = c[3 + a]; }
    o.t0 = c[7]; o.dt = c[8];
    o.abs_r = radii[(size_t)i];
  }
  // the bounds segment (kind 3): records live in KParams.bnd, one per BOX (fanout blocks), padded to 8
  const size_t n_bnd_p = (out.n_boxes + kPad - 1) / kPad * kPad;
  if (out.always.n_segs == 0) out.always.segs.clear();
  // few blocks: the wave-uniform loop tests the block boxes themselves (kind 3); otherwise it tests
  // the super boxes (kind 4: first record n_bnd_p + 1 of the bounds array) and the lanes descend
  const size_t n_super = n_bnd_p / kPad;
  const size_t n_super_p = (n_super + kPad - 1) / kPad * kPad;
  const size_t two_level_min = kTwoLevelMinBlocks;
  out.two_level = out.n_boxes > two_level_min;  // measured: 61 blocks are faster flat (2125 vs 1725 Msamples/s), 200 blocks faster with two levels (1916 vs 1663)
  if (out.two_level)
    out.always.segs.insert(out.always.segs.end(), {4.0, (double)(n_bnd_p + 1), (double)n_super_p, 0.0, 0.0, 0.0, 0.0, 0.0});
  else
    out.always.segs.insert(out.always.segs.end(), {3.0, 0.0, (double)n_bnd_p, 0.0, 0.0, 0.0, 0.0, 0.0});
  out.always.n_segs += 1;
  out.hot_stride = 8;
  if (out.groups.empty()) {
    out.groups.assign(2, 1.0);
    // all static: 32-byte records (a 1600-object scene then fits LDS next to the queues)
    const size_t slots = n_bnd_slots * kPad;
    std::vector<double> packed(4 * slots + 8, 0.0);
    for (size_t k = 0; k < slots; ++k)
      for (int a = 0; a < 4; ++a) packed[4 * k + a] = out.hot[8 * k + a];
    out.hot.swap(packed);
    out.hot_stride = 4;
  }
  // float32 records for the per-lane block expansion
  if (f32 && out.groups.size() <= 2) {
    bool ok = true, any_mover = false, y_only = true;
    double mc0max = 0.0, dcmax = 0.0;
    for (const HostAccel::Obj& o : out.spatial) {
      if (!o.valid) continue;
      const double mc0 = norm3(o.c0[0] - f32->origin[0], o.c0[1] - f32->origin[1], o.c0[2] - f32->origin[2]);
      const double dcn = o.moving ? norm3(o.dc[0], o.dc[1], o.dc[2]) : 0.0;
      ok = ok && f32_eligible(mc0, dcn, o.abs_r * o.abs_r) && mc0 <= f32->far_limit;
      mc0max = std::max(mc0max, mc0);
      dcmax = std::max(dcmax, dcn);
      if (o.moving) {
        any_mover = true;
        y_only = y_only && o.dc[0] == 0.0 && o.dc[2] == 0.0;
      }
    }
    if (ok) {
      out.sp32 = true;
      out.hot32_stride = !any_mover ? 10 : (y_only ? 12 : 16);
      const int st = out.hot32_stride;
      const size_t slots = n_bnd_slots * kPad;
      // A block's four pair records are followed by 4 floats of padding: the per-lane reads of the block expansion
      // (64 lanes, 64 different blocks) then start at 16 different LDS bank groups instead of 4 (st = 12: 48 floats
      // per block, 48 mod 64 takes 4 values; 52 takes 16) -- half of the LDS cycles were bank conflicts.
      out.hot32_block_stride = 4 * st + 4;
      out.hot32.assign(slots / kPad * (size_t)out.hot32_block_stride + 32, 0.0f);
      for (size_t k = 0; k < slots; ++k) {
        float* rec = &out.hot32[(k / kPad) * (size_t)out.hot32_block_stride + ((k % kPad) / 2) * (size_t)st];
        const int h = (int)(k & 1);
        const HostAccel::Obj* o = k < out.spatial.size() && out.spatial[k].valid ? &out.spatial[k] : nullptr;
        if (!o) {
          rec[6 + h] = kF32PadR2;
          continue;
        }
        const double q[3] = {o->c0[0] - f32->origin[0], o->c0[1] - f32->origin[1], o->c0[2] - f32->origin[2]};
        rec[0 + h] = (float)q[0]; rec[2 + h] = (float)q[1]; rec[4 + h] = (float)q[2];
        rec[6 + h] = (float)(o->abs_r * o->abs_r);
        rec[8 + h] = f32_object_k(norm3(q[0], q[1], q[2]), o->abs_r * o->abs_r, o->moving);
        if (o->moving) {
          if (st == 12) rec[10 + h] = (float)o->dc[1];
          else { rec[10 + h] = (float)o->dc[0]; rec[12 + h] = (float)o->dc[1]; rec[14 + h] = (float)o->dc[2]; }
          out.sp_t0 = o->t0; out.sp_dt = o->dt;
        }
      }
      out.sp_mc0max = f32_round_up(mc0max * 1.000001);
      out.sp_dcmax = f32_round_up(dcmax * 1.000001);
    }
  }
  out.available = true;
}

bool compute_block_bounds(const HostAccel& acc, double t_lo, double t_hi, std::vector<double>& bnd) {
  if (!std::isfinite(t_lo) || !std::isfinite(t_hi)) return false;
  const size_t F = (size_t)(acc.fanout > 0 ? acc.fanout : 1);
  const size_t n_bnd_p = (acc.n_boxes + kPad - 1) / kPad * kPad;
  bnd.assign(8 * n_bnd_p + 16, 0.0);
  // padding / empty block: NaN bounds.  Every slab product is NaN, v_min/v_max drop NaN operands, so
  // t_in = 0 and t_out = NaN and `t_in <= t_out` is false.  (An inverted box would NOT do: the slab
  // test is symmetric in lo/hi.)  The cold slots behind padding bounds hold never-hit records anyway.
  const double qnan = std::nan("");
  for (size_t b = 0; b < n_bnd_p + 1; ++b)
    for (int a = 0; a < 6; ++a) bnd[8 * b + a] = qnan;
  for (size_t b = 0; b < acc.n_boxes; ++b) {  // box b: blocks [b F, (b + 1) F)
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    bool any = false;
    for (size_t j = 0; j < F * (size_t)kPad; ++j) {
      const size_t slot = b * F * (size_t)kPad + j;
      if (slot >= acc.spatial.size()) break;
      const HostAccel::Obj& o = acc.spatial[slot];
      if (!o.valid) continue;
      // the centre moves linearly in t: the swept sphere lies in the hull of its two end positions
      const int ends = o.moving ? 2 : 1;
      for (int e = 0; e < ends; ++e) {
        const double f = o.moving ? ((e == 0 ? t_lo : t_hi) - o.t0) / o.dt : 0.0;
        if (!std::isfinite(f)) return false;
        for (int a = 0; a < 3; ++a) {
          const double c = o.c0[a] + f * o.dc[a];
          lo[a] = std::min(lo[a], c - o.abs_r);
          hi[a] = std::max(hi[a], c + o.abs_r);
        }
      }
      any = true;
    }
    if (!any) continue;
    for (int a = 0; a < 3; ++a) {
      // inflate far beyond any rounding of the float64 slab test (relative 1e-6 vs ~1e-15)
      const double pad = 1e-6 * (1.0 + std::fabs(lo[a]) + std::fabs(hi[a]) + (hi[a] - lo[a]));
      if (!std::isfinite(pad)) return false;
      bnd[8 * b + a] = lo[a] - pad;
      bnd[8 * b + 3 + a] = hi[a] + pad;
    }
  }
  // second level: one box around every 8 consecutive block boxes (the build aligns its upper splits to 64 objects
  // curve, so the groups are compact); stored behind the block boxes, padded with NaN boxes
  const size_t n_super = n_bnd_p / kPad;
  const size_t n_super_p = (n_super + kPad - 1) / kPad * kPad;
  const size_t super0 = n_bnd_p + 1;  // record index of the first super box (one slack record in between)
  bnd.resize(8 * (super0 + n_super_p + 1), 0.0);
  for (size_t sidx = 0; sidx < n_super_p + 1; ++sidx)
    for (int a = 0; a < 6; ++a) bnd[8 * (super0 + sidx) + a] = qnan;
  for (size_t sidx = 0; sidx < n_super; ++sidx) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    bool any = false;
    for (int j = 0; j < kPad; ++j) {
      const double* c = &bnd[8 * (sidx * kPad + j)];
      if (c[0] != c[0]) continue;  // NaN: empty child
      for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], c[a]); hi[a] = std::max(hi[a], c[3 + a]); }
      any = true;
    }
    if (!any) continue;
    for (int a = 0; a < 3; ++a) { bnd[8 * (super0 + sidx) + a] = lo[a]; bnd[8 * (super0 + sidx) + 3 + a] = hi[a]; }
  }
  return true;
}

float block_bounds_f32(const std::vector<double>& bnd, const double origin[3], std::vector<float>& bnd32, float* hmin_out) {
  const size_t n = bnd.size() / 8;
  bnd32.assign(8 * n, 0.0f);
  double bmax = 0.0, hmin = INFINITY;
  for (size_t b = 0; b < n; ++b) {
    const double* c = &bnd[8 * b];
    float* r = &bnd32[8 * b];
    if (c[0] != c[0]) {  // NaN box (padding): never entered
      for (int k = 0; k < 6; ++k) r[k] = std::nanf("");
      continue;
    }
    for (int ax = 0; ax < 3; ++ax) {
      // {centre, half-extent}: the float32 interval [fc - fh, fc + fh] (exact arithmetic on the float32 values) contains [lo, hi]
      const double lo = c[ax] - origin[ax], hi = c[3 + ax] - origin[ax];
      const float fc = (float)(0.5 * (lo + hi));
      const double need = std::max((double)fc - lo, hi - (double)fc);
      float fh = (float)need;
      if ((double)fh < need) fh = std::nextafterf(fh, INFINITY);
      fh = std::nextafterf(fh, INFINITY);
      r[2 * ax] = fc;
      r[2 * ax + 1] = fh;
      bmax = std::max(bmax, std::fabs((double)fc) + (double)fh);
      hmin = std::min(hmin, (double)fh);
    }
  }
  if (hmin_out) *hmin_out = std::isfinite(hmin) ? (float)hmin : 0.0f;   // (fh is a float32 already: exact)
  return f32_round_up(bmax);
}

}  // namespace tor
