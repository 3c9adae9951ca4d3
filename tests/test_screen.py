"""The conservative FMA screen of the strict float64 object loop (csrc/tor_screen.hpp, integrate_kernel variants ARITH 2) may keep
too much, never too little: everything it keeps is re-tested with the reference's own operations (spheres.nim:29-48), so a wrong
"keep" costs time and a wrong "drop" would cost a pixel.  Host build of the kernel's own source against the reference's unfused
test on random and adversarial ray / sphere pairs -- tangent rays to the last bit, origins on the surface (every scattered ray),
huge offsets (random_scene's ground sphere: centre 1000 below, radius 1000), tiny and huge directions, moving centres with the
time outside the object's interval.  CPU only."""
import numpy as np
import pytest

from test_filter32 import _cases, _unit


@pytest.mark.parametrize("scale,r_lo,r_hi,origin", [
    (12.0, 0.2, 0.2, (0.0, 0.0, 0.0)),              # random_scene-like
    (12.0, 0.05, 1.0, (3.0, 1.0, -2.0)),
    (300.0, 0.01, 5.0, (1000.0, -2000.0, 500.0)),   # far from the world origin: cancellation in o - c
    (2e4, 1.0, 100.0, (0.0, 0.0, 0.0)),
    (0.01, 1e-4, 1e-3, (0.0, 0.0, 0.0)),            # tiny scene
    (5.0, 900.0, 1100.0, (0.0, -1000.0, 0.0)),      # ground-sphere-like: radius ~ distance
])
def test_screen_never_drops_a_needed_object(tor, scale, r_lo, r_hi, origin):
    rng = np.random.default_rng(int(scale * 1000) + 11)
    origin = np.asarray(origin, dtype=np.float64)
    n = 400_000
    o, d, c0, dc, moving, f, r2 = _cases(rng, n, scale, r_lo, r_hi, origin)
    keep, need = tor.selftest_screen(o, d, c0, dc, moving, f, r2)
    missed = np.flatnonzero((need != 0) & (keep == 0))
    assert missed.size == 0, (missed[:5], o[missed[:1]], d[missed[:1]])
    assert np.count_nonzero(need) > n // 10
    # the second form (expanded quadratic, normalised direction): general static records, common-height static records, and the
    # movers cut down to motion along y (the only movers it takes)
    for variant in (0, 1):
        keep2, need2 = tor.selftest_screen2(o, d, c0, dc, moving, f, r2, variant)
        missed = np.flatnonzero((need2 != 0) & (keep2 == 0))
        assert missed.size == 0, (variant, missed[:5], o[missed[:1]], d[missed[:1]], c0[missed[:1]])
    dcy = dc.copy()
    dcy[:, 0] = 0.0
    dcy[:, 2] = 0.0
    keep2, need2 = tor.selftest_screen2(o, d, c0, dcy, moving, f, r2, 0)
    missed = np.flatnonzero((need2 != 0) & (keep2 == 0))
    assert missed.size == 0, (missed[:5], o[missed[:1]], d[missed[:1]], c0[missed[:1]], dcy[missed[:1]], f[missed[:1]])
    assert np.count_nonzero(need2) > n // 10
    # stage one of the common-height segments: the plane screen alone, statics and movers along y
    keep3, need3 = tor.selftest_screen2(o, d, c0, dcy, moving, f, r2, 2)
    missed = np.flatnonzero((need3 != 0) & (keep3 == 0))
    assert missed.size == 0, (missed[:5], o[missed[:1]], d[missed[:1]], c0[missed[:1]], dcy[missed[:1]], f[missed[:1]])
    assert np.array_equal(need3, need2)
    # round 5: stage one for movers in GENERAL position (xkind 13: the centre's x and z travel too, evaluated inside the chain)
    keep4, need4 = tor.selftest_screen2(o, d, c0, dc, moving, f, r2, 2)
    missed = np.flatnonzero((need4 != 0) & (keep4 == 0))
    assert missed.size == 0, (missed[:5], o[missed[:1]], d[missed[:1]], c0[missed[:1]], dc[missed[:1]], f[missed[:1]])
    assert np.array_equal(need4, need) and np.count_nonzero(need4) > n // 10


def test_screen_on_the_decision_boundary(tor):
    """Tangency to the last bit: directions perturbed by single ulps around the exact tangent, origins ON the sphere with
    directions in the tangent plane, and the reference's own noise region |disc| ~ 1e-16 x |hb|^2."""
    rng = np.random.default_rng(5)
    n = 300_000
    c0 = rng.uniform(-10, 10, size=(n, 3))
    r = rng.choice([0.2, 1.0, 1000.0], size=n)
    c0[r == 1000.0] = [0.0, -1000.0, 0.0]
    nrm = _unit(rng, n)
    o = c0 + nrm * r[:, None]                               # on the surface
    tang = np.cross(nrm, _unit(rng, n))
    tang /= np.linalg.norm(tang, axis=1, keepdims=True)
    lean = rng.choice([0.0, 1e-17, -1e-17, 1e-13, -1e-13, 1e-9, -1e-9, 1e-3, -1e-3], size=(n, 1))
    d = (tang + nrm * lean) * rng.choice([1.0, 1e-4, 1e4, 1e-9], size=(n, 1))
    # a second family: origin far away, ray through the exact tangent point, nudged by ulps
    far = rng.random(n) < 0.5
    o[far] = c0[far] + _unit(rng, int(far.sum())) * (r[far] * rng.uniform(1.5, 40.0, int(far.sum())))[:, None]
    to_c = c0[far] - o[far]
    dist = np.linalg.norm(to_c, axis=1)
    w = to_c / dist[:, None]
    perp = np.cross(w, _unit(rng, int(far.sum())))
    perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    target = c0[far] + perp * r[far, None]
    dd = target - o[far]
    k = rng.integers(-3, 4, size=dd.shape)
    d[far] = dd * (1.0 + k * 2.0 ** -52)
    moving = np.zeros(n, dtype=np.int32)
    keep, need = tor.selftest_screen(o, d, c0, np.zeros((n, 3)), moving, np.zeros(n), r * r)
    missed = np.flatnonzero((need != 0) & (keep == 0))
    assert missed.size == 0, (missed[:5], o[missed[:1]], d[missed[:1]])
    assert 0 < np.count_nonzero(need) < n            # both sides of the boundary are present
    for variant in (0, 1, 2):
        keep2, need2 = tor.selftest_screen2(o, d, c0, np.zeros((n, 3)), moving, np.zeros(n), r * r, variant)
        missed = np.flatnonzero((need2 != 0) & (keep2 == 0))
        assert missed.size == 0, (variant, missed[:5], o[missed[:1]], d[missed[:1]])
    # the same pairs as movers along y caught at a time fraction f: centre = c0 - f dc + f dc
    fm = rng.uniform(-0.5, 1.5, n)
    dcm = np.column_stack([np.zeros(n), rng.uniform(-0.5, 0.5, n), np.zeros(n)])
    for variant in (0, 2):
        keep2, need2 = tor.selftest_screen2(o, d, c0 - dcm * fm[:, None], dcm, np.ones(n, dtype=np.int32), fm, r * r, variant)
        missed = np.flatnonzero((need2 != 0) & (keep2 == 0))
        assert missed.size == 0, (variant, missed[:5], o[missed[:1]], d[missed[:1]])
        assert 0 < np.count_nonzero(need2) < n


def test_screen_drops_the_obvious(tor):
    """The margins are a few hundred ulps: on a realistic distribution the screen keeps (almost) exactly what the
    reference's sign filter keeps -- the FMA form is not a blunter test."""
    rng = np.random.default_rng(9)
    n = 400_000
    c0 = np.column_stack([rng.uniform(-11, 11, n), np.full(n, 0.2), rng.uniform(-11, 11, n)])
    dc = np.column_stack([np.zeros(n), rng.uniform(0, 0.5, n), np.zeros(n)])
    moving = np.ones(n, dtype=np.int32)
    o = np.column_stack([rng.uniform(-11, 11, n), rng.uniform(0.0, 2.0, n), rng.uniform(-11, 11, n)])
    d = _unit(rng, n)
    fr = rng.uniform(0, 1, n)
    keep, need = tor.selftest_screen(o, d, c0, dc, moving, fr, np.full(n, 0.04))
    assert np.count_nonzero((need != 0) & (keep == 0)) == 0
    extra = np.count_nonzero((keep != 0) & (need == 0))
    assert extra <= 4, extra        # only pairs within rounding distance of the boundary
    assert 50 < np.count_nonzero(keep) < 0.01 * n        # a ray meets few of the small spheres
    keep2, need2 = tor.selftest_screen2(o, d, c0, dc, moving, fr, np.full(n, 0.04), 0)   # the second form is no blunter
    assert np.array_equal(need2, need) and np.count_nonzero((need2 != 0) & (keep2 == 0)) == 0
    assert np.count_nonzero((keep2 != 0) & (need2 == 0)) <= 4
    keep3, need3 = tor.selftest_screen2(o, d, c0, dc * 0.0, moving * 0, fr, np.full(n, 0.04), 1)
    assert np.count_nonzero((need3 != 0) & (keep3 == 0)) == 0 and np.count_nonzero((keep3 != 0) & (need3 == 0)) <= 4


def test_screen_degenerate_rays(tor):
    """Zero, denormal-scale, huge and non-finite rays: the screen keeps, or the reference's own test rejects as well."""
    n = 7
    o = np.zeros((n, 3)); d = np.tile([0.0, 0.0, -1.0], (n, 1))
    c0 = np.tile([0.0, 0.0, -5.0], (n, 1)); dc = np.zeros((n, 3)); moving = np.zeros(n, dtype=np.int32)
    f = np.zeros(n); r2 = np.full(n, 0.25)
    d[0] = 0.0
    d[1] = [0, 0, -1e-160]
    d[2] = [0, 0, -1e150]
    o[3] = [np.inf, 0, 0]
    o[4] = [np.nan, 0, 0]
    moving[5] = 1; f[5] = 1e30; dc[5] = [0, 1, 0]
    o[6] = [0, 0, 1e200]
    keep, need = tor.selftest_screen(o, d, c0, dc, moving, f, r2)
    assert np.all((need == 0) | (keep != 0)), (keep, need)
    assert keep[1] and keep[2] and need[1] and need[2]
    for variant in (0, 1, 2):
        keep2, need2 = tor.selftest_screen2(o, d, c0, dc, moving, f, r2, variant)
        assert np.all((need2 == 0) | (keep2 != 0)), (variant, keep2, need2)
        assert keep2[1] and keep2[2]
    # a camera INSIDE a sphere so far from the origin that |c|^2 overflows (K = +inf on the host, clamped) -- o - c is
    # small, the reference hits it, the ray is wild for the second form (B^2 overflows) and must keep the object, not produce
    # 0 x inf = NaN of whatever sign
    c_far = np.array([[2e154, 0.0, 0.0], [0.0, -3e154, 1e154], [1e155, 1e155, 1e155]])
    o_far = c_far + np.array([[1e138, 0, 0], [0, 2e138, 0], [0, 0, -1e139]])
    d_far = np.array([[1.0, 0.2, 0.1], [0.0, 1.0, 0.0], [0.3, 0.3, 1.0]])
    r2_far = np.full(3, 1e300)                                  # radius 1e150: the camera is deep inside
    z3, zi, zf = np.zeros((3, 3)), np.zeros(3, dtype=np.int32), np.zeros(3)
    keep, need = tor.selftest_screen(o_far, d_far, c_far, z3, zi, zf, r2_far)
    assert np.all(need != 0) and np.all(keep != 0), (keep, need)
    for variant in (0, 1, 2):
        keep2, need2 = tor.selftest_screen2(o_far, d_far, c_far, z3, zi, zf, r2_far, variant)
        assert np.all(need2 != 0) and np.all(keep2 != 0), (variant, keep2, need2)



def test_plane_screen_on_its_own_boundary(tor):
    """Stage one of the common-height segments keeps what lies within R of the ray's GROUND TRACK.  Its own decision boundary:
    rays whose ground track is tangent to the sphere's ground circle while the ray itself passes through the sphere's equator
    (the 3D tangent point IS the 2D one), nudged by ulps either way; near-vertical rays (no ground track: everything is kept);
    rays with a huge / tiny horizontal direction; and what it buys: a thin band on random_scene-like input."""
    rng = np.random.default_rng(21)
    n = 300_000
    c0 = np.column_stack([rng.uniform(-11, 11, n), rng.choice([0.2, 0.7, -3.0], n), rng.uniform(-11, 11, n)])
    r = rng.choice([0.2, 1.0, 7.5], size=n)
    ang = rng.uniform(0, 2 * np.pi, n)
    nrm = np.column_stack([np.cos(ang), np.zeros(n), np.sin(ang)])           # horizontal normal: a point of the equator
    tang = np.column_stack([-np.sin(ang), np.zeros(n), np.cos(ang)])
    up = np.array([0.0, 1.0, 0.0])
    slope = rng.choice([0.0, 1e-3, 0.5, 3.0, 1e3, 1e8], size=(n, 1)) * rng.choice([-1.0, 1.0], size=(n, 1))
    dirn = tang + slope * up                                                # tangent to the sphere at the equator point, any slope
    target = c0 + nrm * r[:, None]
    back = rng.uniform(0.5, 30.0, (n, 1))
    o = target - dirn / np.linalg.norm(dirn, axis=1, keepdims=True) * back
    k = rng.integers(-4, 5, size=(n, 3))
    d = (target - o) * (1.0 + k * 2.0 ** -52) * rng.choice([1.0, 1e-6, 1e6], size=(n, 1))
    z3 = np.zeros((n, 3)); zi = np.zeros(n, dtype=np.int32); zf = np.zeros(n)
    keep, need = tor.selftest_screen2(o, d, c0, z3, zi, zf, r * r, 2)
    assert np.count_nonzero((need != 0) & (keep == 0)) == 0
    assert 0 < np.count_nonzero(need) < n
    # the same as movers along y caught at f (centre = c0 - f dc + f dc): x and z do not move
    fm = rng.uniform(-0.5, 1.5, n)
    dcm = np.column_stack([np.zeros(n), rng.uniform(-0.5, 0.5, n), np.zeros(n)])
    keep, need = tor.selftest_screen2(o, d, c0 - dcm * fm[:, None], dcm, np.ones(n, dtype=np.int32), fm, r * r, 2)
    assert np.count_nonzero((need != 0) & (keep == 0)) == 0
    assert 0 < np.count_nonzero(need) < n
    # round 5: ... and as movers in GENERAL position caught at f (xkind 13): the ground projection of the centre moves, the chain
    # evaluates c0_xz + f dc_xz itself (centre = c0 - f dc + f dc up to the rounding of the products, so the rays sit on the boundary
    # to within a few ulps of the scene size instead of one)
    dcg = np.column_stack([rng.uniform(-0.7, 0.7, n), rng.uniform(-0.5, 0.5, n), rng.uniform(-0.7, 0.7, n)])
    keep, need = tor.selftest_screen2(o, d, c0 - dcg * fm[:, None], dcg, np.ones(n, dtype=np.int32), fm, r * r, 2)
    assert np.count_nonzero((need != 0) & (keep == 0)) == 0
    assert 0 < np.count_nonzero(need) < n
    # near-vertical rays through the sphere: the ground track degenerates
    m = 50_000
    c1 = np.column_stack([rng.uniform(-11, 11, m), np.full(m, 0.2), rng.uniform(-11, 11, m)])
    off = _unit(rng, m) * 0.19
    o1 = c1 + off + np.array([0.0, 5.0, 0.0])
    tilt = rng.choice([0.0, 1e-300, 1e-120, 1e-101, 1e-99, 1e-30, 1e-9], size=(m, 2)) * rng.choice([-1.0, 1.0], size=(m, 2))
    d1 = np.column_stack([tilt[:, 0], -np.ones(m), tilt[:, 1]])
    keep, need = tor.selftest_screen2(o1, d1, c1, np.zeros((m, 3)), np.zeros(m, dtype=np.int32), np.zeros(m), np.full(m, 0.04), 2)
    assert np.count_nonzero((need != 0) & (keep == 0)) == 0 and np.count_nonzero(need) > m // 2
    # what it buys: random rays over a field of small spheres keep a thin band, a superset of the second form's candidates
    c2 = np.column_stack([rng.uniform(-11, 11, n), np.full(n, 0.2), rng.uniform(-11, 11, n)])
    o2 = np.column_stack([rng.uniform(-11, 11, n), rng.uniform(0.0, 2.0, n), rng.uniform(-11, 11, n)])
    d2 = _unit(rng, n)
    keep_p, need_p = tor.selftest_screen2(o2, d2, c2, z3, zi, zf, np.full(n, 0.04), 2)
    keep_2, need_2 = tor.selftest_screen2(o2, d2, c2, z3, zi, zf, np.full(n, 0.04), 1)
    assert np.count_nonzero((need_p != 0) & (keep_p == 0)) == 0
    assert np.count_nonzero(keep_p) < 0.05 * n                      # band of width 0.4 across a field 22 wide
    assert np.count_nonzero(keep_2) < np.count_nonzero(keep_p)      # the second form is the finer test


def test_strict_layout_walk_on_the_host(tor):
    """The strict brute-force layout as tor_scene_upload builds it (common-height segments, their {cx, cz} plane table, the
    segment's largest radius, padding) walked on the host the way the ARITH 2 object loop walks it: plane screen first, second
    form on what it keeps.  For every ray x object: what the reference's test needs (tor_selftest_screen_host's `need`) is a
    candidate; objects end up on the segments they belong to; the plane screen is the coarser test."""
    import importlib
    rng = np.random.default_rng(31)
    base = tor.random_scene(0xFACADE)
    recs = base.to_records()
    # second scene: three heights, two time groups, general movers, radii mixed inside a height, group sizes below / above 8
    extra = []
    def add(n, y, mover, t0=0.0, t1=1.0):
        for i in range(n):
            x, z = rng.uniform(-9, 9, 2)
            r = float(rng.choice([0.1, 0.2, 0.4]))
            if mover == 0:
                extra.append([0, x, y, z, x, y, z, 0, 1, r, 0, .5, .5, .5, 0, 0])
            elif mover == 1:
                extra.append([1, x, y, z, x, y + rng.uniform(0, .5), z, t0, t1, r, 0, .5, .5, .5, 0, 0])
            else:
                extra.append([1, x, y, z, x + .3, y, z - .2, t0, t1, r, 0, .5, .5, .5, 0, 0])
    add(41, 0.2, 0); add(7, 0.9, 0); add(19, 0.4, 0); add(53, 0.2, 1); add(12, 0.4, 1, 0.25, 0.75); add(5, 0.3, 1); add(9, 0.2, 2)
    extra.append([0, 0, -1000, 0, 0, -1000, 0, 0, 1, 1000, 0, .5, .5, .5, 0, 0])
    n_sane = len(extra)
    add(3, 0.5, 1, 0.5, 0.5)          # time0 == time1: the time fraction is never finite -> no table (xkind 0), the first form alone
    mixed = np.asarray(extra, dtype=np.float64)
    mixed = mixed[rng.permutation(len(mixed))]
    # third scene (round 5): nothing in common -- a 3-D cloud, every height and radius different, a third static, a third moving
    # along y, a third moving anywhere
    cloud = [[0, 0, -1000, 0, 0, -1000, 0, 0, 1, 1000, 0, .5, .5, .5, 0, 0]]
    for i in range(240):
        x, z = rng.uniform(-8, 8, 2); y = rng.uniform(0.2, 6.0); r = rng.uniform(0.12, 0.3)
        if i % 3 == 0: cloud.append([0, x, y, z, x, y, z, 0, 1, r, 0, .5, .5, .5, 0, 0])
        elif i % 3 == 1: cloud.append([1, x, y, z, x, y + rng.uniform(0, .5), z, 0.0, 1.0, r, 0, .5, .5, .5, 0, 0])
        else: cloud.append([1, x, y, z, x + rng.uniform(-.4, .4), y + rng.uniform(-.3, .3), z + rng.uniform(-.4, .4), 0.0, 1.0, r, 0, .5, .5, .5, 0, 0])
    cloud = np.asarray(cloud, dtype=np.float64)
    # (round 5: statics resting at the common height of a segment of movers along y join it when their radii are of one class --
    # random_scene's 86 small statics are xkind 12 now; which of `mixed`'s static groups join depends on the radii `add` drew, so
    # its set of kinds is asserted loosely below)
    for recs_k, want_kinds in ((recs, {10, 12}), (mixed, None), (cloud, {10, 13, 14})):   # (mixed: the 5 movers along y at 0.3 are below a segment's 8 -> xkind 14)
        scene = tor.Scene.from_records(recs_k)
        n_obj = len(recs_k)
        n_rays = 400
        # rays that start on or near objects (scattered rays) and camera-like rays
        pick = rng.integers(0, n_obj, n_rays)
        c = recs_k[pick, 1:4]
        o = c + _unit(rng, n_rays) * (np.abs(recs_k[pick, 9])[:, None] * rng.choice([1.0, 1.0, 3.0, 40.0], size=(n_rays, 1)))
        d = _unit(rng, n_rays) * rng.choice([1.0, 1e-3, 1e3], size=(n_rays, 1))
        aim = rng.random(n_rays) < 0.5                      # half of them aimed at some other object
        tgt = recs_k[rng.integers(0, n_obj, n_rays), 1:4] + rng.normal(0, 0.1, (n_rays, 3))
        d[aim] = (tgt - o)[aim]
        d[:4] = [[0, -1, 0], [0, 1, 0], [1e-200, -1, 0], [0, -1, 1e-40]]   # vertical: no ground track
        t = rng.uniform(-0.2, 1.2, n_rays)
        keep, kind, pays = tor.debug_screen2_scene(scene.list(), o, d, t, max_segs=16)
        if want_kinds is not None:
            assert set(np.unique(kind)) == want_kinds, np.unique(kind)
        else:
            assert {0, 10, 12, 13, 14} <= set(np.unique(kind)) <= {0, 10, 11, 12, 13, 14}, np.unique(kind)
        is_static = recs_k[:, 0] == 0
        y_mover = (~is_static) & (recs_k[:, 4] == recs_k[:, 1]) & (recs_k[:, 6] == recs_k[:, 3])
        degenerate = (~is_static) & (recs_k[:, 7] == recs_k[:, 8])
        assert np.all(kind[degenerate] == 0) and np.all(kind[~degenerate] >= 10)
        assert np.all(kind[~is_static & ~y_mover & ~degenerate] == 13) and np.all(np.isin(kind[is_static], (10, 11, 12)))
        assert np.all(np.isin(kind[y_mover & ~degenerate], (12, 14)))
        assert np.all((keep == 3) == (kind == 0)[None, :])
        assert np.all(pays[4:, 0] >= 0)                                # a vote per ray and segment
        assert np.all(pays[:4][pays[:4] >= 0] == 0)                    # vertical rays keep everything: they vote against stage one
        # need: the reference's own test per pair
        R, O = np.meshgrid(np.arange(n_rays), np.arange(n_obj), indexing="ij")
        R, O = R.ravel(), O.ravel()
        c0 = recs_k[O, 1:4]; dc = recs_k[O, 4:7] - c0
        mv = (recs_k[O, 0] != 0).astype(np.int32)
        with np.errstate(divide="ignore", invalid="ignore"):      # (the degenerate time group: time0 == time1)
            f = np.where(mv != 0, (t[R] - recs_k[O, 7]) / (recs_k[O, 8] - recs_k[O, 7]), 0.0)
        _, need = tor.selftest_screen(o[R], d[R], c0, dc, mv, f, recs_k[O, 9] ** 2)
        need = need.reshape(n_rays, n_obj) != 0
        on_second_form = keep != 3
        assert np.count_nonzero(need & on_second_form) > 100
        assert np.all(keep[need & on_second_form] == 2), np.argwhere(need & on_second_form & (keep != 2))[:5]
        small = np.abs(recs_k[:, 9]) < 5.0                             # (the ground sphere's band is the whole scene)
        plane_segments = (np.isin(kind, (10, 11, 12, 13, 14)) & small)[None, :] & on_second_form
        assert np.count_nonzero(keep[plane_segments] == 0) > 0.8 * np.count_nonzero(plane_segments)   # most pairs end at stage one
        assert np.all(keep[:4][plane_segments[:4]] >= 1)            # vertical rays: the plane screen keeps everything


def test_layout_segments_largest_first_and_tails_padded_to_words(tor):
    """tor_scene.cpp, round 5: segments are laid out largest first (the largest starts on a word boundary of the sorted list); a
    plane-screened segment (>= 48 objects) whose tail would be three or four blocks of 8 is padded to the next word of 32 slots,
    unless that costs the list one more 512-slot pass; everything else is padded to blocks of 8."""
    segs = tor.debug_layout_segments(tor.random_scene(0xFACADE).list())
    assert segs == [(12, 481, 488, 0), (10, 4, 8, 488)]          # 15 whole words + one block, then {ground, three big spheres}
    rng = np.random.default_rng(3)
    def scene(sizes):   # statics at distinct common heights, one segment per height; radii alike
        recs = []
        for k, n in enumerate(sizes):
            for _ in range(n):
                x, z = rng.uniform(-9, 9, 2)
                recs.append([0, x, 0.2 + 0.01 * k, z, x, 0.2 + 0.01 * k, z, 0, 1, 0.2, 0, .5, .5, .5, 0, 0])
        return tor.Scene.from_records(np.asarray(recs, dtype=np.float64)).list()
    segs = tor.debug_layout_segments(scene([50, 83, 36, 9, 100]))
    assert [s[1] for s in segs] == [100, 83, 50, 36, 9]                 # largest first
    by_n = {s[1]: s for s in segs}
    assert by_n[100][2] == 104      # tail of 4 (+4 padding): one block, not padded to a word
    assert by_n[83][2] == 96        # 88 slots, tail of 24 -> a whole word
    assert by_n[50][2] == 64        # 56 slots, tail of 24 -> a whole word
    assert by_n[36][2] == 40 and by_n[9][2] == 16   # below the plane screen's 48 objects: blocks only
    first = 0
    for s in segs:
        assert s[3] == first and s[2] % 8 == 0
        first += s[2]
    # padding must not cost a pass: 9 segments of 50 are 504 slots in blocks, 576 in words -> stay in blocks
    segs = tor.debug_layout_segments(scene([50] * 9))
    assert all(s[2] == 56 for s in segs) and sum(s[2] for s in segs) == 504


def test_resting_static_in_a_mover_segment_at_an_overflowing_time_fraction(tor):
    """statics that rest at the common height of a segment of movers along y join it with dcy = 0.  A finite time
    fraction whose SQUARE overflows (|f| > 1.3e154: shutter_close ~ 1e155) made f2n = sigma^2 f^2 = inf meet dcy = 0 as NaN in
    screen2_movy_y, and the NaN's clear sign bit dropped a sphere the reference hits (travel |f| stays small: B is not wild).
    The ray is wild for such an f now -- it keeps everything.  (This host build cannot fail on the old code: x86's default NaN has
    its sign bit SET and happened to read as "keep", gfx950's is positive and read as "drop" -- the test that fails without the fix
    is tests/test_gpu_round5.py::test_non_finite_ray_time_hits_statics_only with shutter_close = 1e155; this one pins the host side.)"""
    rng = np.random.default_rng(77)
    n = 20000
    c0 = rng.uniform(-10, 10, size=(n, 3))
    r = rng.uniform(0.1, 0.5, n)
    o = rng.uniform(-15, 15, size=(n, 3))
    target = c0 + rng.normal(size=(n, 3)) * (0.6 * r)[:, None]      # most rays hit
    d = target - o
    dc = np.zeros((n, 3))
    moving = np.ones(n, dtype=np.int32)
    for fval in (1e155, -3e154, 1.4e154, 1.2e154, 1e300):
        f = np.full(n, fval)
        keep, need = tor.selftest_screen2(o, d, c0, dc, moving, f, r * r, 0)
        assert np.count_nonzero(need) > n // 2
        missed = np.flatnonzero((need != 0) & (keep == 0))
        assert missed.size == 0, (fval, missed[:5])
