"""TOR_ACCEL_BLOCKS is exact only if its boxes are conservative.  This CPU test (no GPU) takes the layout the
library builds and checks the invariant directly: every object sits in exactly one place (a spatial slot or
the always-list), every spatial object's swept sphere over the ray-time range lies strictly inside its block's
box, and every block box inside its super box."""
import numpy as np
import pytest


def _scene(tor, rng, n, spread):
    recs = [[0, 0, -500, 0, 0, -500, 0, 0, 1, 500, 0, .5, .5, .5, 0, 0]]
    groups = [(0.0, 1.0), (-0.5, 0.5), (0.25, 2.0), (1.0, 0.0)]
    while len(recs) < n:
        x, y, z = rng.uniform(-spread, spread), rng.uniform(0, 0.3 * spread), rng.uniform(-spread, spread)
        r = float(rng.choice([0.15, 0.2, 0.3, 0.45])) * (1 if rng.random() > 0.05 else -1)
        if rng.random() < 0.4:
            recs.append([0, x, y, z, x, y, z, 0, 1, r, 0, .5, .5, .5, 0, 0])
        else:
            t0, t1 = groups[int(rng.integers(0, 4))]
            d = rng.uniform(-0.6, 0.6, 3)
            recs.append([1, x, y, z, x + d[0], y + d[1], z + d[2], t0, t1, r, 1, .5, .5, .5, 0.1, 0])
    recs.append([1, 1, 1, 1, 2, 2, 2, 0.5, 0.5, 0.3, 0, .1, .9, .1, 0, 0])   # time0 == time1 -> always-list
    return np.asarray(recs, dtype=np.float64)


@pytest.mark.parametrize("n,spread,shutter", [(120, 4.0, (0.0, 1.0)), (485, 11.0, (0.0, 1.0)), (700, 8.0, (-2.0, 3.0)),
                                              (1500, 20.0, (0.0, 0.0)), (2100, 15.0, (0.3, 0.1))])
def test_boxes_are_conservative(tor, n, spread, shutter):
    rng = np.random.default_rng(n)
    recs = _scene(tor, rng, n, spread)
    scene = tor.Scene.from_records(recs)
    t_lo, t_hi = min(0.0, *shutter), max(0.0, *shutter)
    lay = tor.debug_accel_layout(scene.list(), t_lo, t_hi)
    assert lay is not None
    slots, boxes, supers, two_level = lay
    # a culling box stands for F consecutive blocks (csrc/tor_kernels.hpp kBoxFanout = 1 since the round-4 sweep; the view
    # reports every block with the box that covers it, so the check below holds for any F the library is built with)
    F = 1
    n_boxes = (slots.shape[0] + F - 1) // F
    assert two_level == (n_boxes > 96)
    placed = slots[slots >= 0]
    assert len(set(placed.tolist())) == len(placed)                       # no object twice
    always = sorted(set(range(len(recs))) - set(placed.tolist()))
    big_or_degenerate = [i for i in range(len(recs)) if abs(recs[i, 9]) > 2.5 * np.median(np.abs(recs[:, 9]))
                         or (recs[i, 0] == 1 and recs[i, 8] == recs[i, 7])]
    assert always == sorted(big_or_degenerate)                            # only those stay brute force
    ts = np.linspace(t_lo, t_hi, 7)
    for b in range(slots.shape[0]):
        lo, hi = boxes[b, :3], boxes[b, 3:]
        for i in slots[b][slots[b] >= 0]:
            r = recs[i]
            c0, c1, rad = r[1:4], r[4:7], abs(r[9])
            for t in ts:
                f = (t - r[7]) / (r[8] - r[7]) if r[0] == 1 else 0.0
                c = c0 + f * (c1 - c0)
                assert np.all(c - rad > lo) and np.all(c + rad < hi), (b, i, t)
        s = supers[b // 8]
        assert np.all(s[:3] <= lo) and np.all(s[3:] >= hi)
    # padding BOXES are NaN boxes (never entered; an inverted box would be); a padding block behind a real box holds never-hit records
    assert np.all(np.isnan(boxes[n_boxes * F:]))
    for b in range(0, slots.shape[0] - slots.shape[0] % F, F):              # the F blocks of a box share it
        assert all(np.array_equal(boxes[b], boxes[b + k]) for k in range(F))


def test_small_scenes_have_no_second_level(tor):
    scene = tor.Scene.from_records(_scene(tor, np.random.default_rng(1), 40, 3.0))
    assert tor.debug_accel_layout(scene.list(), 0.0, 1.0) is None
    # a non-finite shutter cannot be bounded: the library falls back to brute force
    scene = tor.Scene.from_records(_scene(tor, np.random.default_rng(2), 200, 3.0))
    assert tor.debug_accel_layout(scene.list(), 0.0, float("inf")) is None


def test_surface_area_build_on_degenerate_and_ordered_inputs(tor):
    """Round 6: the top-down surface-area build (tor_scene.cpp build_accel).  Identical spheres (every split costs the same: ties go
    to the most balanced one), spheres on a line in sorted order, and a big random set: built in well under a second each, every
    object placed exactly once, blocks tighter than the Morton order of rounds 1-5 (TOR_ACCEL_ORDER=morton)."""
    import os
    import time
    rng = np.random.default_rng(3)

    def recs_of(c):
        n = len(c)
        r = np.zeros((n, 16))
        r[:, 1:4] = c
        r[:, 4:7] = c
        r[:, 8] = 1
        r[:, 9] = 0.2
        r[:, 11:14] = 0.5
        return r

    def area(boxes):
        d = boxes[:, 3:] - boxes[:, :3]
        return float(np.nansum(d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 0] * d[:, 2]))

    tor.debug_accel_layout(tor.Scene.from_records(recs_of(rng.uniform(-5, 5, (64, 3)))).list(), 0.0, 1.0)   # (loads the library)
    cases = {"identical": np.tile(np.array([[1.0, 2.0, 3.0]]), (20000, 1)),
             "on a line, sorted": np.stack([np.arange(20000) * 0.5, np.zeros(20000), np.zeros(20000)], axis=1),
             "random cloud": rng.uniform(-50, 50, (20000, 3))}
    for name, c in cases.items():
        scene = tor.Scene.from_records(recs_of(c))
        t = time.time()
        slots, boxes, supers, two_level = tor.debug_accel_layout(scene.list(), 0.0, 1.0)
        dt = time.time() - t
        assert dt < 2.0, (name, dt)
        placed = slots[slots >= 0]
        assert len(placed) == len(c) and len(set(placed.tolist())) == len(c), name
        assert two_level
        os.environ["TOR_ACCEL_ORDER"] = "morton"
        try:
            _, mboxes, msupers, _ = tor.debug_accel_layout(scene.list(), 0.0, 1.0)
        finally:
            os.environ.pop("TOR_ACCEL_ORDER", None)
        n_blocks = slots.shape[0]
        assert area(boxes[:n_blocks]) <= area(mboxes[:n_blocks]) * 1.0001, name
        if name == "random cloud":
            assert area(boxes[:n_blocks]) < 0.6 * area(mboxes[:n_blocks])
            assert area(supers[:(n_blocks + 7) // 8]) < 0.6 * area(msupers[:(n_blocks + 7) // 8])
