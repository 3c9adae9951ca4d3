"""GPU tests added in round 6 (VERDICT r5 / ADVICE r5):
 * the top-down surface-area build behind TOR_ACCEL_BLOCKS (tor_scene.cpp build_accel): same canvas as the Morton order of rounds 1-5
   and as the brute force (== the oracle), fewer blocks entered per query;
 * TorStats.block_tests / exact_tests: what the float32 block filter looks at is counted apart from what reaches the reference's test;
 * the SEED_PIXEL cost probe counts the same closest-hit queries whatever layout it walks (the round-3 test of this
   went away with TOR_PROBE_ACCEL although the probe now ALWAYS runs on the culling layout)."""
import numpy as np
import pytest

from test_gpu_round3 import _render_with_env
from test_gpu_round4 import _env, _exact

pytestmark = pytest.mark.gpu


def _stats(tor, scene, cam, h, w, spp, accel, seeding, **env):
    import torch
    with _env(**env):
        ctx = tor.Context(0)
        ctx.upload(scene.list())
        ctx.set_stats(True)
        buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
        ctx.render_device(cam, h, w, spp, 2.2, 50, tor.make_options(seeding=seeding, accel=accel, pixel_kernel=tor.PIXEL_KERNEL_LANE), buf.data_ptr(),
                          torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        st = ctx.last_stats()
        ctx.close()
    return st, buf


def test_surface_area_order_never_changes_a_pixel_and_enters_fewer_blocks(tor, oracle, ref_scene, ref_camera):
    import torch
    objs, _ = ref_scene
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    anim = tor.Animation(108, 192)
    a_cam, a_scene, _ = next(iter(anim.scenes(skip=37)))
    h, w, spp = 108, 192, 16
    for name, sc, cm in (("random_scene", scene, cam), ("animation frame (two-level)", a_scene, a_cam)):
        for seeding in (tor.SEED_SAMPLE, tor.SEED_PIXEL):
            brute, _ = _render_with_env(tor, sc, cm, h, w, spp, {}, seeding=seeding, accel=0)
            per_order = {}
            for order in ("sah", "morton"):
                for accel in (1, 3):
                    got, _ = _render_with_env(tor, sc, cm, h, w, spp, {"TOR_ACCEL_ORDER": order}, seeding=seeding, accel=accel)
                    assert torch.equal(got, brute), (name, seeding, order, accel)
                st, _ = _stats(tor, sc, cm, h, w, spp, 3, seeding, TOR_ACCEL_ORDER=order)
                per_order[order] = st
            s, m = per_order["sah"], per_order["morton"]
            assert s.hit_queries == m.hit_queries and s.samples == m.samples
            assert s.block_tests < 0.8 * m.block_tests, (name, s.block_tests, m.block_tests)      # measured: 0.61 / 0.49
            assert s.exact_tests <= m.exact_tests * 1.001                                        # the filter sees the same objects or fewer
    # ... and the brute force that arbitrates above is the oracle's canvas
    for seeding in (0, 1):
        want = oracle.render(54, 96, 8, ref_camera, objs, seeding=seeding, math=1, arith=0, accum=seeding).pixels
        for order in ("sah", "morton"):
            got, _ = _render_with_env(tor, scene, cam, 54, 96, 8, {"TOR_ACCEL_ORDER": order}, seeding=seeding, accel=3)
            _exact(got.cpu().numpy(), want)


def test_stats_count_filter_tests_apart_from_exact_tests(tor):
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    h, w, spp = 108, 192, 16
    for seeding in (tor.SEED_SAMPLE, tor.SEED_PIXEL):
        per = {accel: _stats(tor, scene, cam, h, w, spp, accel, seeding)[0] for accel in (0, 1, 2, 3)}
        q = per[0].hit_queries
        assert all(st.hit_queries == q and st.samples == h * w * spp for st in per.values())
        for accel, st in per.items():
            assert st.exact_tests + st.block_tests == st.candidates
        assert per[0].block_tests == 0 and per[2].block_tests == 0
        assert per[3].block_tests % 8 == 0 and per[3].block_tests > 0
        # what reaches the reference's own test (spheres.nim:28-49): ~1.4-1.6 objects per query behind either screen
        assert per[0].exact_tests / q < 2.0 and per[3].exact_tests / q < 2.0
        # the float32 pre-filter over every object (accel 2) and behind the boxes (accel 3) is the same test with margins taken per
        # segment / per spatial set: nearly the same survivors
        assert abs(per[3].exact_tests - per[2].exact_tests) < 0.05 * per[2].exact_tests


def test_cost_probe_counts_the_same_queries_on_every_layout(tor):
    """The probe (integrate_kernel SEEDING 2: per-sample streams, 2 spp) only COUNTS closest-hit queries per pixel; the count does
    not depend on how a hit is found, so it runs with both exact accelerations whatever the frame launch uses.  Same counts behind
    a brute-force frame, an accelerated frame, either object order -- and their sum is the kernel's own query counter of a 2-spp
    sample-stream launch."""
    import torch
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    h, w, spp = 135, 240, 64
    costs = {}
    for key, accel, env in (("brute frame", 0, {}), ("accelerated frame", 3, {}), ("accelerated frame, morton", 3, {"TOR_ACCEL_ORDER": "morton"})):
        with _env(**env):
            ctx = tor.Context(0)
            ctx.upload(scene.list())
            buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
            ctx.render_device(cam, h, w, spp, 2.2, 50, tor.make_options(seeding=tor.SEED_PIXEL, accel=accel, pixel_kernel=tor.PIXEL_KERNEL_LANE),
                              buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            costs[key] = ctx.last_pixel_cost(h * w).copy()
            ctx.close()
    base = costs["brute frame"]
    assert base.size == h * w and base.min() >= 2          # two samples, at least one query each
    for key, c in costs.items():
        assert np.array_equal(c, base), key
    st, _ = _stats(tor, scene, cam, h, w, 2, 0, tor.SEED_SAMPLE)
    assert int(base.sum()) == st.hit_queries


def _dense_cluster(n_small, n_big, jitter=0.01, movers=False, seed=7):
    """Spheres that nearly coincide: every ray towards the cluster keeps EVERY object of every block it enters (8 survivors per
    (ray, block) pair) and every big sphere (direct survivors) -- the worst case for the pooled lists of the cooperative resolve."""
    rng = np.random.default_rng(seed)
    recs = []
    for k in range(n_big):   # > 2.5 x the median radius: outside the blocks, tested directly; BEHIND the cluster as the camera sees it
        c = rng.uniform(-jitter, jitter, 3) + (0.0, 0.0, -3.0)
        recs.append([0, *c, *c, 0, 1, 1.0 + 0.01 * k, 2 if k % 2 else 0, .6, .7, .8, 0, 1.5])
    for k in range(n_small):
        c = rng.uniform(-jitter, jitter, 3)
        mat = int(rng.integers(0, 3))
        if movers and k % 3 == 0:
            recs.append([1, *c, *(c + (0, 0.05, 0)), 0, 1, 0.2 + 0.0005 * k, mat, .8, .5, .3, 0.1, 1.5])
        else:
            recs.append([0, *c, *c, 0, 1, 0.2 + 0.0005 * k, mat, .8, .5, .3, 0.1, 1.5])
    return np.asarray(recs, dtype=np.float64)


def test_pooled_lists_hold_the_worst_case(tor, oracle):
    """Cooperative resolve (integrate_resolve_coop.inc): the survivor list takes up to 8 entries per lane from ONE block-expansion trip
    on top of what the list-building step left there.  Scenes of nearly coincident spheres fill both at once: 8 direct survivors and
    8 entered blocks per ray, 8 kept objects per (ray, block) pair.  Every accel mode must still equal the brute force, and the brute
    force the oracle."""
    import torch
    cam = tor.camera(look_from=(0.3, 0.2, 5.0), look_at=(0, 0, 0), vertical_field_of_view=12.0, aperture=0.0, focus_distance=5.0)
    ocam = np.frombuffer(bytes(cam), dtype=np.float64).copy()
    h, w, spp, depth = 24, 32, 8, 6
    for name, recs in (("8 big + 64 small", _dense_cluster(64, 8)),
                       ("4 big + 72 small, movers", _dense_cluster(72, 4, movers=True)),
                       ("8 big + 200 small", _dense_cluster(200, 8)),
                       ("40 big + 900 small (two levels)", _dense_cluster(900, 40, jitter=0.05))):
        scene = tor.Scene.from_records(recs)
        for seeding in (tor.SEED_SAMPLE, tor.SEED_PIXEL):
            brute, _ = _render_with_env(tor, scene, cam, h, w, spp, {}, depth=depth, seeding=seeding, accel=0, pixel_kernel=tor.PIXEL_KERNEL_LANE)
            want = oracle.render(h, w, spp, ocam, recs, max_depth=depth, seeding=seeding, math=oracle.MATH_PORTABLE, accum=seeding).pixels
            _exact(brute.cpu().numpy(), want)
            for accel in (1, 2, 3):
                got, _ = _render_with_env(tor, scene, cam, h, w, spp, {}, depth=depth, seeding=seeding, accel=accel, pixel_kernel=tor.PIXEL_KERNEL_LANE)
                assert torch.equal(got, brute), (name, seeding, accel, int((got != brute).sum().item()))
