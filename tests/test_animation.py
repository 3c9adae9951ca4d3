"""SURVEY 8 f2 / BASELINE config 5: the animated bouncing-spheres scene
(trace_of_radiance/scenes_animated.nim).  The product's host mirror and the oracle restate the
generator, the physics step and the `scenes` iterator independently; every frame's camera and
object list must agree bit for bit, and GPU frames must equal oracle frames.

Parity for this generator is UNPINNED against the reference itself: it holds no rendered frame, and
its `rng.random(float32)` does not compile at HEAD (both sides read it as float32(uniform(float64)))."""
import numpy as np
import pytest


def test_animation_host_mirror_matches_oracle(tor, oracle):
    anim = tor.Animation(216, 384, 0.005, 0.0, 2.0)
    assert anim.n_objects == 1601
    prod = list(anim.scenes(6))
    orc = list(oracle.animation_scenes(216, 384, 0.005, 0.0, 2.0, 6))
    assert len(prod) == len(orc) == 67                     # (2.0 / 0.005) / 6 steps, float32 clock
    for (cam, scene, t), (ocam, oobjs, ot) in zip(prod, orc):
        assert t == ot
        assert np.array_equal(cam.as_array(), ocam)
        assert np.array_equal(scene.to_records(), oobjs)
    # physics sanity: spheres never sink below the ground plane, the camera orbits clockwise
    ys = np.array([s.to_records()[1:-3, 2] for _, s, _ in prod])
    assert ys.min() >= 0.0 and ys.max() < 0.2 + 12.0 ** 2 / (2 * 9.80665) + 0.1  # apex of v0 <= 12 m/s
    ang = [np.arctan2(c.as_array()[2], c.as_array()[0]) for c, _, _ in prod]
    assert all(b < a for a, b in zip(ang, ang[1:]))
    # t_min skips frames without rendering them: same state as stepping through
    late = list(tor.Animation(216, 384, 0.005, 0.99, 2.0).scenes(6))
    olate = list(oracle.animation_scenes(216, 384, 0.005, 0.99, 2.0, 6))
    assert len(late) == len(olate) and np.array_equal(late[0][1].to_records(), olate[0][1])


def test_frame_parallel_split_covers_every_frame(tor):
    for n, world in [(240, 8), (67, 3), (5, 8)]:
        seen = sorted(f for r in range(world) for f in tor.frames_of_rank(n, r, world))
        assert seen == list(range(n))


@pytest.mark.gpu
def test_animation_frames_match_oracle_on_gpu(tor, oracle):
    """~1600 static spheres per frame, shutter [0,0]: frames 0, 1 and 20 at 48x27."""
    h, w, spp = 27, 48, 8
    prod = list(tor.Animation(h, w, 0.005, 0.0, 0.7).scenes(6))
    orc = list(oracle.animation_scenes(h, w, 0.005, 0.0, 0.7, 6))
    ctx_frames = [0, 1, 20]
    for f in ctx_frames:
        cam, scene, _ = prod[f]
        ocam, oobjs, _ = orc[f]
        for seeding in (tor.SEED_PIXEL, tor.SEED_SAMPLE):
            cv = tor.new_canvas(h, w, spp, 2.2)
            tor.render(cv, cam, scene.list(), 50, tor.make_options(seeding=seeding))
            want = oracle.render(h, w, spp, ocam, oobjs, seeding=seeding, math=1, arith=0, accum=seeding).pixels
            err = float(np.max(np.abs(cv.pixels - want)))
            assert err <= 1e-5 and np.array_equal(cv.pixels, want), (f, seeding, err)
