"""GPU tests added in round 5:
 * stage one of the FMA screen (the plane screen) on EVERY segment kind -- statics and movers at any height, movers in general
   position, degenerate time groups, tiny segments, a wall seen edge-on (the gate's vote) -- gated (default), forced on every
   segment (TOR_PLANE=2) and off (TOR_PLANE=0): same canvas bit for bit, same candidates in the resolve pass, == the oracle;
 * one row of BASELINE configs[3] (3840x2160x4096 spp) in the per-sample streams against the PNG-pinned LIBM / SEQUENTIAL oracle mode;
 * TOR_ARITH_FUSED is gone: every entry point refuses it with the reason."""
import os

import numpy as np
import pytest

from test_gpu_round3 import _render_with_env, _screen_scenes
from test_gpu_round4 import _env, _exact, _many_heights_scene

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _cloud_scene(tor, rng, n=420, degenerate=True):
    """A 3-D cloud: no common height, no common radius; a third static (xkind 10), a third moving along y (14), a third moving
    anywhere (13); two time groups; optionally a few movers with time0 == time1 (their time fraction is never finite: no table)."""
    recs = [[0, 0, -1000, 0, 0, -1000, 0, 0, 1, 1000, 0, .5, .5, .5, 0, 0]]
    for i in range(n):
        x, z = rng.uniform(-8, 8, 2)
        y = rng.uniform(0.2, 6.0)
        r = rng.uniform(0.12, 0.3) * (-1.0 if i % 29 == 0 else 1.0)     # a few hollow spheres
        mat = i % 3
        t0, t1 = ((0.0, 1.0), (0.25, 0.75))[(i // 3) % 2]
        if i % 3 == 0:
            recs.append([0, x, y, z, x, y, z, 0, 1, r, mat, .6, .5, .4, 0.2, 1.5])
        elif i % 3 == 1:
            recs.append([1, x, y, z, x, y + rng.uniform(0, .5), z, t0, t1, r, mat, .3, .7, .4, 0.1, 1.5])
        else:
            recs.append([1, x, y, z, x + rng.uniform(-.4, .4), y + rng.uniform(-.3, .3), z + rng.uniform(-.4, .4), t0, t1, r, mat, .3, .3, .8, 0.0, 1.4])
    if degenerate:
        for i in range(5):
            x, z = rng.uniform(-3, 3, 2)
            recs.append([1, x, 1.0, z, x + .2, 1.3, z, 0.5, 0.5, 0.3, 0, .8, .2, .2, 0, 0])
    recs.append([0, 0, 2, 0, 0, 2, 0, 0, 1, 1.5, 2, 0, 0, 0, 0, 1.5])
    order = rng.permutation(len(recs))
    return tor.Scene.from_records(np.asarray(recs, dtype=np.float64)[order])


def _wall_scene(tor, n=20):
    """A vertical wall of spheres in the plane z = 0, looked at along x: every centre lies on the ground track of most rays --
    stage one would keep whole rows -- so the segments (one per row: common height) carry gate 2 and the waves vote."""
    recs = [[0, 0, -1000, 0, 0, -1000, 0, 0, 1, 1000, 0, .5, .5, .5, 0, 0]]
    for i in range(3 * n):
        for j in range(8):
            recs.append([0, 0.5 * i - 15, 0.25 + 0.5 * j, 0.0, 0, 0, 0, 0, 1, 0.22, (i + j) % 3, .6, .5, .4, 0.1, 1.5])
    return tor.Scene.from_records(np.asarray(recs, dtype=np.float64))


def _ragged_segments_scene(tor, rng):
    """Segment sizes that exercise the layout's tails (tor_scene.cpp, round 5): 50 statics at one height (tail of 18 -> padded to
    a word of stage one), 83 movers along y at another (88 slots, tail of 24 -> padded to 96), 36 movers in general position (below
    the plane screen's 48: blocks only), 9 statics at a third height (one whole block + a last block with ONE real slot: the
    half-block path), the ground and two big spheres (a half-real only block)."""
    recs = [[0, 0, -1000, 0, 0, -1000, 0, 0, 1, 1000, 0, .5, .5, .5, 0, 0],
            [0, 0, 1, 0, 0, 1, 0, 0, 1, 1.0, 2, 0, 0, 0, 0, 1.5], [0, 4, 1, 0, 4, 1, 0, 0, 1, 1.0, 1, .7, .6, .5, 0.0, 0]]
    def xz():
        return rng.uniform(-9, 9, 2)
    for i in range(50):
        x, z = xz(); recs.append([0, x, 0.2, z, x, 0.2, z, 0, 1, 0.2, i % 3, .6, .5, .4, 0.2, 1.5])
    for i in range(83):
        x, z = xz(); recs.append([1, x, 0.25, z, x, 0.25 + rng.uniform(0, .5), z, 0.0, 1.0, 0.25, i % 3, .3, .7, .4, 0.1, 1.5])
    for i in range(36):
        x, z = xz(); y = rng.uniform(0.3, 3.0)
        recs.append([1, x, y, z, x + rng.uniform(-.4, .4), y + rng.uniform(-.3, .3), z + rng.uniform(-.4, .4), 0.0, 1.0, 0.2, i % 3, .3, .3, .8, 0.0, 1.4])
    for i in range(9):
        x, z = xz(); recs.append([0, x, 0.3, z, x, 0.3, z, 0, 1, 0.3, i % 3, .5, .5, .7, 0.3, 1.5])
    order = rng.permutation(len(recs))
    return tor.Scene.from_records(np.asarray(recs, dtype=np.float64)[order])


def _big_ragged_scene(tor, rng):
    """More than 512 slots of the sorted list -- the word queue holds 16 words, so the object loop takes two passes -- with ragged
    tails in every family: 301 statics at one height, 151 movers along y at another, 91 movers in general position, 53 statics
    at heights of their own, 9 statics at a third common height, the ground and two big spheres."""
    recs = [[0, 0, -1000, 0, 0, -1000, 0, 0, 1, 1000, 0, .5, .5, .5, 0, 0],
            [0, 0, 1, 0, 0, 1, 0, 0, 1, 1.0, 2, 0, 0, 0, 0, 1.5], [0, 4, 1, 0, 4, 1, 0, 0, 1, 1.0, 1, .7, .6, .5, 0.0, 0]]
    def xz():
        return rng.uniform(-11, 11, 2)
    for i in range(301):
        x, z = xz(); recs.append([0, x, 0.2, z, x, 0.2, z, 0, 1, 0.2, i % 3, .6, .5, .4, 0.2, 1.5])
    for i in range(151):
        x, z = xz(); recs.append([1, x, 0.25, z, x, 0.25 + rng.uniform(0, .5), z, 0.0, 1.0, 0.25, i % 3, .3, .7, .4, 0.1, 1.5])
    for i in range(91):
        x, z = xz(); y = rng.uniform(0.3, 3.0)
        recs.append([1, x, y, z, x + rng.uniform(-.4, .4), y + rng.uniform(-.3, .3), z + rng.uniform(-.4, .4), 0.0, 1.0, 0.2, i % 3, .3, .3, .8, 0.0, 1.4])
    for i in range(53):
        x, z = xz(); y = rng.uniform(0.3, 4.0); recs.append([0, x, y, z, x, y, z, 0, 1, 0.22, i % 3, .5, .6, .7, 0.3, 1.5])
    for i in range(9):
        x, z = xz(); recs.append([0, x, 0.3, z, x, 0.3, z, 0, 1, 0.3, i % 3, .5, .5, .7, 0.3, 1.5])
    order = rng.permutation(len(recs))
    return tor.Scene.from_records(np.asarray(recs, dtype=np.float64)[order])


def test_plane_screen_on_every_segment_kind(tor, oracle):
    """Round 4's stage one ran on segments that share c0.y bit for bit with movers along y only -- two coincidences of scenes.nim:24-36.
    The test itself never reads y (tor_screen.hpp), so round 5 runs it on every float64 segment: statics (xkind 10 / 11), movers
    along y (12 / 14), movers in general position (13: the chain evaluates c0_xz + f dc_xz), with the segment's own wave-uniform test
    as the per-lane stage two.  Claim: what is left is EXACTLY what that test alone leaves -- same canvas bit for bit and the same
    number of candidates with TOR_PLANE=0 (no stage one), the default (the host's gate and the waves' votes decide per segment)
    and TOR_PLANE=2 (stage one on every segment with a table) -- and the canvas is the oracle's."""
    import torch
    rng = np.random.default_rng(55)
    anim = tor.Animation(108, 192)
    a_cam, a_scene, _ = next(iter(anim.scenes(skip=37)))
    scenes = [("3-D cloud", _cloud_scene(tor, rng), tor.camera(look_from=(14, 5, 6), look_at=(0, 3, 0), aperture=0.05)),
              ("3-D cloud seen from below", _cloud_scene(tor, rng, n=200, degenerate=False), tor.camera(look_from=(1, 0.3, 1), look_at=(0, 5, 0.5), vertical_field_of_view=70.0, aperture=0.0)),
              ("wall seen edge-on", _wall_scene(tor), tor.camera(look_from=(16, 3, 0.3), look_at=(0, 2, 0), aperture=0.02)),
              ("many heights", _many_heights_scene(tor, rng), tor.camera(look_from=(11, 2.2, 5), aperture=0.05)),
              ("three time groups, hollow spheres, glass", _screen_scenes(tor)[1][1], tor.camera(look_from=(10, 2.5, 4), aperture=0.05)),
              ("ragged segment tails", _ragged_segments_scene(tor, rng), tor.camera(look_from=(11, 2.5, 5), aperture=0.05)),
              ("ragged tails, more than 512 slots (two passes)", _big_ragged_scene(tor, rng), tor.camera(look_from=(12, 2.5, 5), aperture=0.05)),
              ("animation frame 37 (1601 statics at distinct heights)", a_scene, a_cam)]
    h, w = 108, 192
    for name, scene, cam in scenes:
        for seeding in (tor.SEED_SAMPLE, tor.SEED_PIXEL):
            off, _ = _render_with_env(tor, scene, cam, h, w, 16, {"TOR_PLANE": "0"}, seeding=seeding, accel=0)
            on, _ = _render_with_env(tor, scene, cam, h, w, 16, {}, seeding=seeding, accel=0)
            forced, _ = _render_with_env(tor, scene, cam, h, w, 16, {"TOR_PLANE": "2"}, seeding=seeding, accel=0)
            bare, _ = _render_with_env(tor, scene, cam, h, w, 16, {"TOR_SCREEN": "0"}, seeding=seeding, accel=0)
            assert torch.equal(on, off) and torch.equal(forced, off) and torch.equal(bare, off), (name, seeding)
            assert float(on.abs().sum()) > 0.0
        # candidates of the resolve pass: identical with, without and with forced stage one
        stats = {}
        for key, env in (("off", {"TOR_PLANE": "0"}), ("default", {"TOR_PLANE": None}), ("forced", {"TOR_PLANE": "2"})):
            with _env(**env):
                ctx = tor.Context(0)
            ctx.upload(scene.list())
            ctx.set_stats(True)
            buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
            ctx.render_device(cam, h, w, 8, 2.2, 50, tor.make_options(seeding=tor.SEED_SAMPLE, accel=0), buf.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            stats[key] = ctx.last_stats()
            ctx.close()
        a, b, c = stats["off"], stats["default"], stats["forced"]
        assert a.hit_queries == b.hit_queries == c.hit_queries and a.samples == b.samples == c.samples, name
        assert a.candidates == b.candidates == c.candidates, (name, a.candidates, b.candidates, c.candidates)
    # == the ORACLE on every one of the scenes (round 6; VERDICT r5 weak #1: round 5 tied only the first to it): the restatement of
    # the reference walks the caller's list in the caller's order (hittables_lists.nim:48-55) and knows nothing of segments, padding
    # or passes -- a defect in what all the GPU's loop families share is invisible to the GPU-vs-GPU comparisons above.
    # (TorCamera = 24 float64 in cameras.nim's field order = the oracle's camera)
    for name, scene, cam in scenes:
        recs = scene.to_records()
        ocam = np.frombuffer(bytes(cam), dtype=np.float64).copy()
        for seeding in (0, 1):
            want = oracle.render(54, 96, 8, ocam, recs, seeding=seeding, math=1, arith=0, accum=seeding).pixels
            got, _ = _render_with_env(tor, scene, cam, 54, 96, 8, {}, seeding=seeding, accel=0)
            assert float(np.abs(want).sum()) > 0.0, name
            _exact(got.cpu().numpy(), want)
            fast, _ = _render_with_env(tor, scene, cam, 54, 96, 8, {}, seeding=seeding, accel=3)   # ... and both exact accelerations
            _exact(fast.cpu().numpy(), want)


@pytest.mark.timeout(300)
def test_configs3_row_against_the_pinned_libm_oracle(tor, oracle, ref_scene, ref_camera):
    """BASELINE configs[3] (3840x2160, 4096 spp, depth 50) in the per-sample streams: one row of the frame against the oracle in
    its PNG-pinned LIBM mode with the plain SEQUENTIAL float64 sum (render.nim:59-67, canvas.nim:47-54) -- within the stated 1e-5,
    measured value printed -- and bit-exact against the PORTABLE / QUANTIZED twin.  The analytic bound beside quantize36
    (oracle/tor_oracle.c): every sample is rounded to 2^-36, at most 2^-37 each, so the sum of 4096 samples is within 4096 x 2^-37
    = 3e-8 of the sequential sum before the 1/spp, 7.3e-12 after it; the libm / portable sin, cos, pow differ in the last ulp on top."""
    import torch
    objs, _ = ref_scene
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    h, w, spp = 2160, 3840, 4096
    row = 640                        # through the small spheres and the glass ball
    ctx = tor.Context(0)
    ctx.upload(scene.list())
    buf = torch.zeros((1, w, 3), dtype=torch.float64, device="cuda")
    # shard `row` of h one-row shards is exactly image row `row`
    o = tor.make_options(seeding=tor.SEED_SAMPLE, shard_index=row, shard_count=h, row_tile=1)
    assert list(tor.shard_rows(h, 1, row, h)) == [row]
    ctx.render_device(cam, h, w, spp, 2.2, 50, o, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = buf.cpu().numpy()[0]
    ctx.close()
    exact = oracle.render(h, w, spp, ref_camera, objs, seeding=1, math=1, accum=1, rows=(row, row + 1), col_block=8).pixels[row]
    _exact(got, exact)
    libm = oracle.render(h, w, spp, ref_camera, objs, seeding=1, math=0, accum=0, rows=(row, row + 1), col_block=8).pixels[row]
    err = float(np.max(np.abs(got - libm)))
    print(f"configs[3] row {row}: max |GPU - oracle(LIBM, SEQUENTIAL)| = {err:.3e} (stated tolerance {TOL})")
    assert err <= TOL, err
    assert err < 1e-8, err           # what is actually observed: the bound above plus the last-ulp differences of sin / cos / pow


def test_non_finite_ray_time_hits_statics_only(tor, oracle, ref_scene):
    """A camera whose shutter never closes (shutter_close = inf) gives every camera ray time = inf (cameras.nim:56): the
    reference's moving spheres get non-finite centres and can never be hit, its static spheres are hit as ever -- and rays
    scattered by metal / glass carry time 0 again (rays.nim:19).  Round 5 put random_scene's resting statics into the resting
    movers' segment (dc = 0): the screened loop, the wave-uniform loops and the unscreened loop must still find them at a
    non-finite time fraction (integrate_loop_f64_movers.inc; tor_screen.hpp: a wild ray keeps everything).  == the oracle.
    Round 6: ... and at a FINITE time fraction whose square overflows (shutter_close = 1e155: f^2 = inf met the resting
    statics' dcy = 0 as NaN in the second form's chain and the screen dropped them; |f| |dc| stays far below the wild limit because
    random_scene's movers travel at most 0.5)."""
    import torch
    objs, _ = ref_scene
    scene = tor.random_scene(0xFACADE)
    h, w, spp = 54, 96, 8
    for close in (float("inf"), 1e155, 3e154):
      cam = tor.camera(shutter_open=0.0, shutter_close=close)
      ocam = oracle.camera(shutter_open=0.0, shutter_close=close)
      for seeding in (0, 1):
          want = oracle.render(h, w, spp, ocam, objs, seeding=seeding, math=1, arith=0, accum=seeding).pixels
          assert np.isfinite(want).all() and float(want.sum()) > 0.0
          for env in ({}, {"TOR_PLANE": "0"}, {"TOR_PLANE": "2"}, {"TOR_SCREEN": "0"}):
              got, _ = _render_with_env(tor, scene, cam, h, w, spp, env, seeding=seeding, accel=0, pixel_kernel=tor.PIXEL_KERNEL_LANE)
              _exact(got.cpu().numpy(), want)
          for accel in (1, 2, 3):
              got, _ = _render_with_env(tor, scene, cam, h, w, spp, {}, seeding=seeding, accel=accel, pixel_kernel=tor.PIXEL_KERNEL_LANE)
              _exact(got.cpu().numpy(), want)


def test_fused_arithmetic_is_refused_with_the_reason(tor):
    """Round 5 removed the TOR_ARITH_FUSED kernel variants (not the reference's rounding, README.md:82); the enum value stays
    reserved so that a caller of rounds 1-4 fails loudly instead of silently getting another arithmetic."""
    import torch
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    cv = tor.new_canvas(8, 8, 2, 2.2)
    cv.pixels[:] = 9.0
    with pytest.raises(tor.TorError) as e:
        tor.render(cv, cam, scene.list(), 5, tor.make_options(arith=tor.ARITH_FUSED))
    assert e.value.code == -1 and "TOR_ARITH_FUSED was removed" in str(e.value)
    assert np.all(cv.pixels == 9.0)
    ctx = tor.Context(0)
    ctx.upload(scene.list())
    buf = torch.zeros((8, 8, 3), dtype=torch.float64, device="cuda")
    with pytest.raises(tor.TorError) as e:
        ctx.render_device(cam, 8, 8, 2, 2.2, 5, tor.make_options(arith=tor.ARITH_FUSED), buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert "TOR_ARITH_FUSED was removed" in str(e.value)
    ctx.close()
