"""GPU tests added in round 3:
 * the SEED_PIXEL chain hand-off (DESIGN 4.7 (HISTORY 4.10)): lanes push long pixel chains to server waves inside the same launch --
   same canvas bit for bit under every setting of its knobs, == oracle, == the reference PNG at C1;
 * the multi-GPU path's fallback chain (TOR_FAULT_INJECT), still on one GPU;
 * ABI nits: tor_last_render_timing measured with events, the one-stream-per-context rule, option validation;
 * an empty region A in the SEED_PIXEL schedule with every wave in a slow slot;
 * the conservative FMA screen of the strict float64 object loop (csrc/tor_screen.hpp): on / off, same canvas, == oracle;
 * full-size coverage of the two multi-GPU configs on one GPU: the whole configs[3] frame on one context == its 8 shares
   assembled through the device-list path; configs[4] over 8 consecutive frames through the real frame loop."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HANDOFF_KNOBS = ("TOR_MIGRATE", "TOR_SRV_FRAC", "TOR_SRV_MIN_FRAC", "TOR_SRV_PATIENCE_US", "TOR_PUSH_THETA", "TOR_CHAIN_THETA", "TOR_FLOOR_THETA",
                 "TOR_TAIL_LANES", "TOR_TAIL_REST", "TOR_MIG_FLAGS", "TOR_SCREEN")


def _exact(got, want):
    err = float(np.max(np.abs(got - want)))
    assert err <= TOL, f"max per-channel error {err} > {TOL}"
    assert np.array_equal(got, want), f"not bit-exact (max err {err}, {(got != want).sum()} values differ)"


def _render_with_env(tor, scene, cam, h, w, spp, env, depth=50, **opt):
    """One frame through a FRESH context created under `env` (the library's knobs are read at tor_context_create)."""
    import torch
    # (every knob this helper may set is restored afterwards -- round 5: a key of `env` outside HANDOFF_KNOBS used to leak into
    # the contexts created later in the same process, e.g. TOR_PLANE=0 into the run that was meant to be the default)
    keys = tuple(HANDOFF_KNOBS) + ("TOR_PLANE",) + tuple(k for k in env if k not in HANDOFF_KNOBS)
    saved = {k: os.environ.get(k) for k in keys}
    try:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        ctx = tor.Context(0)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    ctx.upload(scene.list())
    o = tor.make_options(**opt)
    rows = len(tor.shard_rows(h, o.row_tile, o.shard_index, max(o.shard_count, 1)))
    buf = torch.zeros((rows, w, 3), dtype=torch.float64, device="cuda")
    ctx.render_device(cam, h, w, spp, 2.2, depth, o, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    counters = ctx.last_handoff_counters()
    ctx.close()
    return buf, counters


def test_chain_handoff_never_changes_a_pixel(tor, oracle, ref_scene, ref_camera):
    """The hand-off only decides WHO continues a pixel chain: a lane, or -- from a sample boundary on -- a server wave.
    Same canvas with it off, on, and under settings that push almost everything / nothing / only in the tail / with every
    third workgroup a server; == the oracle (reference streams, render.nim:59-67)."""
    import torch
    objs, _ = ref_scene
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    h, w, spp = 90, 160, 64
    want = oracle.render(h, w, spp, ref_camera, objs, seeding=0, math=1, arith=0).pixels
    base, c0 = _render_with_env(tor, scene, cam, h, w, spp, {"TOR_MIGRATE": "0"}, seeding=tor.SEED_PIXEL, accel=3)
    assert c0["pushed"] == 0 and c0["served"] == 0
    _exact(base.cpu().numpy(), want)
    settings = [
        {},                                                              # defaults
        {"TOR_PUSH_THETA": "0.01", "TOR_CHAIN_THETA": "0.01", "TOR_FLOOR_THETA": "0.01"},  # nearly every chain is "hot"
        {"TOR_PUSH_THETA": "1e9", "TOR_TAIL_LANES": "64"},                # only tail pushes, from the first exhausted wave on
        {"TOR_PUSH_THETA": "1e9", "TOR_TAIL_LANES": "-1"},                # nobody pushes: servers only wait
        {"TOR_SRV_MIN_FRAC": "0.3", "TOR_SRV_FRAC": "0.3", "TOR_TAIL_REST": "0", "TOR_SRV_PATIENCE_US": "100"},
        {"TOR_SRV_MIN_FRAC": "0.3", "TOR_SRV_FRAC": "0.3", "TOR_SRV_PATIENCE_US": "0"},
        {"TOR_MIG_FLAGS": "0x101"},                                       # acquire polling, no adaptive threshold
        {"TOR_TAIL_LANES": "0"},
    ]
    pushed_some = False
    for env in settings:
        got, c = _render_with_env(tor, scene, cam, h, w, spp, env, seeding=tor.SEED_PIXEL, accel=3)
        assert torch.equal(got, base), (env, c)
        assert c["served"] == c["pushed"] and c["lane_waves_left"] == 0, (env, c)
        pushed_some = pushed_some or c["pushed"] > 0
    assert pushed_some
    # a frame large enough that every wave has lane work (dedicated servers matter), row shard of a bigger frame, odd sizes
    for (hh, ww, s, kw) in [(540, 960, 48, {}), (1080, 1920, 64, dict(shard_index=3, shard_count=8)), (333, 517, 40, {})]:
        off, _ = _render_with_env(tor, scene, cam, hh, ww, s, {"TOR_MIGRATE": "0"}, seeding=tor.SEED_PIXEL, accel=3, **kw)
        on, c = _render_with_env(tor, scene, cam, hh, ww, s, {}, seeding=tor.SEED_PIXEL, accel=3, **kw)
        hot, c2 = _render_with_env(tor, scene, cam, hh, ww, s, {"TOR_PUSH_THETA": "0.5", "TOR_CHAIN_THETA": "2", "TOR_FLOOR_THETA": "0.3"},
                                   seeding=tor.SEED_PIXEL, accel=3, **kw)
        assert torch.equal(on, off) and torch.equal(hot, off), (hh, ww, c, c2)
        assert c2["hot_pushes"] > 0 and c2["served"] == c2["pushed"]
    # (rounds 3-4 repeated this for the TOR_ARITH_FUSED variants, which carried their own copy of the server code; removed in round 5)
    off, _ = _render_with_env(tor, scene, cam, 270, 480, 48, {"TOR_MIGRATE": "0"}, seeding=tor.SEED_PIXEL, accel=3)
    on, c = _render_with_env(tor, scene, cam, 270, 480, 48, {"TOR_PUSH_THETA": "0.5", "TOR_CHAIN_THETA": "2", "TOR_FLOOR_THETA": "0.3"},
                             seeding=tor.SEED_PIXEL, accel=3)
    assert torch.equal(on, off) and c["pushed"] > 0 and c["served"] == c["pushed"]
    want = oracle.render(270, 480, 48, ref_camera, objs, seeding=0, math=1, arith=0, rows=(100, 101)).pixels[100]
    _exact(on[100].cpu().numpy(), want)


def test_chain_handoff_c1_is_the_reference_image(tor, golden_dir):
    """C1 (the reference's own main(): 384x216, 100 spp) through tor_render() with the hand-off on (default): the 8-bit image
    is the reference's PNG, 0 of 248 832 channels differ -- and chains really were handed over."""
    from PIL import Image
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    cv = tor.new_canvas(216, 384, 100, 2.2)
    tor.render(cv, cam, scene.list(), 50)
    png = np.array(Image.open(os.path.join(golden_dir, "book2_motion_blur.png")).convert("RGB"))
    rgb = tor.export_rgb8(cv)
    assert int((rgb != png).sum()) == 0
    got, c = _render_with_env(tor, scene, cam, 216, 384, 100, {}, seeding=tor.SEED_PIXEL, accel=3)
    assert np.array_equal(got.cpu().numpy(), cv.pixels) and c["pushed"] > 100 and c["served"] == c["pushed"]


def test_gather_fallback_chain_under_fault_injection(tor):
    """TOR_GATHER_AUTO walks RCCL -> peer copies -> per-device D2H; a leg that fails is followed by the next, never a wrong
    canvas.  On this 1-GPU box the device list repeats ordinal 0; TOR_FAULT_INJECT makes the named legs fail."""
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    h, w, spp = 64, 96, 4
    one = tor.new_canvas(h, w, spp, 2.2)
    tor.render(one, cam, scene.list(), 50, tor.make_options(seeding=tor.SEED_SAMPLE, accel=3))
    try:
        for inject, expect in (("", "gather: peer"), ("rccl_init", "gather: peer"), ("peer", "gather: host"), ("rccl_init,peer", "gather: host")):
            os.environ["TOR_FAULT_INJECT"] = inject
            cv = tor.new_canvas(h, w, spp, 2.2)
            cv.pixels[:] = -1.0
            tor.render(cv, cam, scene.list(), 50, tor.make_options(seeding=tor.SEED_SAMPLE, accel=3, devices=[0, 0, 0]))
            assert np.array_equal(cv.pixels, one.pixels), inject
            note = tor.last_note()
            assert note.endswith(expect), (inject, note)
            if "rccl_init" in inject:
                assert "rccl failed" in note
            if "peer" in inject.split(","):
                assert "peer failed" in note
        # an EXPLICIT mode is tried alone: its failure is the caller's answer, and the canvas is left untouched
        os.environ["TOR_FAULT_INJECT"] = "peer"
        cv = tor.new_canvas(h, w, spp, 2.2)
        cv.pixels[:] = -1.0
        with pytest.raises(tor.TorError) as e:
            tor.render(cv, cam, scene.list(), 50, tor.make_options(seeding=tor.SEED_SAMPLE, accel=3, devices=[0, 0], gather=tor.GATHER_PEER))
        assert "peer failed" in str(e.value) and np.all(cv.pixels == -1.0)
    finally:
        os.environ.pop("TOR_FAULT_INJECT", None)


def test_option_validation_says_why(tor):
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    cv = tor.new_canvas(8, 8, 1)
    for kw, word in ((dict(devices=[0, 0], shard_index=1, shard_count=2), "device list"), (dict(devices=[0, 0], device=3), "device"),
                     (dict(row_tile=-3), "row_tile"), (dict(accel=9), "accel")):
        with pytest.raises(tor.TorError) as e:
            tor.render(cv, cam, scene.list(), 5, tor.make_options(**kw))
        assert e.value.code == -1 and word in str(e.value), (kw, str(e.value))
    for name, val in (("TOR_DEVICES", "0,abc"), ("TOR_DEVICES", "99"), ("TOR_GATHER", "carrier-pigeon"), ("TOR_DEFAULT_ACCEL", "7")):
        os.environ[name] = val
        try:
            with pytest.raises(tor.TorError) as e:
                tor.render(cv, cam, scene.list(), 5)
            assert e.value.code == -1 and name in str(e.value)
        finally:
            os.environ.pop(name, None)
    tor.render(cv, cam, scene.list(), 5)   # and a clean environment renders


def test_one_stream_per_context_is_enforced(tor):
    """Launches of one context that may overlap must use ONE stream (per-launch state is ordered by it): a launch on a second
    stream while the first is still running is refused, and accepted once the first has finished."""
    import torch
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    ctx = tor.Context(0)
    ctx.upload(scene.list())
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    h, w = 540, 960
    a = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    b = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    o = tor.make_options(seeding=tor.SEED_SAMPLE)
    ctx.render_device(cam, h, w, 200, 2.2, 50, o, a.data_ptr(), s1.cuda_stream)      # ~90 ms of float64 brute force
    with pytest.raises(tor.TorError) as e:
        ctx.render_device(cam, h, w, 2, 2.2, 50, o, b.data_ptr(), s2.cuda_stream)
    assert e.value.code == -1 and "ONE stream" in str(e.value)
    ctx.render_device(cam, h, w, 2, 2.2, 50, o, b.data_ptr(), s1.cuda_stream)        # the same stream may queue
    torch.cuda.synchronize()
    ctx.render_device(cam, h, w, 2, 2.2, 50, o, a.data_ptr(), s2.cuda_stream)        # idle context: any stream
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    ctx.close()


def test_last_render_timing_means_what_the_header_says(tor):
    """out[1] = launch + kernels until the device is done (HIP events), out[2] = the download -- without TOR_TIMING_SPLIT."""
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    cv = tor.new_canvas(540, 960, 300, 2.2)
    opt = tor.make_options(seeding=tor.SEED_SAMPLE)
    tor.render(cv, cam, scene.list(), 50, opt)
    tor.render(cv, cam, scene.list(), 50, opt)     # ~50 ms of kernel (round 5; 130 ms when the test was written), ~0.5 ms of D2H
    t = tor.last_render_timing()
    assert t["scene_cache_hit"] and t["upload_ms"] < 5.0
    assert t["render_ms"] > 20.0 and t["render_ms"] > 10.0 * t["download_ms"], t
    assert abs(t["upload_ms"] + t["render_ms"] + t["download_ms"] - t["total_ms"]) < 0.1 * t["total_ms"] + 1.0, t


def test_configs3_whole_frame_equals_its_eight_shares(tor):
    """BASELINE configs[3] at FULL size on one GPU: the whole 3840x2160x4096 frame on one context
    (3.4e10 samples: the work-index space is far past 2^32) == the frame tor_render_opt assembles from 8 row shards rendered
    on 8 contexts ({0} x 8: render_multi_device, peer gather, gather_rows_kernel at 199 MB, D2H)."""
    import torch
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    h, w, spp = 2160, 3840, 4096
    ctx = tor.Context(0)
    ctx.upload(scene.list())
    whole = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    ctx.render_device(cam, h, w, spp, 2.2, 50, tor.make_options(seeding=tor.SEED_SAMPLE, accel=3), whole.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ctx.close()
    assert bool(torch.isfinite(whole).all()) and float(whole.min()) >= 0.0
    cv = tor.new_canvas(h, w, spp, 2.2)
    tor.render(cv, cam, scene.list(), 50, tor.make_options(seeding=tor.SEED_SAMPLE, accel=3, devices=[0] * 8, row_tile=1))
    assert tor.last_note().endswith("gather: peer")
    got = torch.from_numpy(cv.pixels).cuda()
    assert torch.equal(got, whole)
    del got
    # and the other gather (every device copies its rows straight into the canvas), ragged tiles of 7 rows
    cv2 = tor.new_canvas(h, w, 64, 2.2)
    tor.render(cv2, cam, scene.list(), 50, tor.make_options(seeding=tor.SEED_SAMPLE, accel=3, devices=[0] * 8, row_tile=7, gather=tor.GATHER_HOST))
    one = tor.new_canvas(h, w, 64, 2.2)
    tor.render(one, cam, scene.list(), 50, tor.make_options(seeding=tor.SEED_SAMPLE, accel=3))
    assert np.array_equal(cv2.pixels, one.pixels)


def test_configs4_frame_loop_over_eight_frames(tor, oracle):
    """BASELINE configs[4] through the real frame loop (trace_of_radiance_animation.nim:173-196) at full size: 8 consecutive
    frames of the animated scene (1601 objects, 1920x1080, 256 spp), frame f -> context f mod 2 on {0, 0}:
    tor_animation_next -> tor_scene_upload -> tor_render_frame_h264.  Every slice decodes with the spec-derived decoder; for
    3 frames one macroblock row of the slice == the oracle's encoder on the oracle's render of those 16 canvas rows, and one
    float64 row == oracle.  (Parity of the animation generator itself is unpinned: the reference holds no frame.)"""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import h264_spec_decoder as dec
    h, w, spp = 1080, 1920, 256
    first = 30
    it = iter(tor.Animation(h, w, 0.005, 0.0, 7.2).scenes(6))
    oit = oracle.animation_scenes(h, w, 0.005, 0.0, 7.2, 6)
    ctxs = [tor.Context(0), tor.Context(0)]
    opt = tor.make_options(seeding=tor.SEED_SAMPLE, accel=3)
    header = tor.h264_stream_header(w, h)
    stream = bytearray(header)
    checked = 0
    for f in range(first + 8):
        cam, scene, t = next(it)
        ocam, oobjs, ot = next(oit)
        if f < first:
            continue
        assert t == ot and np.array_equal(cam.as_array(), ocam) and np.array_equal(scene.to_records(), oobjs)
        ctx = ctxs[f % 2]
        ctx.upload(scene.list())
        sl = ctx.render_frame_h264(cam, h, w, spp, 2.2, 50, opt)
        assert len(sl) == tor.h264_frame_bytes(w, h)
        stream += sl
        if f in (first, first + 3, first + 7):
            mi = (11, 40, 63)[checked]                         # macroblock row of the video frame (0 = top)
            r0 = h - 16 - 16 * mi                              # its 16 canvas rows (row 0 = bottom)
            sub = oracle.render(h, w, spp, ocam, oobjs, seeding=1, math=1, arith=0, accum=1, rows=(r0, r0 + 16), col_block=8).pixels[r0:r0 + 16]
            _, _, _, _, osl = oracle.encode_frame(np.ascontiguousarray(sub))
            n_mb = w // 16
            for mj in range(n_mb):
                a = sl[9 + (mi * n_mb + mj) * 386: 9 + (mi * n_mb + mj) * 386 + 384]
                b = osl[9 + mj * 386: 9 + mj * 386 + 384]
                assert a == b, (f, mi, mj)
            buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
            ctx.render_device(cam, h, w, spp, 2.2, 50, opt, buf.data_ptr(), 0)
            torch.cuda.synchronize()
            _exact(buf[r0 + 5].cpu().numpy(), sub[5])
            checked += 1
    assert checked == 3
    sps, pps, pictures = dec.decode_stream(bytes(stream))
    assert len(pictures) == 8 and sps is not None and pps is not None
    for _, y, cb, cr in pictures:
        assert y.shape == (h, w) and cb.shape == (h // 2, w // 2) and cr.shape == (h // 2, w // 2)
        assert 16 <= int(y.min()) and int(y.max()) <= 235       # limited-range BT.601 luma
    for c in ctxs:
        c.close()


def _screen_scenes(tor):
    """(name, scene, camera): the bench scene; objects of every kind in three time groups around a big glass sphere, seen
    through a lens; the bench scene from INSIDE the cloud; the bench scene moved 3e5 units away from the origin (o - c cancels
    11 digits: the reference's own discriminant is noise at the 1e-6 level there, and so are the screen's margins)."""
    rng = np.random.default_rng(5)
    recs = [[0, 0, -1000, 0, 0, -1000, 0, 0, 1, 1000, 0, .5, .5, .5, 0, 0]]
    for i in range(120):
        x, z = rng.uniform(-9, 9, 2)
        kind, mat = i % 4, [0, 1, 2][i % 3]
        r = 0.25 if i % 17 else -0.25
        if kind == 0:
            recs.append([0, x, .25, z, x, .25, z, 0, 1, r, mat, .6, .5, .4, 0.2, 1.5])
        elif kind == 1:
            recs.append([1, x, .25, z, x, .25 + rng.uniform(0, .6), z, 0.0, 1.0, r, mat, .3, .7, .4, 0.1, 1.5])
        elif kind == 2:
            recs.append([1, x, .25, z, x + rng.uniform(-.5, .5), .4, z + rng.uniform(-.5, .5), 0.25, 0.75, r, mat, .3, .3, .8, 0.0, 1.4])
        else:
            recs.append([1, x, .25, z, x, .25, z + .3, -1.0, 2.0, r, mat, .8, .3, .3, 0.4, 1.3])
    recs.append([0, 0, 1, 0, 0, 1, 0, 0, 1, 1.0, 2, 0, 0, 0, 0, 1.5])
    mixed = np.asarray(recs, dtype=np.float64)
    base = tor.random_scene(0xFACADE)
    out = [("random_scene", base, tor.camera()),
           ("three time groups, hollow spheres, glass", tor.Scene.from_records(mixed), tor.camera(look_from=(10, 2.5, 4), aperture=0.05)),
           ("random_scene from inside", base, tor.camera(look_from=(0.6, 0.7, 0.9), look_at=(4, 0.6, 0), vertical_field_of_view=60.0, aperture=0.02,
                                                         focus_distance=3.0))]
    shift = np.array([3e5, -2e5, 1e5])
    moved = mixed.copy()
    moved[:, 1:4] += shift
    moved[:, 4:7] += shift
    out.append(("everything 3e5 units from the origin", tor.Scene.from_records(moved),
                tor.camera(look_from=tuple(float(x) for x in np.array([10, 2.5, 4]) + shift), look_at=tuple(float(x) for x in shift), aperture=0.05)))
    return out


def test_fma_screen_never_changes_a_pixel(tor, oracle, ref_scene, ref_camera):
    """Strict launches of the brute-force layouts run the wave-uniform object loop as a conservative FMA screen (11 / 12 / 14
    float64 instructions per ray x object instead of the reference's 17 / 19 / 23 unfused operations); what it keeps is re-tested
    with the reference's own operations (spheres.nim:29-48).  TOR_SCREEN=0 is rounds 1-2: the unfused discriminant for every
    object.  Same canvas bit for bit, both stream layouts, four scenes; == the oracle; and the screen keeps a superset."""
    import torch
    objs, _ = ref_scene
    for name, scene, cam in _screen_scenes(tor):
        for seeding in (tor.SEED_SAMPLE, tor.SEED_PIXEL):
            off, _ = _render_with_env(tor, scene, cam, 108, 192, 24, {"TOR_SCREEN": "0"}, seeding=seeding, accel=0)
            on, _ = _render_with_env(tor, scene, cam, 108, 192, 24, {}, seeding=seeding, accel=0)
            assert torch.equal(on, off), (name, seeding, int((on != off).sum()))
            assert float(on.abs().sum()) > 0.0
    h, w, spp = 90, 160, 32
    for seeding in (0, 1):
        want = oracle.render(h, w, spp, ref_camera, objs, seeding=seeding, math=1, arith=0, accum=seeding).pixels   # (accum 1: 2^-36 quanta, as SEED_SAMPLE sums)
        got, _ = _render_with_env(tor, tor.random_scene(0xFACADE), tor.camera(), h, w, spp, {}, seeding=seeding, accel=0)
        _exact(got.cpu().numpy(), want)
    # the candidate counts: same queries, the screen keeps at least what the sign filter of the unfused discriminant keeps
    stats = {}
    for key, env in (("off", {"TOR_SCREEN": "0"}), ("on", {})):
        os.environ.update(env)
        try:
            ctx = tor.Context(0)
        finally:
            os.environ.pop("TOR_SCREEN", None)
        ctx.upload(tor.random_scene(0xFACADE).list())
        ctx.set_stats(True)
        buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
        ctx.render_device(tor.camera(), h, w, spp, 2.2, 50, tor.make_options(seeding=tor.SEED_SAMPLE, accel=0), buf.data_ptr(),
                          torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        stats[key] = ctx.last_stats()
        ctx.close()
    assert stats["on"].hit_queries == stats["off"].hit_queries and stats["on"].samples == stats["off"].samples
    assert stats["off"].candidates <= stats["on"].candidates <= 1.6 * stats["off"].candidates, (stats["on"].candidates, stats["off"].candidates)


def test_servers_walk_the_box_bits_when_a_ray_enters_64_boxes_or_more(tor):
    """serve_chains compacts the boxes a ray can touch with lane permutes -- for up to 63 of them.  A `wild` ray (outside the
    float32 test's guarded ranges) enters every box; with more than 63 blocks the servers fall back to the walk over the set
    bits.  700 small spheres = 88 blocks on one culling level, camera 3e6 units away (|o - P| > 2^20: every camera ray is
    wild): hand-off on == off == the float64 brute force, with chains really served."""
    import torch
    rng = np.random.default_rng(11)
    recs = [[0, 0, -1000, 0, 0, -1000, 0, 0, 1, 1000, 0, .5, .5, .5, 0, 0]]
    for i in range(700):
        x, z = rng.uniform(-14, 14, 2)
        mat = [0, 1, 2][i % 3]
        if i % 2:
            recs.append([0, x, .2, z, x, .2, z, 0, 1, .2, mat, .6, .5, .4, 0.2, 1.5])
        else:
            recs.append([1, x, .2, z, x, .2 + rng.uniform(0, .5), z, 0.0, 1.0, .2, mat, .3, .7, .4, 0.1, 1.5])
    scene = tor.Scene.from_records(np.asarray(recs, dtype=np.float64))
    dist = 3.0e6
    look_from = (dist * 0.8, dist * 0.3, dist * 0.52)
    cam = tor.camera(look_from=look_from, look_at=(0, 0.5, 0), vertical_field_of_view=float(np.degrees(2 * np.arctan(9.0 / dist))),
                     aperture=0.0, focus_distance=dist)
    h, w, spp = 36, 64, 40
    brute, _ = _render_with_env(tor, scene, cam, h, w, spp, {}, seeding=tor.SEED_PIXEL, accel=0, pixel_kernel=tor.PIXEL_KERNEL_LANE)
    # (the hand-off belongs to TorOptions.pixel_kernel = AUTO; TOR_MIGRATE=0 + LANE is the lane kernel alone)
    off, c0 = _render_with_env(tor, scene, cam, h, w, spp, {"TOR_MIGRATE": "0"}, seeding=tor.SEED_PIXEL, accel=3, pixel_kernel=tor.PIXEL_KERNEL_LANE)
    on, c1 = _render_with_env(tor, scene, cam, h, w, spp, {"TOR_PUSH_THETA": "0.01", "TOR_CHAIN_THETA": "0.01", "TOR_FLOOR_THETA": "0.01"},
                              seeding=tor.SEED_PIXEL, accel=3)
    assert float(brute.abs().sum()) > 0.0 and float(brute.std()) > 0.0     # the camera does see the scene
    assert torch.equal(off, brute) and torch.equal(on, brute), (c0, c1)
    assert c0["pushed"] == 0 and c1["pushed"] > 0 and c1["served"] == c1["pushed"]
