"""GPU tests added in round 2: BASELINE's full-size configurations (configs[2], [3], [4]) on the device, the
multi-GPU path behind the C ABI, the scene cache / host-canvas region, the RCCL communicator inside the library,
the Nim-shim emulation and the wild-ray guard of the two-level culling layout.

Full-size frames cannot be rendered by the CPU oracle in seconds, so they are covered by (a) oracle renders of
selected ROWS at the full resolution and sample count (every pixel owns its stream, render.nim:59-60, so a row is
a complete unit) and (b) size-independent properties: determinism, shard invariance, accel invariance."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest
from PIL import Image

pytestmark = pytest.mark.gpu
TOL = 1e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exact(got, want):
    err = float(np.max(np.abs(got - want)))
    assert err <= TOL, f"max per-channel error {err} > {TOL}"
    assert np.array_equal(got, want), f"not bit-exact (max err {err}, {(got != want).sum()} values differ)"


def _device_render(tor, ctx, cam, h, w, spp, depth=50, **opt):
    import torch
    o = tor.make_options(**opt)
    rows = len(tor.shard_rows(h, o.row_tile, o.shard_index, max(o.shard_count, 1)))
    buf = torch.empty((rows, w, 3), dtype=torch.float64, device="cuda")
    ctx.render_device(cam, h, w, spp, 2.2, depth, o, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return buf


def test_configs2_full_size(tor, oracle, ref_scene, ref_camera):
    """BASELINE configs[2]: 1920x1080, 1000 spp, depth 50 -- the configuration the target is quoted on.
    Per-sample streams (the bench's mode): two renders identical (atomics order, schedule), brute force == both
    exact accelerations on the whole frame, five rows == oracle at the full 1000 spp.  Per-pixel streams (what
    tor_render() runs, default accelerations): the same five rows == oracle."""
    import torch
    objs, _ = ref_scene
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    ctx = tor.Context()
    ctx.upload(scene.list())
    h, w, spp = 1080, 1920, 1000
    rows = [0, 377, 540, 731, 1079]
    a = _device_render(tor, ctx, cam, h, w, spp, seeding=tor.SEED_SAMPLE)
    b = _device_render(tor, ctx, cam, h, w, spp, seeding=tor.SEED_SAMPLE)
    assert torch.equal(a, b)
    c = _device_render(tor, ctx, cam, h, w, spp, seeding=tor.SEED_SAMPLE, accel=3)
    assert torch.equal(a, c)
    del b, c
    got = a.cpu().numpy()
    assert np.all(np.isfinite(got)) and got.min() >= 0.0
    for r in rows:
        want = oracle.render(h, w, spp, ref_camera, objs, seeding=1, math=1, arith=0, accum=1, rows=(r, r + 1), col_block=8).pixels[r]
        _exact(got[r], want)
    p = _device_render(tor, ctx, cam, h, w, spp, seeding=tor.SEED_PIXEL, accel=3).cpu().numpy()
    for r in rows:
        want = oracle.render(h, w, spp, ref_camera, objs, seeding=0, math=1, arith=0, rows=(r, r + 1), col_block=8).pixels[r]
        _exact(p[r], want)
    # ... and against the oracle's PINNED mode (LIBM: the one that reproduces the reference PNG at C1) at this size too: the
    # portable math twin the rows above are bit-equal to differs from it by last-bit events only
    for r in (540, 731):
        pinned = oracle.render(h, w, spp, ref_camera, objs, seeding=0, math=0, arith=0, rows=(r, r + 1), col_block=8).pixels[r]
        err = float(np.max(np.abs(p[r] - pinned)))
        print(f"configs[2] row {r} GPU vs oracle(LIBM): max |delta| = {err:.3e}")
        assert err < 1e-12
    ctx.close()


def test_configs3_rank_shares(tor, oracle, ref_scene, ref_camera):
    """BASELINE configs[3]: 3840x2160, 4096 spp, rows sharded across 8 GPUs.  On one GPU: rank 0's share of the
    8-way split (270 rows, 4.25e9 samples -- just under 2^32) rendered twice -> identical; the same rows inside a
    2-way split (1080 rows, 1.7e10 samples -- PAST 2^32, so any 32-bit wrap in the work-index arithmetic would show)
    -> identical to the 8-way result where the two splits share rows; two rows against the oracle at the full
    4096 spp (one of them the LAST local row of the 2-way share, whose work indices are the largest)."""
    import torch
    objs, _ = ref_scene
    scene = tor.random_scene(0xFACADE)
    cam = tor.camera()
    ctx = tor.Context()
    ctx.upload(scene.list())
    h, w, spp = 2160, 3840, 4096
    r8 = tor.shard_rows(h, 1, 0, 8)
    assert len(r8) == 270
    a = _device_render(tor, ctx, cam, h, w, spp, seeding=tor.SEED_SAMPLE, accel=3, shard_index=0, shard_count=8, row_tile=1)
    b = _device_render(tor, ctx, cam, h, w, spp, seeding=tor.SEED_SAMPLE, accel=3, shard_index=0, shard_count=8, row_tile=1)
    assert torch.equal(a, b)
    del b
    brute = _device_render(tor, ctx, cam, h, w, spp, seeding=tor.SEED_SAMPLE, accel=0, shard_index=0, shard_count=8, row_tile=1)
    assert torch.equal(a, brute)      # 4.25e9 samples of float64 brute force == the accelerated render
    del brute
    r2 = tor.shard_rows(h, 1, 0, 2)
    assert len(r2) == 1080 and len(r2) * w * spp > 2 ** 32
    big = _device_render(tor, ctx, cam, h, w, spp, seeding=tor.SEED_SAMPLE, accel=3, shard_index=0, shard_count=2, row_tile=1)
    pos2 = {int(r): i for i, r in enumerate(r2)}
    idx2 = torch.tensor([pos2[int(r)] for r in r8], device="cuda")
    assert torch.equal(big[idx2], a)  # rows 0, 8, 16, ... belong to both shares
    got = big.cpu().numpy()
    for r in (int(r2[3]), int(r2[-1])):
        want = oracle.render(h, w, spp, ref_camera, objs, seeding=1, math=1, arith=0, accum=1, rows=(r, r + 1), col_block=8).pixels[r]
        _exact(got[pos2[r]], want)
    # the reference's stream layout on the same share (a pixel = a chain of 4096 samples)
    p = _device_render(tor, ctx, cam, h, w, spp, seeding=tor.SEED_PIXEL, accel=3, shard_index=0, shard_count=8, row_tile=1).cpu().numpy()
    r = int(r8[100])
    want = oracle.render(h, w, spp, ref_camera, objs, seeding=0, math=1, arith=0, rows=(r, r + 1), col_block=8).pixels[r]
    _exact(p[100], want)
    ctx.close()


def test_configs4_full_size_frame(tor, oracle):
    """BASELINE configs[4]: one full-size frame of the animated scene (1601 objects, 1920x1080, 256 spp) --
    frame 37 of the 240 (camera and spheres have moved): determinism, accel invariance, three rows == oracle.
    (The animation driver itself is pinned only to the builder's restatement: the reference holds no frame.)"""
    import torch
    h, w, spp = 1080, 1920, 256
    it = iter(tor.Animation(h, w, 0.005, 0.0, 7.2).scenes(6))
    oit = oracle.animation_scenes(h, w, 0.005, 0.0, 7.2, 6)
    for _ in range(38):
        cam, scene, t = next(it)
        ocam, oobjs, ot = next(oit)
    assert t == ot and np.array_equal(cam.as_array(), ocam) and np.array_equal(scene.to_records(), oobjs)
    ctx = tor.Context()
    ctx.upload(scene.list())
    a = _device_render(tor, ctx, cam, h, w, spp, seeding=tor.SEED_SAMPLE, accel=3)
    b = _device_render(tor, ctx, cam, h, w, spp, seeding=tor.SEED_SAMPLE, accel=3)
    assert torch.equal(a, b)
    c = _device_render(tor, ctx, cam, h, w, spp, seeding=tor.SEED_SAMPLE, accel=0)
    assert torch.equal(a, c)
    got = a.cpu().numpy()
    for r in (5, 400, 900):
        want = oracle.render(h, w, spp, ocam, oobjs, seeding=1, math=1, arith=0, accum=1, rows=(r, r + 1), col_block=8).pixels[r]
        _exact(got[r], want)
    ctx.close()


def test_multi_device_behind_the_abi(tor):
    """tor_render_opt with a device list (SURVEY 8e behind the drop-in): on a 1-GPU box the list names device 0
    several times -- separate contexts, host threads and streams, row-cyclic shards, framebuffer gather -- and
    the canvas must be bit-identical to the single-device one, for every gather path that can run here."""
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    for seeding in (tor.SEED_PIXEL, tor.SEED_SAMPLE):
        one = tor.new_canvas(45, 80, 8, 2.2)
        tor.render(one, cam, scene.list(), 50, tor.make_options(seeding=seeding, accel=3))
        for devices, row_tile in (([0, 0], 1), ([0, 0, 0], 4), ([0] * 8, 1)):
            for gather in (tor.GATHER_AUTO, tor.GATHER_PEER, tor.GATHER_HOST):
                cv = tor.new_canvas(45, 80, 8, 2.2)
                cv.pixels[:] = -1.0
                tor.render(cv, cam, scene.list(), 50, tor.make_options(seeding=seeding, accel=3, devices=devices, gather=gather,
                                                                          row_tile=row_tile))
                assert np.array_equal(cv.pixels, one.pixels), (seeding, devices, gather)
    # duplicate ordinals cannot form an RCCL communicator: explicit RCCL is refused, loudly
    with pytest.raises(tor.TorError):
        tor.render(tor.new_canvas(8, 8, 1), cam, scene.list(), 50, tor.make_options(devices=[0, 0], gather=tor.GATHER_RCCL))
    with pytest.raises(tor.TorError):
        tor.render(tor.new_canvas(8, 8, 1), cam, scene.list(), 50, tor.make_options(devices=[0, 99]))
    # a 1080p frame through two contexts (host-side assembly of 50 MB)
    one = tor.new_canvas(1080, 1920, 4, 2.2)
    tor.render(one, cam, scene.list(), 50, tor.make_options(seeding=tor.SEED_SAMPLE, accel=3))
    for gather in (tor.GATHER_PEER, tor.GATHER_HOST):
        two = tor.new_canvas(1080, 1920, 4, 2.2)
        tor.render(two, cam, scene.list(), 50, tor.make_options(seeding=tor.SEED_SAMPLE, accel=3, devices=[0, 0], gather=gather))
        assert np.array_equal(one.pixels, two.pixels)


def test_device_list_from_environment(tor):
    """A host that keeps the reference's signature names its GPUs through TOR_DEVICES (tools: the C examples)."""
    exe_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(exe_dir, exist_ok=True)
    exe = os.path.join(exe_dir, "nim_shim_emulation_env")
    libdir = os.path.dirname(tor.LIB_PATH)
    subprocess.run(["gcc", "-O2", os.path.join(ROOT, "examples", "nim_shim_emulation.c"), "-L", libdir, "-ltor_mi355x", "-lm",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True, capture_output=True)
    outs = []
    for env_extra in ({}, {"TOR_DEVICES": "0,0,0", "TOR_GATHER": "peer"}, {"TOR_DEVICES": "all"}, {"TOR_DEFAULT_ACCEL": "0"}):
        r = subprocess.run([exe, "ptr"], capture_output=True, timeout=300, env=dict(os.environ, **env_extra))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(r.stdout)
    assert outs[0] == outs[1] == outs[2] == outs[3]


def test_nim_shim_emulation_reproduces_reference_image(tor, golden_dir, tmp_path):
    """examples/nim_shim_emulation.c: a C unit that never includes tor_render.h -- its own structs (Nim's layouts
    under Nim's names) and its own prototypes, as Nim emits them for the shim of INTEGRATION.md -- renders the
    reference's main() through tor_render (HittableList by value) and tor_render_ptr; both PPMs must be the
    reference's PNG."""
    if shutil.which("gcc") is None:
        pytest.skip("no host C compiler on this box")
    exe = str(tmp_path / "nim_shim_emulation")
    libdir = os.path.dirname(tor.LIB_PATH)
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "examples", "nim_shim_emulation.c"), "-L", libdir,
                    "-ltor_mi355x", "-lm", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True, capture_output=True)
    g = np.array(Image.open(os.path.join(golden_dir, "book2_motion_blur.png")).convert("RGB")).astype(np.int64)
    for mode in ("value", "ptr"):
        r = subprocess.run([exe, mode], capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        tok = r.stdout.split()
        assert tok[0] == b"P3" and (int(tok[1]), int(tok[2]), int(tok[3])) == (384, 216, 255)
        rgb = np.array(tok[4:], dtype=np.int64).reshape(216, 384, 3)
        assert int((rgb != g).sum()) == 0, mode


def test_scene_cache_and_host_canvas_region(tor):
    """'callee may cache device copies'.  An identical object list uploads nothing; layouts a launch
    does not use are never built; a changed list is noticed.  And the SURVEY 8(d) region on a host canvas (what a
    Nim caller pays) stays close to the resident rate at C2 -- recorded in gpurun_out/host_canvas_rate.json."""
    import json
    import time
    import torch
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    ctx = tor.Context()
    ctx.upload(scene.list())
    assert ctx.scene_counters() == (1, 0, 0)          # nothing built yet
    buf = torch.empty((27, 48, 3), dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    ctx.render_device(cam, 27, 48, 4, 2.2, 50, tor.make_options(seeding=tor.SEED_SAMPLE), buf.data_ptr(), s)
    torch.cuda.synchronize()
    assert ctx.scene_counters() == (1, 0, 1)          # only the float64 flat layout
    ctx.upload(scene.list())
    ctx.upload(tor.random_scene(0xFACADE).list())     # another buffer, same bytes
    assert ctx.scene_counters() == (3, 2, 1)
    base = buf.clone()
    ctx.render_device(cam, 27, 48, 4, 2.2, 50, tor.make_options(seeding=tor.SEED_SAMPLE, accel=3), buf.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(buf, base) and ctx.scene_counters()[2] == 2
    other = tor.random_scene(0xBEEF)
    ctx.upload(other.list())
    assert ctx.scene_counters()[1] == 2
    ctx.render_device(cam, 27, 48, 4, 2.2, 50, tor.make_options(seeding=tor.SEED_SAMPLE, accel=3), buf.data_ptr(), s)
    torch.cuda.synchronize()
    assert not torch.equal(buf, base)
    ctx.upload(scene.list())
    ctx.render_device(cam, 27, 48, 4, 2.2, 50, tor.make_options(seeding=tor.SEED_SAMPLE, accel=1), buf.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(buf, base)
    ctx.close()
    # host canvas vs resident, C2 (1920x1080x100, per-sample streams, float64 brute force)
    h, w, spp = 1080, 1920, 100
    opt = tor.make_options(seeding=tor.SEED_SAMPLE)
    cv = tor.new_canvas(h, w, spp, 2.2)
    tor.render(cv, cam, scene.list(), 50, opt)
    first = tor.last_render_timing()
    t = time.perf_counter()
    for _ in range(3):
        tor.render(cv, cam, scene.list(), 50, opt)
    host = h * w * spp * 3 / (time.perf_counter() - t) / 1e6
    steady = tor.last_render_timing()
    assert steady["scene_cache_hit"] and steady["upload_ms"] < 1.0
    ctx = tor.Context()
    ctx.upload(scene.list())
    dev = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
    ctx.render_device(cam, h, w, spp, 2.2, 50, opt, dev.data_ptr(), s)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        ctx.render_device(cam, h, w, spp, 2.2, 50, opt, dev.data_ptr(), s)
    torch.cuda.synchronize()
    resident = h * w * spp * 3 / (time.perf_counter() - t) / 1e6
    assert np.array_equal(cv.pixels, dev.cpu().numpy())
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "host_canvas_rate.json"), "w") as f:
        json.dump({"workload": f"{w}x{h}x{spp}spp sample/strict brute force", "host_canvas_msamples_per_s": host,
                   "resident_msamples_per_s": resident, "ratio": host / resident, "first_call_ms": first, "steady_call_ms": steady}, f)
    assert host > 0.9 * resident, (host, resident)
    ctx.close()


def test_library_rccl_gather_single_rank(tor):
    """tor_comm_* + tor_render_gather_device: RCCL is loaded (dlopen) and a communicator is created inside the
    library on this GPU (world = 1 is all a 1-GPU box can form: RCCL refuses two ranks on one device); the
    gathered frame == the plain render, for the send/recv gather (root 0) and for ncclAllGather (root -1).
    Without a communicator the call degenerates to render + de-interleave (also checked)."""
    import torch
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    ctx = tor.Context()
    ctx.upload(scene.list())
    h, w, spp = 45, 80, 8
    s = torch.cuda.current_stream().cuda_stream
    opt = tor.make_options(seeding=tor.SEED_SAMPLE, accel=3, row_tile=4)
    plain = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
    ctx.render_device(cam, h, w, spp, 2.2, 50, tor.make_options(seeding=tor.SEED_SAMPLE, accel=3), plain.data_ptr(), s)
    frame = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    ctx.render_gather_device(cam, h, w, spp, 2.2, 50, opt, 0, frame.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(frame, plain)
    uid = tor.comm_unique_id()
    assert len(uid) == 128
    ctx.comm_init_rank(uid, 0, 1)
    for root in (0, -1):
        frame.zero_()
        ctx.render_gather_device(cam, h, w, spp, 2.2, 50, opt, root, frame.data_ptr(), s)
        torch.cuda.synchronize()
        assert torch.equal(frame, plain), root
    ctx.comm_destroy()
    ctx.close()


def test_gather_rows_kernel_inverts_the_shard_mapping(tor):
    """The de-interleave kernel of the multi-GPU assembly against tor_shard_rows for ragged splits (world sizes
    that do not divide the row count, row tiles with a partial last tile): shards rendered one by one on this GPU,
    packed rank-major exactly as the gather delivers them, must de-interleave to the single-device frame.
    (Drives the kernel through tor_render_opt's TOR_GATHER_PEER path with repeated ordinals.)"""
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    one = tor.new_canvas(37, 45, 2, 2.2)
    tor.render(one, cam, scene.list(), 8, tor.make_options(seeding=tor.SEED_SAMPLE))
    for n_dev, tile in ((3, 1), (5, 4), (7, 3), (16, 1), (2, 37), (4, 64)):
        cv = tor.new_canvas(37, 45, 2, 2.2)
        cv.pixels[:] = -3.0
        tor.render(cv, cam, scene.list(), 8, tor.make_options(seeding=tor.SEED_SAMPLE, devices=[0] * n_dev, row_tile=tile,
                                                              gather=tor.GATHER_PEER))
        assert np.array_equal(cv.pixels, one.pixels), (n_dev, tile)


def test_wild_rays_never_enter_padding_super_boxes(tor):
    """with TOR_ACCEL_BLOCKS|TOR_ACCEL_F32 on a two-level scene (> 96 blocks) whose number of
    super boxes is not a multiple of 8, a 'wild' ray (origin farther than 2^20 from the scene: outside the float32
    filter's guarded range) used to be sent into the PADDING super boxes, whose block boxes and records do not
    exist.  A camera 3e6 units away makes every primary ray wild; the canvas must equal the brute-force one (and
    the run must not fault)."""
    rng = np.random.default_rng(11)
    n = 900                                             # 113 blocks -> 15 super boxes (padded to 16)
    recs = []
    for i in range(n):
        x, z = rng.uniform(-20, 20, 2)
        y = rng.uniform(0.0, 3.0)
        mat = i % 3
        recs.append([0, x, y, z, x, y, z, 0, 1, 0.3, mat, .6, .5, .4, 0.1, 1.5])
    scene = tor.Scene.from_records(np.asarray(recs, dtype=np.float64))
    lay = tor.debug_accel_layout(scene.list(), 0.0, 1.0)
    assert lay is not None and lay[3] and (len(lay[2]) % 8) != 0     # two levels, ragged super count
    for look_from in ((3.0e6, 1.0e6, 2.0e6), (40.0, 12.0, 30.0)):
        dist = float(np.linalg.norm(look_from))
        cam = tor.camera(look_from=look_from, look_at=(0, 1, 0), vertical_field_of_view=0.002 if dist > 1e5 else 40.0,
                         aperture=0.0, focus_distance=dist)
        for seeding in (0, 1):
            base = tor.new_canvas(24, 40, 4, 2.2)
            tor.render(base, cam, scene.list(), 6, tor.make_options(seeding=seeding, accel=0))
            for accel in (1, 2, 3):
                cv = tor.new_canvas(24, 40, 4, 2.2)
                tor.render(cv, cam, scene.list(), 6, tor.make_options(seeding=seeding, accel=accel))
                assert np.array_equal(cv.pixels, base.pixels), (look_from, seeding, accel)
        assert base.pixels.std() > 1e-3


def test_wave_per_pixel_kernel_matches_lane_kernel_and_oracle(tor, oracle, ref_scene, ref_camera):
    """TOR_SEED_PIXEL has two kernels (TorOptions.pixel_kernel): one LANE per pixel chain (large frames) and one
    WAVE per pixel chain with the object loop split across the 64 lanes (small frames, where the chain latency
    binds).  Same canvas, bit for bit, and == oracle: random_scene, row shards, depth limits,
    the edge-case scene (time groups, duplicates -> tie to the lowest index, hollow sphere, time0 == time1), a
    1300-object scene (one workgroup per CU of LDS) and a 2100-object scene (does not fit LDS: WAVE falls back)."""
    objs, _ = ref_scene
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    K = (tor.PIXEL_KERNEL_LANE, tor.PIXEL_KERNEL_WAVE)

    def both(scene_, cam_, h, w, spp, depth=50, **opt):
        out = []
        for k in K:
            cv = tor.new_canvas(h, w, spp, 2.2)
            tor.render(cv, cam_, scene_.list(), depth, tor.make_options(seeding=tor.SEED_PIXEL, pixel_kernel=k, **opt))
            out.append(cv.pixels)
        assert np.array_equal(out[0], out[1]), f"lane and wave kernels differ in {(out[0] != out[1]).sum()} values"
        return out[1]

    for arith in (0,):
        got = both(scene, cam, 36, 64, 16, arith=arith)
        _exact(got, oracle.render(36, 64, 16, ref_camera, objs, seeding=0, math=1, arith=arith).pixels)
    got = both(scene, cam, 37, 45, 33, shard_index=1, shard_count=3, row_tile=4)
    rows = tor.shard_rows(37, 4, 1, 3)
    _exact(got[rows], oracle.render(37, 45, 33, ref_camera, objs, seeding=0, math=1, arith=0).pixels[rows])
    for depth in (1, 2, 7):
        _exact(both(scene, cam, 9, 16, 4, depth), oracle.render(9, 16, 4, ref_camera, objs, max_depth=depth, seeding=0, math=1).pixels)
    recs = np.asarray([
        [0, 0, -1000, 0, 0, -1000, 0, 0, 1, 1000, 0, .5, .5, .5, 0, 0],
        [1, 0, 1, 0, 0, 1.5, 0, 0.0, 1.0, 1.0, 0, .8, .3, .3, 0, 0],
        [1, -4, 1, 0, -4, 1, 1, 0.25, 0.75, 1.0, 1, .7, .6, .5, 0.3, 0],
        [1, 4, 1, 0, 5, 1, 0, 0.25, 0.75, 1.0, 2, 0, 0, 0, 0, 1.5],
        [1, 2, .5, 2, 2, .5, 2, 0.5, 0.5, 0.5, 0, .1, .9, .1, 0, 0],
        [0, 1, .4, 3, 1, .4, 3, 0, 1, 0.4, 1, .9, .9, .9, 0.0, 0],
        [0, 1, .4, 3, 1, .4, 3, 0, 1, 0.4, 0, .2, .2, .9, 0.0, 0],
        [0, -1, .3, 2, -1, .3, 2, 0, 1, -0.3, 2, 0, 0, 0, 0, 1.5],
    ], dtype=np.float64)
    edge = tor.Scene.from_records(recs)
    for (h, w, spp, depth) in ((24, 40, 8, 50), (2, 2, 4, 50), (16, 16, 1, 50)):
        _exact(both(edge, cam, h, w, spp, depth), oracle.render(h, w, spp, ref_camera, recs, max_depth=depth, seeding=0, math=1).pixels)
    rng = np.random.default_rng(99)
    for n in (1300, 2100):
        big = []
        for i in range(n):
            x, z = rng.uniform(-15, 15, 2)
            if i % 2:
                big.append([1, x, .2, z, x + rng.uniform(-.3, .3), .2 + rng.uniform(0, .5), z, 0.0, 1.0, .2, i % 3, .6, .5, .4, .2, 1.5])
            else:
                big.append([0, x, .2, z, x, .2, z, 0, 1, .2, i % 3, .3, .6, .8, .1, 1.4])
        big = np.asarray(big, dtype=np.float64)
        bscene = tor.Scene.from_records(big)
        got = both(bscene, cam, 12, 20, 3, 10)
        _exact(got, oracle.render(12, 20, 3, ref_camera, big, max_depth=10, seeding=0, math=1).pixels)
    # the reference's own main() (C1) through both kernels
    c1 = both(scene, cam, 216, 384, 100, accel=3)
    _exact(c1, oracle.render(216, 384, 100, ref_camera, objs, seeding=0, math=1, arith=0).pixels)


def test_split_mode_lane_and_wave_kernels_share_a_frame(tor, oracle, ref_scene, ref_camera):
    """Mid-size TOR_SEED_PIXEL frames with both accelerations (tor_render()'s default) run in split mode: the cost
    probe orders the tiles, the most expensive ones go to the wave-per-pixel kernel on a second stream, the lane
    kernel renders the others at the same time.  Same canvas as the lane kernel alone and as the oracle -- for a
    pixel count that is not a multiple of the 64-pixel tile, for a row shard, and for the reference's main()."""
    objs, _ = ref_scene
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    for (h, w, spp, shard) in ((129, 131, 33, None), (216, 384, 100, None), (300, 171, 32, (1, 2, 4))):
        opt = dict(accel=3)
        if shard:
            opt.update(shard_index=shard[0], shard_count=shard[1], row_tile=shard[2])
        auto = tor.new_canvas(h, w, spp, 2.2)
        auto.pixels[:] = -1.0
        tor.render(auto, cam, scene.list(), 50, tor.make_options(**opt))                      # AUTO -> split mode
        lane = tor.new_canvas(h, w, spp, 2.2)
        lane.pixels[:] = -1.0
        tor.render(lane, cam, scene.list(), 50, tor.make_options(pixel_kernel=tor.PIXEL_KERNEL_LANE, **opt))
        assert np.array_equal(auto.pixels, lane.pixels), (h, w, spp, shard)
        rows = tor.shard_rows(h, shard[2], shard[0], shard[1]) if shard else np.arange(h)
        want = oracle.render(h, w, spp, ref_camera, objs, seeding=0, math=1, arith=0).pixels
        _exact(auto.pixels[rows], want[rows])
    # twice in a row on one context (ring slots, second stream, events), interleaved with a plain launch
    a = tor.new_canvas(216, 384, 40, 2.2); b = tor.new_canvas(216, 384, 40, 2.2); c = tor.new_canvas(64, 64, 4, 2.2)
    tor.render(a, cam, scene.list(), 50)
    tor.render(c, cam, scene.list(), 50)
    tor.render(b, cam, scene.list(), 50)
    assert np.array_equal(a.pixels, b.pixels)


def test_pixel_schedule_never_changes_a_pixel(tor, oracle, ref_scene, ref_camera):
    """The TOR_SEED_PIXEL lane kernel's schedule (tiles ordered by their longest probed chain, arbiter priorities) only
    decides WHO renders a pixel and WHEN.  A frame large enough to occupy every hardware wave slot (> 3072 tiles, pixel count
    not a multiple of the tile) must come out identical for every setting of the knobs -- including ones that make nearly
    every chain hot or switch the machinery off -- and equal to the oracle.  (Round 5: the two-region cut of the order,
    TOR_BACK_SLOT / TOR_TAIL_FRAC / TOR_BACK_ACCEL, is gone.)"""
    objs, _ = ref_scene
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    h, w, spp = 385, 515, 32
    rows = np.array([0, 97, 200, 384])
    want = np.stack([oracle.render(h, w, spp, ref_camera, objs, seeding=0, math=1, arith=0, rows=(int(r), int(r) + 1)).pixels[r]
                     for r in rows])
    knobs = [
        {},
        {"TOR_HOT_FRAC": "0.02", "TOR_PRIO_SHIFT": "6"},
        {"TOR_HOT_FRAC": "0", "TOR_PRIO_SHIFT": "0"},
        {"TOR_LPT_MIN_SPP": "0"},
        {"TOR_BLOCKS_PER_CU": "2", "TOR_WAVES_PER_SIMD": "2"},
    ]
    names = sorted({k for d in knobs for k in d})
    saved = {k: os.environ.get(k) for k in names}
    first = {}
    try:
        for d in knobs:
            for k in names:
                os.environ.pop(k, None)
            os.environ.update(d)
            ctx = tor.Context(0)
            ctx.upload(scene.list())
            for accel in (0, 3):
                got = _device_render(tor, ctx, cam, h, w, spp, seeding=tor.SEED_PIXEL, accel=accel,
                                     pixel_kernel=tor.PIXEL_KERNEL_LANE).cpu().numpy()
                if accel not in first:
                    first[accel] = got
                    _exact(got[rows], want)
                assert np.array_equal(got, first[accel]), (d, accel)
            # a row shard through the same schedule
            part = _device_render(tor, ctx, cam, h, w, spp, seeding=tor.SEED_PIXEL, accel=0, pixel_kernel=tor.PIXEL_KERNEL_LANE,
                                  shard_index=1, shard_count=3, row_tile=2).cpu().numpy()
            assert np.array_equal(part, first[0][tor.shard_rows(h, 2, 1, 3)]), d
            if d == {}:
                cost = ctx.last_pixel_cost(h * w)   # the probe of the last launch: the shard's pixels
                assert len(cost) == len(tor.shard_rows(h, 2, 1, 3)) * w and cost.min() >= 2 and cost.max() <= 2 * 50
            del ctx
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    assert np.array_equal(first[0], first[3])
