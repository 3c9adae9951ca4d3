"""The drop-in boundary without a GPU: libtor_mi355x.so loads, exports every symbol that
include/tor_render.h declares, mirrors the reference's struct layouts, and refuses to render on a
machine without a HIP device (there is no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "tor_render.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"TOR_API\s+[\w\s\*]+?\b(tor_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol(tor):
    names = _declared_symbols()
    assert len(names) >= 25 and "tor_render" in names and "tor_render_device" in names
    L = tor.lib()
    for n in names:
        assert hasattr(L, n), f"{n} is declared in include/tor_render.h but not exported"
    assert sorted(tor.EXPORTED_SYMBOLS) == names, "the Python binding list drifted from the header"
    assert L.tor_version().startswith(b"tor_mi355x")


def test_struct_layouts_mirror_the_reference(tor):
    # sizes Nim's C backend derives on x86-64: Canvas 24, Camera 192, HittableList 16,
    # HittableVariant 120 (kind @0, union @8), Sphere 72, MovingSphere 112, Material 40 (kind @0, union @8)
    assert C.sizeof(tor.CanvasStruct) == 24 and C.sizeof(tor.Camera) == 192 and C.sizeof(tor.HittableList) == 16
    assert C.sizeof(tor.HittableVariant) == 120 and tor.HittableVariant.u.offset == 8
    assert C.sizeof(tor.Sphere) == 72 and C.sizeof(tor.MovingSphere) == 112
    assert C.sizeof(tor.Material) == 40 and tor.Material.u.offset == 8
    assert tor.MovingSphere.time0.offset == 48 and tor.MovingSphere.radius.offset == 64 and tor.MovingSphere.material.offset == 72
    assert tor.Camera.lens_radius.offset == 168 and tor.Camera.shutter_close.offset == 184


def test_host_mirrors_and_row_sharding_work_without_a_gpu(tor, oracle):
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    objs, _ = oracle.random_scene(0xFACADE)
    assert len(scene) == 485 and np.array_equal(scene.to_records(), objs)
    assert np.array_equal(cam.as_array(), oracle.camera())
    assert tor.selftest_rng(1, 0xFACADE, n=1)[1] == [0xff30ded049ef2d99]
    rows = [tor.shard_rows(1080, 1, k, 8) for k in range(8)]      # row-cyclic: the bench default
    assert sorted(np.concatenate(rows).tolist()) == list(range(1080)) and all(len(r) == 135 for r in rows)
    rows = [tor.shard_rows(1080, 8, k, 8) for k in range(8)]      # tiles of 8 rows do not divide evenly
    assert sorted(np.concatenate(rows).tolist()) == list(range(1080)) and sorted({len(r) for r in rows}) == [128, 136]
    cv = tor.new_canvas(4, 4, 1)
    cv.pixels[:] = [[[0.0, 0.5, 1.0]] * 4] * 4
    assert tor.export_rgb8(cv)[0, 0].tolist() == [0, 128, 255]


def test_rendering_fails_loudly_without_a_device(tor):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    cv = tor.new_canvas(8, 8, 1)
    cv.pixels[:] = 3.0
    with pytest.raises(tor.TorError) as e:
        tor.render(cv, cam, scene.list(), 5)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    assert np.all(cv.pixels == 3.0)
    with pytest.raises(tor.TorError):
        tor.Context()


def test_options_layout_and_versions(tor):
    """TorOptions: 108 bytes; the 32-byte round-1 layout is still accepted (its fields are a prefix), anything else
    is refused before a device is touched (this box has none: a well-formed call gets TOR_ERR_NO_DEVICE, never a
    CPU fallback)."""
    import ctypes as C
    assert C.sizeof(tor.Options) == 108 and tor.Options.device_count.offset == 32 and tor.Options.pixel_kernel.offset == 104
    scene = tor.random_scene(0xFACADE)
    cam = tor.camera()
    cv = tor.new_canvas(4, 4, 1, 2.2)
    cv.pixels[:] = 5.0
    L = tor.lib()
    cs = cv.struct()
    codes = {}
    for size in (33, 0, 200, 32, 108):
        o = tor.make_options()
        o.struct_size = size
        codes[size] = L.tor_render_opt(C.byref(cs), C.byref(cam), scene.list(), 50, C.byref(o))
    assert codes[33] == -1 and codes[0] == -1 and codes[200] == -1            # TOR_ERR_INVALID_ARGUMENT
    import torch
    if not torch.cuda.is_available():
        assert codes[108] == -2 and codes[32] == -2                          # TOR_ERR_NO_DEVICE: there is no CPU fallback
        assert b"no CPU fallback" in L.tor_last_error()
        assert (cv.pixels == 5.0).all()                                       # never a partial canvas
    for bad in (dict(device_count=17), dict(gather=9), dict(pixel_kernel=3), dict(device_count=2, shard_count=2)):
        o = tor.make_options()
        for k, v in bad.items():
            setattr(o, k, v)
        assert L.tor_render_opt(C.byref(cs), C.byref(cam), scene.list(), 50, C.byref(o)) == -1, bad
    assert L.tor_render_ptr(C.byref(cs), C.byref(cam), None, 50) == -1


def test_bad_settings_say_why_before_a_device_is_touched(tor, monkeypatch):
    """Round 3: malformed TorOptions and TOR_* settings are TOR_ERR_INVALID_ARGUMENT with a reason in
    tor_last_error() -- never a silent fall-back -- and the check comes before any device work (so it runs here, without a GPU)."""
    import ctypes as C
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    cv = tor.new_canvas(4, 4, 1, 2.2)
    cv.pixels[:] = 7.0
    for kw, word in ((dict(devices=[0, 0], shard_index=1, shard_count=2), "device list"), (dict(devices=[0, 1], device=5), "device"),
                     (dict(row_tile=-3), "row_tile"), (dict(accel=9), "accel"), (dict(seeding=5), "seeding"),
                     # round 5: the fused-arithmetic variants are gone; the enum value stays reserved and old callers fail loudly
                     (dict(arith=tor.ARITH_FUSED), "TOR_ARITH_FUSED was removed"), (dict(arith=2), "arith")):
        with pytest.raises(tor.TorError) as e:
            tor.render(cv, cam, scene.list(), 5, tor.make_options(**kw))
        assert e.value.code == -1 and word in str(e.value), (kw, str(e.value))
    for name, val in (("TOR_GATHER", "carrier-pigeon"), ("TOR_DEFAULT_ACCEL", "7"), ("TOR_DEFAULT_ACCEL", "x"), ("TOR_DEVICES", "0,abc"),
                      ("TOR_DEFAULT_SEEDING", "per-photon")):
        monkeypatch.setenv(name, val)
        with pytest.raises(tor.TorError) as e:
            tor.render(cv, cam, scene.list(), 5)            # tor_render(): the settings come from the environment
        assert e.value.code == -1 and name in str(e.value), (name, str(e.value))
        monkeypatch.delenv(name)
    assert np.all(cv.pixels == 7.0)
    assert tor.last_note() == "" or tor.last_note().startswith("gather:") or "gather" in tor.last_note()
    out = (C.c_uint64 * 16)()
    assert tor.lib().tor_last_handoff_counters(None, out) == -1


def test_h264_sizes_that_are_not_whole_macroblocks(tor):
    """Round 3: 1080 rows (BASELINE configs[4]) is 67.5 macroblocks.  Padded + cropped by the SPS (H.264 7.4.2.1.1); odd sizes
    have no 4:2:0 frame."""
    assert tor.h264_frame_bytes(1920, 1080) == 120 * 68 * 386 + 8
    assert tor.h264_frame_bytes(1920, 1088) == 120 * 68 * 386 + 8
    assert len(tor.h264_stream_header(1920, 1080)) > len(tor.h264_stream_header(1920, 1088))   # frame_cropping_flag + offsets
    for w, h in ((1921, 1080), (1920, 1081), (0, 0), (-16, 16)):
        with pytest.raises(tor.TorError):
            tor.h264_frame_bytes(w, h)
