"""ctypes loader for the CPU oracle (oracle/tor_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

SEED_PIXEL, SEED_SAMPLE = 0, 1
MATH_LIBM, MATH_PORTABLE = 0, 1
ARITH_STRICT, ARITH_FUSED = 0, 1
ACCUM_SEQUENTIAL, ACCUM_QUANTIZED = 0, 1


class OracleOptions(C.Structure):
    _fields_ = [
        ("seeding", C.c_int32), ("math", C.c_int32), ("arith", C.c_int32), ("accum", C.c_int32),
        ("row_begin", C.c_int32), ("row_end", C.c_int32), ("threads", C.c_int32),
        ("collect_stats", C.c_int32), ("row_step", C.c_int32), ("col_block", C.c_int32),
    ]


class OracleStats(C.Structure):
    _fields_ = [
        ("hit_calls", C.c_uint64), ("object_tests", C.c_uint64), ("rng_draws", C.c_uint64),
        ("scatter_lambertian", C.c_uint64), ("scatter_metal", C.c_uint64),
        ("scatter_dielectric", C.c_uint64), ("depth_exhausted", C.c_uint64),
        ("absorbed", C.c_uint64), ("depth_hist", C.c_uint64 * 64),
    ]


def build(force: bool = False) -> None:
    """Compile the oracle (gcc; seconds)."""
    target = os.path.join(_BUILD, "liboracle.so")
    src = os.path.join(_HERE, "tor_oracle.c")
    if force or not os.path.exists(target) or os.path.getmtime(target) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "all"], check=True, capture_output=True)


def _cpu_has_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            txt = f.read()
        flags = txt.split("flags", 1)[1].split("\n", 1)[0]
        return " fma" in flags and " avx2" in flags
    except Exception:
        return False


_lib = None


def lib(variant: str | None = None):
    """Load the oracle library. variant: None (auto), 'generic' or 'fma'."""
    global _lib
    if variant is None and _lib is not None:
        return _lib
    if not os.path.exists(os.path.join(_BUILD, "liboracle.so")):
        build()
    use_fma = _cpu_has_fma() if variant is None else (variant == "fma")
    name = "liboracle_fma.so" if use_fma else "liboracle.so"
    L = C.CDLL(os.path.join(_BUILD, name))
    dp = C.POINTER(C.c_double)
    u64p = C.POINTER(C.c_uint64)
    L.oracle_camera.argtypes = [dp, dp, dp] + [C.c_double] * 6 + [dp]
    L.oracle_camera.restype = None
    L.oracle_render.argtypes = [dp, C.c_int32, C.c_int32, C.c_int32, C.c_float, dp, dp, C.c_int64,
                                C.c_int32, C.POINTER(OracleOptions), C.POINTER(OracleStats)]
    L.oracle_render.restype = C.c_int
    L.oracle_random_scene.argtypes = [C.c_uint64, dp, C.c_int64, u64p]
    L.oracle_random_scene.restype = C.c_int64
    L.oracle_rng_seed1.argtypes = [C.c_uint64, u64p]
    L.oracle_rng_seed2.argtypes = [C.c_uint64, C.c_uint64, u64p]
    L.oracle_rng_seed3.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, u64p]
    L.oracle_rng_next.argtypes = [u64p]
    L.oracle_rng_next.restype = C.c_uint64
    L.oracle_rng_uniform01.argtypes = [u64p]
    L.oracle_rng_uniform01.restype = C.c_double
    L.oracle_rng_uniform_range.argtypes = [u64p, C.c_double, C.c_double]
    L.oracle_rng_uniform_range.restype = C.c_double
    L.oracle_port_sincos.argtypes = [dp, dp, dp, C.c_int64]
    L.oracle_libm_sincos.argtypes = [dp, dp, dp, C.c_int64]
    L.oracle_port_sincos_slow_dd.argtypes = [dp, dp, dp, dp, C.c_int64]
    L.oracle_port_pow5.argtypes = [dp, dp, C.c_int64]
    L.oracle_port_pow.argtypes = [dp, C.c_double, dp, C.c_int64]
    L.oracle_libm_pow.argtypes = [dp, C.c_double, dp, C.c_int64]
    L.oracle_quantize36.argtypes = [C.c_double]
    L.oracle_quantize36.restype = C.c_double
    L.oracle_quantize_ppm.argtypes = [dp, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]
    L.oracle_num_threads.restype = C.c_int
    u8p = C.POINTER(C.c_uint8)
    L.oracle_to_rgb_raw.argtypes = [dp, C.c_int32, C.c_int32, u8p]
    L.oracle_rgb_to_ycbcr420.argtypes = [C.c_int32, C.c_int32, u8p, u8p, u8p, u8p]
    L.oracle_h264_stream_header.argtypes = [C.c_int32, C.c_int32, u8p]
    L.oracle_h264_stream_header.restype = C.c_int
    L.oracle_h264_flush_frame.argtypes = [C.c_int32, C.c_int32, u8p, u8p, u8p, u8p]
    L.oracle_h264_flush_frame.restype = C.c_int64
    L.oracle_animation_create.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float]
    L.oracle_animation_create.restype = C.c_void_p
    L.oracle_animation_destroy.argtypes = [C.c_void_p]
    L.oracle_animation_object_count.argtypes = [C.c_void_p]
    L.oracle_animation_object_count.restype = C.c_int64
    L.oracle_animation_next.argtypes = [C.c_void_p, C.c_int32, dp, dp, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_float)]
    L.oracle_animation_next.restype = C.c_int
    if variant is None:
        _lib = L
    return L


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


# ---------------------------------------------------------------------------
# Mirrors of the reference's constructors (trace_of_radiance.nim:26-57)
# ---------------------------------------------------------------------------

OBJ_FIELDS = ("kind", "c0x", "c0y", "c0z", "c1x", "c1y", "c1z", "t0", "t1", "radius",
              "mat", "ax", "ay", "az", "fuzz", "ri")


def random_scene(seed: int = 0xFACADE, variant=None):
    """scenes.nim:13-50 -> (objs[n,16] float64, rng draws consumed)."""
    buf = np.zeros((2048, 16), dtype=np.float64)
    draws = C.c_uint64(0)
    n = lib(variant).oracle_random_scene(seed, _dp(buf), buf.shape[0], C.byref(draws))
    assert n > 0
    return np.ascontiguousarray(buf[:n]), int(draws.value)


def camera(look_from=(13, 2, 3), look_at=(0, 0, 0), vup=(0, 1, 0), vfov=20.0, aspect=16.0 / 9.0,
           aperture=0.1, focus_dist=10.0, shutter_open=0.0, shutter_close=1.0, variant=None):
    """cameras.nim:24-45 -> 24 float64 in the reference's field order."""
    out = np.zeros(24, dtype=np.float64)
    a = np.asarray(look_from, dtype=np.float64)
    b = np.asarray(look_at, dtype=np.float64)
    c = np.asarray(vup, dtype=np.float64)
    lib(variant).oracle_camera(_dp(a), _dp(b), _dp(c), vfov, aspect, aperture, focus_dist,
                               shutter_open, shutter_close, _dp(out))
    return out


@dataclass
class RenderResult:
    pixels: np.ndarray  # (nrows, ncols, 3) float64, row 0 = bottom scanline
    stats: OracleStats | None


def render(nrows, ncols, spp, cam, objs, max_depth=50, gamma=2.2, seeding=SEED_PIXEL,
           math=MATH_LIBM, arith=ARITH_STRICT, accum=ACCUM_SEQUENTIAL, rows=None, threads=0,
           collect_stats=False, variant=None, row_step=1, col_block=0) -> RenderResult:
    """render.nim:49-68 on the CPU.  col_block > 0 parallelises over (row, column block) tiles -- same pixels."""
    pixels = np.zeros((nrows, ncols, 3), dtype=np.float64)
    opt = OracleOptions(seeding, math, arith, accum, 0, nrows, threads, int(collect_stats), int(row_step), int(col_block))
    if rows is not None:
        opt.row_begin, opt.row_end = rows
    st = OracleStats() if collect_stats else None
    objs = np.ascontiguousarray(objs, dtype=np.float64)
    cam = np.ascontiguousarray(cam, dtype=np.float64)
    rc = lib(variant).oracle_render(_dp(pixels), nrows, ncols, spp, gamma, _dp(cam), _dp(objs),
                                    objs.shape[0], max_depth, C.byref(opt),
                                    C.byref(st) if st is not None else None)
    assert rc == 0
    return RenderResult(pixels, st)


def quantize_ppm(pixels: np.ndarray) -> np.ndarray:
    """io/ppm.nim:14-27 -> uint8 (nrows, ncols, 3), first row = top scanline."""
    nrows, ncols, _ = pixels.shape
    out = np.zeros((nrows, ncols, 3), dtype=np.uint8)
    p = np.ascontiguousarray(pixels, dtype=np.float64)
    lib().oracle_quantize_ppm(_dp(p), nrows, ncols, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def animation_scenes(height, width, dt=0.005, t_min=0.0, t_max=2.0, skip=6, seed=0xFACADE, max_frames=None):
    """scenes_animated.nim:90-225 -> yields (cam[24], objs[n,16], t) per frame."""
    L = lib()
    h = L.oracle_animation_create(seed, height, width, dt, t_min, t_max)
    try:
        n_obj = int(L.oracle_animation_object_count(h))
        k = 0
        while max_frames is None or k < max_frames:
            cam = np.zeros(24, dtype=np.float64)
            objs = np.zeros((n_obj, 16), dtype=np.float64)
            n = C.c_int64(0)
            t = C.c_float(0)
            rc = L.oracle_animation_next(h, skip, _dp(cam), _dp(objs), n_obj, C.byref(n), C.byref(t))
            if rc == 0:
                return
            assert rc == 1 and n.value == n_obj
            yield cam, objs, float(t.value)
            k += 1
    finally:
        L.oracle_animation_destroy(h)


def encode_frame(pixels: np.ndarray):
    """io/rgb.nim + io/color_conversions.nim + io/h264.nim on one canvas -> (rgb, Y, Cb, Cr, slice bytes)."""
    L = lib()
    u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
    nrows, ncols, _ = pixels.shape
    p = np.ascontiguousarray(pixels, dtype=np.float64)
    rgb = np.zeros((nrows, ncols, 3), dtype=np.uint8)
    L.oracle_to_rgb_raw(_dp(p), nrows, ncols, u8(rgb))
    Y = np.zeros((nrows, ncols), dtype=np.uint8)
    Cb = np.zeros((nrows // 2, ncols // 2), dtype=np.uint8)
    Cr = np.zeros((nrows // 2, ncols // 2), dtype=np.uint8)
    L.oracle_rgb_to_ycbcr420(ncols, nrows, u8(rgb), u8(Y), u8(Cb), u8(Cr))
    out = np.zeros(nrows * ncols * 2 + 64, dtype=np.uint8)
    n = L.oracle_h264_flush_frame(ncols, nrows, u8(Y), u8(Cb), u8(Cr), u8(out))
    return rgb, Y, Cb, Cr, out[:n].tobytes()


def h264_stream_header(width: int, height: int) -> bytes:
    buf = np.zeros(64, dtype=np.uint8)
    n = lib().oracle_h264_stream_header(width, height, buf.ctypes.data_as(C.POINTER(C.c_uint8)))
    return buf[:n].tobytes()
