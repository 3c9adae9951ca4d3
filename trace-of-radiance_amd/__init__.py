"""trace-of-radiance on MI355X: host-side mirror of the reference's render() interface.

The product is ``lib/libtor_mi355x.so`` (hand-written HIP kernels for gfx950 behind the C ABI
declared in ``include/tor_render.h``).  This module is the thin host layer above that ABI; it
mirrors the names of the reference's own interface for the path

    camera(...)                     physics/cameras.nim:24-45
    random_scene(seed)              scenes.nim:13-50 (+ trace_of_radiance.nim:34-36)
    new_canvas / Canvas             primitives/canvas.nim:20-41
    render(canvas, cam, world, max_depth)   render.nim:49
    export_rgb8 (exportToPPM's quantiser)   io/ppm.nim:14-27

so the parity tests read like the reference's ``main()`` (trace_of_radiance.nim:26-71).
There is NO CPU fallback: every rendering call raises ``TorError`` when the HIP extension or a
GPU is missing.  (The package directory name contains a hyphen; import it with
``importlib.import_module("trace-of-radiance_amd")``.)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtor_mi355x.so")
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

SEED_PIXEL, SEED_SAMPLE = 0, 1
ARITH_STRICT, ARITH_FUSED = 0, 1   # (ARITH_FUSED: removed in round 5; the library rejects it and says why)
ACCEL_NONE, ACCEL_BLOCKS, ACCEL_F32 = 0, 1, 2
GATHER_AUTO, GATHER_RCCL, GATHER_PEER, GATHER_HOST = 0, 1, 2, 3
PIXEL_KERNEL_AUTO, PIXEL_KERNEL_LANE, PIXEL_KERNEL_WAVE = 0, 1, 2
# return codes (include/tor_render.h)
OK, ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_OUT_OF_MEMORY, ERR_INCOMPLETE = 0, -1, -2, -3, -4, -5
MAX_DEVICES = 16
LAMBERTIAN, METAL, DIELECTRIC = 0, 1, 2
SPHERE, MOVING_SPHERE = 0, 1


class TorError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"tor_mi355x error {code}: {msg}")
        self.code = code


# --------------------------------------------------------------------------------------
# ABI structs (include/tor_render.h)
# --------------------------------------------------------------------------------------
class Vec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]


class _Lambertian(C.Structure):
    _fields_ = [("albedo", Vec3)]


class _Metal(C.Structure):
    _fields_ = [("albedo", Vec3), ("fuzz", C.c_double)]


class _Dielectric(C.Structure):
    _fields_ = [("refraction_index", C.c_double)]


class _MaterialU(C.Union):
    _fields_ = [("lambertian", _Lambertian), ("metal", _Metal), ("dielectric", _Dielectric)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("_pad", C.c_uint8 * 7), ("u", _MaterialU)]


class Sphere(C.Structure):
    _fields_ = [("center", Vec3), ("radius", C.c_double), ("material", Material)]


class MovingSphere(C.Structure):
    _fields_ = [("center0", Vec3), ("center1", Vec3), ("time0", C.c_double), ("time1", C.c_double),
                ("radius", C.c_double), ("material", Material)]


class _HittableU(C.Union):
    _fields_ = [("sphere", Sphere), ("moving_sphere", MovingSphere)]


class HittableVariant(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("_pad", C.c_uint8 * 7), ("u", _HittableU)]


class HittableList(C.Structure):
    _fields_ = [("len", C.c_int64), ("objects", C.POINTER(HittableVariant))]


class Camera(C.Structure):
    _fields_ = [("origin", Vec3), ("lower_left_corner", Vec3), ("horizontal", Vec3), ("vertical", Vec3),
                ("u", Vec3), ("v", Vec3), ("w", Vec3), ("lens_radius", C.c_double),
                ("shutter_open", C.c_double), ("shutter_close", C.c_double)]

    def as_array(self) -> np.ndarray:
        return np.frombuffer(bytes(self), dtype=np.float64).copy()


class CanvasStruct(C.Structure):
    _fields_ = [("pixels", C.POINTER(Vec3)), ("nrows", C.c_int32), ("ncols", C.c_int32),
                ("samples_per_pixel", C.c_int32), ("gamma_correction", C.c_float)]


class Options(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("seeding", C.c_int32), ("arith", C.c_int32),
                ("device", C.c_int32), ("shard_index", C.c_int32), ("shard_count", C.c_int32),
                ("row_tile", C.c_int32), ("accel", C.c_int32),
                ("device_count", C.c_int32), ("gather", C.c_int32), ("devices", C.c_int32 * MAX_DEVICES),
                ("pixel_kernel", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("hit_queries", C.c_uint64), ("object_tests", C.c_uint64), ("candidates", C.c_uint64),
                ("wave_iterations", C.c_uint64), ("lane_slots", C.c_uint64), ("samples", C.c_uint64),
                ("block_tests", C.c_uint64), ("exact_tests", C.c_uint64)]


assert C.sizeof(Vec3) == 24 and C.sizeof(Material) == 40 and C.sizeof(Sphere) == 72
assert C.sizeof(MovingSphere) == 112 and C.sizeof(HittableVariant) == 120
assert C.sizeof(HittableList) == 16 and C.sizeof(Camera) == 192 and C.sizeof(CanvasStruct) == 24

EXPORTED_SYMBOLS = [
    "tor_render", "tor_render_opt", "tor_last_error", "tor_context_create", "tor_context_destroy",
    "tor_scene_upload", "tor_shard_rows", "tor_render_device", "tor_quantize_rgb8_device",
    "tor_last_kernel_ms", "tor_kernel_ms_mean", "tor_context_set_stats", "tor_last_stats", "tor_last_wave_log", "tor_camera_init",
    "tor_random_scene", "tor_canvas_to_rgb8", "tor_animation_create", "tor_animation_destroy",
    "tor_animation_object_count", "tor_animation_next", "tor_h264_stream_header", "tor_h264_frame_bytes",
    "tor_encode_frame_device", "tor_render_frame_h264", "tor_mp4_mux_file", "tor_debug_accel_layout", "tor_selftest_filter32_host", "tor_selftest_screen_host", "tor_selftest_slab32_host", "tor_debug_filter32_scene", "tor_selftest_math_device", "tor_selftest_math_host",
    "tor_selftest_rng_host", "tor_version",
    "tor_last_render_timing", "tor_comm_unique_id", "tor_comm_init_rank", "tor_comm_destroy", "tor_render_gather_device",
    "tor_context_scene_counters", "tor_render_ptr", "tor_last_pixel_cost", "tor_last_note", "tor_last_handoff_counters",
    "tor_selftest_screen2_host", "tor_debug_screen2_scene", "tor_debug_layout_segments", "tor_knob_count", "tor_knob_info", "tor_last_gather_info", "tor_last_device_kernel_ms", "tor_comm_abort", "tor_comm_count", "tor_context_handoff_stalled",
]

_lib = None


def build(force: bool = False) -> str:
    """Compile libtor_mi355x.so for gfx950 in-tree (hipcc cross-compiles without a GPU).  Serialised by a
    file lock: the ranks of a multi-GPU launch may all find the library missing at the same time."""
    import fcntl
    src_dir = os.path.join(_HERE, "csrc")
    lock_path = os.path.join(src_dir, ".build.lock")
    if not os.access(src_dir, os.W_OK):  # read-only install: serialise through the temp dir instead
        import hashlib
        import tempfile
        lock_path = os.path.join(tempfile.gettempdir(), "tor_mi355x_" + hashlib.sha1(src_dir.encode()).hexdigest()[:12] + ".lock")
    with open(lock_path, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            cmd = ["make", "-C", src_dir] + (["-B"] if force or not os.path.exists(LIB_PATH) else []) + ["all"]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:  # show the compiler's own words
                raise TorError(-3, f"building libtor_mi355x.so failed ({' '.join(cmd)}):\n{r.stdout[-4000:]}\n{r.stderr[-8000:]}")
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def ensure_built() -> str:
    """Build the HIP extension if the in-tree library is missing (a fresh checkout); never a fallback --
    without hipcc this raises."""
    if not os.path.exists(LIB_PATH):
        build()
    return LIB_PATH


def lib():
    """Load the HIP extension; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TorError(-2, f"HIP extension missing: {LIB_PATH} (run __graft_entry__.build()); "
                           "there is no CPU fallback")
    # PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  Two HIP/HSA
    # runtimes in one process cannot both open the GPU, so when torch is installed it must be
    # loaded FIRST: the dynamic loader then binds this library's NEEDED libamdhip64.so.7 to the
    # already-loaded copy and device pointers / streams are shared with torch.
    if os.environ.get("TOR_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    # (harness-side A/B switch of the tools: TOR_AB_LIB = another build of THIS library -- e.g. last round's kernels -- to measure
    # against in the same gpurun call; never a fallback: unset, the in-tree library above is the only one there is)
    ab = os.environ.get("TOR_AB_LIB")
    if ab:  # never silent: every consumer of this process measures / validates ANOTHER binary
        print(f"trace-of-radiance_amd: TOR_AB_LIB set -- loading {ab} instead of the in-tree library", file=sys.stderr, flush=True)
    L = C.CDLL(ab or LIB_PATH)
    dp = C.POINTER(C.c_double)
    L.tor_last_error.restype = C.c_char_p
    L.tor_version.restype = C.c_char_p
    L.tor_last_note.restype = C.c_char_p
    L.tor_render.argtypes = [C.POINTER(CanvasStruct), C.POINTER(Camera), HittableList, C.c_int64]
    L.tor_render_ptr.argtypes = [C.POINTER(CanvasStruct), C.POINTER(Camera), C.POINTER(HittableList), C.c_int64]
    L.tor_render_opt.argtypes = [C.POINTER(CanvasStruct), C.POINTER(Camera), HittableList, C.c_int64,
                                 C.POINTER(Options)]
    L.tor_context_create.argtypes = [C.c_int32, C.POINTER(C.c_void_p)]
    L.tor_context_destroy.argtypes = [C.c_void_p]
    L.tor_scene_upload.argtypes = [C.c_void_p, HittableList]
    L.tor_shard_rows.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    L.tor_shard_rows.restype = C.c_int32
    L.tor_render_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int32, C.c_int32, C.c_int32,
                                    C.c_float, C.c_int64, C.POINTER(Options), C.c_void_p, C.c_void_p]
    L.tor_quantize_rgb8_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.tor_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int64)]
    L.tor_kernel_ms_mean.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    L.tor_context_set_stats.argtypes = [C.c_void_p, C.c_int32]
    L.tor_last_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.tor_last_wave_log.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
    L.tor_last_pixel_cost.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int64]
    L.tor_last_pixel_cost.restype = C.c_int64
    L.tor_last_handoff_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.tor_camera_init.argtypes = [C.POINTER(Camera), C.POINTER(Vec3), C.POINTER(Vec3), C.POINTER(Vec3)] + \
                                 [C.c_double] * 6
    L.tor_random_scene.argtypes = [C.c_uint64, C.POINTER(HittableVariant), C.c_int64]
    L.tor_random_scene.restype = C.c_int64
    L.tor_canvas_to_rgb8.argtypes = [C.POINTER(CanvasStruct), C.POINTER(C.c_uint8)]
    L.tor_animation_create.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float,
                                       C.POINTER(C.c_void_p)]
    L.tor_animation_destroy.argtypes = [C.c_void_p]
    L.tor_animation_destroy.restype = None
    L.tor_animation_object_count.argtypes = [C.c_void_p]
    L.tor_animation_object_count.restype = C.c_int64
    L.tor_animation_next.argtypes = [C.c_void_p, C.c_int32, C.POINTER(Camera), C.POINTER(HittableVariant), C.c_int64,
                                     C.POINTER(C.c_int64), C.POINTER(C.c_float)]
    L.tor_h264_stream_header.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_uint8), C.c_int32]
    L.tor_h264_frame_bytes.argtypes = [C.c_int32, C.c_int32]
    L.tor_h264_frame_bytes.restype = C.c_int64
    L.tor_encode_frame_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
    L.tor_render_frame_h264.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int64,
                                        C.POINTER(Options), C.POINTER(C.c_uint8), C.c_int64]
    L.tor_mp4_mux_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32]
    L.tor_debug_accel_layout.argtypes = [HittableList, C.c_double, C.c_double, C.POINTER(C.c_int64), C.c_int64,
                                         C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int32)]
    L.tor_selftest_filter32_host.argtypes = [C.c_int64] + [C.POINTER(C.c_double)] * 4 + [C.POINTER(C.c_int32)] + \
        [C.POINTER(C.c_double)] * 3 + [C.POINTER(C.c_int32)] * 2
    L.tor_selftest_screen_host.argtypes = [C.c_int64] + [C.POINTER(C.c_double)] * 4 + [C.POINTER(C.c_int32)] + \
        [C.POINTER(C.c_double)] * 2 + [C.POINTER(C.c_int32)] * 2
    L.tor_selftest_screen2_host.argtypes = [C.c_int64] + [C.POINTER(C.c_double)] * 4 + [C.POINTER(C.c_int32)] + \
        [C.POINTER(C.c_double)] * 2 + [C.c_int32] + [C.POINTER(C.c_int32)] * 2
    L.tor_selftest_slab32_host.argtypes = [C.c_int64] + [C.POINTER(C.c_double)] * 5 + [C.POINTER(C.c_int32)] * 2
    L.tor_debug_filter32_scene.argtypes = [HittableList, C.c_int64] + [C.POINTER(C.c_double)] * 3 + [C.POINTER(C.c_int8)]
    L.tor_debug_screen2_scene.argtypes = [HittableList, C.c_int64] + [C.POINTER(C.c_double)] * 3 + [C.POINTER(C.c_int8), C.POINTER(C.c_int32), C.POINTER(C.c_int8), C.c_int64]
    L.tor_debug_layout_segments.argtypes = [HittableList, C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int64)]
    L.tor_selftest_math_device.argtypes = [C.c_int32, dp, dp, dp, dp, C.c_int64, C.c_int32]
    L.tor_selftest_math_host.argtypes = [C.c_int32, dp, dp, dp, dp, C.c_int64]
    L.tor_selftest_rng_host.argtypes = [C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64,
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int64]
    L.tor_last_render_timing.argtypes = [dp]
    L.tor_comm_unique_id.argtypes = [C.POINTER(C.c_uint8)]
    L.tor_comm_init_rank.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32]
    L.tor_comm_destroy.argtypes = [C.c_void_p]
    L.tor_render_gather_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int64,
                                           C.POINTER(Options), C.c_int32, C.c_void_p, C.c_void_p]
    L.tor_context_scene_counters.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.tor_knob_count.restype = C.c_int32
    L.tor_knob_info.argtypes = [C.c_int32] + [C.POINTER(C.c_char_p)] * 5
    L.tor_last_gather_info.argtypes = [C.POINTER(C.c_int32)]
    L.tor_last_device_kernel_ms.argtypes = [C.POINTER(C.c_float), C.c_int32]
    L.tor_last_device_kernel_ms.restype = C.c_int32
    L.tor_comm_abort.argtypes = [C.c_void_p]
    L.tor_comm_count.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.tor_context_handoff_stalled.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    _lib = L
    return L


def _check(rc: int) -> None:
    if rc != 0:
        raise TorError(rc, lib().tor_last_error().decode("utf-8", "replace"))


# --------------------------------------------------------------------------------------
# Reference-interface mirrors
# --------------------------------------------------------------------------------------
def vec3(x, y, z) -> Vec3:
    return Vec3(float(x), float(y), float(z))


def camera(look_from=(13, 2, 3), look_at=(0, 0, 0), view_up=(0, 1, 0), vertical_field_of_view=20.0,
           aspect_ratio=16.0 / 9.0, aperture=0.1, focus_distance=10.0, shutter_open=0.0,
           shutter_close=1.0) -> Camera:
    """camera() -- physics/cameras.nim:24-45; defaults = trace_of_radiance.nim:38-51."""
    cam = Camera()
    a, b, c = vec3(*look_from), vec3(*look_at), vec3(*view_up)
    _check(lib().tor_camera_init(C.byref(cam), C.byref(a), C.byref(b), C.byref(c),
                                 vertical_field_of_view, aspect_ratio, aperture, focus_distance,
                                 shutter_open, shutter_close))
    return cam


class Scene:
    """Scene / HittableList -- physics/hittables/hittables_lists.nim:15-46 (owner + borrowed view)."""

    def __init__(self, objects=None, n: int = 0):
        self.objects = objects if objects is not None else (HittableVariant * 1)()
        self.n = n

    def list(self) -> HittableList:
        return HittableList(self.n, C.cast(self.objects, C.POINTER(HittableVariant)))

    def __len__(self):
        return self.n

    @staticmethod
    def from_records(recs: np.ndarray) -> "Scene":
        """Build from flat (n,16) float64 records {kind, c0 xyz, c1 xyz, t0, t1, radius, material, albedo rgb, fuzz, ri}
        (the interchange format of the parity tests)."""
        n = int(recs.shape[0])
        arr = (HittableVariant * max(n, 1))()
        for i in range(n):
            r = recs[i]
            h = arr[i]
            mat = Material()
            mat.kind = int(r[10])
            if mat.kind == LAMBERTIAN:
                mat.u.lambertian.albedo = vec3(r[11], r[12], r[13])
            elif mat.kind == METAL:
                mat.u.metal.albedo = vec3(r[11], r[12], r[13])
                mat.u.metal.fuzz = float(r[14])
            else:
                mat.u.dielectric.refraction_index = float(r[15])
            if int(r[0]) == SPHERE:
                h.kind = SPHERE
                h.u.sphere.center = vec3(r[1], r[2], r[3])
                h.u.sphere.radius = float(r[9])
                h.u.sphere.material = mat
            else:
                h.kind = MOVING_SPHERE
                h.u.moving_sphere.center0 = vec3(r[1], r[2], r[3])
                h.u.moving_sphere.center1 = vec3(r[4], r[5], r[6])
                h.u.moving_sphere.time0 = float(r[7])
                h.u.moving_sphere.time1 = float(r[8])
                h.u.moving_sphere.radius = float(r[9])
                h.u.moving_sphere.material = mat
        return Scene(arr, n)

    def to_records(self) -> np.ndarray:
        """The scene as flat (n,16) float64 records (see from_records)."""
        out = np.zeros((self.n, 16), dtype=np.float64)
        for i in range(self.n):
            h = self.objects[i]
            if h.kind == SPHERE:
                s = h.u.sphere
                c0 = c1 = s.center
                t0, t1, rad, m = 0.0, 1.0, s.radius, s.material
            else:
                s = h.u.moving_sphere
                c0, c1, t0, t1, rad, m = s.center0, s.center1, s.time0, s.time1, s.radius, s.material
            out[i, 0] = h.kind
            out[i, 1:4] = (c0.x, c0.y, c0.z)
            out[i, 4:7] = (c1.x, c1.y, c1.z)
            out[i, 7:10] = (t0, t1, rad)
            out[i, 10] = m.kind
            if m.kind == LAMBERTIAN:
                a = m.u.lambertian.albedo
                out[i, 11:14] = (a.x, a.y, a.z)
            elif m.kind == METAL:
                a = m.u.metal.albedo
                out[i, 11:14] = (a.x, a.y, a.z)
                out[i, 14] = m.u.metal.fuzz
            else:
                out[i, 15] = m.u.dielectric.refraction_index
        return out


def random_scene(seed: int = 0xFACADE) -> Scene:
    """random_scene(rng) with rng.seed(seed) -- scenes.nim:13-50."""
    cap = 2048
    arr = (HittableVariant * cap)()
    n = lib().tor_random_scene(seed, arr, cap)
    if n < 0:
        raise TorError(int(n), "tor_random_scene failed")
    return Scene(arr, int(n))


class Animation:
    """random_moving_spheres + `iterator scenes` -- trace_of_radiance/scenes_animated.nim:90-225.

    ``for cam, scene, t in Animation(h, w, dt, t_min, t_max).scenes(skip=6): render(canvas, cam, scene.list(), depth)``
    mirrors trace_of_radiance_animation.nim:84-97."""

    def __init__(self, height: int, width: int, dt: float = 0.005, t_min: float = 0.0, t_max: float = 2.0,
                 seed: int = 0xFACADE):
        self._h = C.c_void_p()
        _check(lib().tor_animation_create(seed, height, width, dt, t_min, t_max, C.byref(self._h)))
        self.n_objects = int(lib().tor_animation_object_count(self._h))

    def scenes(self, skip: int = 6):
        while True:
            cam = Camera()
            arr = (HittableVariant * self.n_objects)()
            n = C.c_int64(0)
            t = C.c_float(0)
            rc = lib().tor_animation_next(self._h, skip, C.byref(cam), arr, self.n_objects, C.byref(n), C.byref(t))
            if rc == 0:
                return
            if rc < 0:
                raise TorError(rc, "tor_animation_next failed")
            yield cam, Scene(arr, int(n.value)), float(t.value)

    def close(self):
        if self._h:
            lib().tor_animation_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def frames_of_rank(n_frames: int, rank: int, world: int):
    """Frame-parallel split of an animation: frame f is rendered by GPU f mod world."""
    return list(range(rank, n_frames, max(world, 1)))


class Canvas:
    """Canvas -- primitives/canvas.nim:20-41: row-major float64 RGB, row 0 = bottom scanline."""

    def __init__(self, height: int, width: int, samples_per_pixel: int, gamma_correction: float = 2.2):
        self.pixels = np.zeros((height, width, 3), dtype=np.float64)
        self.nrows, self.ncols = int(height), int(width)
        self.samples_per_pixel = int(samples_per_pixel)
        self.gamma_correction = float(gamma_correction)

    def struct(self) -> CanvasStruct:
        return CanvasStruct(self.pixels.ctypes.data_as(C.POINTER(Vec3)), self.nrows, self.ncols,
                            self.samples_per_pixel, self.gamma_correction)


def new_canvas(height, width, samples_per_pixel, gamma_correction=2.2) -> Canvas:
    return Canvas(height, width, samples_per_pixel, gamma_correction)


def make_options(seeding=SEED_PIXEL, arith=ARITH_STRICT, device=-1, shard_index=0, shard_count=1,
                 row_tile=1, accel=0, devices=None, gather=GATHER_AUTO, pixel_kernel=PIXEL_KERNEL_AUTO) -> Options:
    """TorOptions.  devices: a list of HIP ordinals -> tor_render_opt renders row shard k on devices[k] and
    assembles the frame in the canvas (`gather`)."""
    o = Options(C.sizeof(Options), seeding, arith, device, shard_index, shard_count, row_tile, accel)
    if devices:
        o.device_count = len(devices)
        for k, d in enumerate(devices):
            o.devices[k] = int(d)
    o.gather = gather
    o.pixel_kernel = pixel_kernel
    return o


def last_render_timing() -> dict:
    """Host-side cost of this thread's last render(): ms for upload, launch+kernels, download/gather, whole call."""
    t = (C.c_double * 5)()
    _check(lib().tor_last_render_timing(t))
    return {"upload_ms": t[0], "render_ms": t[1], "download_ms": t[2], "total_ms": t[3], "scene_cache_hit": bool(t[4])}


def last_note() -> str:
    """What the last multi-device render() on this thread chose / fell back to (tor_last_note)."""
    return lib().tor_last_note().decode("utf-8", "replace")


def knobs() -> list:
    """The library's environment knobs (csrc/tor_knobs.hpp): dicts {name, default, range, when, what}."""
    out = []
    for i in range(int(lib().tor_knob_count())):
        f = [C.c_char_p() for _ in range(5)]
        _check(lib().tor_knob_info(i, *[C.byref(x) for x in f]))
        out.append(dict(zip(("name", "default", "range", "when", "what"), (x.value.decode() for x in f))))
    return out


def last_gather_info() -> dict:
    """Facts about this thread's last multi-device render(): the gather leg, the RCCL communicator's rank count (0: not
    RCCL), entries of the device list, whether they were distinct GPUs (tor_last_gather_info)."""
    out = (C.c_int32 * 4)()
    _check(lib().tor_last_gather_info(out))
    return {"leg": {GATHER_RCCL: "rccl", GATHER_PEER: "peer", GATHER_HOST: "host"}.get(int(out[0]), "none"),
            "rccl_ranks": int(out[1]), "devices": int(out[2]), "distinct_devices": bool(out[3])}


def last_device_kernel_ms() -> list:
    """integrate_kernel's duration on every device of this thread's last multi-device render() (HIP events on each launch stream)."""
    buf = (C.c_float * 64)()
    n = int(lib().tor_last_device_kernel_ms(buf, 64))
    return [float(buf[k]) for k in range(min(n, 64))]


def comm_unique_id() -> bytes:
    buf = (C.c_uint8 * 128)()
    _check(lib().tor_comm_unique_id(buf))
    return bytes(buf)


def render(canvas: Canvas, cam: Camera, world: HittableList, max_depth: int, options: Options | None = None):
    """render(canvas, cam, world, max_depth) -- render.nim:49.  Blocking."""
    cs = canvas.struct()
    if options is None:
        _check(lib().tor_render(C.byref(cs), C.byref(cam), world, int(max_depth)))
    else:
        _check(lib().tor_render_opt(C.byref(cs), C.byref(cam), world, int(max_depth), C.byref(options)))


def export_rgb8(canvas: Canvas) -> np.ndarray:
    """exportToPPM's quantiser (io/ppm.nim:14-27): uint8 (nrows, ncols, 3), first row = top."""
    out = np.zeros((canvas.nrows, canvas.ncols, 3), dtype=np.uint8)
    cs = canvas.struct()
    _check(lib().tor_canvas_to_rgb8(C.byref(cs), out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out


def export_ppm(canvas: Canvas, f) -> None:
    """exportToPPM(canvas, f) -- io/ppm.nim:14-27 (ASCII P3)."""
    rgb = export_rgb8(canvas)
    f.write(f"P3\n{canvas.ncols} {canvas.nrows}\n255\n")
    for r, g, b in rgb.reshape(-1, 3):
        f.write(f"{r} {g} {b}\n")


def h264_stream_header(width: int, height: int) -> bytes:
    """SPS + PPS as H264Encoder.init writes them (io/h264.nim:90-142,37,174-176)."""
    buf = (C.c_uint8 * 64)()
    n = lib().tor_h264_stream_header(width, height, buf, 64)
    if n < 0:
        raise TorError(n, "tor_h264_stream_header failed")
    return bytes(buf[:n])


def h264_frame_bytes(width: int, height: int) -> int:
    n = int(lib().tor_h264_frame_bytes(width, height))
    if n < 0:
        raise TorError(n, "width and height must be multiples of 16")
    return n


def shard_rows(nrows: int, row_tile: int, shard_index: int, shard_count: int) -> np.ndarray:
    buf = (C.c_int32 * max(nrows, 1))()
    n = lib().tor_shard_rows(nrows, row_tile, shard_index, shard_count, buf)
    return np.array(buf[:n], dtype=np.int32)


class Context:
    """Resident device context (tor_context_* / tor_scene_upload / tor_render_device)."""

    def __init__(self, device: int = -1):
        self._h = C.c_void_p()
        self.last_incomplete = False   # set by last_kernel_ms(): the last launch's hand-off stalled, its frame is not whole
        _check(lib().tor_context_create(device, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().tor_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, world: HittableList):
        _check(lib().tor_scene_upload(self._h, world))

    def set_stats(self, enable: bool):
        _check(lib().tor_context_set_stats(self._h, int(enable)))

    def render_device(self, cam: Camera, nrows: int, ncols: int, spp: int, gamma: float, max_depth: int,
                      options: Options, d_pixels_ptr: int, stream_ptr: int = 0):
        """Asynchronous on the given hipStream_t; d_pixels_ptr is a device pointer."""
        _check(lib().tor_render_device(self._h, C.byref(cam), nrows, ncols, spp, gamma, int(max_depth),
                                       C.byref(options), C.c_void_p(d_pixels_ptr), C.c_void_p(stream_ptr)))

    def quantize_rgb8_device(self, d_pixels_ptr: int, n_values: int, d_rgb8_ptr: int, stream_ptr: int = 0):
        _check(lib().tor_quantize_rgb8_device(self._h, C.c_void_p(d_pixels_ptr), n_values,
                                              C.c_void_p(d_rgb8_ptr), C.c_void_p(stream_ptr)))

    def encode_frame_device(self, d_pixels_ptr: int, nrows: int, ncols: int, d_slice_ptr: int, d_y_ptr: int = 0,
                            d_cb_ptr: int = 0, d_cr_ptr: int = 0, stream_ptr: int = 0):
        """canvas -> RGB8 -> Y'CbCr 4:2:0 -> I_PCM slice bytes (io/rgb.nim, color_conversions.nim, h264.nim)."""
        _check(lib().tor_encode_frame_device(self._h, C.c_void_p(d_pixels_ptr), nrows, ncols, C.c_void_p(d_slice_ptr),
                                             C.c_void_p(d_y_ptr), C.c_void_p(d_cb_ptr), C.c_void_p(d_cr_ptr),
                                             C.c_void_p(stream_ptr)))

    def render_frame_h264(self, cam: Camera, nrows: int, ncols: int, spp: int, gamma: float, max_depth: int,
                          options: Options | None = None) -> bytes:
        """Render the uploaded scene and return the frame's I_PCM slice (animation driver loop body)."""
        n = h264_frame_bytes(ncols, nrows)
        buf = (C.c_uint8 * n)()
        _check(lib().tor_render_frame_h264(self._h, C.byref(cam), nrows, ncols, spp, gamma, int(max_depth),
                                           C.byref(options) if options is not None else None, buf, n))
        return bytes(buf)

    def scene_counters(self):
        """(uploads, cache hits, device layouts built)"""
        out = (C.c_int64 * 3)()
        _check(lib().tor_context_scene_counters(self._h, out))
        return int(out[0]), int(out[1]), int(out[2])

    def comm_init_rank(self, unique_id: bytes, rank: int, world: int):
        """ncclCommInitRank inside the library (one process per GPU; id from comm_unique_id() of rank 0)."""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _check(lib().tor_comm_init_rank(self._h, buf, rank, world))

    def comm_destroy(self):
        _check(lib().tor_comm_destroy(self._h))

    def comm_abort(self):
        """ncclCommAbort (a gather that does not complete: RCCL's kernels leave)."""
        _check(lib().tor_comm_abort(self._h))

    def comm_count(self) -> int:
        """Ranks of the context's RCCL communicator (ncclCommCount); 0 without one."""
        n = C.c_int32(0)
        _check(lib().tor_comm_count(self._h, C.byref(n)))
        return int(n.value)

    def handoff_stalled(self):
        """(stalled, frames re-rendered so far) of the last launch's chain hand-off (tor_context_handoff_stalled); blocks."""
        st = C.c_int32(0)
        tot = C.c_int64(0)
        _check(lib().tor_context_handoff_stalled(self._h, C.byref(st), C.byref(tot)))
        return bool(st.value), int(tot.value)

    def render_gather_device(self, cam: Camera, nrows: int, ncols: int, spp: int, gamma: float, max_depth: int,
                             options: Options, root: int, d_frame_ptr: int, stream_ptr: int = 0):
        """This rank's row shard + the RCCL framebuffer gather + de-interleave, asynchronous on the stream."""
        _check(lib().tor_render_gather_device(self._h, C.byref(cam), nrows, ncols, spp, gamma, int(max_depth),
                                              C.byref(options), root, C.c_void_p(d_frame_ptr), C.c_void_p(stream_ptr)))

    def last_kernel_ms(self):
        """(kernel ms by HIP events, pixel-samples traced) of the last launch.  TOR_ERR_INCOMPLETE -- the launch's chain hand-off
        stalled and flagged the frame -- does not raise: the timing is valid, `self.last_incomplete` says the frame is not."""
        ms = C.c_float(0)
        n = C.c_int64(0)
        rc = lib().tor_last_kernel_ms(self._h, C.byref(ms), C.byref(n))
        self.last_incomplete = (rc == ERR_INCOMPLETE)
        if rc != ERR_INCOMPLETE:
            _check(rc)
        return float(ms.value), int(n.value)

    def kernel_ms_mean(self, last_n: int):
        ms = C.c_float(0)
        n = C.c_int32(0)
        _check(lib().tor_kernel_ms_mean(self._h, last_n, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def last_pixel_cost(self, n_pixels: int) -> np.ndarray:
        """Per-pixel closest-hit query counts of the last launch's cost probe (debug; tor_last_pixel_cost)."""
        buf = np.zeros(n_pixels, dtype=np.uint32)
        n = lib().tor_last_pixel_cost(self._h, buf.ctypes.data_as(C.POINTER(C.c_uint32)), n_pixels)
        if n < 0:
            _check(int(n))
        return buf[:n]

    def last_handoff_counters(self) -> dict:
        """Chain hand-off of the last SEED_PIXEL launch (tor_last_handoff_counters)."""
        out = (C.c_uint64 * 16)()
        _check(lib().tor_last_handoff_counters(self._h, out))
        names = ("tickets", "pushed", "lane_waves_left", "server_workgroups", "push_threshold", "served", "hot_pushes", "tail_pushes",
                 "us_counter_dry", "us_lane_end", "us_hot_done", "us_tail_done", "its_hot", "its_tail", "servers_converted", "push_threshold_end")
        return {k: int(v) for k, v in zip(names, out)}

    def last_wave_log(self, cap_waves: int = 16384) -> np.ndarray:
        buf = np.zeros((cap_waves, 8), dtype=np.uint64)
        n = lib().tor_last_wave_log(self._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), cap_waves)
        if n < 0:
            _check(n)
        return buf[:n]

    def last_stats(self) -> Stats:
        st = Stats()
        _check(lib().tor_last_stats(self._h, C.byref(st)))
        return st


def mp4_mux_file(src_annexb_path: str, dst_mp4_path: str, width: int, height: int, fps: int = 30) -> int:
    """MP4Muxer (io/mp4.nim:113-163): Annex-B .264 -> .mp4; returns the number of samples."""
    n = lib().tor_mp4_mux_file(os.fsencode(src_annexb_path), os.fsencode(dst_mp4_path), width, height, fps)
    if n < 0:
        raise TorError(n, lib().tor_last_error().decode())
    return n


def debug_accel_layout(world: HittableList, t_lo: float, t_hi: float):
    """(slot_object[n_blocks, 8], block_boxes[n_blocks_padded, 6], super_boxes[n, 6], two_level) or None."""
    cap = int(world.len) + 64
    slots = np.full(cap, -1, dtype=np.int64)
    boxes = np.zeros((cap // 8 + 16, 6), dtype=np.float64)
    supers = np.zeros((cap // 64 + 16, 6), dtype=np.float64)
    two = C.c_int32(0)
    n = lib().tor_debug_accel_layout(world, t_lo, t_hi, slots.ctypes.data_as(C.POINTER(C.c_int64)), cap,
                                     boxes.ctypes.data_as(C.POINTER(C.c_double)),
                                     supers.ctypes.data_as(C.POINTER(C.c_double)), boxes.shape[0], C.byref(two))
    if n < 0:
        raise TorError(n, "tor_debug_accel_layout failed")
    if n == 0:
        return None
    n_p = (n + 7) // 8 * 8
    return slots[: n * 8].reshape(n, 8), boxes[:n_p], supers[: n_p // 8], bool(two.value)


def selftest_filter32(o, d, c0, dc, moving, f, r2, origin):
    """(keep, need) int32 arrays: host build of the TOR_ACCEL_F32 pre-filter vs the float64 test."""
    dp = lambda x: np.ascontiguousarray(x, dtype=np.float64)
    o, d, c0, dc, f, r2, origin = dp(o), dp(d), dp(c0), dp(dc), dp(f), dp(r2), dp(origin)
    moving = np.ascontiguousarray(moving, dtype=np.int32)
    n = len(f)
    keep = np.zeros(n, dtype=np.int32)
    need = np.zeros(n, dtype=np.int32)
    P = C.POINTER(C.c_double)
    I = C.POINTER(C.c_int32)
    _check(lib().tor_selftest_filter32_host(n, o.ctypes.data_as(P), d.ctypes.data_as(P), c0.ctypes.data_as(P),
                                            dc.ctypes.data_as(P), moving.ctypes.data_as(I), f.ctypes.data_as(P),
                                            r2.ctypes.data_as(P), origin.ctypes.data_as(P), keep.ctypes.data_as(I),
                                            need.ctypes.data_as(I)))
    return keep, need


def selftest_screen(o, d, c0, dc, moving, f, r2):
    """(keep, need) int32 arrays: host build of the strict object loop's conservative FMA screen vs the reference's test."""
    dp = lambda x: np.ascontiguousarray(x, dtype=np.float64)
    o, d, c0, dc, f, r2 = dp(o), dp(d), dp(c0), dp(dc), dp(f), dp(r2)
    moving = np.ascontiguousarray(moving, dtype=np.int32)
    n = len(f)
    keep = np.zeros(n, dtype=np.int32)
    need = np.zeros(n, dtype=np.int32)
    P = C.POINTER(C.c_double)
    I = C.POINTER(C.c_int32)
    _check(lib().tor_selftest_screen_host(n, o.ctypes.data_as(P), d.ctypes.data_as(P), c0.ctypes.data_as(P), dc.ctypes.data_as(P),
                                          moving.ctypes.data_as(I), f.ctypes.data_as(P), r2.ctypes.data_as(P), keep.ctypes.data_as(I),
                                          need.ctypes.data_as(I)))
    return keep, need


def selftest_screen2(o, d, c0, dc, moving, f, r2, variant=0):
    """(keep, need): the screen's second form (expanded quadratic, normalised direction) on the host; variant 1 routes static
    spheres through the common-height record."""
    dp = lambda x: np.ascontiguousarray(x, dtype=np.float64)
    o, d, c0, dc, f, r2 = dp(o), dp(d), dp(c0), dp(dc), dp(f), dp(r2)
    moving = np.ascontiguousarray(moving, dtype=np.int32)
    n = len(f)
    keep = np.zeros(n, dtype=np.int32)
    need = np.zeros(n, dtype=np.int32)
    P = C.POINTER(C.c_double)
    I = C.POINTER(C.c_int32)
    _check(lib().tor_selftest_screen2_host(n, o.ctypes.data_as(P), d.ctypes.data_as(P), c0.ctypes.data_as(P), dc.ctypes.data_as(P),
                                           moving.ctypes.data_as(I), f.ctypes.data_as(P), r2.ctypes.data_as(P), int(variant),
                                           keep.ctypes.data_as(I), need.ctypes.data_as(I)))
    return keep, need


def selftest_slab32(o, d, lo, hi, origin):
    """(keep, need) int32 arrays: host build of the float32 box test vs the float64 slab test."""
    dp = lambda x: np.ascontiguousarray(x, dtype=np.float64)
    o, d, lo, hi, origin = dp(o), dp(d), dp(lo), dp(hi), dp(origin)
    n = len(o)
    keep = np.zeros(n, dtype=np.int32)
    need = np.zeros(n, dtype=np.int32)
    P = C.POINTER(C.c_double)
    I = C.POINTER(C.c_int32)
    _check(lib().tor_selftest_slab32_host(n, o.ctypes.data_as(P), d.ctypes.data_as(P), lo.ctypes.data_as(P), hi.ctypes.data_as(P),
                                          origin.ctypes.data_as(P), keep.ctypes.data_as(I), need.ctypes.data_as(I)))
    return keep, need


def debug_filter32_scene(world: HittableList, o, d, time):
    """keep[n_rays, n_objects] int8 (1 kept, 0 dropped, 2 float64 loop): the TOR_ACCEL_F32 segment walk on the host."""
    dp = lambda x: np.ascontiguousarray(x, dtype=np.float64)
    o, d, time = dp(o), dp(d), dp(time)
    keep = np.zeros((len(time), int(world.len)), dtype=np.int8)
    P = C.POINTER(C.c_double)
    _check(lib().tor_debug_filter32_scene(world, len(time), o.ctypes.data_as(P), d.ctypes.data_as(P), time.ctypes.data_as(P),
                                          keep.ctypes.data_as(C.POINTER(C.c_int8))))
    return keep


def debug_layout_segments(world: HittableList, max_segs: int = 64):
    """[(xkind, objects, slots, first slot), ...]: the float64 layout's segments in the order the kernel walks them (host only)."""
    out = np.zeros((max_segs, 4), dtype=np.int32)
    n = C.c_int64(0)
    _check(lib().tor_debug_layout_segments(world, out.ctypes.data_as(C.POINTER(C.c_int32)), max_segs, C.byref(n)))
    return [tuple(int(v) for v in row) for row in out[:min(int(n.value), max_segs)]]


def debug_screen2_scene(world: HittableList, o, d, time, max_segs: int = 0):
    """(keep[n_rays, n_objects] int8, kind[n_objects] int32 [, pays[n_rays, max_segs] int8]): the strict layout's screened
    segments walked on the host -- 0 dropped by the plane screen, 1 dropped by stage two, 2 candidate, 3 no plane table; kind
    0 | 10..14; pays (with max_segs > 0): the rays' votes for stage one per segment (plane_pays)."""
    dp = lambda x: np.ascontiguousarray(x, dtype=np.float64)
    o, d, time = dp(o), dp(d), dp(time)
    keep = np.zeros((len(time), int(world.len)), dtype=np.int8)
    kind = np.zeros(int(world.len), dtype=np.int32)
    pays = np.full((len(time), max(max_segs, 1)), -1, dtype=np.int8)
    P = C.POINTER(C.c_double)
    _check(lib().tor_debug_screen2_scene(world, len(time), o.ctypes.data_as(P), d.ctypes.data_as(P), time.ctypes.data_as(P),
                                         keep.ctypes.data_as(C.POINTER(C.c_int8)), kind.ctypes.data_as(C.POINTER(C.c_int32)),
                                         pays.ctypes.data_as(C.POINTER(C.c_int8)) if max_segs > 0 else None, max_segs))
    return (keep, kind, pays) if max_segs > 0 else (keep, kind)


def selftest_math(op: int, x: np.ndarray, y: np.ndarray | None = None, where: str = "device", device: int = -1):
    """Run the kernel's math routines on arrays (op: 0 sincos, 1 x^5, 2 pow, 3 sqrt, 4 div, 5 quantize)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    yy = np.ascontiguousarray(y, dtype=np.float64) if y is not None else None
    o0 = np.zeros_like(x)
    o1 = np.zeros_like(x)
    dp = C.POINTER(C.c_double)
    yp = yy.ctypes.data_as(dp) if yy is not None else None
    if where == "device":
        _check(lib().tor_selftest_math_device(op, x.ctypes.data_as(dp), yp, o0.ctypes.data_as(dp),
                                              o1.ctypes.data_as(dp), x.size, device))
    else:
        _check(lib().tor_selftest_math_host(op, x.ctypes.data_as(dp), yp, o0.ctypes.data_as(dp),
                                            o1.ctypes.data_as(dp), x.size))
    return o0, o1


def selftest_rng(mode: int, a: int, b: int = 0, c: int = 0, n: int = 4):
    state = (C.c_uint64 * 4)()
    draws = (C.c_uint64 * max(n, 1))()
    _check(lib().tor_selftest_rng_host(mode, a, b, c, state, draws, n))
    return [int(v) for v in state], [int(v) for v in draws[:n]]
