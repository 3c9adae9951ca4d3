/*
 * tor_render.h -- C ABI of libtor_mi355x.so: the MI355X (gfx950) implementation of the
 * trace-of-radiance integrator hot path
 *     render() -> radiance() -> HittableList.hit -> Material.scatter
 * (reference: trace_of_radiance/render.nim:21-68).  Plain pointers and sizes only; every
 * struct is a bit-for-bit mirror of the value type Nim's C backend emits for the reference
 * type named next to it (x86-64), so a Nim caller passes `unsafeAddr` of its own objects
 * (see INTEGRATION.md for the {.importc.} shim).
 *
 * All arithmetic on the path is IEEE float64 (vec3s.nim:12-14).
 */
#ifndef TOR_RENDER_H
#define TOR_RENDER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TOR_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------ */
/* POD mirrors of the reference's value types                                            */
/* ------------------------------------------------------------------------------------ */

/* Vec3 / Point3 / Color / Attenuation / UnitVector -- primitives/vec3s.nim:12-14 (24 B) */
typedef struct TorVec3 { double x, y, z; } TorVec3;

/* MaterialKind -- physics/core.nim:25-27 (enum order of registerSubType) */
enum { TOR_LAMBERTIAN = 0, TOR_METAL = 1, TOR_DIELECTRIC = 2 };

/* Material -- physics/core.nim:16-28 (object variant: kind:uint8 @0, union @8; 40 B) */
typedef struct TorMaterial {
  uint8_t kind;
  uint8_t _pad[7];
  union {
    struct { TorVec3 albedo; } lambertian;              /* core.nim:16-17 */
    struct { TorVec3 albedo; double fuzz; } metal;      /* core.nim:18-20 */
    struct { double refraction_index; } dielectric;     /* core.nim:21-22 */
  } u;
} TorMaterial;

/* Sphere -- physics/hittables/spheres.nim:15-18 (72 B) */
typedef struct TorSphere {
  TorVec3 center;
  double radius;
  TorMaterial material;
} TorSphere;

/* MovingSphere -- physics/hittables/moving_spheres.nim:15-20 (112 B) */
typedef struct TorMovingSphere {
  TorVec3 center0, center1;
  double time0, time1;
  double radius;
  TorMaterial material;
} TorMovingSphere;

/* HittableVariantKind -- physics/hittables/hittables_variants.nim:53-54 */
enum { TOR_SPHERE = 0, TOR_MOVING_SPHERE = 1 };

/* HittableVariant -- hittables_variants.nim:50-57 (kind:uint8 @0, union @8; 120 B) */
typedef struct TorHittableVariant {
  uint8_t kind;
  uint8_t _pad[7];
  union {
    TorSphere sphere;
    TorMovingSphere moving_sphere;
  } u;
} TorHittableVariant;

/* HittableList -- physics/hittables/hittables_lists.nim:20-24 (borrowed view; 16 B) */
typedef struct TorHittableList {
  int64_t len;
  const TorHittableVariant* objects;
} TorHittableList;

/* Camera -- physics/cameras.nim:15-22 (24 float64 in declaration order; 192 B) */
typedef struct TorCamera {
  TorVec3 origin, lower_left_corner, horizontal, vertical, u, v, w;
  double lens_radius, shutter_open, shutter_close;
} TorCamera;

/* Canvas -- primitives/canvas.nim:20-28 (24 B).  pixels: nrows*ncols Colors, row-major,
 * row 0 = BOTTOM scanline (io/ppm.nim:20); caller-allocated, fully overwritten. */
typedef struct TorCanvas {
  TorVec3* pixels;
  int32_t nrows, ncols;
  int32_t samples_per_pixel;
  float gamma_correction;
} TorCanvas;

/* ------------------------------------------------------------------------------------ */
/* Options                                                                               */
/* ------------------------------------------------------------------------------------ */

/* How the per-pixel RNG streams are laid out. */
enum {
  /* render.nim:59-67: rng.seed(row,col) once per pixel, the spp samples share one stream.
   * One GPU lane per PIXEL.  This is the reference's behaviour. */
  TOR_SEED_PIXEL = 0,
  /* Counter-based extension (BASELINE.json north_star): the stream is re-seeded per
   * (row,col,sample): sm=pair(row,col); h=splitMix64(sm); seed(h xor sample).
   * One GPU lane per PIXEL-SAMPLE; per-pixel sums use exact (2^-36-quantised) float64
   * addition so the result does not depend on scheduling or device count. */
  TOR_SEED_SAMPLE = 1
};

/* Rounding of the ray/sphere quadratic. */
enum {
  TOR_ARITH_STRICT = 0, /* reference operation order, no FMA (README.md:82) -- the only arithmetic     */
  TOR_ARITH_FUSED = 1   /* REMOVED in round 5 (rounds 1-4: the same formulas with explicit fma(); not the
                           reference's rounding, and slower than STRICT behind the conservative screen).  The value
                           stays reserved: every entry point rejects it with TOR_ERR_INVALID_ARGUMENT and says why. */
};

/* Exact accelerations of the closest-hit query; a bit mask.  They never change a pixel:
 * closest hit is order independent and every kept object goes through the reference's float64 test. */
enum {
  TOR_ACCEL_NONE = 0,   /* the reference's algorithm and arithmetic: every ray against every object in
                           float64 (default)                                                          */
  TOR_ACCEL_BLOCKS = 1, /* objects in spatial blocks of 8 inside conservative boxes; a ray only looks
                           into the blocks whose box it can touch                                     */
  TOR_ACCEL_F32 = 2     /* still every ray against every object, but first through a conservative
                           packed-float32 discriminant with an a-priori error margin; only the objects
                           it cannot rule out get the float64 test                                    */
};

/* How the row shards of a multi-device tor_render_opt() reach the caller's canvas. */
enum {
  TOR_GATHER_AUTO = 0,  /* RCCL when the devices are distinct, librccl loads and the communicator passes its
                           self-check; if that leg fails -- or does not COMPLETE: every wait of the RCCL leg has a
                           deadline (TOR_RCCL_TIMEOUT_MS per transfer, default 10 s + 1 ms / MB; TOR_RCCL_INIT_TIMEOUT_MS
                           for communicator creation + self-check, default 120 s), past it the communicators are
                           aborted -- peer copies; if those fail: HOST (tor_last_note)                           */
  TOR_GATHER_RCCL = 1,  /* single-process RCCL (ncclCommInitAll): every device sends its shard to
                           devices[0] over xGMI, one de-interleave kernel, one D2H (BASELINE north_star) */
  TOR_GATHER_PEER = 2,  /* the same with hipMemcpyPeerAsync instead of RCCL                              */
  TOR_GATHER_HOST = 3   /* no device-side gather: every device copies its rows straight into the
                           canvas over its own PCIe link (SURVEY 8e "alternative")                       */
};

#define TOR_MAX_DEVICES 16

enum { TOR_PIXEL_KERNEL_AUTO = 0, TOR_PIXEL_KERNEL_LANE = 1, TOR_PIXEL_KERNEL_WAVE = 2 };

typedef struct TorOptions {
  uint32_t struct_size; /* = sizeof(TorOptions); the 32-byte round-1 layout (up to `accel`) is accepted too */
  int32_t seeding;      /* TOR_SEED_*  (default TOR_SEED_PIXEL)  */
  int32_t arith;        /* TOR_ARITH_* (default TOR_ARITH_STRICT) */
  int32_t device;       /* HIP device ordinal; -1 = current device */
  /* Row sharding (render.nim:55 `parallelFor row` across GPUs): image rows are cut into
   * tiles of row_tile rows; tile t is rendered by shard (t mod shard_count).  The shard's
   * rows are written compactly, in increasing row order.  shard_count <= 1: whole image. */
  int32_t shard_index, shard_count, row_tile;
  int32_t accel;        /* TOR_ACCEL_* bits (explicit options: default TOR_ACCEL_NONE) */
  /* Multi-GPU behind the drop-in (tor_render / tor_render_opt only): device_count > 1 renders the frame on
   * devices[0 .. device_count) -- one host thread and one HIP stream per entry, entry k renders row shard
   * (k, device_count, row_tile) -- and assembles it in canvas->pixels (`gather`).  An ordinal may appear more
   * than once (several contexts on one GPU: how a 1-GPU box tests the path).  shard_index/shard_count/device
   * must then be left at their defaults.  The canvas is bit-identical for every device list. */
  int32_t device_count;
  int32_t gather;       /* TOR_GATHER_* */
  int32_t devices[TOR_MAX_DEVICES];
  /* TOR_SEED_PIXEL only: which kernel walks the pixel chains (same canvas either way).
   * TOR_PIXEL_KERNEL_AUTO: with both exact accelerations, >= 32 spp and a single-level culling layout (<= 128 block
   * boxes) every frame size runs the one-lane-per-pixel kernel with the CHAIN HAND-OFF -- lanes push their long pixel
   * chains to server waves inside the same launch (DESIGN 4.7 (HISTORY 4.7-4.8); the launch covers the whole GPU and assumes exclusive
   * use of it: tor_context_handoff_stalled; TOR_MIGRATE=0 turns it off).  Where the hand-off cannot run: frames of
   * 16 K pixels and more (per device) with both accelerations and >= 32 spp are SHARED -- the tiles that carry the
   * largest part of a probed cost go to the one-wave-per-pixel kernel on a second stream, the lane kernel renders the
   * rest at the same time (TOR_SPLIT_FRAC overrides the fraction, 0 = off); otherwise one wave per pixel up to 114688
   * pixels (TOR_COOP_MAX_PIXELS), one lane per pixel above.  LANE / WAVE force one kernel for the whole frame, without
   * hand-off (WAVE falls back to LANE when the scene does not fit LDS). */
  int32_t pixel_kernel;
} TorOptions;

/* Status codes (the reference's render() returns void and has no error path; this ABI
 * returns 0 on success and never writes a partial canvas on failure). */
enum {
  TOR_OK = 0,
  TOR_ERR_INVALID_ARGUMENT = -1,
  TOR_ERR_NO_DEVICE = -2,  /* no HIP device / kernels unavailable: there is NO CPU fallback */
  TOR_ERR_HIP = -3,
  TOR_ERR_OUT_OF_MEMORY = -4,
  TOR_ERR_INCOMPLETE = -5  /* tor_last_kernel_ms after an asynchronous tor_render_device whose chain hand-off stalled: the frame
                              has holes and must be rendered again (the blocking entry points and tor_render_gather_device
                              do that themselves)                                                                     */
};

/* ------------------------------------------------------------------------------------ */
/* The drop-in entry point                                                               */
/* ------------------------------------------------------------------------------------ */

/* Replaces `proc render*(canvas: var Canvas, cam: Camera, world: HittableList,
 * max_depth: int)` -- render.nim:49.  Blocking: canvas.pixels is complete on return (the
 * reference's canvas is complete only after exit(Weave)/syncRoot(Weave),
 * trace_of_radiance.nim:61-63).  Reference semantics: TOR_SEED_PIXEL, TOR_ARITH_STRICT.
 *
 * A host that keeps the reference's signature cannot pass TorOptions, so tor_render() takes its speed knobs
 * from the environment -- none of them changes a pixel:
 *   TOR_DEFAULT_ACCEL = 0..3  TOR_ACCEL_* bits.  Unset: 3 -- both exact accelerations are ON for tor_render()
 *                             (bit-identical canvases by construction, parity tests and differential fuzzing);
 *                             0 restores the reference's float64 brute force.
 *   TOR_SCREEN = 0            (float64 brute force only) every ray x object through the reference's unfused discriminant;
 *                             unset: the object loop is a conservative FMA screen of the same quadratic and only its
 *                             candidates see the unfused operations -- same canvas (csrc/tor_screen.hpp), read at context creation
 *   TOR_DEFAULT_SEEDING = pixel | sample   the ONE knob here that selects a different (equally valid) image: `pixel`
 *                             (default) = the reference's streams, one per pixel (render.nim:59-67); `sample` = the
 *                             counter-based per-sample streams of TOR_SEED_SAMPLE -- what lets 8 GPUs share a 1080p frame
 *                             (a pixel stream is a sequential chain) and the mode the headline Msamples/s is quoted on
 *   TOR_DEVICES = "all" | "0,1,2,3"   render on several GPUs (TorOptions.device_count / devices)
 *   TOR_GATHER  = rccl | peer | host  (TorOptions.gather)
 * The device scene is cached: a call whose object list is byte-identical to the previous call's (on that
 * device) uploads nothing (the host pointer is never retained; the library keeps its own copy). */
TOR_API int tor_render(TorCanvas* canvas, const TorCamera* cam, TorHittableList world,
                       int64_t max_depth);

/* tor_render with the HittableList behind a pointer: for FFIs that would rather not pass a 16-byte struct by
 * value (Nim passes small objects by value and large ones by hidden pointer -- README.md:232-235 -- unless the
 * type is marked {.bycopy.}; a pointer leaves nothing to the calling convention). */
TOR_API int tor_render_ptr(TorCanvas* canvas, const TorCamera* cam, const TorHittableList* world, int64_t max_depth);

/* Same with explicit options (NULL = tor_render's defaults).  With shard_count > 1 only this shard's
 * rows of canvas->pixels are written (in place, at their image positions). */
TOR_API int tor_render_opt(TorCanvas* canvas, const TorCamera* cam, TorHittableList world,
                           int64_t max_depth, const TorOptions* opt);

/* Host-side cost of the last tor_render / tor_render_opt call on this thread, in milliseconds:
 * out[0] scene upload (0 on a cache hit), out[1] launch + kernels until the device is done (single device: measured
 * with HIP events on the call's stream, first launch to last kernel), out[2] the rest of the call's device section:
 * D2H / gather into canvas->pixels, out[3] whole call.  out[4] = 1 when the scene came from the cache. */
TOR_API int tor_last_render_timing(double out[5]);

/* The library's environment knobs -- ONE table (csrc/tor_knobs.hpp; KNOBS.md is generated from it): name, default, accepted
 * values, when it is read ("call" | "context" | "upload") and what it does.  Strings are static. */
TOR_API int32_t tor_knob_count(void);
TOR_API int tor_knob_info(int32_t i, const char** name, const char** dflt, const char** range, const char** when, const char** what);

/* Thread-local description of the last failure (never NULL). */
TOR_API const char* tor_last_error(void);

/* Thread-local note of the last successful multi-device tor_render / tor_render_opt on this thread (never NULL): which
 * gather ran ("gather: rccl" | "gather: peer" | "gather: host"), preceded by the legs TOR_GATHER_AUTO tried first and
 * why they failed -- AUTO walks RCCL -> peer copies -> per-device D2H and never returns a wrong canvas.  Not an error. */
TOR_API const char* tor_last_note(void);

/* Facts about the last successful multi-device tor_render / tor_render_opt on this thread (what a scaling log needs):
 * out[0] = the TOR_GATHER_* leg that assembled the frame, out[1] = ranks of the RCCL communicator that carried it
 * (ncclCommCount; 0 when the leg was not RCCL), out[2] = entries of the device list, out[3] = 1 when they were distinct GPUs. */
TOR_API int tor_last_gather_info(int32_t out[4]);
/* Duration of the dominant kernel (integrate_kernel) on every device of that call, in milliseconds: HIP events around the
 * launch on the launch's own stream.  Writes min(cap, n) values, returns n = entries of the device list. */
TOR_API int32_t tor_last_device_kernel_ms(float* out, int32_t cap);

/* ------------------------------------------------------------------------------------ */
/* Resident-context API (frame loops: trace_of_radiance_animation.nim:173-196; benchmarks; */
/* multi-GPU hosts that own device buffers)                                              */
/* ------------------------------------------------------------------------------------ */

typedef struct TorContext TorContext;

TOR_API int tor_context_create(int32_t device, TorContext** out);
TOR_API int tor_context_destroy(TorContext* ctx);

/* Flattens the AoS HittableVariant list (hittables_lists.nim:41-46) into the device SoA
 * scene.  The host pointer is not retained (the context keeps a byte copy: a later upload of an identical
 * list is a no-op, and the layouts a launch does not use -- the float32 and block-culling variants -- are
 * only built when a launch first asks for them). */
TOR_API int tor_scene_upload(TorContext* ctx, TorHittableList world);

/* Number of rows / list of rows shard (index,count,row_tile) owns. rows_out may be NULL. */
TOR_API int32_t tor_shard_rows(int32_t nrows, int32_t row_tile, int32_t shard_index,
                               int32_t shard_count, int32_t* rows_out);

/* Renders this shard's rows into d_pixels, a DEVICE buffer of tor_shard_rows()*ncols*3
 * float64 (rows in increasing order, gamma-corrected exactly like Canvas.draw,
 * canvas.nim:47-54).  Asynchronous on hip_stream (a hipStream_t; NULL = default stream);
 * the buffer is complete when the stream reaches the end of the enqueued work. */
TOR_API int tor_render_device(TorContext* ctx, const TorCamera* cam, int32_t nrows, int32_t ncols,
                              int32_t samples_per_pixel, float gamma_correction,
                              int64_t max_depth, const TorOptions* opt, double* d_pixels,
                              void* hip_stream);

/* ---- multi-process hosts: one process per GPU, the framebuffer gather inside the library (RCCL) ----------
 * rank 0 calls tor_comm_unique_id and hands the 128 bytes to the other ranks by its own means (bench.py:
 * torch.distributed broadcast); every rank then calls tor_comm_init_rank on its context (ncclCommInitRank).
 * tor_render_gather_device = tor_render_device for shard (rank, world, opt->row_tile) + the gather of the row
 * shards over xGMI + a de-interleave kernel: d_frame (nrows*ncols*3 float64, rows in image order, DEVICE memory)
 * is complete on `root` (root >= 0: send/recv gather, each rank's own link to the root) or on every rank
 * (root < 0: ncclAllGather).  Asynchronous on hip_stream.  librccl.so.1 is loaded on first use (dlopen), the
 * library has no link-time dependency on it. */
TOR_API int tor_comm_unique_id(uint8_t id_out[128]);
TOR_API int tor_comm_init_rank(TorContext* ctx, const uint8_t id[128], int32_t rank, int32_t world);
TOR_API int tor_comm_destroy(TorContext* ctx);
/* ncclCommAbort: for a host whose watchdog saw a gather that does not complete (bench.py polls its stream with a deadline);
 * RCCL's kernels leave, the context has no communicator afterwards.  tor_comm_count: ranks of the communicator
 * (ncclCommCount), 0 without one. */
TOR_API int tor_comm_abort(TorContext* ctx);
TOR_API int tor_comm_count(TorContext* ctx, int32_t* ranks_out);
TOR_API int tor_render_gather_device(TorContext* ctx, const TorCamera* cam, int32_t nrows, int32_t ncols,
                                     int32_t samples_per_pixel, float gamma_correction, int64_t max_depth,
                                     const TorOptions* opt, int32_t root, double* d_frame, void* hip_stream);

/* Chain hand-off of the last launch of `ctx` (TOR_SEED_PIXEL with both exact accelerations, DESIGN 4.7 (HISTORY 4.7-4.8)): the launch covers
 * the whole GPU and its waves wait for each other, so it assumes exclusive use of the device's compute units (hand-off
 * launches of ONE process are chained per device by the library; TOR_MIGRATE=0 turns the hand-off off).  If some of its
 * workgroups never become resident (another process's persistent kernel, a CU mask) the waiting waves give up after
 * TOR_SRV_STALL_S seconds without progress (default 60, 0 = never) and the frame is INCOMPLETE: tor_render / tor_render_opt /
 * tor_render_frame_h264 notice and render the frame again without the hand-off; a caller of the asynchronous
 * tor_render_device asks here.  Blocks until the last launch's stream is idle.  *stalled_out = 1: render again with
 * TOR_PIXEL_KERNEL_LANE.  total_out (nullable): frames the blocking entry points re-rendered so far. */
TOR_API int tor_context_handoff_stalled(TorContext* ctx, int32_t* stalled_out, int64_t* total_out);

/* Scene-cache counters of a context: out[0] = tor_scene_upload calls, out[1] = calls that found the device
 * scene up to date (nothing rebuilt, nothing copied), out[2] = device layouts built so far. */
TOR_API int tor_context_scene_counters(TorContext* ctx, int64_t out[3]);

/* Device-side output stage (io/ppm.nim:15-16 quantiser int(256*clamp(c,0,0.999))):
 * d_pixels (n_rows*ncols*3 float64) -> d_rgb8 (n_rows*ncols*3 bytes), same row order. */
TOR_API int tor_quantize_rgb8_device(TorContext* ctx, const double* d_pixels, int64_t n_values,
                                     uint8_t* d_rgb8, void* hip_stream);

/* Video output stage of the animation driver (trace_of_radiance_animation.nim:186-196), one fused
 * device kernel per frame: Canvas -> RGB8 (io/rgb.nim:17-31, top scanline first) -> BT.601 Y'CbCr
 * 4:2:0 (io/color_conversions.nim:180-252) -> one I_PCM slice (io/h264.nim:249-259).
 *   tor_h264_stream_header : SPS (h264.nim:90-142) + PPS (h264.nim:37), written once per stream
 *   tor_h264_frame_bytes   : size of one frame's slice NAL unit (width, height even; a size that is not a multiple of 16 --
 *                            1080 rows: BASELINE configs[4] -- is padded to whole macroblocks by edge replication and the SPS
 *                            crops the padding away (H.264 7.4.2.1.1).  The reference has a TODO there (h264.nim:178): its
 *                            SPS announces ceil(size/16) macroblocks, its flushFrame writes floor(size/16).)
 *   tor_encode_frame_device: d_pixels = finished canvas (nrows*ncols*3 float64, row 0 = bottom);
 *                            d_slice receives tor_h264_frame_bytes() bytes; d_y/d_cb/d_cr (nullable)
 *                            receive the planes (H264Encoder.getFrameBuffers, h264.nim:206-224).
 * Concatenating header + slices gives the Annex-B .264 file of main_animation_mp4;
 * tor_mp4_mux_file wraps it into the .mp4 (host side). */
TOR_API int tor_h264_stream_header(int32_t width, int32_t height, uint8_t* out, int32_t cap);
TOR_API int64_t tor_h264_frame_bytes(int32_t width, int32_t height);
/* MP4Muxer.initialize + writeMP4_from + close (io/mp4.nim:113-163, driver trace_of_radiance_animation.nim:
 * 203-210): reads the Annex-B stream src_annexb_path and writes an MP4 file with one avc1 video track --
 * one sample per slice NAL unit, 90 kHz time base, 90000/fps ticks per sample (the reference: fps = 30).
 * Host code, streams file to file.  Returns the number of samples written, or a negative status. */
TOR_API int tor_mp4_mux_file(const char* src_annexb_path, const char* dst_mp4_path, int32_t width,
                             int32_t height, int32_t fps);
TOR_API int tor_encode_frame_device(TorContext* ctx, const double* d_pixels, int32_t nrows, int32_t ncols,
                                    uint8_t* d_slice, uint8_t* d_y, uint8_t* d_cb, uint8_t* d_cr,
                                    void* hip_stream);

/* The animation driver's loop body in one blocking call (trace_of_radiance_animation.nim:181-196):
 * render the frame with the uploaded scene, convert + pack it on the device, copy only the slice NAL
 * unit (tor_h264_frame_bytes() bytes) to slice_out. */
TOR_API int tor_render_frame_h264(TorContext* ctx, const TorCamera* cam, int32_t nrows, int32_t ncols,
                                  int32_t samples_per_pixel, float gamma_correction, int64_t max_depth,
                                  const TorOptions* opt, uint8_t* slice_out, int64_t cap);

/* Timing of the last tor_render_device call on this context, measured with HIP events
 * recorded on the launch stream around the integrator kernel only (ms; the SEED_PIXEL cost probe and
 * the tile sort run before the start event); blocks until the kernel has finished.  samples_out
 * (nullable) = pixel-samples that launch traced.  All launches of one context that may be in flight
 * together must use ONE stream (per-launch state lives in a ring of 64 slots; with statistics enabled
 * launches must not overlap at all): tor_render_device returns TOR_ERR_INVALID_ARGUMENT for a launch on a different
 * stream while the context's previous launch is still running (use one context per stream for concurrency). */
TOR_API int tor_last_kernel_ms(TorContext* ctx, float* ms_out, int64_t* samples_out);
/* Mean duration (ms) of the integrator kernel over the last `last_n` tor_render_device calls
 * (at most 64 are remembered), from the same per-launch HIP events. */
TOR_API int tor_kernel_ms_mean(TorContext* ctx, int32_t last_n, float* mean_ms_out, int32_t* n_used_out);

/* Workload counters of the last tor_render_device call (only when the context was told to
 * collect them, see tor_context_set_stats): closest-hit queries, candidate resolves. */
typedef struct TorStats {
  uint64_t hit_queries;       /* world.hit() calls (hittables_lists.nim:48-55)          */
  uint64_t object_tests;      /* hit_queries * n_objects                                 */
  uint64_t candidates;        /* objects that went past the wave-uniform object loop: block_tests + exact_tests */
  uint64_t wave_iterations;   /* bounce-loop trips summed over waves                     */
  uint64_t lane_slots;        /* 64 * wave_iterations                                    */
  uint64_t samples;           /* pixel-samples traced                                    */
  uint64_t block_tests;       /* TOR_ACCEL_BLOCKS | TOR_ACCEL_F32: objects the float32 filter looked at in the blocks a ray entered (8 per block) */
  uint64_t exact_tests;       /* candidates - block_tests: objects that reached the reference's own test (spheres.nim:28-49) */
} TorStats;
TOR_API int tor_context_set_stats(TorContext* ctx, int32_t enable);
TOR_API int tor_last_stats(TorContext* ctx, TorStats* out);
/* Debug: 8 x u64 per wave of the last launch (stats enabled): {start, end (100 MHz wall clock), bounce
 * iterations (low 40 bits) | stage two of the plane-screened segments << 43 (a 21-bit field like the six below, part of
 * the object loop's), closest-hit queries | HW_ID << 44, time when the work counter ran dry, iteration count at that
 * time | trips of the resolve loop << 32, and two words of six 21-bit fields in units of 4096 shader cycles:
 * refill + camera ray, object loop, exact resolve | shade, deposit, total}.
 * Returns the number of waves copied (<= cap_waves) or < 0. */
TOR_API int tor_last_wave_log(TorContext* ctx, uint64_t* out, int64_t cap_waves);
/* Debug: chain hand-off of the last TOR_SEED_PIXEL launch on this context (all 0 when the launch ran without it):
 * out[0] tickets taken by server waves, out[1] chains pushed by lanes, out[2] waves still in the lane loop (0 after the
 * launch), out[3] workgroups that were servers from the start, out[4] push threshold (bounce iterations), out[5] chains
 * served, out[6] pushes of hot chains, out[7] pushes in the tail of the frame; out[8..11] microseconds after the
 * kernel's start at which the work counter ran dry, the last wave left the lane loop, the last hot chain and the last
 * tail chain were finished by a server; out[12], out[13] bounce iterations served for hot / tail chains; out[14] dedicated
 * server waves that found nothing to do and turned into lane waves; out[15] the adaptive push threshold at the end. */
TOR_API int tor_last_handoff_counters(TorContext* ctx, uint64_t out[16]);
/* Debug: the cost probe of the last TOR_SEED_PIXEL launch (it runs from 32 spp on): closest-hit queries per pixel over the
 * probe's samples (2 per pixel; per-sample streams, so only statistically what the frame's samples do), in the shard's
 * local pixel order.  Returns the number of pixels copied (<= cap_pixels) or < 0 (no probe ran). */
TOR_API int64_t tor_last_pixel_cost(TorContext* ctx, uint32_t* out, int64_t cap_pixels);

/* ------------------------------------------------------------------------------------ */
/* Host-side mirrors of the reference constructors on either side of the path            */
/* ------------------------------------------------------------------------------------ */

/* camera(...) -- physics/cameras.nim:24-45 */
TOR_API int tor_camera_init(TorCamera* out, const TorVec3* look_from, const TorVec3* look_at,
                            const TorVec3* view_up, double vertical_fov_degrees,
                            double aspect_ratio, double aperture, double focus_distance,
                            double shutter_open, double shutter_close);

/* random_scene(rng) with rng.seed(seed) -- scenes.nim:13-50, trace_of_radiance.nim:34-36.
 * Returns the number of objects written, or a negative status if cap is too small. */
TOR_API int64_t tor_random_scene(uint64_t seed, TorHittableVariant* out, int64_t cap);

/* Animated scene (BASELINE config 5) -- trace_of_radiance/scenes_animated.nim.
 * tor_animation_create = random_moving_spheres (:90-154) with rng.seed(seed);
 * tor_animation_next   = one turn of `iterator scenes(anim, skip)` (:176-225): returns 1 and fills
 *                        (cam, objects[0 .. *n_out)) when a frame is due, 0 once anim.t >= t_max.
 * Frames are independent given (cam, objects): a frame-parallel host gives frame f to GPU f mod N. */
typedef struct TorAnimation TorAnimation;
TOR_API int tor_animation_create(uint64_t seed, int32_t height, int32_t width, float dt, float t_min,
                                 float t_max, TorAnimation** out);
TOR_API void tor_animation_destroy(TorAnimation* anim);
TOR_API int64_t tor_animation_object_count(const TorAnimation* anim);
TOR_API int tor_animation_next(TorAnimation* anim, int32_t skip, TorCamera* cam, TorHittableVariant* objects,
                               int64_t cap, int64_t* n_out, float* t_out);

/* exportToPPM's quantiser on a host canvas -- io/ppm.nim:14-27.  out: nrows*ncols*3 bytes,
 * first row = top scanline. */
TOR_API int tor_canvas_to_rgb8(const TorCanvas* canvas, uint8_t* out);

/* ------------------------------------------------------------------------------------ */
/* Self-test probes (used by the parity tests; no effect on rendering)                   */
/* ------------------------------------------------------------------------------------ */

/* Host-only debug view of the TOR_ACCEL_BLOCKS layout for a ray-time range: slot_object[k] = original index
 * of the object in spatial slot k (block k/8) or -1; block_boxes / super_boxes = 6 float64 {lo xyz, hi xyz}
 * per (padded) block / per group of 8 blocks.  Returns the number of blocks (0 when no culling layout is
 * built for this scene) or a negative status. */
TOR_API int tor_debug_accel_layout(TorHittableList world, double t_lo, double t_hi, int64_t* slot_object,
                                   int64_t slot_cap, double* block_boxes, double* super_boxes, int64_t box_cap,
                                   int32_t* two_level_out);

/* TOR_ACCEL_F32 over a whole scene on the HOST: builds the layout tor_scene_upload builds and walks its float32
 * segments for each ray (origin o, direction d, time) exactly as the kernel does.
 * keep[ray * world.len + object] = 1 kept, 0 dropped, 2 object stays on the float64 loop. */
TOR_API int tor_debug_filter32_scene(TorHittableList world, int64_t n_rays, const double* o, const double* d,
                                     const double* time, int8_t* keep);

/* The strict brute-force layout's screened segments over a whole scene on the HOST (csrc/tor_screen.hpp; the layout
 * tor_scene_upload builds, walked as the ARITH 2 object loop walks it when stage one runs: plane screen, then the segment's own
 * test per object -- second form for xkind 10 / 11 / 12, first form for 13 / 14).
 * keep[ray * world.len + object] = 0 dropped by the plane screen, 1 dropped by stage two, 2 candidate of the exact test,
 * 3 the object sits on a segment without a plane table; kind_out[object] (nullable) = 0 | 10..14, its segment's xkind;
 * pays_out[ray * n_segs_out + segment] (nullable) = 1 when that ray votes for stage one on that segment (plane_pays). */
TOR_API int tor_debug_screen2_scene(TorHittableList world, int64_t n_rays, const double* o, const double* d,
                                    const double* time, int8_t* keep, int32_t* kind_out, int8_t* pays_out, int64_t n_segs_out);

/* The float64 layout's segments in the order the kernel walks them (tor_scene.cpp: largest first; a plane-screened segment's long
 * tail padded to a whole word of 32 slots): out[4 * s + {0, 1, 2, 3}] = {xkind, objects, slots, first slot} for the first
 * `max_segs` segments; *n_segs_out = the number of segments.  Host only. */
TOR_API int tor_debug_layout_segments(TorHittableList world, int32_t* out, int64_t max_segs, int64_t* n_segs_out);

/* Float32 slab test of the culling boxes on the HOST (same source as the kernel): ray i against the box [lo_i, hi_i].
 * keep[i] = the float32 test keeps the box, need[i] = the float64 slab test of the float64 path passes.
 * Correct iff need[i] != 0 implies keep[i] != 0. */
TOR_API int tor_selftest_slab32_host(int64_t n, const double* o, const double* d, const double* lo, const double* hi,
                                     const double* origin, int32_t* keep, int32_t* need);

/* TOR_ACCEL_F32 self test on the HOST (same source as the kernel's pre-filter): ray i against sphere i with
 * centre c0 + dc * f (moving != 0) or c0.  keep[i] = pre-filter keeps the object; need[i] bit 0 = the
 * float64 test D > 0 and (half_b < 0 or c < 0) holds, bit 1 = the reference's hit() accepts a root.
 * The filter is correct iff need[i] != 0 implies keep[i] != 0. */
TOR_API int tor_selftest_filter32_host(int64_t n, const double* o, const double* d, const double* c0,
                                       const double* dc, const int32_t* moving, const double* f, const double* r2,
                                       const double* origin, int32_t* keep, int32_t* need);

/* The conservative FMA screen of the strict float64 object loop on the HOST (same source as the kernel, csrc/tor_screen.hpp):
 * ray i against sphere i, margins as for a segment that holds only this object.  keep / need as above; correct iff
 * need[i] != 0 implies keep[i] != 0.  (The screen only selects candidates for the exact test of spheres.nim:29-48.) */
TOR_API int tor_selftest_screen_host(int64_t n, const double* o, const double* d, const double* c0,
                                     const double* dc, const int32_t* moving, const double* f, const double* r2,
                                     int32_t* keep, int32_t* need);
/* The screen's SECOND form (expanded quadratic, direction normalised per ray: csrc/tor_screen.hpp).  variant 0: static
 * spheres through the general record, movers along y through the common-height record, other movers through the first form
 * (as the kernel routes them); variant 1: static spheres through the common-height record; variant 2: static spheres and
 * movers along y through the plane screen alone (stage one of the common-height segments). */
TOR_API int tor_selftest_screen2_host(int64_t n, const double* o, const double* d, const double* c0,
                                      const double* dc, const int32_t* moving, const double* f, const double* r2,
                                      int32_t variant, int32_t* keep, int32_t* need);

/* Runs the kernel's own math on the DEVICE: op 0: sin,cos(a)  1: x^5  2: pow(x,y)
 * 3: sqrt(x)  4: x/y  5: uniform01 of seed(row=x,col=y) first n draws... see tests. */
TOR_API int tor_selftest_math_device(int32_t op, const double* x, const double* y, double* out0,
                                     double* out1, int64_t n, int32_t device);
/* Same routines compiled for the HOST from the same source (no GPU needed). */
TOR_API int tor_selftest_math_host(int32_t op, const double* x, const double* y, double* out0,
                                   double* out1, int64_t n);
/* RNG probes (host build of the kernel's RNG): state after seed, then n draws. */
TOR_API int tor_selftest_rng_host(int32_t mode, uint64_t a, uint64_t b, uint64_t c,
                                  uint64_t state_out[4], uint64_t* draws_out, int64_t n);

TOR_API const char* tor_version(void);

#ifdef __cplusplus
}
#endif

#endif /* TOR_RENDER_H */
