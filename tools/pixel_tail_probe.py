"""Where the SEED_PIXEL brute-force launch loses time against SEED_SAMPLE: per-wave timeline (wave log) per launch shape.
usage (GPU box): python tools/pixel_tail_probe.py [spp]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
H, W = 1080, 1920
for waves in ("2", "3"):
    os.environ["TOR_WAVES_PER_SIMD"] = waves
    os.environ["TOR_BLOCKS_PER_CU"] = waves
    tor = importlib.import_module("trace-of-radiance_amd")
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    ctx = tor.Context(0)
    ctx.upload(scene.list())
    buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
    for mode in (tor.SEED_PIXEL, tor.SEED_SAMPLE):
        opt = tor.make_options(seeding=mode, accel=0, pixel_kernel=tor.PIXEL_KERNEL_LANE)
        ctx.set_stats(False)
        for rep in range(2):
            ctx.render_device(cam, H, W, spp, 2.2, 50, opt, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        ms = ctx.last_kernel_ms()[0]
        ctx.set_stats(True)
        ctx.render_device(cam, H, W, spp, 2.2, 50, opt, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        raw = ctx.last_wave_log()
        wl = raw.astype(np.float64)
        t0 = wl[:, 0].min()
        start, end = (wl[:, 0] - t0) / 100e3, (wl[:, 1] - t0) / 100e3
        it = wl[:, 2] & ((1 << 40) - 1)
        q = (raw[:, 3] & np.uint64((1 << 44) - 1)).astype(np.float64)
        texh = (wl[:, 4] - t0) / 100e3
        busy = (end - start).sum() / (len(wl) * end.max())
        print(f"waves/SIMD {waves} seeding {mode}: {H * W * spp / ms / 1e3:7.1f} Msamples/s kernel {ms:.1f} ms, {len(wl)} waves; "
              f"end p10 {np.percentile(end, 10):.1f} p50 {np.median(end):.1f} p90 {np.percentile(end, 90):.1f} max {end.max():.1f} ms; "
              f"wave-residency {busy:.3f}; lane util {q.sum() / (64 * it.sum()):.3f}; counter dry p50 {np.median(texh):.1f} ms; "
              f"iterations total {it.sum():.3e}, us/iter p10 {np.percentile((end - start) * 1e3 / it, 10):.1f} p50 {np.median((end - start) * 1e3 / it):.1f} "
              f"p90 {np.percentile((end - start) * 1e3 / it, 90):.1f}", flush=True)
    del ctx
