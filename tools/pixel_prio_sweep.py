"""SEED_PIXEL lane kernel schedule: workgroups per CU x slow wave slots (region B only) x priority period x
region cut x hot-chain threshold, each in a fresh context; canvases must be identical.
usage (GPU box): python tools/pixel_prio_sweep.py [spp] [accel]"""
import hashlib, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
accel = int(sys.argv[2]) if len(sys.argv) > 2 else 0
H, W = (int(os.environ.get("H", 1080)), int(os.environ.get("W", 1920)))
tor = importlib.import_module("trace-of-radiance_amd")
scene, cam = tor.random_scene(0xFACADE), tor.camera()
ref = None
for waves, back, shift, stag, hot in [("3", "2", "16", "0.2", "0.4"), ("3", "2", "0", "0.2", "0"), ("3", "2", "16", "0.2", "0"), ("3", "2", "0", "0.2", "0.4"), ("3", "2", "14", "0.2", "0.4"), ("3", "2", "16", "0.2", "0.3"), ("3", "2", "16", "0.2", "0.6"), ("3", "2", "16", "0.3", "0.4")]:
    os.environ["TOR_TAIL_FRAC"] = stag
    os.environ["TOR_HOT_FRAC"] = hot
    os.environ["TOR_BACK_SLOT"] = back
    os.environ["TOR_WAVES_PER_SIMD"] = waves
    os.environ["TOR_BLOCKS_PER_CU"] = waves
    os.environ["TOR_PRIO_SHIFT"] = shift
    ctx = tor.Context(0)
    ctx.upload(scene.list())
    buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
    opt = tor.make_options(seeding=tor.SEED_PIXEL, accel=accel, pixel_kernel=tor.PIXEL_KERNEL_LANE)
    times = []
    for rep in range(9):
        ctx.render_device(cam, H, W, spp, 2.2, 50, opt, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        times.append(ctx.last_kernel_ms()[0])
    h = hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:12]
    ref = ref or h
    ctx.set_stats(True)
    ctx.render_device(cam, H, W, spp, 2.2, 50, opt, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    raw = ctx.last_wave_log()
    wl = raw.astype(np.float64)
    t0 = wl[:, 0].min()
    end, dry = (wl[:, 1] - t0) / 100e3, (wl[:, 4] - t0) / 100e3
    slot = (raw[:, 3] >> np.uint64(44)).astype(np.int64) & 15
    it_dry = (raw[:, 5] & np.uint64(0xffffffff)).astype(np.float64)
    rates = [f"{np.median((dry[slot == sl] - (wl[slot == sl, 0] - t0) / 100e3) * 1e3 / np.maximum(it_dry[slot == sl], 1)):.0f}" for sl in sorted(set(slot.tolist()))]
    t = np.array(times[1:])
    print(f"accel {accel} wg/CU {waves} back slot {back} prio shift {shift:>2} tail {stag:>4} hot {hot:>4}: {H * W * spp / t.mean() / 1e3:7.1f} Msamples/s (kernel ms {t.min():.1f}..{t.max():.1f}) "
          f"{'same' if h == ref else 'DIFFERENT ' + h}; us/iter by slot {'/'.join(rates)}; dry p50 {np.median(dry):.1f}; end p50 {np.median(end):.1f} p99 {np.percentile(end, 99):.1f} max {end.max():.1f}", flush=True)
    del ctx
