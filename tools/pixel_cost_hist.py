"""Distribution of per-pixel chain lengths (closest-hit queries per pixel) from the cost probe run at full spp.
usage (GPU box): TOR_PROBE_SPP=100 python tools/pixel_cost_hist.py [spp]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
os.environ.setdefault("TOR_PROBE_SPP", str(spp))
H, W = 1080, 1920
tor = importlib.import_module("trace-of-radiance_amd")
scene, cam = tor.random_scene(0xFACADE), tor.camera()
ctx = tor.Context(0)
ctx.upload(scene.list())
buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
opt = tor.make_options(seeding=tor.SEED_PIXEL, accel=3, pixel_kernel=tor.PIXEL_KERNEL_LANE)
ctx.render_device(cam, H, W, spp, 2.2, 50, opt, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
c = ctx.last_pixel_cost(H * W).astype(np.float64)
probe = int(os.environ["TOR_PROBE_SPP"])
print(f"probe spp {probe}: queries per pixel: mean {c.mean():.1f} (per sample {c.mean() / probe:.2f}); p50 {np.median(c):.0f} p90 {np.percentile(c, 90):.0f} "
      f"p99 {np.percentile(c, 99):.0f} p99.9 {np.percentile(c, 99.9):.0f} p99.99 {np.percentile(c, 99.99):.0f} max {c.max():.0f}")
t = c[: (H * W // 64) * 64].reshape(-1, 64)
print(f"tiles: sum mean {t.sum(1).mean():.0f} max {t.sum(1).max():.0f}; tile max-pixel p50 {np.median(t.max(1)):.0f} p99 {np.percentile(t.max(1), 99):.0f} max {t.max(1).max():.0f}")
order = np.argsort(-t.sum(1))
mx = t.max(1)[order]
n = len(order)
for frac in (0.0, 0.05, 0.1, 0.25, 0.5, 0.75, 0.9):
    k = int(frac * n)
    print(f"  by-sum order, tiles from position {frac:.2f} on: longest chain {mx[k:].max():.0f} (mean pixel {c.mean():.0f})")
print(f"  pixels above 4x mean: {(c > 4 * c.mean()).sum()}, above 8x: {(c > 8 * c.mean()).sum()}; work share of pixels above 4x mean: {c[c > 4 * c.mean()].sum() / c.sum():.3f}")
