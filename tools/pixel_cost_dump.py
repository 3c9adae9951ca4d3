"""Per-pixel chain lengths (closest-hit queries per pixel over PROBE_SPP per-sample-stream samples) of the bench frame,
saved as uint32 .npy for offline analysis of the SEED_PIXEL hand-off policy.
usage (GPU box): python tools/pixel_cost_dump.py [probe_spp] [H W] -> gpurun_out/pixel_cost_<W>x<H>_<spp>.npy"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1080, 1920)
os.environ["TOR_PROBE_SPP"] = str(spp)
tor = importlib.import_module("trace-of-radiance_amd")
scene, cam = tor.random_scene(0xFACADE), tor.camera()
ctx = tor.Context(0)
ctx.upload(scene.list())
buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
opt = tor.make_options(seeding=tor.SEED_PIXEL, accel=3, pixel_kernel=tor.PIXEL_KERNEL_LANE)
ctx.render_device(cam, H, W, 32, 2.2, 50, opt, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
c = ctx.last_pixel_cost(H * W)
os.makedirs("gpurun_out", exist_ok=True)
np.save(f"gpurun_out/pixel_cost_{W}x{H}_{spp}.npy", c.reshape(H, W))
q = c.astype(np.float64) / spp
print(f"{W}x{H} probe spp {spp}: queries/sample per pixel: mean {q.mean():.3f} p50 {np.median(q):.2f} p90 {np.percentile(q, 90):.2f} "
      f"p99 {np.percentile(q, 99):.2f} p99.9 {np.percentile(q, 99.9):.2f} max {q.max():.2f}")
for thr in (3, 4, 5, 6, 8, 10, 15, 20, 25, 30):
    m = q > thr
    print(f"  pixels with > {thr:2d} queries/sample: {m.mean() * 100:7.3f} % of pixels, {q[m].sum() / q.sum() * 100:6.2f} % of the work")
