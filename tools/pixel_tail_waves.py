"""The waves that end a SEED_PIXEL launch last: who they are (hardware slot), how fast they run and what they still had to do
when the work counter ran dry.  usage (GPU box): python tools/pixel_tail_waves.py [spp] [accel]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
accel = int(sys.argv[2]) if len(sys.argv) > 2 else 0
os.environ.setdefault("TOR_WAVES_PER_SIMD", "3")
os.environ.setdefault("TOR_BLOCKS_PER_CU", "3")
H, W = 1080, 1920
tor = importlib.import_module("trace-of-radiance_amd")
scene, cam = tor.random_scene(0xFACADE), tor.camera()
ctx = tor.Context(0)
ctx.upload(scene.list())
buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
opt = tor.make_options(seeding=tor.SEED_PIXEL, accel=accel, pixel_kernel=tor.PIXEL_KERNEL_LANE)
ctx.set_stats(True)
for rep in range(2):
    ctx.render_device(cam, H, W, spp, 2.2, 50, opt, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
raw = ctx.last_wave_log()
wl = raw.astype(np.float64)
t0 = wl[:, 0].min()
start, end, dry = (wl[:, 0] - t0) / 100e3, (wl[:, 1] - t0) / 100e3, (wl[:, 4] - t0) / 100e3
it = wl[:, 2] & ((1 << 40) - 1)
it_dry = (raw[:, 5] & np.uint64(0xffffffff)).astype(np.float64)
hw = (raw[:, 3] >> np.uint64(44)).astype(np.int64)
q = (raw[:, 3] & np.uint64((1 << 44) - 1)).astype(np.float64)
slot, simd, cu = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15
print(f"kernel {ctx.last_kernel_ms()[0]:.1f} ms; {len(wl)} waves; counter dry p50 {np.median(dry):.1f} ms; end p50 {np.median(end):.1f} p90 {np.percentile(end, 90):.1f} "
      f"p99 {np.percentile(end, 99):.1f} max {end.max():.1f}")
for sl in sorted(set(slot.tolist())):
    m = slot == sl
    print(f"  hw wave slot {sl}: {m.sum():5d} waves; us/iter before dry p50 {np.median((dry[m] - start[m]) * 1e3 / np.maximum(it_dry[m], 1)):6.1f}; "
          f"iterations before dry p50 {np.median(it_dry[m]):6.0f}, after dry p50 {np.median(it[m] - it_dry[m]):5.0f} p90 {np.percentile(it[m] - it_dry[m], 90):5.0f}; "
          f"us/iter after dry p50 {np.median((end[m] - dry[m]) * 1e3 / np.maximum(it[m] - it_dry[m], 1)):6.1f}; end p50 {np.median(end[m]):.1f} p90 {np.percentile(end[m], 90):.1f} max {end[m].max():.1f}")
last = np.argsort(end)[-24:]
print("  the 24 last waves: end ms | dry ms | slot simd | iters before dry | iters after dry | us/iter before | us/iter after | lane util after dry")
for w in last:
    ia = it[w] - it_dry[w]
    print(f"    {end[w]:7.1f} {dry[w]:7.1f}  {slot[w]:2d} {simd[w]:1d}  {it_dry[w]:6.0f} {ia:6.0f}  {(dry[w] - start[w]) * 1e3 / max(it_dry[w], 1):6.1f} {(end[w] - dry[w]) * 1e3 / max(ia, 1):6.1f}")
# histogram of iterations after dry against the average pixel length
print(f"  iterations per wave total p50 {np.median(it):.0f}; lane utilisation {q.sum() / (64 * it.sum()):.3f}; after dry: p50 {np.median(it - it_dry):.0f} p90 {np.percentile(it - it_dry, 90):.0f} "
      f"p99 {np.percentile(it - it_dry, 99):.0f} max {(it - it_dry).max():.0f}")
