"""Latency of a server's bounce (DESIGN 4.7 (HISTORY 4.10)): a frame so small that nearly every wave is idle -- its time is its longest
pixel chain x the time of one dependent bounce in serve_chains.  Prints kernel ms, the hand-off timeline and, from the
probe-independent bookkeeping, microseconds per served bounce of the busiest chain.
usage (GPU box): python tools/server_latency.py [H W spp]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
tor = importlib.import_module("trace-of-radiance_amd")
H, W, spp = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (54, 96, 2000)
scene, cam = tor.random_scene(0xFACADE), tor.camera()
st = torch.cuda.current_stream().cuda_stream
ctx = tor.Context(0)
ctx.upload(scene.list())
buf = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
o = tor.make_options(seeding=tor.SEED_PIXEL, accel=3)
ctx.render_device(cam, H, W, spp, 2.2, 50, o, buf.data_ptr(), st)
torch.cuda.synchronize()
ks = []
for _ in range(3):
    ctx.render_device(cam, H, W, spp, 2.2, 50, o, buf.data_ptr(), st)
    torch.cuda.synchronize()
    ks.append(ctx.last_kernel_ms()[0])
c = ctx.last_handoff_counters()
cost = ctx.last_pixel_cost(H * W)
longest = float(cost.max()) / 2.0 * spp      # (the probe runs 2 samples per pixel)
print(f"{W}x{H}x{spp}: kernel {min(ks):.2f} ms; longest chain ~{longest:.0f} bounces ({cost.max() / 2.0:.1f} queries/sample in the probe) -> "
      f"{min(ks) * 1e3 / longest:.2f} us per bounce if the frame is that chain; hot {c['hot_pushes']} tail {c['tail_pushes']} "
      f"hot_done {c['us_hot_done'] / 1e3:.1f} ms tail_done {c['us_tail_done'] / 1e3:.1f} ms lane_end {c['us_lane_end'] / 1e3:.1f} ms")
