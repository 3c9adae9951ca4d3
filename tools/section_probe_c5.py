import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
tor = importlib.import_module("trace-of-radiance_amd")
H, W, SPP = 1080, 1920, 64
cam, scene, _ = next(iter(tor.Animation(H, W, 0.005, 0.3, 2.0).scenes(6)))
ctx = tor.Context(); ctx.upload(scene.list())
buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
for accel in (0, 2, 1, 3):
    opt = tor.make_options(seeding=tor.SEED_SAMPLE, accel=accel)
    ctx.set_stats(False)
    ctx.render_device(cam, H, W, SPP, 2.2, 50, opt, buf.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    ms = ctx.last_kernel_ms()[0]
    ctx.set_stats(True)
    ctx.render_device(cam, H, W, SPP, 2.2, 50, opt, buf.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    wl = ctx.last_wave_log(); st = ctx.last_stats()
    m = (1 << 21) - 1
    sec = np.stack([wl[:, 6] & m, (wl[:, 6] >> 21) & m, (wl[:, 6] >> 42) & m, wl[:, 7] & m, (wl[:, 7] >> 21) & m, (wl[:, 7] >> 42) & m], axis=1).astype(np.float64)
    tot = sec[:, 5].sum()
    names = ["refill", "loop", "resolve", "shade", "deposit"]
    trips = float((wl[:, 5] >> 32).sum()) / max(float((wl[:, 2] & ((1 << 40) - 1)).sum()), 1.0)
    print(f"c5 accel {accel}: {H*W*SPP/ms/1e3:7.0f} Msamples/s trips/iter {trips:.1f} cand/query {st.candidates/max(st.hit_queries,1):.1f} queries/sample {st.hit_queries/max(st.samples,1):.2f}", {n: round(float(sec[:, i].sum()/tot), 3) for i, n in enumerate(names)}, flush=True)
