"""Chain hand-off (DESIGN 4.7 (HISTORY 4.10)) on vs off: same canvas, step time, hand-off counters.
usage (GPU box): python tools/mig_check.py [HxWxSPP[:k/N] ...]   (k/N = row shard k of N)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
tor = importlib.import_module("trace-of-radiance_amd")
scene, cam = tor.random_scene(0xFACADE), tor.camera()
st = torch.cuda.current_stream().cuda_stream
specs = sys.argv[1:] or ["216x384x100", "216x384x1000", "1080x1920x100", "1080x1920x1000:0/8", "1080x1920x1000:3/4", "1080x1920x1000"]
for spec in specs:
    size, _, shard = spec.partition(":")
    H, W, spp = (int(x) for x in size.split("x"))
    k, N = (int(x) for x in shard.split("/")) if shard else (0, 1)
    rows = len(tor.shard_rows(H, 1, k, N))
    out = {}
    for mig in ("0", "1"):
        os.environ["TOR_MIGRATE"] = mig
        ctx = tor.Context(0)
        ctx.upload(scene.list())
        o = tor.make_options(seeding=tor.SEED_PIXEL, accel=3, shard_index=k, shard_count=N, row_tile=1)
        buf = torch.zeros((rows, W, 3), dtype=torch.float64, device="cuda")
        ctx.render_device(cam, H, W, spp, 2.2, 50, o, buf.data_ptr(), st)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            ctx.render_device(cam, H, W, spp, 2.2, 50, o, buf.data_ptr(), st)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
        out[mig] = (buf.clone(), min(ts), ctx.last_handoff_counters())
        ctx.close()
    same = bool(torch.equal(out["0"][0], out["1"][0]))
    print(f"{spec:24s} off {out['0'][1]:8.2f} ms  on {out['1'][1]:8.2f} ms ({rows * W * spp / out['1'][1] / 1e3:7.1f} Msamples/s)  identical={same}  {out['1'][2]}", flush=True)
    if not same:
        d = (out["0"][0] != out["1"][0]).any(dim=2)
        print("   differing pixels:", int(d.sum()), "first:", d.nonzero()[:5].tolist())
