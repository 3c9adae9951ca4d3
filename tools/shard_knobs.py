"""All N shards of a frame under several settings of the hand-off knobs: max / mean step time over the shards (min of 2 runs each).
usage: python tools/shard_knobs.py HxWxSPP N 'ENV=V,ENV=V' ..."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
tor = importlib.import_module("trace-of-radiance_amd")
scene, cam = tor.random_scene(0xFACADE), tor.camera()
st = torch.cuda.current_stream().cuda_stream
H, W, spp = (int(x) for x in sys.argv[1].split("x"))
N = int(sys.argv[2])
KN = ("TOR_MIGRATE", "TOR_SRV_FRAC", "TOR_TAIL_REST", "TOR_TAIL_LANES", "TOR_PUSH_THETA", "TOR_FLOOR_THETA", "TOR_MIG_FLAGS", "TOR_KEY_MODE", "TOR_CHAIN_THETA")
for setting in sys.argv[3:]:
    for k in KN:
        os.environ.pop(k, None)
    for kv in setting.split(","):
        if kv:
            a, b = kv.split("=")
            os.environ[a] = b
    ctx = tor.Context(0)
    ctx.upload(scene.list())
    ts = []
    for k in range(N):
        o = tor.make_options(seeding=tor.SEED_PIXEL, accel=3, shard_index=k, shard_count=N, row_tile=1)
        rows = len(tor.shard_rows(H, 1, k, N))
        buf = torch.zeros((rows, W, 3), dtype=torch.float64, device="cuda")
        best = 1e9
        for rep in range(3):
            t = time.perf_counter()
            ctx.render_device(cam, H, W, spp, 2.2, 50, o, buf.data_ptr(), st)
            torch.cuda.synchronize()
            if rep > 0:
                best = min(best, (time.perf_counter() - t) * 1e3)
        ts.append(best)
    print(f"{sys.argv[1]} N={N} [{setting:40s}] max {max(ts):8.2f} mean {sum(ts) / N:8.2f}  " + " ".join(f"{t:.0f}" for t in ts), flush=True)
    ctx.close()
