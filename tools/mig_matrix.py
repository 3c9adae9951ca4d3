"""Chain hand-off experiments: one frame/shard under several settings of the hand-off knobs (new context per setting).
usage (GPU box): python tools/mig_matrix.py HxWxSPP[:k/N] 'ENV=V,ENV=V' 'ENV=V' ..."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
tor = importlib.import_module("trace-of-radiance_amd")
scene, cam = tor.random_scene(0xFACADE), tor.camera()
st = torch.cuda.current_stream().cuda_stream
spec = sys.argv[1]
size, _, shard = spec.partition(":")
H, W, spp = (int(x) for x in size.split("x"))
k, N = (int(x) for x in shard.split("/")) if shard else (0, 1)
rows = len(tor.shard_rows(H, 1, k, N))
ref = None
KNOBS = ("TOR_FLOOR_THETA", "TOR_KEY_MODE", "TOR_PROBE_SPP", "TOR_TAIL_REST", "TOR_CHAIN_THETA", "TOR_MIG_FLAGS", "TOR_MIGRATE", "TOR_SRV_FRAC", "TOR_SRV_PATIENCE_US", "TOR_SRV_MIN_FRAC", "TOR_PUSH_THETA", "TOR_TAIL_LANES", "TOR_HOT_FRAC", "TOR_PRIO_SHIFT", "TOR_BLOCKS_PER_CU")
for setting in sys.argv[2:]:
    for kn in KNOBS:
        os.environ.pop(kn, None)
    for kv in setting.split(","):
        if kv:
            a, b = kv.split("=")
            os.environ[a] = b
    ctx = tor.Context(0)
    ctx.upload(scene.list())
    o = tor.make_options(seeding=tor.SEED_PIXEL, accel=3, shard_index=k, shard_count=N, row_tile=1)
    buf = torch.zeros((rows, W, 3), dtype=torch.float64, device="cuda")
    ctx.render_device(cam, H, W, spp, 2.2, 50, o, buf.data_ptr(), st)
    torch.cuda.synchronize()
    ts, ks = [], []
    for _ in range(2):
        t = time.perf_counter()
        ctx.render_device(cam, H, W, spp, 2.2, 50, o, buf.data_ptr(), st)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
        ks.append(ctx.last_kernel_ms()[0])
    if ref is None:
        ref = buf.clone()
    c = ctx.last_handoff_counters()
    print(f"{spec} [{setting:48s}] step {min(ts):8.2f} ms kernel {min(ks):8.2f} ms  same={bool(torch.equal(ref, buf))} srvWG {c['server_workgroups']:3d} thr {c['push_threshold']:6d} "
          f"hot {c['hot_pushes']:6d} tail {c['tail_pushes']:6d} | ms: dry {c['us_counter_dry'] / 1e3:6.1f} lane_end {c['us_lane_end'] / 1e3:6.1f} hot_done {c['us_hot_done'] / 1e3:6.1f} "
          f"tail_done {c['us_tail_done'] / 1e3:6.1f} conv {c['servers_converted']:4d} thr_end {c['push_threshold_end']:6d} | Mits hot {c['its_hot'] / 1e6:6.2f} tail {c['its_tail'] / 1e6:6.2f}", flush=True)
    ctx.close()
