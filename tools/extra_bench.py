import importlib, sys, time, json
sys.path.insert(0, '.')
import torch
tor = importlib.import_module("trace-of-radiance_amd")
scene, cam = tor.random_scene(0xFACADE), tor.camera()
ctx = tor.Context(0); ctx.upload(scene.list())
def run(name, H, W, spp, **opt):
    o = tor.make_options(**opt)
    rows = len(tor.shard_rows(H, opt.get("row_tile", 1), opt.get("shard_index", 0), opt.get("shard_count", 1)))
    buf = torch.empty((rows, W, 3), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ctx.render_device(cam, H, W, min(spp, 8), 2.2, 50, o, buf.data_ptr(), st); torch.cuda.synchronize()
    t = time.perf_counter()
    ctx.render_device(cam, H, W, spp, 2.2, 50, o, buf.data_ptr(), st); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(json.dumps({"case": name, "Msamples_per_s": round(rows * W * spp / dt / 1e6, 1), "seconds": round(dt, 3)}), flush=True)
for seeding, sn in ((tor.SEED_SAMPLE, "sample"), (tor.SEED_PIXEL, "pixel")):
    for accel, an in ((tor.ACCEL_NONE, "brute"), (tor.ACCEL_BLOCKS, "blocks")):
        run(f"C3 1920x1080x1000 {sn} {an}", 1080, 1920, 1000, seeding=seeding, accel=accel)
for accel, an in ((tor.ACCEL_NONE, "brute"), (tor.ACCEL_BLOCKS, "blocks")):
    run(f"C4 share: 3840x2160x4096, rank 0 of 8 (270 rows) sample {an}", 2160, 3840, 4096, seeding=tor.SEED_SAMPLE, accel=accel,
        shard_index=0, shard_count=8, row_tile=8)
run("C1 384x216x100 pixel brute (tor_render semantics)", 216, 384, 100, seeding=tor.SEED_PIXEL)
run("C1 384x216x100 sample brute", 216, 384, 100, seeding=tor.SEED_SAMPLE)
