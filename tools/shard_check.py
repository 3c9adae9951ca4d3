import importlib, sys, time, json
sys.path.insert(0, '.')
import torch
tor = importlib.import_module("trace-of-radiance_amd")
scene, cam = tor.random_scene(0xFACADE), tor.camera()
ctx = tor.Context(0); ctx.upload(scene.list())
st = torch.cuda.current_stream().cuda_stream
def run(name, spp, **opt):
    o = tor.make_options(seeding=tor.SEED_SAMPLE, **opt)
    rows = len(tor.shard_rows(1080, opt.get("row_tile", 1), opt.get("shard_index", 0), opt.get("shard_count", 1)))
    buf = torch.empty((rows, 1920, 3), dtype=torch.float64, device="cuda")
    ctx.render_device(cam, 1080, 1920, 8, 2.2, 50, o, buf.data_ptr(), st); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.perf_counter(); ctx.render_device(cam, 1080, 1920, spp, 2.2, 50, o, buf.data_ptr(), st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(name, "rows", rows, "ms", [round(x * 1e3, 1) for x in ts], "Msamples/s", round(rows * 1920 * spp / min(ts) / 1e6, 1), flush=True)
run("single GPU, 100 spp", 100)
for k in (0, 3, 7):
    run(f"rank {k} of 8, row-cyclic, 800 spp", 800, shard_index=k, shard_count=8, row_tile=1)
for k in (0, 7):
    run(f"rank {k} of 8, tiles of 8 rows, 800 spp", 800, shard_index=k, shard_count=8, row_tile=8)
