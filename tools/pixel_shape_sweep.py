import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tor = importlib.import_module("trace-of-radiance_amd")
H, W, SPP = 1080, 1920, int(os.environ.get("SPP", "100"))
scene, cam = tor.random_scene(0xFACADE), tor.camera()
ctx = tor.Context(); ctx.upload(scene.list())
buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
opt = tor.make_options(seeding=tor.SEED_PIXEL, accel=0)
ts = []
for _ in range(6):
    ctx.render_device(cam, H, W, SPP, 2.2, 50, opt, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    ctx.render_device(cam, H, W, SPP, 2.2, 50, opt, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ts.sort()
print(os.environ.get("TAG"), "pixel fp64 step ms (incl. probe):", " ".join(f"{t*1e3:.1f}" for t in ts), f"-> median {H*W*SPP/ts[len(ts)//2]/1e6:.0f} Msamples/s")
