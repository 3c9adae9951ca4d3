"""TOR_BLOCKS_PER_CU 2 / 3 / 4 for the four main modes at configs[1] (is the round-1/2 launch shape still right for the round-4 kernels?)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
tor = importlib.import_module("trace-of-radiance_amd")
H, W, spp = 1080, 1920, int(os.environ.get("SPP", "100"))
st = torch.cuda.current_stream().cuda_stream
scene, cam = tor.random_scene(0xFACADE), tor.camera()
for label, seeding, accel in (("sample brute", tor.SEED_SAMPLE, 0), ("pixel brute", tor.SEED_PIXEL, 0), ("sample b+f32", tor.SEED_SAMPLE, 3), ("pixel b+f32", tor.SEED_PIXEL, 3)):
    line = f"{label:14s}"
    for bpc in ("2", "3", "4"):
        os.environ["TOR_BLOCKS_PER_CU"] = bpc
        ctx = tor.Context(0)
        ctx.upload(scene.list())
        buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
        opt = tor.make_options(seeding=seeding, accel=accel)
        ms = []
        for _ in range(4):
            torch.cuda.synchronize(); t = time.perf_counter()
            ctx.render_device(cam, H, W, spp, 2.2, 50, opt, buf.data_ptr(), st)
            torch.cuda.synchronize(); ms.append((time.perf_counter() - t) * 1e3)
        line += f"   bpc {bpc}: {min(ms[1:]):7.2f} ms {H * W * spp / min(ms[1:]) / 1e3:7.1f}"
        ctx.close()
    print(line, flush=True)
os.environ.pop("TOR_BLOCKS_PER_CU", None)
