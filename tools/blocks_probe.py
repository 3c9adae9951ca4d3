"""Throughput of the accelerated modes on C2 (one line per seeding x accel); env knobs are read by the library."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
tor = importlib.import_module("trace-of-radiance_amd")
H, W, SPP = 1080, 1920, int(os.environ.get("SPP", "100"))
scene, cam = tor.random_scene(0xFACADE), tor.camera()
ctx = tor.Context()
ctx.upload(scene.list())
buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
accels = [int(x) for x in os.environ.get("ACCELS", "0,2,1,3").split(",")]
out = []
for seeding in (tor.SEED_SAMPLE, tor.SEED_PIXEL):
    for accel in accels:
        opt = tor.make_options(seeding=seeding, accel=accel)
        best = 1e9
        for _ in range(3):
            ctx.render_device(cam, H, W, SPP, 2.2, 50, opt, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            best = min(best, ctx.last_kernel_ms()[0])
        out.append(f"s{seeding}a{accel}:{H * W * SPP / best / 1e3:.0f}")
print(os.environ.get("TAG", ""), " ".join(out), flush=True)
