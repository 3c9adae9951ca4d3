"""Workgroups per CU of the metric's kernel (TOR_BLOCKS_PER_CU = 3: the 168-register variant; 4: the 128-register variant
integrate_kernel<1, 2, 4, 0, 0>): kernel time of the strict float64 brute force, per-sample streams; canvases compared."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
tor = importlib.import_module("trace-of-radiance_amd")
H, W = 1080, 1920
st = torch.cuda.current_stream().cuda_stream
scene, cam = tor.random_scene(0xFACADE), tor.camera()
ref = None
for spp in (100, 1000):
    for bpc in ("3", "4", "5"):
        os.environ["TOR_BLOCKS_PER_CU"] = bpc
        ctx = tor.Context(0)
        ctx.upload(scene.list())
        buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
        opt = tor.make_options(seeding=tor.SEED_SAMPLE)
        ms = []
        for _ in range(3):
            ctx.render_device(cam, H, W, spp, 2.2, 50, opt, buf.data_ptr(), st)
            torch.cuda.synchronize()
            ms.append(ctx.last_kernel_ms()[0])
        if bpc == "3":
            ref = buf.clone()
        print(f"spp {spp} workgroups/CU {bpc}: kernel {min(ms):8.2f} ms  {H * W * spp / min(ms) / 1e3:7.1f} Msamples/s  same canvas: {bool(torch.equal(buf, ref))}", flush=True)
        ctx.close()
os.environ.pop("TOR_BLOCKS_PER_CU", None)
