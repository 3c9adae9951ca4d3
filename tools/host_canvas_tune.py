"""SURVEY 8(d) region vs resident region on C2 (1920x1080x100, per-sample streams, float64 brute force) for several
D2H staging settings (TOR_COPY_THREADS x TOR_COPY_CHUNK_KB)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
tor = importlib.import_module("trace-of-radiance_amd")
scene, cam = tor.random_scene(0xFACADE), tor.camera()
h, w, spp = 1080, 1920, int(os.environ.get("SPP", "100"))
opt = tor.make_options(seeding=tor.SEED_SAMPLE)
ctx = tor.Context(); ctx.upload(scene.list())
dev = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def resident(n=5):
    ctx.render_device(cam, h, w, spp, 2.2, 50, opt, dev.data_ptr(), s); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        ctx.render_device(cam, h, w, spp, 2.2, 50, opt, dev.data_ptr(), s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
r = resident()
print(f"resident {r:.3f} ms")
cv = tor.new_canvas(h, w, spp, 2.2)
for threads in (1, 4, 8, 16, 32):
    for chunk in (512, 2048, 8192):
        os.environ["TOR_COPY_THREADS"] = str(threads); os.environ["TOR_COPY_CHUNK_KB"] = str(chunk)
        tor.render(cv, cam, scene.list(), 50, opt)
        t = time.perf_counter()
        for _ in range(5):
            tor.render(cv, cam, scene.list(), 50, opt)
        dt = (time.perf_counter() - t) / 5 * 1e3
        print(f"threads {threads:2d} chunk {chunk:5d} KB: {dt:.3f} ms  (+{dt - r:.3f} ms, ratio {r / dt:.4f})", flush=True)
print(f"resident again {resident():.3f} ms")
