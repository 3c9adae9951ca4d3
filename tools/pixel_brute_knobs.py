"""TOR_SEED_PIXEL float64 brute force under single schedule knobs (which part of the round-2 schedule still pays with the round-4 loop?)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
tor = importlib.import_module("trace-of-radiance_amd")
H, W = 1080, 1920
st = torch.cuda.current_stream().cuda_stream
scene, cam = tor.random_scene(0xFACADE), tor.camera()
SET = [("default", {}), ("BACK_SLOT=0", {"TOR_BACK_SLOT": "0"}), ("PRIO_SHIFT=0", {"TOR_PRIO_SHIFT": "0"}), ("HOT_FRAC=0", {"TOR_HOT_FRAC": "0"}),
       ("BACK_SLOT=0 PRIO_SHIFT=0", {"TOR_BACK_SLOT": "0", "TOR_PRIO_SHIFT": "0"}), ("all three off", {"TOR_BACK_SLOT": "0", "TOR_PRIO_SHIFT": "0", "TOR_HOT_FRAC": "0"}),
       ("LPT_MIN_SPP=0 (no probe, no order)", {"TOR_LPT_MIN_SPP": "0"}), ("TAIL_FRAC=0.1", {"TOR_TAIL_FRAC": "0.1"}), ("TAIL_FRAC=0.35", {"TOR_TAIL_FRAC": "0.35"})]
ref = None
for spp in (100, 1000):
    for name, env in SET:
        for k in ("TOR_BACK_SLOT", "TOR_PRIO_SHIFT", "TOR_HOT_FRAC", "TOR_LPT_MIN_SPP", "TOR_TAIL_FRAC"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ctx = tor.Context(0)
        ctx.upload(scene.list())
        buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
        opt = tor.make_options(seeding=tor.SEED_PIXEL)
        ms = []
        for _ in range(5 if spp == 100 else 3):
            torch.cuda.synchronize()
            import time
            t = time.perf_counter()
            ctx.render_device(cam, H, W, spp, 2.2, 50, opt, buf.data_ptr(), st)
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t) * 1e3)
        if ref is None or ref.shape != buf.shape or name == "default":
            ref = buf.clone()
        print(f"spp {spp:4d} {name:36s} step min {min(ms[1:]):8.2f} ms  median {sorted(ms[1:])[len(ms[1:]) // 2]:8.2f}  {H * W * spp / min(ms[1:]) / 1e3:7.1f} Msamples/s  same: {bool(torch.equal(buf, ref))}", flush=True)
        ctx.close()
