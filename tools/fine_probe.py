"""Finer shares of the bounce iteration of the COOPERATIVE variants (both exact accelerations): ray set-up, box loop, direct float32
segments, segment headers, list building (A), block expansion (B), exact tests (C) -- from the TOR_FINE_PROBE build of the
library (`make -C trace-of-radiance_amd/csrc fine`; loaded through TOR_AB_LIB, never the product).
usage: TOR_AB_LIB=trace-of-radiance_amd/lib/libtor_mi355x_fine.so python tools/fine_probe.py [scene: random | anim]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
tor = importlib.import_module("trace-of-radiance_amd")

H, W, SPP = 1080, 1920, int(os.environ.get("SPP", "20"))
which = sys.argv[1] if len(sys.argv) > 1 else "random"
if which == "anim":
    it = iter(tor.Animation(H, W, 0.005, 0.0, 7.2).scenes(6))
    for _ in range(38):
        cam, scene, _t = next(it)
else:
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
ctx = tor.Context()
ctx.upload(scene.list())
buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
for seeding in (tor.SEED_SAMPLE, tor.SEED_PIXEL):
    opt = tor.make_options(seeding=seeding, accel=3)
    ctx.set_stats(False)
    ctx.render_device(cam, H, W, SPP, 2.2, 50, opt, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ms = ctx.last_kernel_ms()[0]
    ctx.set_stats(True)
    ctx.render_device(cam, H, W, SPP, 2.2, 50, opt, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    wl = ctx.last_wave_log()
    st = ctx.last_stats()
    m = (1 << 21) - 1
    tot = float(((wl[:, 7] >> 42) & m).sum()) * 4096.0
    coarse = {"refill": wl[:, 6] & m, "loop": (wl[:, 6] >> 21) & m, "resolve": (wl[:, 6] >> 42) & m, "shade": wl[:, 7] & m, "deposit": (wl[:, 7] >> 21) & m}
    lo = lambda w: (w & 0xffffffff).astype(np.float64).sum() * 256.0
    hi = lambda w: (w >> 32).astype(np.float64).sum() * 256.0
    fine = {"setup": lo(wl[:, 0]), "boxes": hi(wl[:, 0]), "f32 direct": lo(wl[:, 1]), "headers": hi(wl[:, 1]), "lists (A)": lo(wl[:, 4]),
            "blocks (B)": hi(wl[:, 4]), "exact (C)": lo(wl[:, 5])}
    iters = float((wl[:, 2] & ((1 << 40) - 1)).sum())
    q = max(st.hit_queries, 1)
    print(f"{which} seeding {seeding} accel 3: {H * W * SPP / ms / 1e3:8.1f} Msamples/s  kernel {ms:7.2f} ms   per query: blocks entered "
          f"{st.block_tests / 8 / q:.2f}, exact tests {st.exact_tests / q:.2f}; queries per iteration {st.hit_queries / iters:.1f}; "
          f"cycles per iteration {tot / iters:.0f}")
    print("   coarse:", {k: round(float(v.sum()) * 4096.0 / tot, 3) for k, v in coarse.items()})
    print("   fine  :", {k: round(v / tot, 3) for k, v in fine.items()}, flush=True)
