"""What does a SEGMENT cost the float64 brute force?  The same 480 small movers along y + ground + 3 big spheres (random_scene's shape),
with the movers dealt to G = 1, 2, 3, 4, 6, 8 time groups -- (time0, time1) pairs that differ in the last bits only, so that the
geometry, the candidates and the paths are (all but) the same while the layout gets G segments of 480 / G instead of one.
usage: python tools/segment_cost.py [spp]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
tor = importlib.import_module("trace-of-radiance_amd")
H, W, SPP = 1080, 1920, int(sys.argv[1]) if len(sys.argv) > 1 else 50
base = tor.random_scene(0xFACADE).to_records()
cam = tor.camera()
buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for general in (False, True):
    for G in (1, 2, 3, 4, 6, 8):
        recs = base.copy()
        k = 0
        for r in recs:
            if r[0] == 1:
                g = k % G
                r[8] = 1.0 + g * 2.0 ** -40          # time1 of group g (time0 = 0): another bit pattern, the same motion to 1e-12
                if general:
                    r[4] += 0.05 * ((k * 7919) % 13 - 6) / 6.0   # the centre travels in x too: xkind 13 (4-fma stage one, first form per lane)
                k += 1
            else:
                # statics resting at the movers' height would join ONE of the groups: lift them a hair so that every G has the same statics segment
                pass
        scene = tor.Scene.from_records(recs)
        segs = tor.debug_layout_segments(scene.list())
        ctx = tor.Context(0)
        ctx.upload(scene.list())
        opt = tor.make_options(seeding=tor.SEED_SAMPLE, accel=0)
        best = 1e9
        for _ in range(3):
            ctx.render_device(cam, H, W, SPP, 2.2, 50, opt, buf.data_ptr(), st)
            torch.cuda.synchronize()
            best = min(best, ctx.last_kernel_ms()[0])
        ctx.set_stats(True)
        ctx.render_device(cam, H, W, SPP, 2.2, 50, opt, buf.data_ptr(), st)
        torch.cuda.synchronize()
        s = ctx.last_stats()
        ctx.close()
        print(f"{'general movers' if general else 'movers along y'}  G = {G}: {len(segs)} segments {[(x[0], x[1]) for x in segs]}  kernel {best:7.2f} ms  "
              f"{H * W * SPP / best / 1e3:7.1f} Msamples/s  queries/sample {s.hit_queries / s.samples:.3f}  exact tests/query {s.exact_tests / s.hit_queries:.2f}", flush=True)
