"""Box fan-out of TOR_ACCEL_BLOCKS (blocks of 8 objects per culling box; needs the library of commit c39354b, where TOR_BOX_FANOUT was read at upload): step times of the accelerated modes on
configs[1] (random_scene, 485 objects) and on a configs[4] frame (1601 objects), every setting against the brute-force canvas."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
tor = importlib.import_module("trace-of-radiance_amd")

H, W = 1080, 1920
st = torch.cuda.current_stream().cuda_stream


def run(ctx, cam, spp, opt, buf, reps=3):
    ctx.render_device(cam, H, W, spp, 2.2, 50, opt, buf.data_ptr(), st)
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        ctx.render_device(cam, H, W, spp, 2.2, 50, opt, buf.data_ptr(), st)
        torch.cuda.synchronize()
        ms.append(ctx.last_kernel_ms()[0])
    return min(ms)


scene, cam = tor.random_scene(0xFACADE), tor.camera()
anim = tor.Animation(H, W, 0.005, 0.0, 7.2)
acam, ascene, _ = next(iter(anim.scenes(6)))
buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
ref = {}
for name, sc, cm, spp in (("random_scene", scene, cam, int(os.environ.get("SPP", "100"))), ("animation frame", ascene, acam, 64)):
    base = tor.Context(0)
    base.upload(sc.list())
    for seeding in (tor.SEED_SAMPLE, tor.SEED_PIXEL):
        run(base, cm, spp, tor.make_options(seeding=seeding, accel=0), buf, reps=1)
        ref[(name, seeding)] = buf.clone()
    base.close()
    for fan in ("1", "2", "4", "8"):
        os.environ["TOR_BOX_FANOUT"] = fan
        ctx = tor.Context(0)
        ctx.upload(sc.list())
        line = f"{name:16s} fanout {fan}:"
        for label, seeding, accel in (("sample b+f32", tor.SEED_SAMPLE, 3), ("pixel b+f32", tor.SEED_PIXEL, 3), ("sample blocks", tor.SEED_SAMPLE, 1)):
            ms = run(ctx, cm, spp, tor.make_options(seeding=seeding, accel=accel), buf)
            same = bool(torch.equal(buf, ref[(name, seeding)]))
            line += f"  {label} {ms:7.2f} ms {H * W * spp / ms / 1e3:7.1f} Msamples/s {'==' if same else 'DIFFERENT'}"
        print(line, flush=True)
        ctx.close()
os.environ.pop("TOR_BOX_FANOUT", None)
