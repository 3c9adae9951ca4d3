"""Where does a server's bounce go?  (DESIGN 4.7 (HISTORY 4.10): the floor of a frame with a long pixel chain is chain length x the time of one
dependent bounce in serve_chains.)  Needs the profiling build (make -C trace-of-radiance_amd/csrc prof): shader-clock stamps around
the phases of every served bounce -- closest-hit query (ray set-up, slab tests, candidate records, float64 tests, wave minimum),
shading per material, sky, camera ray of a new sample.
usage (GPU box): python tools/server_phases.py [H W spp]"""
import importlib, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    tor = importlib.import_module("trace-of-radiance_amd")
    tor.LIB_PATH = os.path.join(os.path.dirname(tor.LIB_PATH), "libtor_mi355x_prof.so")
    H, W, spp = (int(x) for x in sys.argv[2:5])
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    st = torch.cuda.current_stream().cuda_stream
    ctx = tor.Context(0)
    ctx.upload(scene.list())
    buf = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    o = tor.make_options(seeding=tor.SEED_PIXEL, accel=3)
    for _ in range(2):
        ctx.render_device(cam, H, W, spp, 2.2, 50, o, buf.data_ptr(), st)
        torch.cuda.synchronize()
    print("kernel_ms", ctx.last_kernel_ms()[0])
    c = ctx.last_handoff_counters()
    print("counters", c)
    sys.exit(0)
H, W, spp = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (216, 384, 1000)
r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(H), str(W), str(spp)], capture_output=True, text=True)
m = re.search(r"serve_prof (.*)", r.stderr)
if not m:
    print(r.stdout[-2000:], r.stderr[-2000:])
    sys.exit(1)
tok = m.group(1).split()
v = {tok[i]: int(tok[i + 1]) for i in range(0, len(tok), 2)}
km = float(re.search(r"kernel_ms ([0-9.]+)", r.stdout).group(1))
bounces = v["n_lambertian"] + v["n_metal"] + v["n_dielectric"] + v["n_miss"]
print(f"{W}x{H}x{spp}: kernel {km:.2f} ms; served {bounces} bounces of {v['samples']} samples; shader-clock ticks:")
print(f"  all phases / bounce             {v['total'] / bounces:8.1f}")
print(f"  closest-hit query / bounce      {v['query'] / bounces:8.1f}   ({v['candidates'] / bounces:.1f} candidates, {v['rounds'] / bounces:.2f} rounds of 64)")
for name in ("lambertian", "metal", "dielectric"):
    n = v["n_" + name]
    print(f"  shade {name:10s}              {v[name] / max(n, 1):8.1f}   x {n} ({100.0 * n / bounces:.1f} % of the bounces)")
print(f"  miss -> sky                     {v['miss'] / max(v['n_miss'], 1):8.1f}   x {v['n_miss']} ({100.0 * v['n_miss'] / bounces:.1f} %)")
print(f"  camera ray of a new sample      {v['setup'] / max(v['samples'], 1):8.1f}   x {v['samples']}")
if v.get("q_slab", 0):
    q = [v[k] / bounces for k in ("q_slab", "q_slots", "q_load", "q_test", "q_min")]
    print(f"  inside the query (make prof PROF_LEVEL=2; the stamps serialise, the sum exceeds the un-stamped query): ray set-up + slab tests + ballot {q[0]:.0f}, "
          f"candidate slots {q[1]:.0f}, record loads {q[2]:.0f}, float64 tests {q[3]:.0f}, minimum + winner {q[4]:.0f}")
print(f"  share of the ticks: query {100.0 * v['query'] / v['total']:.1f} %, shading {100.0 * (v['lambertian'] + v['metal'] + v['dielectric']) / v['total']:.1f} %, "
      f"sky {100.0 * v['miss'] / v['total']:.1f} %, camera {100.0 * v['setup'] / v['total']:.1f} %")
