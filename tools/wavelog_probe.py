import importlib, sys, numpy as np, torch
sys.path.insert(0, '.')
tor = importlib.import_module("trace-of-radiance_amd")
scene, cam = tor.random_scene(0xFACADE), tor.camera()
ctx = tor.Context(0); ctx.upload(scene.list()); ctx.set_stats(True)
H, W, spp = 1080, 1920, 100
buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
for mode in (tor.SEED_PIXEL, tor.SEED_SAMPLE):
    for rep in range(2):
        ctx.render_device(cam, H, W, spp, 2.2, 50, tor.make_options(seeding=mode), buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    ms, _ = ctx.kernel_ms_mean(1)
    wl = ctx.last_wave_log().astype(np.float64)
    t0 = wl[:, 0].min()
    start = (wl[:, 0] - t0) / 100e3; end = (wl[:, 1] - t0) / 100e3   # ms (100 MHz)
    raw = ctx.last_wave_log()
    hw = (raw[:, 3] >> np.uint64(44)).astype(np.int64); q = (raw[:, 3] & np.uint64((1 << 44) - 1)).astype(np.float64)
    it = wl[:, 2] & ((1 << 40) - 1)
    slot = hw & 15; simd = (hw >> 4) & 3
    rate_us = (wl[:, 1] - wl[:, 0]) / 100.0 / it
    for sl in sorted(set(slot.tolist())):
        mk = slot == sl
        print(f'   hw slot {sl}: {mk.sum()} waves, median us/iter {np.median(rate_us[mk]):.1f}, median iters {np.median(it[mk]):.0f}')
    print(f"mode {mode}: kernel {ms:.1f} ms, waves {len(wl)}; start max {start.max():.2f} ms; end: min {end.min():.1f} p10 {np.percentile(end,10):.1f} p50 {np.percentile(end,50):.1f} p90 {np.percentile(end,90):.1f} p99 {np.percentile(end,99):.1f} max {end.max():.1f}")
    print(f"   iterations/wave: min {it.min():.0f} p50 {np.median(it):.0f} p90 {np.percentile(it,90):.0f} max {it.max():.0f}; lane util per wave p10 {np.percentile(q/(64*it),10):.3f} p50 {np.median(q/(64*it)):.3f}")
    # time per iteration vs time
    rate = (end - start) * 1e3 / it
    print(f"   us/iteration per wave: p10 {np.percentile(rate,10):.1f} p50 {np.median(rate):.1f} p90 {np.percentile(rate,90):.1f}")
    texh = (wl[:, 4] - t0) / 100e3; itx = (raw[:, 5] & np.uint64(0xffffffff)).astype(np.float64)   # word 5 = iterations at exhaustion | resolve-loop trips << 32
    print(f'   counter dry seen at: min {texh.min():.1f} p50 {np.median(texh):.1f} max {texh.max():.1f} ms; iterations after dry: p50 {np.median(it - itx):.0f} p90 {np.percentile(it - itx, 90):.0f} max {(it - itx).max():.0f}; tail duration p50 {np.median(end - texh):.1f} p90 {np.percentile(end - texh, 90):.1f} max {(end - texh).max():.1f} ms')
    hist, edges = np.histogram(end, bins=12)
    print("   end-time histogram:", list(zip(np.round(edges[:-1]).astype(int), hist)))
