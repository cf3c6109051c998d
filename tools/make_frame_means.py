"""Writes tests/golden/bench_frame_means.json: the float64 mean of every frame bench.py times (the metric line and BASELINE's
five configurations at their true sizes), rendered on the GPU box by the HIP path.  Pixels are deterministic -- per-path keyed
random streams, sums in sample order, one kernel arithmetic for every schedule -- so these are exact values: bench.py compares
with tolerance 0 and refuses to print a rate for a frame whose mean differs.  Regenerate only when a change is MEANT to move
pixels (it never was in round 3: pruning, filters and schedules leave every pixel bit-identical, and the values of the scenes
round 2 also timed are unchanged).

    python tools/make_frame_means.py            (GPU box; ~15 s)
"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
hrt = importlib.import_module("hai719-raytracing_amd")
hrt.init(0)
out = {}
for name, w, h, spp in [(bench.SCENE, bench.W, bench.H, bench.SPP)] + [(n, w, h, s) for _, n, w, h, s in bench.OTHER_CONFIGS]:
    dev = hrt.DeviceScene(hrt.HostScene().setup(name, w / h, 1).flatten())
    img, st = dev.render(hrt.default_camera(w / h), w, h, spp, bench.SEED)
    # the same frame through the other kernel form where that is affordable: the committed value is schedule-independent
    if w * h * spp <= 1920 * 1080 * 64:
        other, _ = dev.render(hrt.default_camera(w / h), w, h, spp, bench.SEED, flags=hrt.FLAG_WAVE_KERNEL)
        assert (other == img).all(), name
    out[f"{name} {w}x{h}@{spp} seed {bench.SEED}"] = bench.frame_mean(img)
    print(name, w, h, spp, out[f"{name} {w}x{h}@{spp} seed {bench.SEED}"], round(st.kernel_ms, 2), "ms", flush=True)
    dev.close()
path = os.path.join(ROOT, "gpurun_out", "bench_frame_means.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path, "(copy to tests/golden/bench_frame_means.json)")
