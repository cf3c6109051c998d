"""One configuration for rocprofv3 runs: python3 tools/prof_one.py <scene> <spp> [w h [launches]]
Exactly `launches` launches of the trace kernel (default 1: a PMC pass then holds the counters of that launch alone)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hrt = importlib.import_module("hai719-raytracing_amd")
hrt.init(0)
name = sys.argv[1]; spp = int(sys.argv[2])
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
launches = int(sys.argv[5]) if len(sys.argv) > 5 else 1
s = hrt.HostScene().setup(name, w / h, 1); d = s.flatten(); cam = hrt.default_camera(w / h)
dev = hrt.DeviceScene(d)
import torch
buf = torch.zeros((hrt.tiles_total(w, h), 64, 3), dtype=torch.float32, device="cuda")
for _ in range(launches):
    dev.render_tiles(cam, w, h, spp, 1, 0, 0, 1, buf.data_ptr(), 0)
    ms = dev.last_kernel_ms()
    print(name, f"{w}x{h}@{spp}", round(ms, 3), "ms", round(w * h * spp / ms / 1e3, 1), "Msamples/s", flush=True)
