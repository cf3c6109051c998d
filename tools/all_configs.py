"""Kernel time of the five BASELINE.json configurations at their true sizes (one launch each, HIP events)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hrt = importlib.import_module("hai719-raytracing_amd")
hrt.init(0)
CFG = [("cfg1", "cornell_box", 256, 256, 4), ("cfg2", "cornell_mesh", 1920, 1080, 64), ("cfg2 (metric)", "cornell_mesh", 1920, 1080, 256),
       ("cfg3", "random_spheres", 1920, 1080, 256), ("cfg4", "mesh_in_box", 3840, 2160, 512), ("cfg5", "backrooms_pool", 3840, 2160, 1024)]
print("| config | scene | size | kernel ms | Msamples/s |\n|---|---|---|---|---|")
for tag, name, w, h, spp in CFG:
    s = hrt.HostScene().setup(name, w / h, 1); d = s.flatten(); cam = hrt.default_camera(w / h)
    dev = hrt.DeviceScene(d)
    dev.render(cam, 64, 64, 1, 1)  # first-launch costs out of the way
    _, st = dev.render(cam, w, h, spp, 1)
    print(f"| {tag} | {name} | {w}x{h} @ {spp} | {st.kernel_ms:.1f} | {w*h*spp/st.kernel_ms/1e3:.0f} |", flush=True)
