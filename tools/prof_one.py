"""One frame of one scene, for rocprofv3 --pmc runs: python3 tools/prof_one.py <scene> <spp>"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hrt = importlib.import_module("hai719-raytracing_amd")
hrt.init(0)
name = sys.argv[1]; spp = int(sys.argv[2]); w, h = 1920, 1080
s = hrt.HostScene().setup(name, w / h, 1); d = s.flatten(); cam = hrt.default_camera(w / h)
dev = hrt.DeviceScene(d)
img, st = dev.render(cam, w, h, spp, 1)
print(name, spp, st.kernel_ms, "ms", w * h * spp / st.kernel_ms / 1e3, "Msamples/s")
