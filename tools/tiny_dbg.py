import importlib, os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hrt = importlib.import_module("hai719-raytracing_amd")
hrt.init(0)
name = sys.argv[1]; w, h, spp = (int(x) for x in sys.argv[2:5])
s = hrt.HostScene().setup(name, w / h, 1); d = s.flatten(); cam = hrt.default_camera(w / h)
dev = hrt.DeviceScene(d)
img, st = dev.render(cam, w, h, spp, 1)
st16 = (C.c_uint64 * 16)()
lib = hrt.device_lib(); lib.hrt_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
lib.hrt_debug_read_stamps(dev._h, st16)
print(name, w, h, spp, "ms", round(st.kernel_ms, 3), "mean", float(img.mean()), "nonzero px", int((img.sum(axis=2) != 0).sum()), "waves", st.waves_launched, "lds", st.lds_bytes)
print("dbg [cycles, gen, P, M, S, done, ended]:", list(st16)[:7], "snap cP0 cP1 cF0 cF1 parity ngen total cursor:", list(st16)[7:15])
