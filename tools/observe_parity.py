"""Prints what the parity tests observe (not what they assert): worst |gpu - oracle| and the number of differing
first-hit ids per scene, so that the asserted thresholds can be set to the observed ones.  Run on the GPU box."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
hrt = importlib.import_module("hai719-raytracing_amd")
import oracle_lib as O
from scene_util import many_squares
hrt.init(0)
lib = hrt.device_lib()
lib.hrt_render_aov.argtypes = [C.c_void_p, C.POINTER(hrt.Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]

def aov(dev, cam, w, h, which):
    out = np.empty((h, w, 3), np.float32)
    assert lib.hrt_render_aov(dev._h, C.byref(cam), w, h, which, out.ctypes.data) == 0
    return out

def one(label, host, w, h, spp, seed):
    desc = host.flatten(); dev = hrt.DeviceScene(desc); cam = hrt.default_camera(w / h)
    ref = O.OracleScene(desc)
    img, _ = dev.render(cam, w, h, spp, seed=seed)
    want = ref.render(cam, w, h, spp, seed=seed, threads=0)
    d = np.abs(img.astype(np.float64) - want)
    rel = d / np.maximum(1.0, np.abs(want))
    ra = ref.aov(cam, w, h); hit = aov(dev, cam, w, h, 0)
    ids = int(((hit[..., 1] != ra["hit"][..., 1]) | (hit[..., 2] != ra["hit"][..., 2])).sum())
    tdiff = np.abs(hit[..., 0] - ra["hit"][..., 0]).max()
    other = [np.abs(aov(dev, cam, w, h, k) - ra[key]).max() for k, key in ((1, "normal"), (2, "albedo"), (3, "emission"))]
    print(f"{label:28s} {w}x{h}@{spp}: max|d| {d.max():.3g} max rel {rel.max():.3g} px>1e-6 {(rel > 1e-6).any(axis=2).sum()} "
          f"px>2e-7 {(rel > 2e-7).any(axis=2).sum()} | ids differ {ids} t {tdiff:.3g} n/a/e {other[0]:.3g} {other[1]:.3g} {other[2]:.3g}", flush=True)

for name in ["cornell_box", "cornell_mesh", "random_spheres", "mesh_in_box", "backrooms_pool", "single_sphere", "single_square", "mesh",
             "rt_in_a_weekend", "debug_refraction", "flamingo", "raccoon", "flamingo_pond", "flamingo_lake"]:
    for (w, h, spp, seed) in ((64, 36, 4, 1), (160, 90, 3, 4), (240, 135, 2, 9)):
        one(name, hrt.HostScene().setup(name, w / h, 1), w, h, spp, seed)
for nq, nm in ((33, 0), (64, 2), (70, 5), (3, 32)):
    one(f"many_squares:{nq}:{nm}", many_squares(hrt, nq, nm), 72, 40, 3, 2)
