"""Ad-hoc GPU check: render small configs on the GPU, compare with the oracle, time a bigger run."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
hrt = importlib.import_module("hai719-raytracing_amd")
import oracle_lib as O

def compare(name, w, h, spp, seed=1):
    s = hrt.HostScene().setup(name, w / h, 1)
    d = s.flatten()
    cam = hrt.default_camera(w / h)
    dev = hrt.DeviceScene(d)
    img, st = dev.render(cam, w, h, spp, seed)
    ref = O.OracleScene(d).render(cam, w, h, spp, seed=seed, threads=0)
    diff = np.abs(img - ref)
    tol = 1e-3 * np.maximum(1.0, np.abs(ref))
    bad = (diff > tol).any(axis=2)
    print(f"{name} {w}x{h}@{spp}: kernel {st.kernel_ms:.2f} ms, mean gpu {img.mean():.5f} ref {ref.mean():.5f}, "
          f"bad px {bad.mean()*100:.3f}%, max diff {diff.max():.4g}, mean abs diff {diff.mean():.3g}, nan {np.isnan(img).sum()}, vgpr {st.vgprs}")
    for k, nm in enumerate(["hit", "normal", "albedo", "emission"]):
        pass
    return dev, cam

def aov(name, w, h):
    import ctypes as C
    s = hrt.HostScene().setup(name, w / h, 1)
    d = s.flatten()
    cam = hrt.default_camera(w / h)
    dev = hrt.DeviceScene(d)
    ref = O.OracleScene(d).aov(cam, w, h)
    lib = hrt.device_lib()
    lib.hrt_render_aov.argtypes = [C.c_void_p, C.POINTER(hrt.Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    for k, nm in enumerate(["hit", "normal", "albedo", "emission"]):
        out = np.empty((h, w, 3), np.float32)
        rc = lib.hrt_render_aov(dev._h, C.byref(cam), w, h, k, out.ctypes.data)
        assert rc == 0, lib.hrt_last_error()
        df = np.abs(out - ref[nm])
        if nm == "hit":
            same = (out[..., 1] == ref[nm][..., 1]) & (out[..., 2] == ref[nm][..., 2])
            print(f"  aov {name} hit: kind/index equal {same.mean()*100:.3f}%  max |dt| {df[...,0][same].max():.3g}")
        else:
            print(f"  aov {name} {nm}: max diff {df.max():.3g}  >1e-4: {(df>1e-4).any(axis=2).mean()*100:.3f}%")

if __name__ == "__main__":
    hrt.init(0)
    for name in ["cornell_box", "cornell_mesh", "random_spheres", "mesh_in_box"]:
        aov(name, 160, 90)
    compare("cornell_box", 64, 64, 4)
    compare("cornell_mesh", 128, 72, 4)
    compare("random_spheres", 128, 72, 2)
    compare("mesh_in_box", 128, 72, 4)
    # timing
    for name, w, h, spp in [("cornell_mesh", 1920, 1080, 16), ("random_spheres", 1920, 1080, 4)]:
        s = hrt.HostScene().setup(name, w / h, 1); d = s.flatten(); cam = hrt.default_camera(w / h)
        dev = hrt.DeviceScene(d)
        dev.render(cam, w, h, 1, 1)
        t = time.time(); img, st = dev.render(cam, w, h, spp, 1); dt = time.time() - t
        print(f"TIMING {name} {w}x{h}@{spp}: kernel {st.kernel_ms:.1f} ms -> {w*h*spp/st.kernel_ms/1e3:.1f} Msamples/s (wall {dt*1e3:.1f} ms) waves {st.waves_launched} lds {st.lds_bytes}")
