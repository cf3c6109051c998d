"""Times the trace kernel on the bench scenes (several rounds, reports min/median) and checks parity at small size."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
hrt = importlib.import_module("hai719-raytracing_amd")
import oracle_lib as O
hrt.init(0)
tag = os.environ.get("HRT_LIBNAME", "libhrt.so") + " lds=" + os.environ.get("HRT_LDS_KB", "32") + " leaf=" + os.environ.get("HRT_KD_LEAF", "4")
scenes = sys.argv[1:] or ["cornell_mesh", "random_spheres"]
for name in scenes:
    w, h = 1920, 1080
    spp = {"cornell_mesh": 32, "random_spheres": 8, "mesh_in_box": 32, "cornell_box": 32, "backrooms_pool": 16}.get(name, 16)
    if os.environ.get("HRT_SPP"): spp = int(os.environ["HRT_SPP"])
    s = hrt.HostScene().setup(name, w / h, 1)
    if os.environ.get('HRT_KD_LEAF'): s.set_kd_params(int(os.environ['HRT_KD_LEAF']), int(os.environ.get('HRT_KD_DEPTH', '0')))
    d = s.flatten(); cam = hrt.default_camera(w / h)
    dev = hrt.DeviceScene(d)
    # parity spot check
    img, st = dev.render(cam, 96, 54, 2, 5)
    ref = O.OracleScene(d).render(cam, 96, 54, 2, seed=5, threads=0)
    bad = (np.abs(img - ref) > 1e-5 * np.maximum(1, np.abs(ref))).any(axis=2).mean()
    dev.render(cam, w, h, 2, 1)
    ms = []
    for _ in range(5):
        _, st = dev.render(cam, w, h, spp, 1); ms.append(st.kernel_ms)
    ms = np.array(ms)
    import ctypes as C
    st16 = (C.c_uint64 * 16)()
    hrt.device_lib().hrt_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
    hrt.device_lib().hrt_debug_read_stamps(dev._h, st16)
    tot = sum(st16) or 1
    if sum(st16):
        names = ["regen", "spheres", "quad_filter", "quad_refine", "gates", "mesh_stage", "shade", "direct", "scatter", "end", "tile_io"]
        print("   stamps %: " + "  ".join(f"{n} {100*st16[i]/tot:.1f}" for i, n in enumerate(names)), flush=True)
        print("   raw stamps: " + " ".join(str(int(x)) for x in st16), flush=True)
    print(f"{tag:28s} {name:15s} {w}x{h}@{spp}: min {ms.min():8.2f} ms  med {np.median(ms):8.2f} ms -> {w*h*spp/ms.min()/1e3:8.1f} Msamples/s  vgpr {st.vgprs} waves {st.waves_launched} lds {st.lds_bytes}  bad_px {bad*100:.3f}%", flush=True)
