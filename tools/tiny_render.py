import importlib, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hrt = importlib.import_module("hai719-raytracing_amd")
hrt.init(0)
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
w, h, spp = (int(x) for x in (sys.argv[2:5] if len(sys.argv) > 4 else (64, 64, 2)))
s = hrt.HostScene().setup(name, w / h, 1); d = s.flatten(); cam = hrt.default_camera(w / h)
dev = hrt.DeviceScene(d)
img, st = dev.render(cam, w, h, spp, 1)
st16 = (C.c_uint64 * 16)()
lib = hrt.device_lib(); lib.hrt_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
lib.hrt_debug_read_stamps(dev._h, st16)
tot = sum(st16) or 1
names = ["regen", "spheres", "quad_filter", "quad_refine", "gates", "mesh_stage", "shade", "direct", "scatter", "end", "tile_io"]
print(name, w, h, spp, "kernel ms", round(st.kernel_ms, 3), "mean", float(img.mean()))
print("stamps %: " + "  ".join(f"{n} {100*st16[i]/tot:.1f}" for i, n in enumerate(names)))
