import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hrt = importlib.import_module("hai719-raytracing_amd")
hrt.init(0)
name, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = hrt.DeviceScene(hrt.HostScene().setup(name, w / h, 1).flatten()); cam = hrt.default_camera(w / h)
lib = hrt.device_lib(); lib.hrt_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
try:
    img, st = dev.render(cam, w, h, spp, 11, flags=hrt.FLAG_STREAM_KERNEL)
    print("ok", st.kernel_ms)
except Exception as e:
    print("ERR", e)
o = (C.c_uint64 * 16)(); lib.hrt_debug_read_stamps(dev._h, o)
print([hex(x) for x in o])
