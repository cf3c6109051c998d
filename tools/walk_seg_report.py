"""Diagnostic (libhrt_var_wseg.so, built with -DHRT_WALK_SEG): clocks of the segments of a KD-walk trip in the streaming kernel's
T chunks, summed over waves (every stamp drains the wave's memory counters, so segments are serialised), and lane occupancy."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hrt = importlib.import_module("hai719-raytracing_amd")
hrt.init(0)
w, h = 1920, 1080
lib = hrt.device_lib(); lib.hrt_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
for spec in sys.argv[1:] or ["cornell_mesh:32"]:
    name, spp = spec.split(":"); spp = int(spp)
    dev = hrt.DeviceScene(hrt.HostScene().setup(name, w / h, 1).flatten()); cam = hrt.default_camera(w / h)
    dev.render(cam, w, h, 2, 1)
    _, st = dev.render(cam, w, h, spp, 1)
    o = (C.c_uint64 * 16)(); lib.hrt_debug_read_stamps(dev._h, o)
    seg = list(o[0:5]); tot = max(1, sum(seg)); trips, leaf_trips, batches = max(1, o[8]), max(1, o[9]), max(1, o[10])
    print(f"{name} {w}x{h}@{spp}: kernel {st.kernel_ms:.1f} ms; {trips / (w * h * spp):.3f} wave-trips per sample x 64 = {64 * trips / (w * h * spp):.2f}; "
          f"{tot / trips:.0f} clocks per wave-trip; lanes active per trip {o[11] / trips:.1f}, at a leaf {o[5] / leaf_trips:.1f} "
          f"({100 * leaf_trips / trips:.0f} % of trips), testing triangles {o[6] / batches:.1f} ({100 * batches / trips:.0f} % of trips)")
    for k, nm in enumerate(["descent (2 levels)", "leaf nodelets", "triangle batch (planes, rows)", "exit face + rope", "walk start (root clip, irregular triangles)"]):
        print(f"  {nm:44s} {seg[k] / trips:8.0f} clocks per trip  {100.0 * seg[k] / tot:5.1f} %")
