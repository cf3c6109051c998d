"""Diagnostic (libhrt_var_dbg.so, -DHRT_SP_DEBUG): where the streaming kernel's waves spend their clocks, by chunk class."""
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
    work, alive, cycles, chunks, serial = o[0], o[1], o[2], o[3], o[4]
    cls = list(o[5:11]); tot = max(1, sum(cls))
    names = ["T (KD walk)", "mesh hits", "sphere hits", "square hits", "misses", "G (new paths)"]
    print(f"{name} {w}x{h}@{spp}: kernel {st.kernel_ms:.1f} ms; waves in chunk loops {100 * work / max(1, alive):.1f} % of their life, "
          f"serial section {100 * serial / max(1, alive):.1f} %, waiting for a cycle {100 * o[11] / max(1, alive):.1f} %, {cycles / 4096:.0f} cycles per workgroup-wave, {chunks / max(1, cycles):.2f} chunks per wave per cycle")
    print(f"   cycles that started no path (every unit in flight draining, or the end of the launch): {100 * o[14] / max(1, cycles):.1f} % of the cycles, {100 * o[13] / max(1, work):.1f} % of the clocks in chunk loops")
    print("   " + "  ".join(f"{n} {100 * c / tot:.1f}%" for n, c in zip(names, cls)))
