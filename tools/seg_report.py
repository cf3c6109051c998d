"""Diagnostic (libhrt_var_seg.so, built with -DHRT_SP_SEG): clocks of the segments of a square-hit chunk, summed over waves."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hrt = importlib.import_module("hai719-raytracing_amd")
hrt.init(0)
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_mesh"; spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32; w, h = 1920, 1080
dev = hrt.DeviceScene(hrt.HostScene().setup(name, w / h, 1).flatten()); cam = hrt.default_camera(w / h)
dev.render(cam, w, h, 2, 1)
_, st = dev.render(cam, w, h, spp, 1)
out = (C.c_uint64 * 16)()
lib = hrt.device_lib(); lib.hrt_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]; lib.hrt_debug_read_stamps(dev._h, out)
n = max(1, out[7])
if os.environ.get("HRT_SEG_KIND") == "2":  # T chunks
    tot = sum(out[:3])
    print(f"{name} {w}x{h}@{spp}: kernel {st.kernel_ms:.1f} ms, {n} T chunks, {tot / n:.0f} clocks per chunk, {out[6] / n:.1f} lanes active, "
          f"{out[5] / max(1, out[6]) * 100:.1f} % of the visits finish their walk")
    for k, nm in enumerate(["record load", "walk (6 trips)", "stores + appends"]):
        print(f"  {nm:32s} {out[k] / n:8.0f} clocks  {100.0 * out[k] / max(1, tot):5.1f} %")
else:
    names = ["record load / camera ray", "shade (mat rows, texel, nmap)", "scatter", "write-back", "spheres + squares", "mesh gates", "sample store + stores + appends", None, "direct light (shadow rays)"]
    tot = sum(out[:7]) + out[8]
    kind = {"1": "square-hit", "3": "sphere-hit", "4": "G (new path)", "5": "mesh-hit"}.get(os.environ.get("HRT_SEG_KIND", "1"), "?")
    print(f"{name} {w}x{h}@{spp}: kernel {st.kernel_ms:.1f} ms, {n} {kind} chunks, {tot / n:.0f} clocks per chunk")
    for k, nm in enumerate(names):
        if nm: print(f"  {nm:32s} {out[k] / n:8.0f} clocks  {100.0 * out[k] / max(1, tot):5.1f} %")
