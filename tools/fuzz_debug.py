"""One fuzz scene (tools/fuzz_exact.py) rendered by several builds / flags, to find which component changes a pixel."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hrt = importlib.import_module("hai719-raytracing_amd")
hrt.init(0)
src = open(os.path.join(ROOT, "tools", "fuzz_exact.py")).read().split("bad = 0")[0].replace("hrt.init(0)", "")
exec(src)
seed = int(sys.argv[1])
host, ns, nq = scene(seed)
dev = hrt.DeviceScene(host.flatten()); cam = hrt.default_camera(w / h)
frames = {}
for tag, flags in (("shipped", 0), ("stream", hrt.FLAG_STREAM_KERNEL), ("wave", hrt.FLAG_WAVE_KERNEL), ("exact", hrt.FLAG_EXACT_ONLY),
                   ("exact_brute", hrt.FLAG_EXACT_ONLY | hrt.FLAG_MESH_BRUTE), ("exact_wave", hrt.FLAG_EXACT_ONLY | hrt.FLAG_WAVE_KERNEL)):
    try:
        frames[tag], _ = dev.render(cam, w, h, spp, seed=seed, flags=flags)
    except hrt.HrtError as e:
        print(tag, "->", e)
ref = frames["exact"]
for tag, f in frames.items():
    d = (f != ref).any(axis=2)
    print(f"{os.environ.get('HRT_LIBNAME', 'libhrt.so')} {tag}: {int(d.sum())} pixels differ from exact", np.argwhere(d)[:3].tolist(), flush=True)
# first hits under the differing pixels: the shipped filters (hrt_render_aov) against the oracle's closest hit
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import oracle_lib
lib = hrt.device_lib()
lib.hrt_render_aov.argtypes = [C.c_void_p, C.POINTER(hrt.Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
hit = np.empty((h, w, 3), np.float32)
assert lib.hrt_render_aov(dev._h, C.byref(cam), w, h, 0, hit.ctypes.data) == 0
want = oracle_lib.OracleScene(host.flatten()).aov(cam, w, h)["hit"]
dd = (hit != want).any(axis=2)
print("AOV first hits differing from the oracle:", int(dd.sum()), np.argwhere(dd)[:5].tolist())
d = (frames["shipped"] != ref).any(axis=2)
ys, xs = np.nonzero(d)
print("differing frame pixels: y", ys.min(), ys.max(), "x", xs.min(), xs.max())
kinds = {}
for y, x in zip(ys, xs):
    k = (int(want[y, x, 1]), int(want[y, x, 2]))
    kinds[k] = kinds.get(k, 0) + 1
print("first-hit (kind, id) under differing pixels:", sorted(kinds.items(), key=lambda kv: -kv[1])[:8])
for y, x in np.argwhere(dd)[:5]:
    print("  aov", y, x, "gpu", hit[y, x].tolist(), "oracle", want[y, x].tolist())
orc = oracle_lib.OracleScene(host.flatten()).render(cam, w, h, spp, seed=seed, threads=0)
for tag in ("shipped", "exact"):
    bad = (np.abs(frames[tag].astype(np.float64) - orc) > 1e-6 * np.maximum(1.0, np.abs(orc))).any(axis=2)
    print(tag, "vs oracle: pixels beyond 1e-6:", int(bad.sum()))
st = (C.c_uint32 * 8)()
for m in range(2):
    host._lib.hrt_host_scene_irregular_stats(host._h, m, st)
    print("mesh", m, "irregular stats [kept out, slivers, dropped, pairs, ref leaves, ref depth, dead, entries]:", list(st), host.kd_stats(m))
