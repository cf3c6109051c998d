import importlib, os, sys, time
sys.path.insert(0, '.')
hrt = importlib.import_module("hai719-raytracing_amd"); hrt.init(0)
os.environ["HRT_KD_VERBOSE"] = "1"
for name in ["backrooms_pool", "mesh_in_box"]:
    s = hrt.HostScene().setup(name, 16 / 9, 1)
    for b in (None, "gpu", "gpu"):
        s.set_kd_builder(b)
        t0 = time.perf_counter(); s.flatten(); print(name, b, "flatten", round((time.perf_counter() - t0) * 1e3, 1), "ms", flush=True)
