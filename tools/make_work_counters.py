"""Per-sample work counters of the trace path -> tests/golden/work_counters.json.

Counts come from the CPU oracle walking the SAME flattened structures as the kernel (spheres and
quads in object order, the rope KD-tree of include/hrt.h for meshes) on a bounded sample of each
config scene at seed 1.  bench.py multiplies them by the fixed record sizes to price the kernel's
algorithmic bytes; the reference-shaped tree's counts are stored next to them for comparison.
"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
hrt = importlib.import_module("hai719-raytracing_amd")
import oracle_lib as O

SAMPLE = (240, 135, 16)
out = {"_sample": {"w": SAMPLE[0], "h": SAMPLE[1], "spp": SAMPLE[2], "seed": 1,
                   "record_bytes": {"sphere_test": 32, "quad_test": 48, "node_visit": 32, "tri_test": 40,
                                    "material_fetch": 64, "texel": 3, "framebuffer_per_pixel": 12}}}
for name in ["cornell_box", "cornell_mesh", "random_spheres", "mesh_in_box", "backrooms_pool"]:
    w, h, spp = SAMPLE
    host = hrt.HostScene().setup(name, 16 / 9, 1)
    desc = host.flatten()
    cam = hrt.default_camera(16 / 9)
    entry = {}
    for label, mode in (("per_sample", O.MESH_ROPE_TREE), ("per_sample_reference_tree", O.MESH_REF_TREE)):
        _, c = O.OracleScene(desc, mode).render(cam, w, h, spp, seed=1, threads=0, counters=True)
        n = c.pop("samples")
        entry[label] = {k: round(v / n, 4) for k, v in c.items()}
    ps = entry["per_sample"]
    entry["algorithmic_bytes_per_sample"] = round(
        32 * ps["sphere_tests"] + 48 * ps["quad_tests"] + 32 * ps["node_visits"] + 40 * ps["tri_tests"]
        + 64 * ps["shaded_hits"] + 3 * ps["texel_lookups"], 1)
    if name in ("cornell_mesh", "mesh_in_box", "backrooms_pool"):
        entry["kd_tree"] = host.kd_stats(0)
    out[name] = entry
    print(name, entry)
with open(os.path.join(ROOT, "tests", "golden", "work_counters.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
