"""Randomised search for a pixel that a filter, the pruning or the KD walk changes (GPU box):

    python tools/fuzz_exact.py [scenes] [first seed]

Every scene is random -- 0-130 spheres with radii over four decades (some enclosing the camera, some moving, mirror / glass /
diffuse / emissive), 1-70 squares (axis-aligned walls and tilted, glass and emissive ones), 0-3 meshes of random triangles (slivers
included), 0-2 point lights, dark or gradient sky -- and is rendered by the shipped build (the kernel form the library picks) and by the proof
build (HRT_FLAG_EXACT_ONLY: no filter, no pruning, IEEE divisions; every third scene also HRT_FLAG_MESH_BRUTE: no tree).  The two
frames must be identical; the lane-per-pixel kernel's frame too.  Prints one line per scene and a summary; exits non-zero on the
first difference.  FUZZ_ORACLE=1 additionally compares first hits and a small render with the CPU oracle (FUZZ_MAX_ENTRIES bounds the
reference-box entries of the scenes that are rendered: the oracle walks the reference-shaped tree)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hrt = importlib.import_module("hai719-raytracing_amd")
hrt.init(0)
if os.environ.get("FUZZ_ORACLE") == "1":
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    lib = hrt.device_lib()
    lib.hrt_render_aov.argtypes = [C.c_void_p, C.POINTER(hrt.Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
M = hrt.Material.make
w, h, spp = 640, 360, 4   # (every fourth scene: a random frame shape instead -- ragged tiles, one-pixel rows, hundreds of samples per pixel)


def material(rng, allow_emissive=True, tex=None):
    kind = rng.integers(0, 6)
    typ = hrt.MAT_GLASS if kind == 0 else (hrt.MAT_MIRROR if kind == 1 else hrt.MAT_DIFFUSE)
    kw = dict(albedo=tuple(rng.uniform(0.0, 1.0, 3)) if rng.random() > 0.1 else (0.0, 0.0, 0.0), type=typ,
              transparency=float(rng.choice([0.0, 0.3, 0.9])), index_medium=float(rng.uniform(1.05, 2.0)),
              motion=(0.0, float(rng.uniform(0, 1.0)), float(rng.uniform(-0.5, 0.5))) if rng.random() < 0.25 else (0, 0, 0))
    if allow_emissive and kind == 5:
        kw.update(emissive=1, light_color=tuple(rng.uniform(0.2, 1.0, 3)), light_intensity=float(rng.uniform(1, 12)))
    if tex is not None:   # (textures, normal maps, checkers: Material.cpp:63-130)
        textures, nmaps = tex
        r = rng.random()
        if r < 0.25 and textures:
            kw.update(texture_type=hrt.TEX_IMAGE, image=int(rng.choice(textures)), tex_scale=(float(rng.uniform(0.3, 6)), float(rng.uniform(0.3, 6))))
        elif r < 0.4:
            kw.update(texture_type=hrt.TEX_CHECKER, checker1=tuple(rng.uniform(0, 1, 3)), checker2=tuple(rng.uniform(0, 1, 3)),
                      tex_scale=(float(rng.uniform(0.5, 12)), float(rng.uniform(0.5, 12))))
        if rng.random() < 0.25 and nmaps:
            kw.update(normal_map=int(rng.choice(nmaps)))
            kw.setdefault("tex_scale", (float(rng.uniform(0.3, 6)), float(rng.uniform(0.3, 6))))
    return M(**kw)


def scene(seed):
    rng = np.random.default_rng(seed)
    s = hrt.HostScene()
    s.set_sky(bool(rng.integers(0, 2)))
    textures = [s.add_texture(rng.integers(0, 256, (int(rng.integers(1, 40)), int(rng.integers(1, 40)), 3), dtype=np.uint8)) for _ in range(int(rng.integers(0, 3)))]
    nmaps = [s.add_normal_map(rng.integers(0, 256, (int(rng.integers(1, 40)), int(rng.integers(1, 40)), 3), dtype=np.uint8)) for _ in range(int(rng.integers(0, 3)))]
    tex = (textures, nmaps)
    if rng.random() < 0.15:
        s.set_skybox(rng.integers(0, 256, (int(rng.integers(2, 30)), int(rng.integers(2, 60)), 3), dtype=np.uint8))
    for _ in range(int(rng.integers(0, 3))):
        s.add_light(tuple(rng.uniform((-6, 2, -8), (6, 10, 4))), float(rng.uniform(0.2, 3.0)), tuple(rng.uniform(0.3, 1.0, 3)))
    ns = int(rng.choice([0, 1, 2, 7, 8, 9, 31, 64, 65, 127, 128, 130]))
    for _ in range(ns):
        r = float(10.0 ** rng.uniform(-2.5, 1.3))
        c = rng.uniform((-15, -3, -45), (15, 8, 8))
        if rng.random() < 0.03:
            c, r = np.array([0.0, 0.0, 5.0]) + rng.normal(size=3), float(rng.uniform(3, 40))   # around the camera
        s.add_sphere(tuple(float(x) for x in c), r, material(rng, tex=tex))
    nq = int(rng.choice([1, 2, 6, 11, 28, 33, 64, 70]))
    if rng.random() < 0.7:   # a room of axis-aligned walls first
        e = float(rng.uniform(3, 9))
        for (p, r_, u_) in (((-e, -2, -2 * e), (1, 0, 0), (0, 0, 1)), ((-e, 2 * e - 2, -2 * e), (0, 0, 1), (1, 0, 0)), ((-e, -2, -2 * e), (0, 1, 0), (1, 0, 0)),
                            ((-e, -2, -2 * e), (0, 0, 1), (0, 1, 0)), ((e, -2, -2 * e), (0, 1, 0), (0, 0, 1))):
            s.add_quad(p, r_, u_, 2 * e, 2 * e, material(rng, tex=tex))
    for i in range(nq):
        c = rng.uniform((-6, -2, -12), (6, 5, 1))
        if rng.random() < 0.5:
            ax = int(rng.integers(0, 3)); r_ = np.eye(3)[(ax + 1) % 3]; u_ = np.eye(3)[(ax + 2) % 3]
            if rng.random() < 0.5: r_, u_ = u_, r_
        else:
            r_ = rng.normal(size=3); u_ = np.cross(r_, rng.normal(size=3))
        s.add_quad(tuple(float(x) for x in c), tuple(float(x) for x in r_), tuple(float(x) for x in u_), float(10.0 ** rng.uniform(-1.5, 0.9)),
                   float(10.0 ** rng.uniform(-1.5, 0.9)), material(rng, tex=tex))
    for _ in range(int(rng.integers(0, 4))):
        nt = int(rng.choice([1, 4, 60, 700]))
        base = rng.uniform((-4, -1.5, -9), (4, 3, -1))
        spread = float(rng.uniform(0.2, 1.5))
        centres = rng.normal(scale=spread, size=(nt, 1, 3))
        size = spread * float(rng.choice([0.5, 0.15, 0.04]))  # (a soup of LARGE overlapping triangles is refused by flatten: the reference's builder cannot finish it)
        pos = (base + centres + rng.normal(scale=size, size=(nt, 3, 3))).reshape(-1, 3).astype(np.float32)
        if rng.random() < 0.3:
            k = rng.random(nt) < 0.2   # some slivers
            v = pos.reshape(nt, 3, 3)
            v[k, 1] = v[k, 0] + (v[k, 2] - v[k, 0]) * np.float32(0.5) + np.float32(1e-6)
            pos = v.reshape(-1, 3)
        tri = np.arange(3 * nt, dtype=np.uint32).reshape(nt, 3)
        s.add_mesh(pos, tri, material(rng, allow_emissive=False), face_colors=rng.uniform(0, 1, (nt, 3)).astype(np.float32) if rng.random() < 0.5 else None)
    return s, ns, nq


def host_meshes(desc):
    import ctypes as C
    n = C.cast(desc, C.POINTER(C.c_uint32 * 16)).contents[12]   # hrt_scene_desc: {u32 n, pointer} pairs of materials, spheres, quads, meshes: n_meshes is dword 12
    return range(n)


bad = 0
ties = 0
skipped = 0
refused = 0
for k in range(n_scenes):
    seed = seed0 + k
    host, ns, nq = scene(seed)
    try:
        desc = host.flatten()
    except hrt.HrtError as e:
        refused += 1
        print(f"seed {seed}: refused by flatten ({str(e)[:90]}...)", flush=True)
        continue
    entries = sum(host.irregular_stats(m)["entries"] for m in range(len(host_meshes(desc))))
    if entries > int(os.environ.get("FUZZ_MAX_ENTRIES", "300000")):   # a reference tree of depth 100 with tens of thousands of leaves: correct, and minutes per frame
        skipped += 1
        print(f"seed {seed}: skipped ({entries} reference-box entries: the reference's own tree is degenerate)", flush=True)
        continue
    dev = hrt.DeviceScene(desc)
    w, h, spp = 640, 360, 4
    if k % 4 == 1:
        frng = np.random.default_rng(seed + 77)
        w, h = int(frng.integers(1, 500)), int(frng.integers(1, 300))
        spp = int(frng.choice([1, 2, 3, 7, 33, 130, 300])) if w * h < 20000 else int(frng.choice([1, 2, 5]))
    cam = hrt.default_camera(w / h)
    a, _ = dev.render(cam, w, h, spp, seed=seed)   # the form the library picks (streaming, unless the object tables exceed its LDS budget)
    exact = hrt.FLAG_EXACT_ONLY | (hrt.FLAG_MESH_BRUTE if k % 3 == 0 else 0)
    b, _ = dev.render(cam, w, h, spp, seed=seed, flags=exact)
    c, _ = dev.render(cam, w, h, spp, seed=seed, flags=hrt.FLAG_WAVE_KERNEL)
    same = np.array_equal(a, b, equal_nan=True) and np.array_equal(a, c, equal_nan=True)
    if not same and k % 3 == 0 and np.array_equal(a, c, equal_nan=True):
        # the all-triangles loop against the walk: two triangles at the SAME fp32 distance are a tie that the two orders of testing break
        # differently (SURVEY N11 "up to fp ties"; the reference's own tree breaks it a third way).  Counted, and a failure beyond 2 pixels.
        b2, _ = dev.render(cam, w, h, spp, seed=seed, flags=hrt.FLAG_EXACT_ONLY)
        n_tie = int((b != b2).any(axis=2).sum())
        if np.array_equal(a, b2, equal_nan=True) and n_tie <= 2:
            ties += n_tie
            print(f"seed {seed}: walk vs all-triangles loop differ on {n_tie} pixel(s) {np.argwhere((b != b2).any(axis=2)).tolist()}: a tie between triangles; filters identical", flush=True)
            same = True
    if same and os.environ.get("FUZZ_ORACLE") == "1":
        # ... and against the CPU oracle (the reference's algorithm restated, reference-shaped KD-tree): first hits identical on every
        # pixel, every pixel of a small render within 1e-6 (tests/test_gpu_parity.py's rule)
        ow, oh, ospp = 72, 40, 2
        ocam = hrt.default_camera(ow / oh)
        osc = oracle_lib.OracleScene(desc)
        hit = np.empty((oh, ow, 3), np.float32)
        assert lib.hrt_render_aov(dev._h, C.byref(ocam), ow, oh, 0, hit.ctypes.data) == 0
        want = osc.aov(ocam, ow, oh)["hit"]
        g, _ = dev.render(ocam, ow, oh, ospp, seed=seed)
        o = osc.render(ocam, ow, oh, ospp, seed=seed, threads=0)
        far = (np.abs(g.astype(np.float64) - o) > 1e-6 * np.maximum(1.0, np.abs(o))).any(axis=2)
        if not np.array_equal(hit, want) or far.any():
            same = False
            print(f"   ORACLE: first hits differ on {int((hit != want).any(axis=2).sum())} pixels, render beyond 1e-6 on {int(far.sum())} pixels {np.argwhere(far)[:3].tolist()}", flush=True)
    print(f"seed {seed}: {ns} spheres, {nq}+ squares: {'identical' if same else 'DIFFERENT'}  (mean {float(np.nanmean(a)):.4f})", flush=True)
    if not same:
        d = (a != b).any(axis=2) | (a != c).any(axis=2)
        print("   pixels", int(d.sum()), "first", np.argwhere(d)[:5].tolist(), "shipped vs exact", int((a != b).any(axis=2).sum()), "vs lane-per-pixel", int((a != c).any(axis=2).sum()))
        bad += 1
        break
    dev.close()
print(f"{k + 1} random scenes ({refused} refused by the host layer, {skipped} skipped as degenerate), {ties} tie pixel(s) between walk and all-triangles loop, {bad} with a difference")
sys.exit(1 if bad else 0)
