"""Synthetic scenes shared by the GPU tests (host layer only: no GPU, no oracle)."""
import numpy as np


def many_squares(gpu, n_quads, n_meshes):
    """A random cloud of small tilted squares (some glass, some mirror, some moving) over a floor, tetrahedra as meshes,
    one point light: exercises the 32-bit / 64-bit candidate masks of the square filter and the bypass beyond 64."""
    M = gpu.Material.make
    rng = np.random.default_rng(100 + n_quads)
    s = gpu.HostScene()
    s.set_sky(False)
    s.add_light((0.0, 4.0, 3.0), 1.0)
    s.add_quad((-6, -2, -8), (1, 0, 0), (0, 0, 1), 12, 12, M(albedo=(0.8, 0.8, 0.8)))
    for i in range(n_quads - 1):
        c = rng.uniform((-3, -1.5, -6), (3, 2.0, -1))
        r = rng.normal(size=3); u = np.cross(r, rng.normal(size=3))
        kind = i % 5
        mat = M(albedo=tuple(rng.uniform(0.2, 1, 3)), type=gpu.MAT_GLASS if kind == 0 else (gpu.MAT_MIRROR if kind == 1 else gpu.MAT_DIFFUSE),
                transparency=0.5 if kind == 0 else 0.0, index_medium=1.4, motion=(0.0, 0.3, 0.0) if kind == 2 else (0, 0, 0))
        s.add_quad(tuple(c), tuple(r), tuple(u), float(rng.uniform(0.3, 0.9)), float(rng.uniform(0.3, 0.9)), mat)
    tet = np.array([[0, 0, 0], [0.6, 0, 0], [0.3, 0.6, 0.1], [0.3, 0.2, 0.6]], np.float32)
    tri = np.array([[0, 2, 1], [0, 1, 3], [1, 2, 3], [0, 3, 2]], np.uint32)
    for i in range(n_meshes):
        s.add_mesh(tet + rng.uniform((-3, -1.5, -5), (3, 1.5, -1.5)).astype(np.float32), tri, M(albedo=tuple(rng.uniform(0.2, 1, 3))))
    return s


def many_spheres(gpu, n, n_lights, dark=False):
    """A crowd of n spheres over a floor (mirror / glass / diffuse, some moving, some tiny, some overlapping, one enclosing the
    camera's side of the scene partly), n_lights point lights: exercises the packed pair filter of the closest-hit loop and of
    the shadow rays on both sides of its limits (8 <= n <= 128), odd counts (the last sphere pairs with itself) and far / near /
    tangent geometry."""
    M = gpu.Material.make
    rng = np.random.default_rng(7000 + 13 * n + n_lights)
    s = gpu.HostScene()
    s.set_sky(dark)
    for i in range(n_lights):
        s.add_light((float(rng.uniform(-4, 4)), float(rng.uniform(4, 9)), float(rng.uniform(-6, 3))), float(rng.uniform(0.5, 2.0)))
    s.add_quad((-40, -2, -60), (1, 0, 0), (0, 0, 1), 80, 70, M(albedo=(0.7, 0.7, 0.7)))
    for i in range(n):
        kind = i % 4
        r = float(rng.choice([0.05, 0.3, 0.8, 1.5, 4.0], p=[0.1, 0.3, 0.3, 0.25, 0.05]))
        c = (float(rng.uniform(-12, 12)), float(-2 + r * rng.uniform(0.6, 1.4)), float(rng.uniform(-40, 1)))
        mat = M(albedo=tuple(rng.uniform(0.2, 1, 3)), type=gpu.MAT_GLASS if kind == 0 else (gpu.MAT_MIRROR if kind == 1 else gpu.MAT_DIFFUSE),
                transparency=0.6 if kind == 0 else (0.3 if kind == 3 else 0.0), index_medium=1.5,
                motion=(0.0, float(rng.uniform(0, 0.8)), 0.0) if i % 3 == 0 else (0, 0, 0))
        s.add_sphere(c, r, mat)
    return s


def overlapping_soup(gpu, nt=200, seed=1, scale=0.5, slivers=12):
    """A triangle soup on which the REFERENCE's builder degenerates (found by tools/fuzz_exact.py, round 3): triangles larger
    than their spacing straddle every median cut, the tree reaches depth 100 and drops triangles below it (KDTree.cpp:101), and
    because the cut is the median of UNCLIPPED bounds (:87-98) planes fall outside their nodes -- children stick out of their
    parents, and a leaf is then reached only through its own box AND those ancestors' (hrt_tri_exception::group).  With
    nt = 200, seed = 1: 117 dropped triangles, 12 797 (triangle, leaf) pairs, 27 711 box entries."""
    rng = np.random.default_rng(1000 * nt + seed)
    centres = rng.normal(scale=1.0, size=(nt, 1, 3))
    v = (centres + rng.normal(scale=scale, size=(nt, 3, 3))).astype(np.float32)
    v[:slivers, 1] = v[:slivers, 0] + (v[:slivers, 2] - v[:slivers, 0]) * np.float32(0.5) + np.float32(1e-6)   # a few (near) collinear ones
    pos = v.reshape(-1, 3) + np.float32([0, 0.5, -4])
    tri = np.arange(3 * nt, dtype=np.uint32).reshape(nt, 3)
    s = gpu.HostScene()
    s.set_sky(False)
    s.add_quad((-6, -3, -10), (1, 0, 0), (0, 0, 1), 12, 12, gpu.Material.make(albedo=(0.8, 0.8, 0.8)))
    s.add_quad((-1, 4, -5), (1, 0, 0), (0, 0, 1), 2, 2, gpu.Material.make(albedo=(0, 0, 0), emissive=True, light_color=(1, 1, 1), light_intensity=8.0))
    s.add_mesh(pos, tri, gpu.Material.make(albedo=(0.7, 0.6, 0.5)), face_colors=rng.uniform(0.1, 1, (nt, 3)).astype(np.float32))
    return s


def describe_difference(a, b):
    """Text for an assertion message: how many pixels differ and by how much."""
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    bad = (a != b).any(axis=2)
    n = int(bad.sum())
    where = np.argwhere(bad)[:5].tolist()
    return f"{n} of {bad.size} pixels differ, max |diff| {d.max():.3g}, first at (y, x) {where}"
