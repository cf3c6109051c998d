"""ctypes access to the CPU oracle (oracle/liboracle.so) and, when it was built in
the build container, to the glut-free reference parts (oracle/_ref/libref_parts.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")

MESH_REF_TREE, MESH_BRUTE, MESH_ROPE_TREE = 0, 1, 2
COUNTER_NAMES = ["closest_queries", "shadow_queries", "sphere_tests", "quad_tests", "node_visits", "tri_tests",
                 "shaded_hits", "texel_lookups", "rng_draws", "samples"]

_lib = None
_ref = None


def build():
    subprocess.run(["make", "-s", "-C", _ORACLE_DIR], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.path.join(_ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oracle_scene_create.restype = C.c_void_p
        L.oracle_scene_create.argtypes = [C.c_void_p, C.c_int]
        L.oracle_scene_destroy.argtypes = [C.c_void_p]
        L.oracle_scene_destroy.restype = None
        L.oracle_ref_tree_stats.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.oracle_render.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                    C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_aov.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_void_p] * 4
        L.oracle_mesh_query.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
        for name in ("oracle_kat_triangle", "oracle_kat_aabb", "oracle_kat_sphere", "oracle_kat_quad"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
            getattr(L, name).restype = None
        L.oracle_kat_optics.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.oracle_kat_optics.restype = None
        L.oracle_kat_random.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_kat_random.restype = None
        L.oracle_path_stream.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_path_stream.restype = None
        L.oracle_camera_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.oracle_camera_rays.restype = None
        L.oracle_camera_matrices.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_camera_matrices.restype = None
        L.oracle_kat_normalize.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.oracle_kat_normalize.restype = None
        L.oracle_trace_path.argtypes = [C.c_void_p, C.c_void_p] + [C.c_uint32] * 5 + [C.c_uint64, C.c_void_p, C.c_uint32]
        L.oracle_trace_path.restype = C.c_uint32
        _lib = L
    return _lib


def ref_parts():
    """The reference's own glut-free code, or None when oracle/_ref was not built (e.g. on the GPU box)."""
    global _ref
    if _ref is None:
        path = os.path.join(_ORACLE_DIR, "_ref", "libref_parts.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_fixed_seed.restype = C.c_uint32
        for name in ("ref_kat_triangle", "ref_kat_aabb"):
            getattr(R, name).argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
            getattr(R, name).restype = None
        R.ref_kat_optics.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        R.ref_kat_optics.restype = None
        R.ref_kat_random.argtypes = [C.c_uint32, C.c_void_p]
        R.ref_kat_random.restype = None
        R.ref_kat_normalize.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        R.ref_kat_normalize.restype = None
        R.ref_camera_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        R.ref_camera_rays.restype = C.c_int
        R.ref_camera_inverses.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        R.ref_camera_inverses.restype = C.c_int
        R.ref_ppm_info.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        R.ref_ppm_info.restype = C.c_uint64
        _ref = R
    return _ref


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class OracleScene:
    def __init__(self, desc, mesh_mode: int = MESH_REF_TREE):
        self._L = lib()
        self._desc = desc  # keep the description alive
        self._h = self._L.oracle_scene_create(desc, mesh_mode)

    def close(self):
        if self._h:
            self._L.oracle_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, cam, w, h, spp, seed=1, flags=0, threads=0, counters=False):
        out = np.empty((h, w, 3), dtype=np.float32)
        cnt = np.zeros(10, dtype=np.uint64) if counters else None
        rc = self._L.oracle_render(self._h, C.byref(cam), w, h, spp, seed, flags, threads, out.ctypes.data,
                                   None if cnt is None else cnt.ctypes.data)
        assert rc == 0
        if counters:
            return out, dict(zip(COUNTER_NAMES, (int(c) for c in cnt)))
        return out

    def aov(self, cam, w, h):
        bufs = [np.empty((h, w, 3), dtype=np.float32) for _ in range(4)]
        rc = self._L.oracle_aov(self._h, C.byref(cam), w, h, *[b.ctypes.data for b in bufs])
        assert rc == 0
        return dict(zip(["hit", "normal", "albedo", "emission"], bufs))

    def ref_tree_stats(self, mesh=0):
        out = (C.c_uint32 * 4)()
        self._L.oracle_ref_tree_stats(self._h, mesh, out)
        return dict(zip(["nodes", "leaves", "tri_refs", "max_leaf"], list(out)))


def mesh_query(desc, mesh, mode, rays):
    rays = _f32(rays)
    out = np.empty((rays.shape[0], 3), dtype=np.float32)
    rc = lib().oracle_mesh_query(desc, mesh, mode, rays.ctypes.data, rays.shape[0], out.ctypes.data)
    assert rc == 0
    return out


def kat(which: str, prim, rays, use_ref=False):
    widths = {"triangle": 8, "aabb": 1, "sphere": 9, "quad": 7}
    L = ref_parts() if use_ref else lib()
    fn = getattr(L, ("ref_kat_" if use_ref else "oracle_kat_") + which)
    prim, rays = _f32(prim), _f32(rays)
    out = np.empty((rays.shape[0], widths[which]), dtype=np.float32)
    fn(prim.ctypes.data, rays.ctypes.data, rays.shape[0], out.ctypes.data)
    return out


def kat_optics(inp, use_ref=False):
    inp = _f32(inp)
    out = np.empty((inp.shape[0], 8), dtype=np.float32)
    fn = ref_parts().ref_kat_optics if use_ref else lib().oracle_kat_optics
    fn(inp.ctypes.data, inp.shape[0], out.ctypes.data)
    return out


def kat_random(n, seed=None, use_ref=False):
    out = np.empty((n, 4), dtype=np.float32)
    if use_ref:
        ref_parts().ref_kat_random(n, out.ctypes.data)
    else:
        lib().oracle_kat_random(seed, n, out.ctypes.data)
    return out


def path_stream(seed, pixel, sample, n):
    out = np.empty(n, dtype=np.float32)
    lib().oracle_path_stream(seed, pixel, sample, n, out.ctypes.data)
    return out


def camera_rays(cam, uv):
    uv = _f32(uv)
    out = np.empty((uv.shape[0], 6), dtype=np.float32)
    lib().oracle_camera_rays(C.byref(cam), uv.ctypes.data, uv.shape[0], out.ctypes.data)
    return out


def camera_matrices(cam):
    """(modelview, projection, modelview^-1, projection^-1), column-major doubles, as the oracle builds them."""
    m = np.zeros(64, np.float64)
    lib().oracle_camera_matrices(C.byref(cam), m.ctypes.data)
    return m[:16].copy(), m[16:32].copy(), m[32:48].copy(), m[48:].copy()


def ref_camera_rays(cam, uv):
    """The reference's own gluInvertMatrix + screen_space_to_world_space_ray + Ray constructor (oracle/_ref) on the GL
    matrices of `cam`: (n, 6) origin + direction."""
    mv, pr, _, _ = camera_matrices(cam)
    uv = _f32(uv)
    out = np.empty((uv.shape[0], 6), dtype=np.float32)
    assert ref_parts().ref_camera_rays(mv.ctypes.data, pr.ctypes.data, uv.ctypes.data, uv.shape[0], out.ctypes.data) == 1
    return out


def ref_camera_inverses(cam):
    mv, pr, _, _ = camera_matrices(cam)
    out = np.zeros(32, np.float64)
    assert ref_parts().ref_camera_inverses(mv.ctypes.data, pr.ctypes.data, out.ctypes.data) == 1
    return out[:16], out[16:]


def kat_normalize(v):
    v = _f32(v)
    out = np.empty_like(v)
    lib().oracle_kat_normalize(v.ctypes.data, v.shape[0], out.ctypes.data)
    return out


CAMERA_FIELDS = 16  # eye, right, up, forward (3 each), fovy_deg, aspect, znear, zfar


def camera_to_row(cam):
    return np.array(list(cam.eye) + list(cam.right) + list(cam.up) + list(cam.forward) + [cam.fovy_deg, cam.aspect, cam.znear, cam.zfar],
                    np.float32)


def camera_from_row(hrt, row):
    cam = hrt.Camera()
    cam.eye[:] = row[0:3]; cam.right[:] = row[3:6]; cam.up[:] = row[6:9]; cam.forward[:] = row[9:12]
    cam.fovy_deg, cam.aspect, cam.znear, cam.zfar = (float(x) for x in row[12:16])
    return cam


def trace_path(scene, cam, w, h, x, y, sample, seed, cap=8):
    """Debug: the closest-hit queries of one path, rows of {o, d, time, kind, index, t, triangle, 0}."""
    out = np.zeros((cap, 12), np.float32)
    n = lib().oracle_trace_path(scene._h, C.byref(cam), w, h, x, y, sample, seed, out.ctypes.data, cap)
    return out[:n]
