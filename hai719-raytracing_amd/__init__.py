"""ctypes bindings over the two C-ABI libraries of the MI355X ray-trace path.

* ``libhrt_host.so`` -- host scene layer (``include/hrt_host.h``): Scene / Mesh /
  Material set-up with the reference's interface, flattened KD-tree builder.
* ``libhrt.so``      -- HIP kernels for gfx950 behind ``include/hrt.h``; the drop-in
  for the reference's ``ray_trace_from_camera()`` (/root/reference/main.cpp:200-263).

There is NO CPU fallback: if ``libhrt.so`` is missing or no GPU is present every
render entry point raises.  The package directory name contains a hyphen, so
import it with ``importlib.import_module("hai719-raytracing_amd")``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
ASSET_ROOT = os.path.join(REPO_ROOT, "assets")
TILE = 8  # HRT_TILE

FLAG_GAMMA = 1
FLAG_NO_LDS_TREE = 2
FLAG_WAVE_KERNEL = 4
FLAG_STREAM_KERNEL = 8
FLAG_NO_SHADOW_CULL = 16
FLAG_DUAL_KERNEL = 32
FLAG_EXACT_ONLY = 64   # proof builds: no filters, no reciprocal approximations (include/hrt.h)
FLAG_MESH_BRUTE = 128  # with FLAG_EXACT_ONLY: every triangle of a gated mesh, no KD walk

MAT_DIFFUSE, MAT_GLASS, MAT_MIRROR = 0, 1, 2
TEX_NONE, TEX_CHECKER, TEX_IMAGE = 0, 1, 2


class HrtError(RuntimeError):
    pass


class SceneDescPtr(C.c_void_p):
    """``const hrt_scene_desc*`` that keeps the HostScene owning the memory alive."""


# ----------------------------------------------------------------- PODs (hrt.h)
class Material(C.Structure):
    _fields_ = [
        ("albedo", C.c_float * 3), ("transparency", C.c_float), ("index_medium", C.c_float),
        ("type", C.c_int32), ("texture_type", C.c_int32),
        ("checker1", C.c_float * 3), ("checker2", C.c_float * 3),
        ("tex_scale_x", C.c_float), ("tex_scale_y", C.c_float),
        ("emissive", C.c_int32), ("light_color", C.c_float * 3), ("light_intensity", C.c_float),
        ("image", C.c_int32), ("normal_map", C.c_int32), ("motion", C.c_float * 3),
    ]

    @staticmethod
    def make(albedo=(0.8, 0.8, 0.8), type=MAT_DIFFUSE, transparency=0.0, index_medium=1.0,
             emissive=False, light_color=(0, 0, 0), light_intensity=0.0, motion=(0, 0, 0),
             texture_type=TEX_NONE, image=-1, normal_map=-1, checker1=(0, 0, 0), checker2=(0, 0, 0),
             tex_scale=(1.0, 1.0)) -> "Material":
        m = Material()
        m.albedo[:] = albedo
        m.type = type
        m.transparency = transparency
        m.index_medium = index_medium
        m.emissive = int(bool(emissive))
        m.light_color[:] = light_color
        m.light_intensity = light_intensity
        m.motion[:] = motion
        m.texture_type = texture_type
        m.image = image
        m.normal_map = normal_map
        m.checker1[:] = checker1
        m.checker2[:] = checker2
        m.tex_scale_x, m.tex_scale_y = tex_scale
        return m


class Camera(C.Structure):
    _fields_ = [
        ("eye", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3), ("forward", C.c_float * 3),
        ("fovy_deg", C.c_float), ("aspect", C.c_float), ("znear", C.c_float), ("zfar", C.c_float),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("kernel_ms", C.c_double), ("total_ms", C.c_double), ("samples", C.c_uint64),
        ("vgprs", C.c_uint32), ("sgprs", C.c_uint32), ("lds_bytes", C.c_uint32), ("waves_launched", C.c_uint32),
    ]


def _load(name: str) -> C.CDLL:
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        raise HrtError(f"{path} is missing: run `python __graft_entry__.py` (build()) or `make -C {_HERE}` first")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


_host: Optional[C.CDLL] = None
_dev: Optional[C.CDLL] = None


def host_lib() -> C.CDLL:
    global _host
    if _host is None:
        lib = _load("libhrt_host.so")
        lib.hrt_host_last_error.restype = C.c_char_p
        lib.hrt_host_scene_new.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.hrt_host_scene_free.argtypes = [C.c_void_p]
        lib.hrt_host_scene_free.restype = None
        lib.hrt_host_scene_setup.argtypes = [C.c_void_p, C.c_char_p, C.c_float, C.c_uint64]
        lib.hrt_host_scene_clear.argtypes = [C.c_void_p]
        lib.hrt_host_scene_add_texture.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        lib.hrt_host_scene_add_normal_map.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        lib.hrt_host_scene_set_skybox.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        lib.hrt_host_scene_add_sphere.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_float, C.POINTER(Material)]
        lib.hrt_host_scene_add_quad.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                                C.POINTER(C.c_float), C.c_float, C.c_float, C.POINTER(Material)]
        lib.hrt_host_scene_add_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                                C.c_void_p, C.POINTER(Material)]
        lib.hrt_host_scene_add_mesh_off.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Material)]
        lib.hrt_host_scene_add_light.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float)]
        lib.hrt_host_scene_set_sky.argtypes = [C.c_void_p, C.c_int32]
        lib.hrt_host_scene_set_kd_params.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        lib.hrt_host_scene_set_kd_builder.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.hrt_host_scene_flatten.argtypes = [C.c_void_p, C.c_void_p]
        lib.hrt_host_scene_kd_stats.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.hrt_host_scene_irregular_stats.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.hrt_host_default_camera.argtypes = [C.c_float, C.POINTER(Camera)]
        lib.hrt_host_default_camera.restype = None
        _host = lib
    return _host


def device_lib() -> C.CDLL:
    """The HIP library.  Raises if it has not been built -- there is no fallback."""
    global _dev
    if _dev is None:
        lib = _load(os.environ.get("HRT_LIBNAME", "libhrt.so"))  # HRT_LIBNAME: A/B builds of the same ABI (tools/variants.sh)
        lib.hrt_last_error.restype = C.c_char_p
        lib.hrt_init.argtypes = [C.c_int]
        lib.hrt_shutdown.restype = None
        lib.hrt_scene_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        lib.hrt_scene_destroy.argtypes = [C.c_void_p]
        lib.hrt_scene_destroy.restype = None
        lib.hrt_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                   C.c_uint32, C.c_void_p, C.POINTER(Stats)]
        lib.hrt_tiles_total.argtypes = [C.c_uint32, C.c_uint32]
        lib.hrt_tiles_total.restype = C.c_uint32
        lib.hrt_tiles_owned.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.hrt_tiles_owned.restype = C.c_uint32
        lib.hrt_render_tiles.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                         C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.hrt_assemble_frame.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                           C.c_void_p]
        lib.hrt_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        lib.hrt_check_last_launch.argtypes = [C.c_void_p]
        lib.hrt_kernel_info.argtypes = [C.POINTER(Stats)]
        lib.hrt_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.hrt_render_accumulate.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.hrt_finalize_tiles.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.hrt_encode_ppm.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t,
                                       C.POINTER(C.c_size_t), C.c_void_p]
        lib.hrt_multi_create.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        lib.hrt_multi_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                         C.c_uint32, C.c_void_p, C.POINTER(Stats)]
        lib.hrt_multi_destroy.argtypes = [C.c_void_p]
        lib.hrt_multi_destroy.restype = None
        lib.hrt_multi_gather.argtypes = [C.c_void_p]
        lib.hrt_multi_gather.restype = C.c_char_p
        lib.hrt_render_multi.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                         C.c_uint32, C.c_uint32, C.POINTER(C.c_int), C.c_void_p, C.POINTER(Stats)]
        lib.hrt_debug_kat.argtypes = [C.c_uint32, C.POINTER(Camera), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        _dev = lib
    return _dev


def _fp(a):
    return np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(C.POINTER(C.c_float))


# ------------------------------------------------------------------ host scene
class HostScene:
    """Scene of the host layer (mirrors the reference's ``Scene`` set-up API)."""

    def __init__(self, asset_root: str = ASSET_ROOT):
        self._lib = host_lib()
        self._h = C.c_void_p()
        self._keep = []
        rc = self._lib.hrt_host_scene_new(asset_root.encode(), C.byref(self._h))
        self._check(rc)

    def _check(self, rc: int):
        if rc < 0:
            raise HrtError(f"hrt_host error {rc}: {self._lib.hrt_host_last_error().decode()}")
        return rc

    def close(self):
        if self._h:
            self._lib.hrt_host_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setup(self, name: str, aspect: float = 1.0, seed: int = 1) -> "HostScene":
        self._check(self._lib.hrt_host_scene_setup(self._h, name.encode(), aspect, seed))
        return self

    def clear(self):
        self._check(self._lib.hrt_host_scene_clear(self._h))
        return self

    def add_texture(self, rgb: np.ndarray) -> int:
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        return self._check(self._lib.hrt_host_scene_add_texture(self._h, rgb.shape[1], rgb.shape[0], rgb.ctypes.data))

    def add_normal_map(self, rgb: np.ndarray) -> int:
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        return self._check(self._lib.hrt_host_scene_add_normal_map(self._h, rgb.shape[1], rgb.shape[0], rgb.ctypes.data))

    def add_sphere(self, center, radius, material: Material):
        self._check(self._lib.hrt_host_scene_add_sphere(self._h, _fp(center), radius, C.byref(material)))

    def add_quad(self, bottom_left, right, up, width, height, material: Material):
        self._check(self._lib.hrt_host_scene_add_quad(self._h, _fp(bottom_left), _fp(right), _fp(up), width, height,
                                                      C.byref(material)))

    def add_mesh(self, positions: np.ndarray, indices: np.ndarray, material: Material, face_colors=None):
        p = np.ascontiguousarray(positions, dtype=np.float32)
        i = np.ascontiguousarray(indices, dtype=np.uint32)
        fc = None if face_colors is None else np.ascontiguousarray(face_colors, dtype=np.float32)
        self._check(self._lib.hrt_host_scene_add_mesh(self._h, p.ctypes.data, p.shape[0], i.ctypes.data, i.shape[0],
                                                      None if fc is None else fc.ctypes.data, C.byref(material)))

    def add_mesh_off(self, rel_path: str, material: Material):
        self._check(self._lib.hrt_host_scene_add_mesh_off(self._h, rel_path.encode(), C.byref(material)))

    def add_light(self, pos, radius, color=(1, 1, 1)):
        self._check(self._lib.hrt_host_scene_add_light(self._h, _fp(pos), radius, _fp(color)))

    def set_skybox(self, rgb: Optional[np.ndarray]):
        """Equirectangular RGB8 skybox (h, w, 3), or None to remove it (Scene::loadSkybox, from memory)."""
        if rgb is None:
            self._check(self._lib.hrt_host_scene_set_skybox(self._h, 0, 0, None))
            return
        a = np.ascontiguousarray(rgb, dtype=np.uint8)
        self._keep.append(a)
        self._check(self._lib.hrt_host_scene_set_skybox(self._h, a.shape[1], a.shape[0], a.ctypes.data))

    def set_sky(self, dark: bool):
        self._check(self._lib.hrt_host_scene_set_sky(self._h, int(dark)))

    def set_kd_params(self, leaf_max: int = 0, max_depth: int = 0):
        self._check(self._lib.hrt_host_scene_set_kd_params(self._h, leaf_max, max_depth))

    def set_kd_builder(self, fn=None, user=None):
        """hrt_host_scene_set_kd_builder: ``fn`` a hrt_kd_builder_fn (a ctypes function pointer, or an address), None = the
        host's own threaded builder.  ``set_kd_builder("gpu")`` selects libhrt.so's hrt_kd_build_gpu (needs ``init``)."""
        if fn == "gpu":
            fn = C.cast(device_lib().hrt_kd_build_gpu, C.c_void_p)
        elif fn is not None and not isinstance(fn, (int, C.c_void_p)):
            self._kd_builder_keepalive = fn  # the ctypes callback object must outlive the flatten
            fn = C.cast(fn, C.c_void_p)
        self._check(self._lib.hrt_host_scene_set_kd_builder(self._h, fn, user))
        return self

    def flatten(self) -> C.c_void_p:
        """Builds the KD-trees; returns ``const hrt_scene_desc*`` (valid until the next flatten / close)."""
        d = SceneDescPtr()
        self._check(self._lib.hrt_host_scene_flatten(self._h, C.byref(d)))
        d._owner = self
        return d

    def kd_stats(self, mesh: int = 0) -> dict:
        out = (C.c_uint32 * 6)()
        self._check(self._lib.hrt_host_scene_kd_stats(self._h, mesh, out))
        keys = ["inner", "leaves", "empty_leaves", "depth", "leaf_tri_refs", "units"]
        return dict(zip(keys, list(out)))


def _irregular_stats(self, mesh: int = 0) -> dict:
    """Triangles kept out of the SAH tree because the reference's own tree treats them specially (host/ref_tree.h)."""
    out = (C.c_uint32 * 8)()
    self._check(self._lib.hrt_host_scene_irregular_stats(self._h, mesh, out))
    return dict(zip(["out_of_tree", "slivers", "dropped", "pairs", "ref_leaves", "ref_depth", "dead", "entries"], list(out)))


HostScene.irregular_stats = _irregular_stats


def default_camera(aspect: float) -> Camera:
    cam = Camera()
    host_lib().hrt_host_default_camera(aspect, C.byref(cam))
    return cam


# ---------------------------------------------------------------- device scene
_inited = False


def init(device: int = 0):
    global _inited
    lib = device_lib()
    rc = lib.hrt_init(device)
    if rc < 0:
        raise HrtError(f"hrt_init({device}) failed ({rc}): {lib.hrt_last_error().decode()}")
    _inited = True


class DeviceScene:
    """``hrt_scene``: the flattened scene resident in HBM."""

    def __init__(self, desc: C.c_void_p):
        self._lib = device_lib()
        if not _inited:
            init(int(os.environ.get("LOCAL_RANK", "0")))
        self._h = C.c_void_p()
        self._check(self._lib.hrt_scene_create(desc, C.byref(self._h)))

    def _check(self, rc: int):
        if rc < 0:
            raise HrtError(f"hrt error {rc}: {self._lib.hrt_last_error().decode()}")
        return rc

    def close(self):
        if self._h:
            self._lib.hrt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, cam: Camera, w: int, h: int, spp: int, seed: int = 1, flags: int = 0):
        """hrt_render: whole frame -> host ndarray (h, w, 3) float32, plus Stats."""
        out = np.empty((h, w, 3), dtype=np.float32)
        st = Stats()
        self._check(self._lib.hrt_render(self._h, C.byref(cam), w, h, spp, seed, flags, out.ctypes.data, C.byref(st)))
        return out, st

    def render_tiles(self, cam: Camera, w: int, h: int, spp: int, seed: int, flags: int, rank: int, world: int,
                     d_tiles_ptr: int, stream_ptr: int = 0):
        """hrt_render_tiles: this rank's tiles into a device buffer (asynchronous)."""
        self._check(self._lib.hrt_render_tiles(self._h, C.byref(cam), w, h, spp, seed, flags, rank, world,
                                               C.c_void_p(d_tiles_ptr), C.c_void_p(stream_ptr)))

    def render_accumulate(self, cam: Camera, w: int, h: int, first_sample: int, n_samples: int, seed: int, flags: int,
                          rank: int, world: int, d_sum_tiles_ptr: int, stream_ptr: int = 0):
        """hrt_render_accumulate: add samples [first_sample, first_sample + n_samples) to the running sums (asynchronous)."""
        self._check(self._lib.hrt_render_accumulate(self._h, C.byref(cam), w, h, first_sample, n_samples, seed, flags, rank,
                                                    world, C.c_void_p(d_sum_tiles_ptr), C.c_void_p(stream_ptr)))

    def check_last_launch(self):
        """hrt_check_last_launch: waits for the last launch; raises if the trace kernel gave up (incomplete tiles)."""
        self._check(self._lib.hrt_check_last_launch(self._h))

    def last_kernel_ms(self) -> float:
        ms = C.c_double()
        self._check(self._lib.hrt_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value


KAT_CAMERA, KAT_TRIANGLE, KAT_AABB, KAT_SPHERE, KAT_QUAD, KAT_OPTICS, KAT_NORMALIZE = range(7)
_KAT_IN = {KAT_CAMERA: 2, KAT_TRIANGLE: 7, KAT_AABB: 7, KAT_SPHERE: 7, KAT_QUAD: 7, KAT_OPTICS: 8, KAT_NORMALIZE: 3}
_KAT_OUT = {KAT_CAMERA: 12, KAT_TRIANGLE: 8, KAT_AABB: 2, KAT_SPHERE: 9, KAT_QUAD: 8, KAT_OPTICS: 8, KAT_NORMALIZE: 3}


def debug_kat(which: int, inp, prim=None, cam: Optional[Camera] = None) -> np.ndarray:
    """hrt_debug_kat: the DEVICE functions of the trace path on caller vectors (include/hrt.h); returns (n, out width)."""
    lib = device_lib()
    a = np.ascontiguousarray(inp, dtype=np.float32).reshape(-1, _KAT_IN[which])
    out = np.empty((a.shape[0], _KAT_OUT[which]), dtype=np.float32)
    pr = None if prim is None else np.ascontiguousarray(prim, dtype=np.float32)
    rc = lib.hrt_debug_kat(which, None if cam is None else C.byref(cam), None if pr is None else pr.ctypes.data, a.ctypes.data,
                           a.shape[0], out.ctypes.data)
    if rc < 0:
        raise HrtError(f"hrt_debug_kat failed ({rc}): {lib.hrt_last_error().decode()}")
    return out


class MultiScene:
    """``hrt_multi``: one replica of the scene per slot of ``devices`` (ordinals; may repeat), image tiles across them."""

    def __init__(self, desc: C.c_void_p, devices):
        global _inited
        self._lib = device_lib()
        self._h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        rc = self._lib.hrt_multi_create(desc, len(devices), arr, C.byref(self._h))
        if rc < 0:
            raise HrtError(f"hrt error {rc}: {self._lib.hrt_last_error().decode()}")
        self.note = self._lib.hrt_last_error().decode()  # what creation fell back from, if anything
        _inited = True

    @property
    def gather(self) -> str:
        """"rccl" (one ncclGather over the handle's communicators) or "peer" (hipMemcpyPeerAsync per slot)."""
        return self._lib.hrt_multi_gather(self._h).decode()

    def close(self):
        if self._h:
            self._lib.hrt_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, cam: Camera, w: int, h: int, spp: int, seed: int = 1, flags: int = 0):
        out = np.empty((h, w, 3), dtype=np.float32)
        st = Stats()
        rc = self._lib.hrt_multi_render(self._h, C.byref(cam), w, h, spp, seed, flags, out.ctypes.data, C.byref(st))
        if rc < 0:
            raise HrtError(f"hrt error {rc}: {self._lib.hrt_last_error().decode()}")
        return out, st


def render_multi(desc, cam: Camera, w: int, h: int, spp: int, seed: int, flags: int, devices):
    """hrt_render_multi: create + render + destroy in one call."""
    lib = device_lib()
    out = np.empty((h, w, 3), dtype=np.float32)
    st = Stats()
    arr = (C.c_int * len(devices))(*devices)
    rc = lib.hrt_render_multi(desc, C.byref(cam), w, h, spp, seed, flags, len(devices), arr, out.ctypes.data, C.byref(st))
    if rc < 0:
        raise HrtError(f"hrt error {rc}: {lib.hrt_last_error().decode()}")
    return out, st


def tiles_total(w: int, h: int) -> int:
    return ((w + TILE - 1) // TILE) * ((h + TILE - 1) // TILE)


def tiles_owned(w: int, h: int, rank: int, world: int) -> int:
    t = tiles_total(w, h)
    return (t - rank + world - 1) // world if rank < t else 0


def assemble_frame(d_gathered_ptr: int, tiles_per_rank_padded: int, w: int, h: int, world: int, d_frame_ptr: int,
                   stream_ptr: int = 0):
    lib = device_lib()
    rc = lib.hrt_assemble_frame(C.c_void_p(d_gathered_ptr), tiles_per_rank_padded, w, h, world,
                                C.c_void_p(d_frame_ptr), C.c_void_p(stream_ptr))
    if rc < 0:
        raise HrtError(f"hrt_assemble_frame failed ({rc}): {lib.hrt_last_error().decode()}")


def assemble_frame_host(gathered: np.ndarray, w: int, h: int, world: int) -> np.ndarray:
    """Host (numpy) statement of the tile -> frame mapping of hrt_assemble_frame; used by the CPU multi-rank test.

    gathered: (world, tiles_per_rank_padded, TILE*TILE, 3)."""
    tx = (w + TILE - 1) // TILE
    frame = np.zeros((h, w, 3), dtype=np.float32)
    for t in range(tiles_total(w, h)):
        rank, slot = t % world, t // world
        x0, y0 = (t % tx) * TILE, (t // tx) * TILE
        tile = gathered[rank, slot].reshape(TILE, TILE, 3)
        hh, ww = min(TILE, h - y0), min(TILE, w - x0)
        frame[y0:y0 + hh, x0:x0 + ww] = tile[:hh, :ww]
    return frame


def finalize_tiles(d_sum_tiles_ptr: int, n_tiles: int, total_samples: int, flags: int, d_tiles_ptr: int, stream_ptr: int = 0):
    """hrt_finalize_tiles: running sums -> pixel means (+ gamma with FLAG_GAMMA); the two pointers may be equal."""
    lib = device_lib()
    rc = lib.hrt_finalize_tiles(C.c_void_p(d_sum_tiles_ptr), n_tiles, total_samples, flags, C.c_void_p(d_tiles_ptr),
                                C.c_void_p(stream_ptr))
    if rc < 0:
        raise HrtError(f"hrt_finalize_tiles failed ({rc}): {lib.hrt_last_error().decode()}")


def encode_ppm(d_frame_ptr: int, w: int, h: int, fmt: int, d_out_ptr: int, capacity: int, stream_ptr: int = 0) -> int:
    """hrt_encode_ppm: the reference's PPM file (fmt 3, byte for byte) or its binary form (fmt 6), encoded on the
    device into d_out; returns the file size in bytes."""
    lib = device_lib()
    n = C.c_size_t(0)
    rc = lib.hrt_encode_ppm(C.c_void_p(d_frame_ptr), w, h, fmt, C.c_void_p(d_out_ptr), capacity, C.byref(n), C.c_void_p(stream_ptr))
    if rc < 0:
        raise HrtError(f"hrt_encode_ppm failed ({rc}): {lib.hrt_last_error().decode()}")
    return int(n.value)


def ppm_text_reference(rgb: np.ndarray) -> bytes:
    """Host statement of main.cpp:258-262 (what `ofstream <<` writes for an (h, w, 3) float32 frame); the
    checker of hrt_encode_ppm in the tests."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w, _ = rgb.shape
    m = np.where(rgb < np.float32(1.0), rgb, np.float32(1.0))          # std::min<float>(1.f, c): NaN -> 1
    v = (np.float32(255.0) * m).astype(np.float32)
    with np.errstate(invalid="ignore"):
        iv = np.where(np.isfinite(v), np.trunc(np.clip(v, -2147483648.0, 255.0)), -2147483648.0).astype(np.int64)
    body = " ".join(str(int(x)) for x in iv.reshape(-1))
    return f"P3\n{w} {h}\n255\n".encode() + body.encode() + b" \n"


def write_ppm(path: str, rgb: np.ndarray):
    """P3 dump with the reference's quantisation (main.cpp:258-261)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    lib = device_lib()
    rc = lib.hrt_write_ppm(path.encode(), rgb.ctypes.data, rgb.shape[1], rgb.shape[0])
    if rc < 0:
        raise HrtError(f"hrt_write_ppm failed ({rc}): {lib.hrt_last_error().decode()}")
