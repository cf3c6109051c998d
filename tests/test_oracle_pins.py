"""The oracle against the reference's own code: committed golden vectors (generated from
oracle/_ref, i.e. the reference's Triangle.h / AABB.h / Functions.cpp / Vec3.h / Ray.h /
imageLoader.cpp compiled from /root/reference) and, where oracle/_ref is present, live.
Everything here is bit-exact: the restatement must BE the reference arithmetic."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(GOLDEN, "ref_kat.npz"))


def test_triangle_matches_reference_golden(oracle, kat):
    for tri, want in zip(kat["tri_prims"], kat["tri_out"]):
        got = oracle.kat("triangle", tri, kat["tri_rays"])
        assert np.array_equal(got, want)
    assert kat["tri_out"][0][:, 0].sum() > 100  # the vectors do exercise hits
    assert kat["tri_out"][1][:, 0].sum() == 0   # back-facing triangle: culled (Triangle.h:87-91)


def test_aabb_matches_reference_golden(oracle, kat):
    for box, want in zip(kat["aabb_prims"], kat["aabb_out"]):
        got = oracle.kat("aabb", box, kat["aabb_rays"])[:, 0]
        assert np.array_equal(got, want[:, 0])
    assert 0 < kat["aabb_out"][0].sum() < kat["aabb_rays"].shape[0]


def test_optics_match_reference_golden(oracle, kat):
    got = oracle.kat_optics(kat["optics_in"])
    assert np.array_equal(got, kat["optics_out"], equal_nan=True)


def test_random_stream_matches_reference_golden(oracle, kat):
    got = oracle.kat_random(kat["random_out"].shape[0], seed=int(kat["random_seed"][0]))
    assert np.array_equal(got, kat["random_out"])
    u = got[:, 0]
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.03
    assert np.allclose(np.linalg.norm(got[:, 1:], axis=1), 1.0, atol=1e-6)


def test_ray_normalisation_matches_reference_golden(oracle, kat):
    # Ray's constructor normalises (Line.h:13-16): push the vectors through a triangle KAT ray and
    # read nothing but the direction handling -> use the camera-free normalise path of oracle_kat_optics?
    # The oracle exposes normalisation through reflect(d, 0): reflect(d,n=0) = d, so check via numpy
    # that the golden outputs are the correctly rounded d / |d| the oracle's `normalized` produces.
    v = kat["normalize_in"].astype(np.float32)
    L = np.sqrt((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2]).astype(np.float32), dtype=np.float32)
    want = (v / L[:, None]).astype(np.float32)
    # numpy's float32 sum order matches Vec3::squareLength (x*x + y*y + z*z), division is IEEE
    assert np.array_equal(want, kat["normalize_out"])


def test_ppm_loader_matches_reference_golden(hrt):
    """Host-layer PPM loader == reference imageLoader.cpp (w, h, every byte) on every shipped image."""
    want = json.load(open(os.path.join(GOLDEN, "ref_ppm.json")))
    assert len(want) >= 5
    lib = hrt.host_lib()
    lib.hrt_host_ppm_info.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
    for rel, info in want.items():
        w, h, s = C.c_int32(), C.c_int32(), C.c_uint64()
        assert lib.hrt_host_ppm_info(os.path.join(hrt.ASSET_ROOT, rel).encode(), C.byref(w), C.byref(h), C.byref(s)) == 0
        assert (w.value, h.value, f"{s.value:016x}") == (info["w"], info["h"], info["fnv1a"]), rel
    assert lib.hrt_host_ppm_info(b"/nonexistent.ppm", C.byref(w), C.byref(h), C.byref(s)) < 0


def test_live_reference_parts_when_present(oracle):
    """In the build container oracle/_ref exists: fresh random inputs, still bit-exact."""
    if oracle.ref_parts() is None:
        pytest.skip("oracle/_ref not built here (the reference does not travel to the GPU box)")
    rng = np.random.default_rng(99)
    n = 3000
    rays = np.concatenate([rng.uniform(-1, 2, (n, 2)), np.full((n, 1), 2.0), rng.normal(0, 0.4, (n, 2)),
                           np.full((n, 1), -1.0), rng.uniform(0, 1, (n, 1))], 1).astype(np.float32)
    for _ in range(5):
        tri = rng.uniform(-1, 1, 9).astype(np.float32)
        assert np.array_equal(oracle.kat("triangle", tri, rays), oracle.kat("triangle", tri, rays, use_ref=True))
        box = np.sort(rng.uniform(-1, 1, (2, 3)), axis=0).astype(np.float32).ravel()
        assert np.array_equal(oracle.kat("aabb", box, rays), oracle.kat("aabb", box, rays, use_ref=True))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    nn = rng.normal(size=(n, 3)); nn /= np.linalg.norm(nn, axis=1, keepdims=True)
    inp = np.concatenate([d, nn, rng.uniform(0.4, 2.5, (n, 1)), rng.uniform(0, 1, (n, 1))], 1).astype(np.float32)
    assert np.array_equal(oracle.kat_optics(inp), oracle.kat_optics(inp, use_ref=True))


def test_oracle_renders_match_committed_images(hrt, oracle):
    """The oracle on this machine reproduces the committed oracle pixels (same compiler flags, no FMA)."""
    g = np.load(os.path.join(GOLDEN, "oracle_images.npz"))
    for name in ["cornell_box", "cornell_mesh", "random_spheres", "mesh_in_box", "backrooms_pool"]:
        w, h, spp, seed = (int(x) for x in g[name + "_shape"])
        host = hrt.HostScene().setup(name, w / h, 1)
        desc = host.flatten()
        cam = hrt.default_camera(w / h)
        sc = oracle.OracleScene(desc)
        img = sc.render(cam, w, h, spp, seed=seed, threads=0)
        assert np.allclose(img, g[name + "_render"], rtol=1e-5, atol=1e-6), name
        aov = sc.aov(cam, w, h)
        assert np.array_equal(aov["hit"][..., 1:], g[name + "_aov_hit"][..., 1:]), name


def test_oracle_thread_modes_agree(hrt, oracle):
    """Row pool, one-thread-per-scanline (main.cpp:232-238) and single thread give identical pixels."""
    host = hrt.HostScene().setup("cornell_box", 1.0, 1)
    desc = host.flatten()
    cam = hrt.default_camera(1.0)
    sc = oracle.OracleScene(desc)
    a = sc.render(cam, 32, 32, 2, seed=3, threads=1)
    b = sc.render(cam, 32, 32, 2, seed=3, threads=4)
    c = sc.render(cam, 32, 32, 2, seed=3, threads=-1)
    assert np.array_equal(a, b) and np.array_equal(a, c)
    d = sc.render(cam, 32, 32, 2, seed=4, threads=4)
    assert not np.array_equal(a, d)


def test_path_stream_is_uniform_and_keyed(oracle):
    s = oracle.path_stream(1, 1000, 3, 4096)
    assert 0.0 <= s.min() and s.max() < 1.0
    assert abs(s.mean() - 0.5) < 0.02 and abs(s.var() - 1 / 12) < 0.01
    assert not np.array_equal(s, oracle.path_stream(1, 1001, 3, 4096))
    assert not np.array_equal(s, oracle.path_stream(1, 1000, 4, 4096))
    assert not np.array_equal(s, oracle.path_stream(2, 1000, 3, 4096))
    assert np.array_equal(s, oracle.path_stream(1, 1000, 3, 4096))
