"""Known answers on the DEVICE (hrt_debug_kat): the HIP functions of the trace path against vectors produced by the
reference's own code (tests/golden/ref_kat.npz: Triangle.h, AABB.h, Functions.cpp, Ray/Line/Vec3, matrixUtilities.h
compiled where they lie), bit for bit -- one hop from the reference to the GPU instead of two through rendered pixels.
Sphere / square vectors are oracle output (tests/golden/oracle_kat.npz): parity unpinned, stated there."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(GOLDEN, "ref_kat.npz"))


def test_camera_rays_on_the_device_match_the_reference(gpu, oracle, kat):
    """Row a2: ray origin and direction for 16 poses x 1024 (u, v), bit-identical to the reference's gluInvertMatrix +
    screen_space_to_world_space_ray + Ray constructor; the shipped multiply-by-reciprocal fast path of the final
    division and the exact-division build agree everywhere."""
    for row, want in zip(kat["camera_rows"], kat["camera_rays"]):
        got = gpu.debug_kat(gpu.KAT_CAMERA, kat["camera_uv"], cam=oracle.camera_from_row(gpu, row))
        assert np.array_equal(got[:, :6], want)
        assert np.array_equal(got[:, 6:], want)
    rng = np.random.default_rng(3)
    uv = rng.uniform(0, 1, (1 << 20, 2)).astype(np.float32)  # 1 M more rays through the default 16:9 camera vs the oracle
    cam = gpu.default_camera(16 / 9)
    got = gpu.debug_kat(gpu.KAT_CAMERA, uv, cam=cam)
    assert np.array_equal(got[:, :6], got[:, 6:]) and np.array_equal(got[:, :6], oracle.camera_rays(cam, uv))


def test_triangle_on_the_device_matches_the_reference(gpu, kat):
    for tri, want in zip(kat["tri_prims"], kat["tri_out"]):
        assert np.array_equal(gpu.debug_kat(gpu.KAT_TRIANGLE, kat["tri_rays"], prim=tri), want)


def test_aabb_on_the_device_matches_the_reference(gpu, kat):
    for box, want in zip(kat["aabb_prims"], kat["aabb_out"]):
        got = gpu.debug_kat(gpu.KAT_AABB, kat["aabb_rays"], prim=box)
        assert np.array_equal(got[:, 0], want[:, 0])          # AABB::intersects, fp64 form
        assert np.array_equal(got[:, 1], want[:, 0])          # the shipped gate: fp32 filter in front of it


def test_optics_on_the_device_match_the_reference(gpu, kat):
    """reflect / refract are bit-identical.  reflectance: the device evaluates pow(1 - c, 5) as m^2 * m^2 * m in fp64 and
    narrows; the reference calls libm pow.  gamma: device pow vs glibc pow.  Both are compared after narrowing to fp32."""
    got = gpu.debug_kat(gpu.KAT_OPTICS, kat["optics_in"])
    want = kat["optics_out"]
    assert np.array_equal(got[:, :6], want[:, :6], equal_nan=True)
    assert np.array_equal(got[:, 6], want[:, 6]), f"reflectance differs on {(got[:, 6] != want[:, 6]).sum()} of {len(want)}"
    bad = got[:, 7] != want[:, 7]
    assert bad.sum() == 0, f"gamma differs on {bad.sum()} of {len(want)} (max {np.abs(got[:, 7] - want[:, 7]).max():.3g})"


def test_normalize_on_the_device_matches_the_reference(gpu, kat):
    assert np.array_equal(gpu.debug_kat(gpu.KAT_NORMALIZE, kat["normalize_in"]), kat["normalize_out"])


def test_sphere_and_square_on_the_device_match_the_oracle_vectors(gpu):
    """Parity unpinned (oracle vectors, see tests/golden/make_golden.py).  theta / phi go through fp64 acos / atan2 of two
    different libms: equal after narrowing on these vectors.  The square filter must let every hit through."""
    k = np.load(os.path.join(GOLDEN, "oracle_kat.npz"))
    for prim, want in zip(k["sphere_prims"], k["sphere_out"]):
        assert np.array_equal(gpu.debug_kat(gpu.KAT_SPHERE, k["sphere_rays"], prim=prim), want)
    for prim, want in zip(k["quad_prims"], k["quad_out"]):
        got = gpu.debug_kat(gpu.KAT_QUAD, k["quad_rays"], prim=prim)
        assert np.array_equal(got[:, :7], want)
        assert (got[:, 7] >= got[:, 0]).all()


def test_kat_arguments_are_validated(gpu):
    with pytest.raises(gpu.HrtError):
        gpu.debug_kat(gpu.KAT_TRIANGLE, np.zeros((1, 7), np.float32))          # no primitive
    with pytest.raises(gpu.HrtError):
        gpu.debug_kat(gpu.KAT_CAMERA, np.zeros((1, 2), np.float32))            # no camera
