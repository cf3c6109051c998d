"""The randomised search of tools/fuzz_exact.py as part of the suite: the seeds that FOUND something in round 3, and a fresh
slice of random scenes.  (The long campaigns -- 14 380 scenes -- are run by hand on the GPU box)"""
import ctypes as C
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from scene_util import overlapping_soup

pytestmark = pytest.mark.gpu


def test_a_reference_leaf_reached_through_several_boxes(gpu, oracle):
    """tests/scene_util.py overlapping_soup: the reference's tree of this mesh goes to depth 100, drops 117 triangles and cuts
    nodes with planes outside them.  Round 3's fuzz (seed 1011 of the first generator) found the shipped build differing from
    the proof build on such a mesh (the cull box of a sliver ignored boxes with reversed faces) and both from the
    reference-shaped tree (a leaf whose box sticks out of an ancestor's is reached only through BOTH boxes).  Now: shipped ==
    proof build == lane-per-pixel kernel, first hits == the oracle's reference-shaped tree, render within 1e-6 of the oracle."""
    host = overlapping_soup(gpu)
    desc = host.flatten()
    st = host.irregular_stats(0)
    assert st["ref_depth"] == 100 and st["dropped"] > 50 and st["entries"] > st["pairs"] > 1000   # leaves that need more than one box
    dev = gpu.DeviceScene(desc)
    w, h = 240, 135
    cam = gpu.default_camera(w / h)
    a, _ = dev.render(cam, w, h, 3, seed=1011)
    assert np.array_equal(a, dev.render(cam, w, h, 3, seed=1011, flags=gpu.FLAG_EXACT_ONLY)[0])
    assert np.array_equal(a, dev.render(cam, w, h, 3, seed=1011, flags=gpu.FLAG_EXACT_ONLY | gpu.FLAG_WAVE_KERNEL)[0])
    assert np.array_equal(a, dev.render(cam, w, h, 3, seed=1011, flags=gpu.FLAG_WAVE_KERNEL)[0])
    lib = gpu.device_lib()
    lib.hrt_render_aov.argtypes = [C.c_void_p, C.POINTER(gpu.Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    hit = np.empty((h, w, 3), np.float32)
    assert lib.hrt_render_aov(dev._h, C.byref(cam), w, h, 0, hit.ctypes.data) == 0
    osc = oracle.OracleScene(desc)
    want = osc.aov(cam, w, h)["hit"]
    assert (want[..., 1] == 3).sum() > 2000                      # the mesh fills a good part of the view
    assert np.array_equal(hit, want), f"{int((hit != want).any(axis=2).sum())} first hits differ from the reference-shaped tree"
    ref = osc.render(cam, 96, 54, 2, seed=5, threads=0)
    img, _ = dev.render(gpu.default_camera(96 / 54), 96, 54, 2, seed=5)
    assert (np.abs(img.astype(np.float64) - ref) <= 1e-6 * np.maximum(1.0, np.abs(ref))).all()


def test_a_slice_of_random_scenes(gpu):
    """60 random scenes (spheres over four decades of radius, squares, triangle soups with slivers, lights, textures, normal maps,
    skybox images, random frame shapes): shipped build == proof build == lane-per-pixel kernel on every pixel."""
    env = dict(os.environ, FUZZ_MAX_ENTRIES="200000")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_exact.py"), "60", "7000"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("60 random scenes") and last.endswith("0 with a difference"), last
    assert r.stdout.count("identical") >= 40, last
