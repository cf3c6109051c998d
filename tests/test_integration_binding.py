"""INTEGRATION.md path A is code a maintainer pastes into the reference.  The two blocks are extracted from the document
verbatim, compiled against a mock of the reference's member names (tests/integration/ref_mock.h) and run:
flatten on the CPU (here), the whole ray_trace_from_camera() on the GPU box."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import PKG, ROOT

HERE = os.path.join(ROOT, "tests", "integration")


def _build(tmp):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in ("scene_to_hrt", "ray_trace_from_camera"):
        m = re.search(rf"<!-- BEGIN {name} -->\s*```cpp\n(.*?)```\s*<!-- END {name} -->", text, re.S)
        assert m, f"INTEGRATION.md lost its {name} block"
        assert "..." not in m.group(1), "the binding must be written out in full"
        open(os.path.join(tmp, name + ".inc"), "w").write(m.group(1))
    exe = os.path.join(tmp, "driver")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", str(tmp), "-I", HERE, "-I", os.path.join(ROOT, "include"),
           os.path.join(HERE, "driver.cpp"), "-o", exe, "-L", PKG, "-lhrt_host", "-lhrt", f"-Wl,-rpath,{PKG}",
           "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_binding_code_of_integration_md_compiles_and_flattens(hrt, tmp_path):
    exe = _build(str(tmp_path))
    r = subprocess.run([exe, "flatten"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out = dict(line.split(" ", 1) for line in r.stdout.strip().splitlines())
    assert out["counts"] == "3 1 1 1 1 2 0"          # materials in object order, 1 texture + 1 normal map, gradient sky
    q = [float(x) for x in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", out["quad"].replace("v0", "").replace("v1", "").replace("v3", ""))]
    # vertices as ROTATED by the set-up code; tangent frame as setQuad left it (N5): (2,0,0) and (0,3,0), not the rotated edges
    assert q[0:3] == [-2, -1, 1] and q[3:6] == [-2, -1, -1] and q[6:9] == [-2, 2, 1]
    assert q[9:12] == [2, 0, 0] and q[12:15] == [0, 3, 0]
    assert out["quadmat"] == "tex 2 image 0 nmap 1 scale 2 0.5 albedo 0.9"   # normal maps are numbered after the textures
    assert out["sphere"] == "0.5 0.75 type 1 eta 1.4 motion 0.5"
    assert out["mesh"].startswith("4 4 color_type 0 vc 0.25 leaf 4 ")        # vertex colours arrived, every triangle in the tree
    assert out["light"] == "3 1.5 0.8"
    assert out["image0"] == "4 2 30 image1 2 2"


@pytest.mark.gpu
def test_binding_code_of_integration_md_renders(gpu, tmp_path):
    exe = _build(str(tmp_path))
    img_path = os.path.join(str(tmp_path), "img.bin")
    r = subprocess.run([exe, "render", img_path], capture_output=True, text=True)
    assert r.returncode == 0 and "Done in" in r.stdout, r.stdout + r.stderr
    img = np.fromfile(img_path, np.float32).reshape(54, 96, 3)
    assert np.isfinite(img).all() and img.max() > 0.2 and (img > 0).mean() > 0.5   # gradient sky + lit objects, gamma applied
