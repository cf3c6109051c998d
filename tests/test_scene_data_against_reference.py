"""Build-container only: the host layer's scene DATA against the reference's set-up code read where it lies
(tests/golden/check_scene_data.py).  Skipped where /root/reference does not exist (the GPU box)."""
import importlib.util
import os

import pytest

from conftest import ROOT

REF_SCENE_H = "/root/reference/src/Scene.h"


def _checker():
    spec = importlib.util.spec_from_file_location("check_scene_data", os.path.join(ROOT, "tests", "golden", "check_scene_data.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(not os.path.exists(REF_SCENE_H), reason="the reference does not travel to the GPU box")
@pytest.mark.parametrize("name,ref_fn,setup,aspect", [("cornell_box", "setup_cornell_box", "cornell_box", 16 / 9), ("cornell_box", "setup_cornell_box", "cornell_box", 1.0),
                                                      ("backrooms_pool", "setup_backrooms_pool", "backrooms_pool", 16 / 9),
                                                      ("random_spheres", "setup_random_spheres", "random_spheres", 16 / 9)])
def test_host_scene_data_is_the_reference_scene_data(hrt, name, ref_fn, setup, aspect):
    """Every square (vertices after the set-up code's transforms, tangent frame as setQuad leaves it), sphere, mesh (transformed
    vertices), light and every Material field the path reads, as the reference's setup_* statements produce them, equals what
    the host layer flattens -- so a transcription slip cannot hide behind the fact that oracle and GPU share the host layer."""
    compared, fails = _checker().check(name, ref_fn, setup, aspect, hrt)
    assert not fails, f"{name}: {fails[:5]}"
    assert compared.split(",")[1].strip() != "0 squares"


@pytest.mark.skipif(not os.path.exists(REF_SCENE_H), reason="the reference does not travel to the GPU box")
def test_the_scene_data_check_is_not_vacuous(hrt):
    """The same comparison with the host scene built for ANOTHER aspect ratio must fail (four walls scale with it)."""
    mod = _checker()
    real_setup = hrt.HostScene.setup

    def wrong(self, name, aspect=1.0, seed=1):
        return real_setup(self, name, 1.0, seed)
    hrt.HostScene.setup = wrong
    try:
        _, fails = mod.check("cornell_box", "setup_cornell_box", "cornell_box", 16 / 9, hrt)
    finally:
        hrt.HostScene.setup = real_setup
    assert len(fails) >= 4
