"""The HIP trace path against the CPU oracle on random scenes (tests/scenegen.py; the oracle itself is checked
against the unmodified reference on the same scenes in test_random_scenes_cpu.py): crowded cells, slivers, glass
and mirrors facing each other, observers inside the grid, fixed and scaled grids.  Bars: primary-hit ids
bit-exact, raw colour equal up to pow() (relative 1e-4), the same frame with the triangle filter off."""
import os

import numpy as np
import pytest

import raytracer_amd as rt
from oracle import oracle as O
from conftest import pow_free_pixels
from tests.scenegen import write_scene

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def renderer():
    os.chdir(ROOT)
    r = rt.Renderer(0)
    yield r
    r.close()


@pytest.mark.parametrize("seed", list(range(1, 25)))
def test_random_scene_matches_oracle(renderer, tmp_path, seed):
    prefix = write_scene(str(tmp_path), seed, 96, 72)
    scene = rt.Scene(prefix)
    renderer.upload(scene)
    renderer.frame_begin()
    g = renderer.render_host()
    osc = O.OracleScene(prefix)
    o = osc.render(shadow_cache=False, threads=1 if osc.npl else 4)
    assert np.array_equal(g["ids"], o["ids"])
    clean = o["taint"] == 0
    assert np.array_equal(np.isfinite(g["rgb"][clean]), np.isfinite(o["rgb"][clean]))
    fin = clean[..., None] & np.isfinite(o["rgb"])
    diff = np.abs(g["rgb"].astype(np.float64) - o["rgb"].astype(np.float64))
    scale = np.maximum(np.abs(o["rgb"]).astype(np.float64), 1e-3)
    rel = (diff / scale)[fin]
    assert rel.size == 0 or rel.max() < 1e-4
    nopow = pow_free_pixels(scene, g["ids"]) & clean  # no pow() in the colour: bit-exact
    assert (g["rgb"].view(np.uint32) == o["rgb"].view(np.uint32))[nopow].all()
    if clean.all() and np.isfinite(o["rgb"]).all():
        assert np.allclose(g["minmax"], o["minmax"], rtol=1e-5)
    # the filter and the batch culling change nothing
    renderer.set_option("filter", 0)
    try:
        renderer.frame_begin()
        g0 = renderer.render_host()
    finally:
        renderer.set_option("filter", 1)
    assert np.array_equal(g0["ids"], g["ids"])
    assert np.array_equal(g0["rgb"].view(np.uint32), g["rgb"].view(np.uint32))
