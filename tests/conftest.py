import os
import sys

import pytest

# The light lists of a scene are built from its second frame on (a caller that renders one frame does not pay for them);
# the tests render one frame of most scenes and want that frame answered from the lists: from the first frame on.
os.environ.setdefault("RTGPU_LIGHT_LISTS_AFTER", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU case, run only with RT_SLOW=1")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


def pow_free_pixels(scene, ids):
    """Pixels whose colour involves no pow(): the only operation of the path whose result may differ between the GPU's
    math library and glibc (src/raytrace.c:227 calls it when ks > 0).  True where the frame has no surface with ks > 0
    at all, where the primary ray hit nothing, or hit a surface with ks = 0 that spawns no secondary ray."""
    import numpy as np
    surf = scene.desc.surfaces
    ks = np.array([surf[i].ks for i in range(scene.ns)])
    kr = np.array([surf[i].kr for i in range(scene.ns)])
    kt = np.array([surf[i].kt for i in range(scene.ns)])
    sid = scene.sids()
    if not (ks[np.unique(sid)] > 0).any():
        return np.ones(ids.shape, bool)
    hit = ids >= 0
    s = sid[np.clip(ids, 0, None)]
    return ~hit | ((ks[s] == 0) & (kr[s] == 0) & (kt[s] == 0))
