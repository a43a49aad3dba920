"""Pins the CPU oracle (oracle/rt_oracle.c) against the reference's own outputs.

The golden .npz files were produced by the unmodified reference compiled in the build container
(tests/golden/make_golden.py); known_answers.json holds digests at each scene's native resolution.
The oracle must reproduce ids AND raw float colours bit-for-bit (the reference is bit-stable across
compilers), the min/max pair, and the bytes of the 24-bpp BMP.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CASES = ["test", "s2", "s2_glass", "s3", "s5", "pokoj", "pokoj_cegly", "pokoj_kafelki", "temple", "ulica"]


def load_case(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    over = {k: os.path.join(ROOT, v) for k, v in json.loads(str(z["overrides"])).items()}
    return z, os.path.join(ROOT, str(z["prefix"])), over


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_dump(name):
    z, prefix, over = load_case(name)
    h, w = z["ids"].shape
    tex = bool(z["texture"])
    s = O.OracleScene(prefix, texture=tex, width=w, height=h, **over)
    o = s.render(shadow_cache=True)
    assert np.array_equal(o["ids"], z["ids"])
    clean = o["taint"] == 0  # pixels where the reference reads uninitialised texture samples are not comparable
    same = (o["rgb"].view(np.uint32) == z["rgb"].view(np.uint32)).all(-1)
    assert same[clean].all()
    if not tex:
        assert clean.all()
        assert np.array_equal(o["minmax"], z["minmax"])
        assert abs(s.gamma - float(z["gamma"])) == 0.0
        bmp = O.bmp24(O.tonemap(o["rgb"], o["minmax"], s.gamma))
        assert hashlib.sha256(bmp).hexdigest() == str(z["bmp_sha256"])
    else:
        assert (~clean).sum() < 0.01 * clean.size


@pytest.mark.parametrize("name", ["test", "s3", "pokoj"])
def test_oracle_native_resolution_digests(name):
    known = json.load(open(os.path.join(GOLD, "known_answers.json")))[name]
    over = {k: os.path.join(ROOT, v) for k, v in known["overrides"].items()}
    s = O.OracleScene(os.path.join(ROOT, known["prefix"]), **over)
    assert (s.w, s.h) == (known["width"], known["height"])
    o = s.render(shadow_cache=True)
    assert hashlib.sha256(o["ids"].tobytes()).hexdigest() == known["ids_sha256"]
    assert int((o["ids"] >= 0).sum()) == known["primary_hits"]
    bmp = O.bmp24(O.tonemap(o["rgb"], o["minmax"], s.gamma))
    assert hashlib.sha256(bmp).hexdigest() == known["bmp_sha256"]


@pytest.mark.skipif(not os.environ.get("RT_SLOW"), reason="23 s + 12 s of single-core rendering; set RT_SLOW=1")
@pytest.mark.parametrize("name", ["s2", "s5", "temple", "ulica"])
def test_oracle_native_resolution_digests_slow(name):
    test_oracle_native_resolution_digests(name)


def test_shadow_cache_does_not_change_the_image():
    """The shadow cache is scan-order state; the GPU path drops it. Dropping it must not move the 8-bit image."""
    for name in ["s3", "temple", "pokoj"]:
        z, prefix, over = load_case(name)
        h, w = z["ids"].shape
        s = O.OracleScene(prefix, width=w, height=h, **over)
        a = s.render(shadow_cache=True)
        b = s.render(shadow_cache=False)
        assert np.array_equal(a["ids"], b["ids"])
        ia = O.rgba_to_u8(O.tonemap(a["rgb"], a["minmax"], s.gamma)).astype(int)
        ib = O.rgba_to_u8(O.tonemap(b["rgb"], b["minmax"], s.gamma)).astype(int)
        assert np.abs(ia - ib).max() <= 1


def test_parallel_rows_equal_serial_without_cache():
    z, prefix, over = load_case("s3")
    h, w = z["ids"].shape
    s = O.OracleScene(prefix, width=w, height=h, **over)
    a = s.render(shadow_cache=False, threads=1)
    b = s.render(shadow_cache=False, threads=4)
    assert np.array_equal(a["ids"], b["ids"])
    assert np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))
    assert np.array_equal(a["minmax"], b["minmax"])


def test_rand_restatement_matches_libc():
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    want = np.array([libc.rand() for _ in range(5000)], np.int32)
    assert np.array_equal(O.rand_values(5000), want)


@pytest.mark.skipif(not O.have_reference(), reason="oracle/_ref/ref_dump not built (no /root/reference here)")
def test_oracle_matches_live_reference_at_odd_resolution():
    prefix = os.path.join(ROOT, "scenes/s3/s3")
    ref = O.run_reference(prefix, width=97, height=61, cwd=os.path.join(ROOT, "scenes"))
    o = O.OracleScene(prefix, width=97, height=61).render(shadow_cache=True)
    assert np.array_equal(o["ids"], ref["ids"])
    assert np.array_equal(o["rgb"].view(np.uint32), ref["rgb"].view(np.uint32))
