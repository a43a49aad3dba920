"""CPU-only checks: the C host layer against the oracle's loaders, the BMP writer, and that the C-ABI
library loads and exports every symbol include/rtgpu.h declares (no compute without a GPU)."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest

import raytracer_amd as rt
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = [("scenes/test/s2", {"atr": "scenes/temple/temple.atr"}), ("scenes/s2/s2", {}), ("scenes/s3/s3", {}),
          ("scenes/s5/s5", {}), ("scenes/pokoj/s2", {}), ("scenes/temple/temple", {}), ("scenes/ulica/ulica", {})]


def test_abi_exports_every_declared_symbol():
    text = open(os.path.join(ROOT, "include", "rtgpu.h")).read()
    names = sorted(set(re.findall(r"\b(rtgpu_[a-z_]+)\s*\(", text)))
    assert len(names) >= 18
    lib = rt.gpu_lib()
    for n in names:
        assert hasattr(lib, n), n


def test_header_describes_every_option_the_library_accepts():
    src = open(os.path.join(ROOT, "raytracer_amd", "csrc", "rtgpu.hip")).read()
    body = src[src.index("int rtgpu_set_option("):]
    body = body[:body.index("unknown option")]
    options = sorted(set(re.findall(r'strcmp\(name, "([a-z_]+)"\)', body)))
    assert len(options) >= 20
    text = open(os.path.join(ROOT, "include", "rtgpu.h")).read()
    comment = text[:text.index("int rtgpu_set_option(")]
    comment = comment[comment.rindex("/*"):]
    for o in options:
        assert '"%s"' % o in comment, o


def test_host_layer_exports_reference_entry_points():
    lib = rt.host_lib()
    for n in ["rtSceneLoad", "rtSceneConfigureRenderer", "rtSceneSetSurfaces", "rtSceneSetLights", "rtSceneSetPlanarLights",
              "rtSceneDestroy", "rtLightLoad", "rtPlanarLightLoad", "rtSurfaceLoad", "rtCameraLoad", "rtCameraDestroy",
              "rtScenePreprocess", "rtVisualizedSceneRaytrace", "rtVisualizedSceneDestroy", "rtVisualizedSceneToBitmap",
              "rtBitmapCreate", "rtBitmapDestroy", "rtBitmapLoad", "rtBitmapSave", "rtGetErrorDesc"]:
        assert hasattr(lib, n), n


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtgpuError) as e:
        rt.Renderer(0)
    assert e.value.code == 100


@pytest.mark.parametrize("prefix,over", SCENES)
def test_host_loader_matches_oracle_loader(prefix, over):
    over = {k: os.path.join(ROOT, v) for k, v in over.items()}
    s = rt.Scene(os.path.join(ROOT, prefix), **over)
    o = O.OracleScene(os.path.join(ROOT, prefix), **over)
    v, sid = o.vertices()
    assert np.array_equal(v.view(np.uint32), s.vertices().view(np.uint32))  # includes the loader's vertex nudge
    assert np.array_equal(sid, s.sids())
    assert (s.nt, s.ns, s.nl, s.npl) == (o.nt, o.ns, o.nl, o.npl)
    assert (s.width, s.height) == (o.w, o.h) and s.gamma == o.gamma and s.desc.vmode == o.vmode


def test_cfg_overrides_and_missing_files():
    base = os.path.join(ROOT, "scenes/pokoj/s2")
    assert rt.Scene(base, cfg=os.path.join(ROOT, "scenes/cegly.cfg")).gamma == 2.0
    assert rt.Scene(base).gamma == 2.5  # no pokoj/s2.cfg: defaults stay (a warning in the reference)
    assert rt.Scene(base, gamma=1.7).gamma == pytest.approx(1.7)
    with pytest.raises(RuntimeError):
        rt.Scene(os.path.join(ROOT, "scenes/nope/nope"))
    with pytest.raises(RuntimeError):  # surface ids 10/20/30 need at least 31 surfaces
        rt.Scene(os.path.join(ROOT, "scenes/test/s2"), atr=os.path.join(ROOT, "scenes/s3/s3.atr"))


def test_texture_switch_follows_cwd_file(tmp_path):
    cwd = os.getcwd()
    try:
        os.chdir(ROOT)
        assert rt.Scene(os.path.join(ROOT, "scenes/pokoj/s2"), texture=True).desc.texture_present == 1
        assert rt.Scene(os.path.join(ROOT, "scenes/pokoj/s2"), texture=False).desc.texture_present == 0
        os.chdir(tmp_path)
        assert rt.Scene(os.path.join(ROOT, "scenes/pokoj/s2"), texture=None).desc.texture_present == 0
        with pytest.raises(RuntimeError):
            rt.Scene(os.path.join(ROOT, "scenes/pokoj/s2"), texture=True)
    finally:
        os.chdir(cwd)


def test_bmp_writer_bytes_match_oracle(tmp_path):
    rng = np.random.default_rng(1)
    for w, h in [(5, 3), (8, 8), (97, 61)]:  # widths with and without row padding
        rgba = (rng.integers(0, 2 ** 32, size=(h, w), dtype=np.uint64) & 0xffffff00).astype(np.uint32)
        p = str(tmp_path / "o.bmp")
        rt.save_bmp24(rgba, p)
        assert open(p, "rb").read() == O.bmp24(rgba)


def test_cli_binary_prints_help():
    import subprocess
    out = subprocess.run([os.path.join(ROOT, "raytracer_amd", "raytrace")], capture_output=True, text=True)
    assert out.returncode == 0 and "usage:" in out.stdout


def test_loader_float_parser_equals_strtof():
    """The geometry loader converts short decimals itself (one exact division) and must return what sscanf("%f")
    = strtof returns, bit for bit, and stop at the same character, on every input."""
    import ctypes as C
    import random
    H = rt.host_lib()
    H.rtParseFloat.restype = C.c_float
    H.rtParseFloat.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    libc = C.CDLL("libc.so.6")
    libc.strtof.restype = C.c_float
    libc.strtof.argtypes = [C.c_char_p, C.POINTER(C.c_char_p)]

    def both(text):
        b = text.encode()
        n = C.c_int(0)
        x = H.rtParseFloat(b, C.byref(n))
        buf = C.create_string_buffer(b)
        end = C.c_char_p()
        y = libc.strtof(buf, C.byref(end))
        used = C.cast(end, C.c_void_p).value - C.addressof(buf)
        return np.float32(x).view(np.uint32), n.value, np.float32(y).view(np.uint32), used

    edge = ["", "abc", " 12.5 7", "1e5", "1E-3", "0x10", "nan", "inf", "-inf", "-0.0000", ".5", "5.", "-.25x", "+3.1400",
            "1234567.8", "0.00000012345678", "00012.3400", "9999999", "16777217", "0.12345678901", "1.00000000000", "--1",
            "\t\n 23.6853 0.0000", "4294967296.5", "0.0000001", "1234567", "123.4567e", "7.5p3"]
    rng = random.Random(7)
    for _ in range(300000):
        m = rng.randrange(0, 20000000)
        d = rng.randrange(0, 12)
        digits = str(m)
        text = ("0." + digits.rjust(d, "0")) if d >= len(digits) else digits[:len(digits) - d] + ("." + digits[len(digits) - d:] if d else "")
        edge.append(("-" if rng.random() < 0.5 else "") + text)
    for text in edge:
        a, na, b, nb = both(text)
        assert (a, na) == (b, nb), text


def test_geometry_cache_gives_the_same_scene_and_notices_a_changed_source(tmp_path):
    import shutil
    import time
    src = tmp_path / "scene"
    for ext in (".brs", ".atr", ".cam", ".lgt"):
        shutil.copy(os.path.join(ROOT, "scenes/s5/s5" + ext), str(src) + ext)
    cache = str(tmp_path / "scene.geom")
    plain = rt.Scene(str(src))
    first = rt.Scene(str(src), geometry_cache=cache)      # writes the cache
    assert os.path.getsize(cache) == 56 + 40 * plain.nt
    t0 = time.time(); second = rt.Scene(str(src), geometry_cache=cache); cached_s = time.time() - t0  # reads it
    for s in (first, second):
        assert s.nt == plain.nt
        assert np.array_equal(s.vertices().view(np.uint32), plain.vertices().view(np.uint32))
        assert np.array_equal(s.sids(), plain.sids())
        assert list(s.desc.dmin) == list(plain.desc.dmin) and list(s.desc.dmax) == list(plain.desc.dmax)
    # a changed source (here: one triangle fewer is not needed; a new modification time is enough) invalidates the cache
    lines = open(str(src) + ".brs").read().split("\n")
    first_vertex = lines[1].split()
    lines[1] = "%.4f %s %s" % (float(first_vertex[0]) + 1.0, first_vertex[1], first_vertex[2])
    open(str(src) + ".brs", "w").write("\n".join(lines))
    os.utime(str(src) + ".brs", (time.time() + 5, time.time() + 5))
    third = rt.Scene(str(src), geometry_cache=cache)
    fresh = rt.Scene(str(src))
    assert np.array_equal(third.vertices().view(np.uint32), fresh.vertices().view(np.uint32))
    assert not np.array_equal(third.vertices(), plain.vertices())


def test_chunk_limit_keeps_items_inside_32_bit_tickets():
    """The walk kernels number their work items with 32-bit tickets of 64 and nodes with int32: the chunk planner's
    bound (needs no GPU) keeps tiles * 64 * fan^(levels-1) * nslots below 2^32 and the node count below 2^31."""
    L = rt.gpu_lib()
    L.rtgpu_chunk_tile_limit.restype = ctypes.c_int64
    L.rtgpu_chunk_tile_limit.argtypes = [ctypes.c_int] * 3
    for fan, levels, nslots in [(0, 1, 6), (0, 1, 26), (0, 1, 64), (1, 5, 6), (2, 5, 64), (2, 5, 80), (0, 1, 0), (2, 5, 100000)]:
        lim = int(L.rtgpu_chunk_tile_limit(fan, levels, nslots))
        assert lim >= 1
        deepest = 64 * fan ** (levels - 1) if levels > 1 else 64
        nodes = sum(64 * fan ** k for k in range(levels)) if levels > 1 else 64
        items = lim * max(deepest, 64) * max(nslots, 1)
        if lim > 1:
            assert items + 64 <= 2 ** 32 - 1 and lim * nodes <= 2 ** 31 - 1
            # and the bound is tight: one more tile would break one of the two
            assert (lim + 1) * max(deepest, 64) * max(nslots, 1) + 64 > 2 ** 32 - 1 or (lim + 1) * nodes > 2 ** 31 - 1
    # the case of the review: a scene without secondary rays, 26 lights, more than 67 M pixels in one chunk
    assert int(L.rtgpu_chunk_tile_limit(0, 1, 26)) * 64 * 26 < 2 ** 32
    assert int(L.rtgpu_chunk_tile_limit(3, 5, 6)) == -1 and int(L.rtgpu_chunk_tile_limit(1, 6, 6)) == -1


@pytest.mark.parametrize("w,h", [(3840, 2160), (97, 61), (8, 8), (7680, 4320), (50, 100), (5, 3)])
def test_tile_plan_of_the_c_multi_gpu_host(w, h):
    """rtTilePlan (host/scene.c), the bookkeeping of `raytrace --devices N`: tiles per device and the offsets of the
    devices' blocks in the gathered buffer are a partition of the image's tiles, in the layout the Python path
    (raytracer_amd/dist.py) and rtgpu_untile use."""
    from raytracer_amd import dist as D
    H = rt.host_lib()
    H.rtTilePlan.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    H.rtTilePlan.restype = None
    ntiles = D.tiles_of(w, h)
    for ndev in (1, 2, 3, 4, 8, 64):
        tiles = (ctypes.c_int64 * ndev)()
        first = (ctypes.c_int64 * (ndev + 1))()
        H.rtTilePlan(w, h, ndev, tiles, first)
        assert list(tiles) == [D.tile_count(ntiles, r, ndev) for r in range(ndev)]
        assert sum(tiles) == ntiles and first[0] == 0 and first[ndev] == ntiles * 64
        assert all(first[r + 1] - first[r] == tiles[r] * 64 for r in range(ndev))
        # the gathered layout de-interleaves to the image: a frame whose pixels hold their own index survives the round trip
        if w * h <= 97 * 61:
            img = np.arange(w * h, dtype=np.int64).reshape(h, w)
            parts = [D.split_tiles_reference(img, r, ndev).reshape(-1) for r in range(ndev)]
            assert all(len(p) == tiles[r] * 64 for r, p in enumerate(parts))
            back = D.untile_reference(np.concatenate(parts), ndev, w, h)
            assert np.array_equal(back, img)


def test_integration_md_shows_the_stub_that_is_compiled():
    """The integration guide prints integration/raytrace_gpu.c, the file that oracle/Makefile compiles against the
    reference's headers: the printed text is the file, not a sketch of it."""
    stub = open(os.path.join(ROOT, "integration", "raytrace_gpu.c")).read()
    assert stub in open(os.path.join(ROOT, "INTEGRATION.md")).read()
