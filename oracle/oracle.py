"""TEST INFRASTRUCTURE: ctypes wrapper over oracle/liboracle.so (CPU restatement of the
reference renderer) and over oracle/_ref/ref_dump (the unmodified reference, compiled here).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import json
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_BIN = os.path.join(HERE, "_ref", "ref_dump")

SHADOW_CACHE = 1
COUNTERS = 2

COUNTER_FIELDS = ["nearest_queries", "shadow_queries", "shadow_backface", "shadow_cached",
                  "nearest_steps", "nearest_tests", "shadow_steps", "shadow_tests", "pixels_traced"]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in COUNTER_FIELDS]

    def asdict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_FIELDS}


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.ora_load.restype = C.c_void_p
        L.ora_load.argtypes = [C.c_char_p] * 6 + [C.c_int]
        L.ora_free.argtypes = [C.c_void_p]
        L.ora_set_resolution.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ora_get_info.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_get_gamma.restype = C.c_float
        L.ora_get_gamma.argtypes = [C.c_void_p]
        L.ora_render.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
        L.ora_render_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
        L.ora_grid_dims.argtypes = [C.c_void_p] * 5
        L.ora_grid_refs.restype = C.c_int64
        L.ora_grid_refs.argtypes = [C.c_void_p] * 3
        L.ora_triangles.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_vertices.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_tonemap.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.ora_tonemap_multi.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ora_bmp24.restype = C.c_int64
        L.ora_bmp24.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ora_rand_fill.argtypes = [C.c_void_p, C.c_int64]
        _lib = L
    return _lib


def scene_paths(prefix, atr=None, cfg=None, lgt=None, cam=None, pnr=None, brs=None):
    """Same expansion as the reference CLI's -s PREFIX (main.c:123-130)."""
    return dict(brs=brs or prefix + ".brs", lgt=lgt or prefix + ".lgt", pnr=pnr or prefix + ".pnr",
                atr=atr or prefix + ".atr", cam=cam or prefix + ".cam", cfg=cfg or prefix + ".cfg")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleScene:
    def __init__(self, prefix, texture=False, width=None, height=None, **over):
        p = scene_paths(prefix, **over)
        enc = lambda s: s.encode() if s else None
        self._h = lib().ora_load(enc(p["brs"]), enc(p["lgt"]), enc(p["pnr"]), enc(p["atr"]), enc(p["cam"]),
                                 enc(p["cfg"]), int(bool(texture)))
        if not self._h:
            raise RuntimeError("oracle: cannot load scene %s" % prefix)
        if width and height:
            lib().ora_set_resolution(self._h, width, height)
        info = np.zeros(8, np.int32)
        lib().ora_get_info(self._h, _p(info))
        self.w, self.h, self.nt, self.nl, self.npl, self.ns, self.vmode, self.texture = [int(x) for x in info]
        self.gamma = float(lib().ora_get_gamma(self._h))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ora_free(self._h)
            self._h = None

    def render(self, shadow_cache=True, counters=False, threads=1, rows=None):
        """Returns dict(rgb [h,w,3] f32, ids [h,w] i32, minmax [6], taint [h,w] u8, counters dict|None)."""
        rgb = np.zeros((self.h, self.w, 3), np.float32)
        ids = np.full((self.h, self.w), -1, np.int32)
        mm = np.zeros(6, np.float32)
        taint = np.zeros((self.h, self.w), np.uint8)
        cnt = Counters()
        flags = (SHADOW_CACHE if shadow_cache else 0) | (COUNTERS if counters else 0)
        y0, y1 = rows if rows else (0, self.h)
        rc = lib().ora_render_rows(self._h, flags, threads, y0, y1, _p(rgb), _p(ids), _p(mm), _p(taint), C.byref(cnt))
        if rc:
            raise RuntimeError("oracle render failed: %d" % rc)
        return dict(rgb=rgb, ids=ids, minmax=mm, taint=taint, counters=cnt.asdict() if counters else None)

    def grid(self):
        nv = np.zeros(3, np.int32); cs = np.zeros(3, np.float32)
        dmin = np.zeros(3, np.float32); dmax = np.zeros(3, np.float32)
        if lib().ora_grid_dims(self._h, _p(nv), _p(cs), _p(dmin), _p(dmax)):
            raise RuntimeError("render first")
        nc = int(nv[0]) * int(nv[1]) * int(nv[2])
        start = np.zeros(nc + 1, np.int32)
        n = lib().ora_grid_refs(self._h, _p(start), None)
        refs = np.zeros(max(int(n), 1), np.int32)
        lib().ora_grid_refs(self._h, _p(start), _p(refs))
        return dict(nv=nv, cs=cs, dmin=dmin, dmax=dmax, cell_start=start, refs=refs[:int(n)])

    def triangles(self):
        out = np.zeros((self.nt, 17), np.float32)
        lib().ora_triangles(self._h, _p(out))
        return out

    def vertices(self):
        v = np.zeros((self.nt, 9), np.float32); sid = np.zeros(self.nt, np.int32)
        lib().ora_vertices(self._h, _p(v), _p(sid))
        return v, sid


def tonemap(rgb, minmax, gamma):
    """gamma: one gamma, or a list of them (the reference's zero-terminated list without its terminator,
    raytrace.c:439-447: the channels are averaged over the list)."""
    h, w, _ = rgb.shape
    out = np.zeros((h, w), np.uint32)
    rgb = np.ascontiguousarray(rgb, np.float32); mm = np.ascontiguousarray(minmax, np.float32)
    if isinstance(gamma, (list, tuple, np.ndarray)):
        g = np.ascontiguousarray(gamma, np.float32)
        lib().ora_tonemap_multi(w, h, _p(rgb), _p(mm), _p(g) if g.size else None, int(g.size), _p(out))
    else:
        lib().ora_tonemap(w, h, _p(rgb), _p(mm), C.c_float(gamma), _p(out))
    return out


def rgba_to_u8(rgba):
    """[h,w] packed R<<24|G<<16|B<<8|A -> [h,w,3] uint8."""
    return np.stack([(rgba >> 24) & 0xff, (rgba >> 16) & 0xff, (rgba >> 8) & 0xff], -1).astype(np.uint8)


def bmp24(rgba):
    h, w = rgba.shape
    rgba = np.ascontiguousarray(rgba, np.uint32)
    n = lib().ora_bmp24(w, h, _p(rgba), None)
    out = np.zeros(int(n), np.uint8)
    lib().ora_bmp24(w, h, _p(rgba), _p(out))
    return out.tobytes()


def rand_values(n):
    out = np.zeros(n, np.int32)
    lib().ora_rand_fill(_p(out), n)
    return out


# ---- the unmodified reference, where oracle/_ref/ref_dump has been built ----

def have_reference():
    return os.path.exists(REF_BIN)


def run_reference(prefix, width=None, height=None, cwd=None, want_bmp=False, gammas=None, **over):
    """Runs the compiled reference; returns dict(rgb, ids, minmax, gamma, seconds, bmp bytes|None).
    gammas: a list handed to the reference's rtVisualizedSceneToBitmap as its zero-terminated param1."""
    p = scene_paths(prefix, **over)
    with tempfile.TemporaryDirectory() as td:
        dump = os.path.join(td, "d.bin"); bmp = os.path.join(td, "o.bmp")
        cmd = [REF_BIN, "-g", p["brs"], "-l", p["lgt"], "-L", p["pnr"], "-a", p["atr"], "-c", p["cam"],
               "-C", p["cfg"], "-d", dump]
        if width and height:
            cmd += ["-W", str(width), "-H", str(height)]
        if want_bmp:
            cmd += ["-o", bmp]
        if gammas is not None:
            cmd += ["-G", ",".join(repr(float(g)) for g in gammas) + ","]
        cmd = [os.path.abspath(c) if os.path.exists(c) else c for c in cmd]
        out = subprocess.run(cmd, cwd=cwd, check=True, capture_output=True, text=True).stdout
        meta = json.loads(out.strip().splitlines()[-1])
        raw = open(dump, "rb").read()
        hdr = np.frombuffer(raw[:16], np.int32)
        w, h = int(hdr[1]), int(hdr[2])
        mm = np.frombuffer(raw[16:48], np.float32)
        px = np.frombuffer(raw[48:], np.int32).reshape(h, w, 4)
        res = dict(ids=px[:, :, 0].copy(), rgb=px[:, :, 1:].copy().view(np.float32), minmax=mm[:6].copy(),
                   gamma=float(mm[6]), seconds=meta["seconds"], meta=meta,
                   bmp=open(bmp, "rb").read() if want_bmp else None)
        return res
