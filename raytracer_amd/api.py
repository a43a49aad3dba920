"""ctypes bindings of librthost.so (C host layer) and librtgpu.so (HIP trace path)."""
import ctypes as C
import os
import subprocess

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
TILE = 8
TILE_PIXELS = 64


class RtgpuError(RuntimeError):
    def __init__(self, code, what, message=""):
        self.code = code
        super().__init__("%s failed: [%d] %s%s" % (what, code, gpu_lib().rtgpu_error_string(code).decode(),
                                                   (": " + message) if message else ""))


# ---- C structs of include/rtgpu.h -------------------------------------------------------------

class Surface(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("kd", C.c_float), ("ks", C.c_float), ("g", C.c_float), ("ka", C.c_float),
                ("kt", C.c_float), ("eta", C.c_float), ("kr", C.c_float)]


class Light(C.Structure):
    _fields_ = [("p", C.c_float * 3), ("flux", C.c_float), ("color", C.c_float * 3)]


class PlanarLight(C.Structure):
    _fields_ = [("flux", C.c_float), ("color", C.c_float * 3), ("a", C.c_float * 3), ("ab", C.c_float * 3),
                ("ac", C.c_float * 3)]


class Camera(C.Structure):
    _fields_ = [("ob", C.c_float * 3), ("ul", C.c_float * 3), ("bl", C.c_float * 3), ("ur", C.c_float * 3),
                ("sw", C.c_int32), ("sh", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("nt", C.c_int32), ("ns", C.c_int32), ("nl", C.c_int32), ("npl", C.c_int32),
                ("vertices", C.POINTER(C.c_float)), ("sid", C.POINTER(C.c_int32)), ("surfaces", C.POINTER(Surface)),
                ("lights", C.POINTER(Light)), ("planar", C.POINTER(PlanarLight)),
                ("dmin", C.c_float * 3), ("dmax", C.c_float * 3), ("gamma", C.c_float), ("distmod", C.c_float),
                ("vmode", C.c_int32), ("vcoeff", C.c_float * 3), ("texture_present", C.c_int32)]


class GridInfo(C.Structure):
    _fields_ = [("nv", C.c_int32 * 3), ("cell", C.c_float * 3), ("dmin", C.c_float * 3), ("dmax", C.c_float * 3),
                ("ncells", C.c_int32), ("nrefs", C.c_int32), ("total_flux", C.c_float)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("pixels", "pixels_traced", "nearest_queries", "shadow_queries", "shadow_walks",
                                         "nearest_steps", "nearest_tests", "shadow_steps", "shadow_tests",
                                         "nearest_lane_tests", "nearest_lane_slots", "shadow_lane_tests",
                                         "shadow_lane_slots", "nearest_stage2", "nearest_stage3", "nearest_exact",
                                         "shadow_stage2", "shadow_stage3", "shadow_exact", "nearest_visits", "shadow_visits", "nearest_batches",
                                         "nearest_batches_culled", "shadow_batches", "shadow_batches_culled",
                                         "nearest_full_tests", "shadow_full_tests", "chunks", "chunks_rerendered",
                                         "shadow_list_tickets", "shadow_list_fallbacks",
                                         "view_list_tickets", "view_list_fallbacks")]

    def asdict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}

    @property
    def rays(self):
        """Rays as the reference issues them: nearest-hit queries + shadow queries."""
        return int(self.nearest_queries + self.shadow_queries)


class Timings(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("preprocess_ms", "grid_ms", "primary_ms", "nearest_ms", "hit_ms", "shadow_ms",
                                         "accumulate_ms", "minmax_ms")] + [("nearest_launches", C.c_int32),
                                                                           ("shadow_launches", C.c_int32)]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class PixelLayout(C.Structure):
    """rtgpu_pixel_layout: a caller's pixel record (the reference's RT_VisualizedScenePixel is (24, 0, 16, &scene->t[0], 232))."""
    _fields_ = [("stride", C.c_int64), ("color_offset", C.c_int64), ("tri_offset", C.c_int64), ("tri_base", C.c_uint64),
                ("tri_stride", C.c_int64)]


class SceneFlat(C.Structure):
    _fields_ = [("desc", SceneDesc), ("vertices", C.POINTER(C.c_float)), ("sid", C.POINTER(C.c_int32)),
                ("surfaces", C.POINTER(Surface)), ("lights", C.POINTER(Light)), ("planar", C.POINTER(PlanarLight))]


def csrc_hash():
    """SHA-256 over the names and contents of the kernel sources (raytracer_amd/csrc/*): what counter profiles are
    stamped with, so that a reader can tell whether they describe the kernels of this tree."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            h.update(name.encode() + b"\0")
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


# ---- library loading --------------------------------------------------------------------------

_gpu = None
_host = None


def build_native(force=False):
    """Builds librtgpu.so (hipcc, gfx950), librthost.so and the CLI in-tree."""
    if force:
        subprocess.check_call(["make", "-s", "-C", ROOT, "clean"])
    subprocess.check_call(["make", "-s", "-C", ROOT, "all"])


def gpu_lib():
    global _gpu
    if _gpu is None:
        path = os.environ.get("RTGPU_LIB") or os.path.join(PKG, "librtgpu.so")  # RTGPU_LIB: kernel A/B experiments
        if not os.path.exists(path):
            raise RuntimeError("librtgpu.so is missing: run `make` (needs hipcc); there is no fallback path")
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
        L.rtgpu_error_string.restype = C.c_char_p
        L.rtgpu_error_string.argtypes = [i32]
        L.rtgpu_last_message.restype = C.c_char_p
        L.rtgpu_last_message.argtypes = [vp]
        L.rtgpu_create.argtypes = [i32, C.POINTER(vp)]
        L.rtgpu_destroy.argtypes = [vp]
        L.rtgpu_destroy.restype = None
        L.rtgpu_set_option.argtypes = [vp, C.c_char_p, i64]
        L.rtgpu_set_stream.argtypes = [vp, vp]
        L.rtgpu_set_null_stream.argtypes = [vp]
        L.rtgpu_scene_upload.argtypes = [vp, C.POINTER(SceneDesc)]
        L.rtgpu_frame_begin.argtypes = [vp, C.POINTER(Camera)]
        L.rtgpu_grid_info_get.argtypes = [vp, C.POINTER(GridInfo)]
        L.rtgpu_tile_count.restype = i64
        L.rtgpu_tile_count.argtypes = [vp, i32, i32]
        L.rtgpu_render.argtypes = [vp, i32, i32, vp, vp, vp]
        L.rtgpu_render_static.argtypes = [vp, i32, i32, vp, vp, vp]
        L.rtgpu_static_scratch_bytes.restype = i64
        L.rtgpu_static_scratch_bytes.argtypes = []
        L.rtgpu_chunk_tile_limit.restype = i64
        L.rtgpu_chunk_tile_limit.argtypes = [i32, i32, i32]
        L.rtgpu_render_trace.argtypes = [vp, i32, i32]
        L.rtgpu_hit_counts.argtypes = [vp, vp]
        L.rtgpu_hit_total.argtypes = [vp, vp]
        L.rtgpu_render_shade.argtypes = [vp, vp, i64, i64, vp, vp, vp]
        L.rtgpu_tonemap.argtypes = [vp, vp, i64, vp, C.c_float, vp]
        L.rtgpu_tonemap_multi.argtypes = [vp, vp, i64, vp, C.POINTER(C.c_float), i32, vp]
        L.rtgpu_tonemap_multi_host.argtypes = [vp, vp, i64, vp, C.POINTER(C.c_float), i32, vp]
        L.rtgpu_untile.argtypes = [vp, vp, i32, i32, vp]
        L.rtgpu_bmp_row_bytes.argtypes = [i32, i32]
        L.rtgpu_bmp_row_bytes.restype = C.c_int64
        L.rtgpu_bmp_rows.argtypes = [vp, vp, i32, vp]
        L.rtgpu_bmp_header.argtypes = [i32, i32, i32, vp]
        L.rtgpu_sync.argtypes = [vp]
        L.rtgpu_counters_get.argtypes = [vp, C.POINTER(Counters)]
        L.rtgpu_timings_get.argtypes = [vp, C.POINTER(Timings)]
        L.rtgpu_render_host.argtypes = [vp, vp, vp, vp]
        L.rtgpu_tonemap_host.argtypes = [vp, vp, i64, vp, C.c_float, vp]
        L.rtgpu_render_pixelmap.argtypes = [vp, vp, C.POINTER(PixelLayout), vp]
        L.rtgpu_tonemap_pixelmap.argtypes = [vp, vp, C.POINTER(PixelLayout), i64, vp, C.POINTER(C.c_float), i32, vp]
        L.rtgpu_pinned_alloc.restype = vp
        L.rtgpu_pinned_alloc.argtypes = [C.c_size_t]
        L.rtgpu_pinned_free.restype = None
        L.rtgpu_pinned_free.argtypes = [vp]
        L.rtgpu_grid_download.argtypes = [vp, vp, vp]
        L.rtgpu_triangles_download.argtypes = [vp, vp]
        _gpu = L
    return _gpu


def host_lib():
    global _host
    if _host is None:
        gpu_lib()
        path = os.path.join(PKG, "librthost.so")
        if not os.path.exists(path):
            raise RuntimeError("librthost.so is missing: run `make`")
        L = C.CDLL(path)
        vp = C.c_void_p
        L.rtSceneLoadAll.restype = vp
        L.rtSceneLoadAll.argtypes = [C.c_char_p] * 5 + [C.c_float] * 3
        L.rtSceneLoadAllCached.restype = vp
        L.rtSceneLoadAllCached.argtypes = [C.c_char_p] * 6 + [C.c_float] * 3
        L.rtSceneDestroy.argtypes = [C.POINTER(vp)]
        L.rtCameraLoad.restype = vp
        L.rtCameraLoad.argtypes = [C.c_char_p]
        L.rtCameraDestroy.argtypes = [C.POINTER(vp)]
        L.rtScenePreprocess.restype = vp
        L.rtScenePreprocess.argtypes = [vp, vp]
        L.rtSceneFlatten.restype = C.POINTER(SceneFlat)
        L.rtSceneFlatten.argtypes = [vp]
        L.rtSceneFlatFree.argtypes = [C.POINTER(C.POINTER(SceneFlat))]
        L.rtCameraFlatten.argtypes = [vp, C.POINTER(Camera)]
        L.rtBitmapCreate.restype = vp
        L.rtBitmapCreate.argtypes = [C.c_int32, C.c_int32, C.c_uint32]
        L.rtBitmapSave.argtypes = [vp, C.c_char_p, C.c_uint16]
        L.rtBitmapDestroy.argtypes = [C.POINTER(vp)]
        L.rtGetErrorDesc.restype = C.c_char_p
        _host = L
    return _host


def scene_paths(prefix, atr=None, cfg=None, lgt=None, cam=None, pnr=None, brs=None):
    """Expansion of the CLI's -s PREFIX (src/main.c:123-130 of the reference)."""
    return dict(brs=brs or prefix + ".brs", lgt=lgt or prefix + ".lgt", pnr=pnr or prefix + ".pnr",
                atr=atr or prefix + ".atr", cam=cam or prefix + ".cam", cfg=cfg or prefix + ".cfg")


class _Bitmap(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("background", C.c_uint32), ("pixels", C.POINTER(C.c_uint32))]


def save_bmp24(rgba, path):
    """Writes a [h, w] uint32 image (R<<24|G<<16|B<<8) with the host layer's BMP writer."""
    H = host_lib()
    h, w = rgba.shape
    bmp = H.rtBitmapCreate(w, h, 0)
    if not bmp:
        raise MemoryError("rtBitmapCreate")
    b = C.cast(bmp, C.POINTER(_Bitmap)).contents
    C.memmove(b.pixels, np.ascontiguousarray(rgba, np.uint32).ctypes.data, 4 * w * h)
    H.rtBitmapSave(bmp, path.encode(), 24)
    p = C.c_void_p(bmp)
    H.rtBitmapDestroy(C.byref(p))


class Scene:
    """A scene loaded by the C host layer (same files and quirks as the reference's loaders)."""

    def __init__(self, prefix, texture=False, width=None, height=None, gamma=2.5, epsilon=0.0, distmod=2.0, geometry_cache=None, **over):
        """texture: False = brick texture off, True = on (needs textures/brickwall.bmp in the cwd, which is the
        reference's on/off switch, src/preprocess.c:13), None = on iff that file can be loaded.
        geometry_cache: path of a binary cache of the geometry file (rtSceneLoadCached), or None."""
        H = host_lib()
        p = scene_paths(prefix, **over)
        enc = lambda s: s.encode() if s else None
        C.set_errno(0)
        self._scene = H.rtSceneLoadAllCached(enc(p["brs"]), enc(geometry_cache), enc(p["lgt"]), enc(p["pnr"]), enc(p["atr"]), enc(p["cfg"]),
                                             gamma, epsilon, distmod)
        if not self._scene:
            raise RuntimeError("cannot load scene %s" % prefix)
        self._cam = H.rtCameraLoad(enc(p["cam"]))
        if not self._cam:
            raise RuntimeError("cannot load camera %s" % p["cam"])
        self.camera = Camera()
        H.rtCameraFlatten(self._cam, C.byref(self.camera))
        if width and height:
            self.camera.sw, self.camera.sh = int(width), int(height)
        if texture is None or texture:
            H.rtScenePreprocess(self._scene, self._cam)
        self._flat = H.rtSceneFlatten(self._scene)
        if not self._flat:
            raise MemoryError("rtSceneFlatten")
        self.desc = self._flat.contents.desc
        if texture and not self.desc.texture_present:
            raise RuntimeError("texture requested but textures/brickwall.bmp cannot be loaded from the cwd")
        d = self.desc
        self.nt, self.ns, self.nl, self.npl = d.nt, d.ns, d.nl, d.npl
        self.gamma = d.gamma if d.gamma > 0 else 2.5

    @property
    def width(self):
        return self.camera.sw

    @property
    def height(self):
        return self.camera.sh

    def vertices(self):
        return np.ctypeslib.as_array(self.desc.vertices, shape=(self.nt, 9)).copy()

    def sids(self):
        return np.ctypeslib.as_array(self.desc.sid, shape=(self.nt,)).copy()

    def __del__(self):
        try:
            H = host_lib()
            if getattr(self, "_flat", None):
                H.rtSceneFlatFree(C.byref(self._flat))
                self._flat = None
            if getattr(self, "_cam", None):
                p = C.c_void_p(self._cam)
                H.rtCameraDestroy(C.byref(p))
                self._cam = None
            if getattr(self, "_scene", None):
                p = C.c_void_p(self._scene)
                H.rtSceneDestroy(C.byref(p))
                self._scene = None
        except Exception:
            pass


def _ptr(x):
    """Device or host pointer of a torch tensor / numpy array / int."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    if isinstance(x, np.ndarray):
        return C.c_void_p(x.ctypes.data)
    return C.c_void_p(int(x))


class Renderer:
    """One GPU context of librtgpu (one per process/GPU)."""

    def __init__(self, device=0, **options):
        self._L = gpu_lib()
        h = C.c_void_p()
        rc = self._L.rtgpu_create(int(device), C.byref(h))
        if rc:
            raise RtgpuError(rc, "rtgpu_create")
        self._h = h
        self.device = int(device)
        self.scene = None
        for k, v in options.items():
            self.set_option(k, v)

    def close(self):
        if getattr(self, "_h", None):
            self._L.rtgpu_destroy(self._h)
            self._h = None

    __del__ = close

    def _chk(self, rc, what):
        if rc:
            raise RtgpuError(rc, what, self._L.rtgpu_last_message(self._h).decode())

    def set_option(self, name, value):
        self._chk(self._L.rtgpu_set_option(self._h, name.encode(), int(value)), "rtgpu_set_option(%s)" % name)

    def set_stream(self, stream):
        """stream: a torch.cuda.Stream, a raw hipStream_t handle, or None for the context's own stream.  A torch
        stream whose handle is 0 is the device's null stream (torch's default): the context then enqueues there."""
        h = getattr(stream, "cuda_stream", stream)
        if stream is not None and h == 0:
            self._chk(self._L.rtgpu_set_null_stream(self._h), "rtgpu_set_null_stream")
        else:
            self._chk(self._L.rtgpu_set_stream(self._h, C.c_void_p(h) if h else None), "rtgpu_set_stream")

    def upload(self, scene):
        self._chk(self._L.rtgpu_scene_upload(self._h, C.byref(scene.desc)), "rtgpu_scene_upload")
        self.scene = scene

    def frame_begin(self, camera=None):
        cam = camera if camera is not None else self.scene.camera
        self._chk(self._L.rtgpu_frame_begin(self._h, C.byref(cam)), "rtgpu_frame_begin")
        self.camera = cam

    def grid_info(self):
        g = GridInfo()
        self._chk(self._L.rtgpu_grid_info_get(self._h, C.byref(g)), "rtgpu_grid_info_get")
        return g

    def tile_count(self, rank=0, world=1):
        return int(self._L.rtgpu_tile_count(self._h, rank, world))

    def render(self, rgb, ids, minmax, rank=0, world=1):
        """rgb/ids/minmax: device buffers (torch CUDA tensors or raw pointers), tile-major."""
        self._chk(self._L.rtgpu_render(self._h, rank, world, _ptr(rgb), _ptr(ids), _ptr(minmax)), "rtgpu_render")

    def render_static(self, rgb, ids, minmax, rank=0, world=1):
        """render() with the queues in the library's static scratch array: no device memory is allocated."""
        self._chk(self._L.rtgpu_render_static(self._h, rank, world, _ptr(rgb), _ptr(ids), _ptr(minmax)), "rtgpu_render_static")

    def render_trace(self, rank=0, world=1):
        """First half of render(): everything up to the hits (see rtgpu_render_trace)."""
        self._chk(self._L.rtgpu_render_trace(self._h, rank, world), "rtgpu_render_trace")

    def hit_counts(self, counts):
        self._chk(self._L.rtgpu_hit_counts(self._h, _ptr(counts)), "rtgpu_hit_counts")

    def hit_total(self, total):
        """total: a device int64 that receives this rank's number of shaded hits (the sum of hit_counts)."""
        self._chk(self._L.rtgpu_hit_total(self._h, _ptr(total)), "rtgpu_hit_total")

    def render_shade(self, rgb, ids, minmax, all_counts=None, per_rank=0, total_hits=-1):
        self._chk(self._L.rtgpu_render_shade(self._h, _ptr(all_counts) if all_counts is not None else None, per_rank, total_hits,
                                             _ptr(rgb), _ptr(ids), _ptr(minmax)), "rtgpu_render_shade")

    def tonemap(self, rgb, npixels, minmax, gamma, rgba):
        """gamma: one gamma, or a list of them (the reference's zero-terminated list without its terminator)."""
        if isinstance(gamma, (list, tuple, np.ndarray)):
            g = (C.c_float * max(len(gamma), 1))(*[float(x) for x in gamma])
            self._chk(self._L.rtgpu_tonemap_multi(self._h, _ptr(rgb), int(npixels), _ptr(minmax), g, len(gamma), _ptr(rgba)),
                      "rtgpu_tonemap_multi")
        else:
            self._chk(self._L.rtgpu_tonemap(self._h, _ptr(rgb), int(npixels), _ptr(minmax), C.c_float(gamma), _ptr(rgba)),
                      "rtgpu_tonemap")

    def untile(self, gathered, world, elem_bytes, image):
        self._chk(self._L.rtgpu_untile(self._h, _ptr(gathered), world, elem_bytes, _ptr(image)), "rtgpu_untile")

    def bmp_rows(self, image, bpp, rows):
        """image: device RGBA8 image (row-major), rows: device byte buffer of rtgpu_bmp_row_bytes(w, bpp) * h bytes."""
        self._chk(self._L.rtgpu_bmp_rows(self._h, _ptr(image), bpp, _ptr(rows)), "rtgpu_bmp_rows")

    def bmp_file(self, image, bpp=24):
        """The complete BMP file (bytes) of a device RGBA8 image: rows packed on the device, header from the library."""
        import torch
        w, h = self.camera.sw, self.camera.sh
        row = int(self._L.rtgpu_bmp_row_bytes(w, bpp))
        rows = torch.empty(row * h, dtype=torch.uint8, device=image.device)
        torch.cuda.synchronize(image.device)  # `image` may still be in flight on torch's stream; the context has its own
        self.bmp_rows(image, bpp, rows)
        self.sync()
        hdr = (C.c_ubyte * 54)()
        self._chk(self._L.rtgpu_bmp_header(w, h, bpp, hdr), "rtgpu_bmp_header")
        return bytes(hdr) + rows.cpu().numpy().tobytes()

    def sync(self):
        self._chk(self._L.rtgpu_sync(self._h), "rtgpu_sync")

    def counters(self):
        c = Counters()
        self._chk(self._L.rtgpu_counters_get(self._h, C.byref(c)), "rtgpu_counters_get")
        return c

    def timings(self):
        t = Timings()
        self._chk(self._L.rtgpu_timings_get(self._h, C.byref(t)), "rtgpu_timings_get")
        return t

    def render_host(self):
        """Single-GPU frame into host arrays: dict(rgb [h,w,3] f32, ids [h,w] i32, minmax [6] f32)."""
        w, h = self.camera.sw, self.camera.sh
        rgb = np.zeros((h, w, 3), np.float32)
        ids = np.zeros((h, w), np.int32)
        mm = np.zeros(6, np.float32)
        self._chk(self._L.rtgpu_render_host(self._h, _ptr(rgb), _ptr(ids), _ptr(mm)), "rtgpu_render_host")
        return dict(rgb=rgb, ids=ids, minmax=mm)

    def render_pixelmap(self, layout, pinned=True):
        """Single-GPU frame into the caller's pixel records (rtgpu_render_pixelmap): returns (bytes [h*w, stride] u8, minmax)."""
        w, h = self.camera.sw, self.camera.sh
        n = w * h * int(layout.stride)
        mm = np.zeros(6, np.float32)
        block = self._L.rtgpu_pinned_alloc(n) if pinned else None
        if pinned and not block:
            raise MemoryError("rtgpu_pinned_alloc")
        try:
            buf = (C.c_ubyte * n).from_address(block) if pinned else (C.c_ubyte * n)()
            self._chk(self._L.rtgpu_render_pixelmap(self._h, C.cast(buf, C.c_void_p), C.byref(layout), _ptr(mm)), "rtgpu_render_pixelmap")
            out = np.frombuffer(buf, np.uint8).reshape(h * w, int(layout.stride)).copy()
        finally:
            if pinned:
                self._L.rtgpu_pinned_free(block)
        return out, mm

    def tonemap_pixelmap(self, records, layout, minmax, gamma):
        """rtgpu_tonemap_pixelmap on records [n, stride] u8; gamma: one gamma or a list."""
        records = np.ascontiguousarray(records, np.uint8)
        n = records.shape[0]
        gl = list(gamma) if isinstance(gamma, (list, tuple, np.ndarray)) else [gamma]
        g = (C.c_float * max(len(gl), 1))(*[float(x) for x in gl])
        mm = np.ascontiguousarray(minmax, np.float32)
        out = np.zeros(n, np.uint32)
        self._chk(self._L.rtgpu_tonemap_pixelmap(self._h, _ptr(records), C.byref(layout), n, _ptr(mm), g, len(gl), _ptr(out)),
                  "rtgpu_tonemap_pixelmap")
        return out

    def tonemap_host(self, rgb, minmax, gamma):
        h, w, _ = rgb.shape
        rgb = np.ascontiguousarray(rgb, np.float32)
        mm = np.ascontiguousarray(minmax, np.float32)
        out = np.zeros((h, w), np.uint32)
        if isinstance(gamma, (list, tuple, np.ndarray)):
            g = (C.c_float * max(len(gamma), 1))(*[float(x) for x in gamma])
            self._chk(self._L.rtgpu_tonemap_multi_host(self._h, _ptr(rgb), w * h, _ptr(mm), g, len(gamma), _ptr(out)),
                      "rtgpu_tonemap_multi_host")
        else:
            self._chk(self._L.rtgpu_tonemap_host(self._h, _ptr(rgb), w * h, _ptr(mm), C.c_float(gamma), _ptr(out)),
                      "rtgpu_tonemap_host")
        return out

    def grid(self):
        g = self.grid_info()
        start = np.zeros(g.ncells + 1, np.int32)
        refs = np.zeros(max(g.nrefs, 1), np.int32)
        self._chk(self._L.rtgpu_grid_download(self._h, _ptr(start), _ptr(refs)), "rtgpu_grid_download")
        return dict(nv=np.array(g.nv[:], np.int32), cs=np.array(g.cell[:], np.float32), dmin=np.array(g.dmin[:], np.float32),
                    dmax=np.array(g.dmax[:], np.float32), cell_start=start, refs=refs[:g.nrefs], total_flux=g.total_flux)

    def triangles(self):
        out = np.zeros((self.scene.nt, 10), np.float32)
        self._chk(self._L.rtgpu_triangles_download(self._h, _ptr(out)), "rtgpu_triangles_download")
        return out


def rgba_to_u8(rgba):
    return np.stack([(rgba >> 24) & 0xff, (rgba >> 16) & 0xff, (rgba >> 8) & 0xff], -1).astype(np.uint8)


def render_bmp(prefix, out_path, device=0, width=None, height=None, **over):
    """Whole pipeline on one GPU: load, render, tone-map, write a 24-bpp BMP.  Returns the frame dict."""
    scene = Scene(prefix, width=width, height=height, **over)
    r = Renderer(device)
    r.upload(scene)
    r.frame_begin()
    frame = r.render_host()
    rgba = r.tonemap_host(frame["rgb"], frame["minmax"], scene.gamma)
    save_bmp24(rgba, out_path)
    frame["rgba"] = rgba
    r.close()
    return frame
