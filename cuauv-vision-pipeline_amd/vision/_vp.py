"""ctypes binding of libvp.so (C ABI: include/vp.h).  Thin: argument marshalling only.

The library is built in-tree by `cuauv-vision-pipeline_amd/build.py` (hipcc --offload-arch=gfx950) into
`cuauv-vision-pipeline_amd/lib/libvp.so`.  If it is missing, or no MI355X is visible, every operator
raises VpError — the product has no CPU path.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libvp.so")   # VP_LIB: measurement builds (tools/build_probe.sh)

BGR2LAB, BGR2HSV, BGR2GRAY, GRAY2BGR, HSV2BGR, BGR2YCRCB, BGR2HLS = 0, 1, 2, 3, 4, 5, 6
CB_EQUALIZE_RGB, CB_RGB_CONTRAST, CB_HSV_CONTRAST, CB_HSI_CONTRAST, CB_EXTREMA_CLIPPING, CB_ADAPTIVE_CAST = 1, 2, 4, 8, 16, 32
CB_DEFAULT = CB_EQUALIZE_RGB | CB_HSV_CONTRAST | CB_EXTREMA_CLIPPING
MORPH_ERODE, MORPH_DILATE, MORPH_OPEN, MORPH_CLOSE, MORPH_GRADIENT = 0, 1, 2, 3, 4
SHAPE_RECT, SHAPE_CROSS, SHAPE_ELLIPSE = 0, 1, 2
CCL_PIXEL, CCL_BLOCK2X2 = 1, 2
RETR_EXTERNAL, RETR_LIST = 0, 1
CHAIN_APPROX_NONE, CHAIN_APPROX_SIMPLE = 1, 2
CHAIN_MAX_MORPH = 8
OPT_CHAIN_STREAMS = 1
OPT_CCL_LEVELS = 2
OPT_CCL_MERGE_CAP = 3
OPT_FLAT_OPS = 4
PROF_KERNELS = 15


class VpError(RuntimeError):
    pass


class ChainDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("color_mode", C.c_int32),
                ("lo", C.c_int32 * 3), ("hi", C.c_int32 * 3), ("n_morph", C.c_int32),
                ("morph_op", C.c_int32 * CHAIN_MAX_MORPH), ("morph_kw", C.c_int32 * CHAIN_MAX_MORPH),
                ("morph_kh", C.c_int32 * CHAIN_MAX_MORPH), ("morph_iter", C.c_int32 * CHAIN_MAX_MORPH),
                ("ccl", C.c_int32), ("numbering", C.c_int32), ("max_labels", C.c_int32)]


class ChainBuffers(C.Structure):
    _fields_ = [("bgr", C.c_void_p), ("threshed", C.c_void_p), ("cleaned", C.c_void_p), ("labels", C.c_void_p),
                ("stats", C.c_void_p), ("centroids", C.c_void_p), ("nlabels", C.c_void_p)]


class ContourDesc(C.Structure):
    _fields_ = [("source", C.c_int32), ("mode", C.c_int32), ("method", C.c_int32), ("max_contours", C.c_int32),
                ("max_points", C.c_int64)]


class ContourBuffers(C.Structure):
    _fields_ = [("info", C.c_void_p), ("counts", C.c_void_p), ("offsets", C.c_void_p), ("is_hole", C.c_void_p),
                ("points", C.c_void_p), ("features", C.c_void_p)]


_lib = None
_lock = threading.Lock()

_SIGS = {
    "vp_version": (C.c_int, []),
    "vp_strerror": (C.c_char_p, [C.c_int]),
    "vp_device_count": (C.c_int, []),
    "vp_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "vp_create": (C.c_void_p, [C.c_int]),
    "vp_destroy": (C.c_int, [C.c_void_p]),
    "vp_last_error": (C.c_char_p, [C.c_void_p]),
    "vp_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vp_get_stream": (C.c_void_p, [C.c_void_p]),
    "vp_synchronize": (C.c_int, [C.c_void_p]),
    "vp_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "vp_timer_start": (C.c_int, [C.c_void_p]),
    "vp_timer_stop": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "vp_get_tables": (C.c_int, [C.c_void_p] * 5),
    "vp_profile_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "vp_profile_end": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vp_profile_kernel_name": (C.c_char_p, [C.c_int]),
    "vp_cvt_color_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vp_color_balance_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vp_color_balance_last_folds": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "vp_color_balance_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vp_threshold_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_int, C.c_void_p]),
    "vp_otsu_threshold_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_int, C.c_void_p, C.c_void_p]),
    "vp_gaussian_blur_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]),
    "vp_resize_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vp_adaptive_threshold_mean_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, C.c_void_p]),
    "vp_canny_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]),
    "vp_warp_affine_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "vp_letterbox_u8_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vp_letterbox_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vp_nms_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vp_nms_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vp_cvt_bgr2lab_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vp_order_stats_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    "vp_inrange_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vp_inrange_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "vp_color_distance_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "vp_structuring_element": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vp_morph_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vp_ccl_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vp_find_contours_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "vp_draw_polyline_u8": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "vp_convex_hull_i32": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    "vp_min_area_rect_i32": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "vp_polygon_sums_i32": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "vp_draw_polylines_u8": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "vp_cvt_color_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vp_inrange_u8_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vp_morph_u8_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vp_find_contours_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "vp_chain_run": (C.c_int, [C.c_void_p, C.POINTER(ChainDesc), C.POINTER(ChainBuffers), C.c_int]),
    "vp_chain_run_host": (C.c_int, [C.c_void_p, C.POINTER(ChainDesc), C.POINTER(ChainBuffers), C.c_int]),
    "vp_chain_run_contours": (C.c_int, [C.c_void_p, C.POINTER(ChainDesc), C.POINTER(ChainBuffers), C.POINTER(ContourDesc),
                                        C.POINTER(ContourBuffers), C.c_int]),
    "vp_chain_run_contours_host": (C.c_int, [C.c_void_p, C.POINTER(ChainDesc), C.POINTER(ChainBuffers), C.POINTER(ContourDesc),
                                             C.POINTER(ContourBuffers), C.c_int]),
    "vp_chain_algorithmic_bytes": (C.c_uint64, [C.POINTER(ChainDesc), C.POINTER(ChainBuffers), C.c_int]),
    "vp_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "vp_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vp_host_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "vp_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vp_host_register": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "vp_host_unregister": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vp_feeder_start": (C.c_void_p, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vp_feeder_take": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "vp_feeder_release": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vp_feeder_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "vp_feeder_stop": (C.c_int, [C.c_void_p]),
    "vp_feeder_destroy": (C.c_int, [C.c_void_p]),
    "vp_post_d2h": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "vp_post_done": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vp_post_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vp_post_fence": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vp_post_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vp_inrange_u8_bits_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "vp_contours_last_heads": (C.c_uint, [C.c_void_p]),
    "vp_find_contours_bits_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int,
                                           C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "vp_memcpy_d2d_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "vp_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "vp_draw_polylines_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "vp_add_weighted_u8_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_double, C.c_size_t, C.c_void_p]),
    "vp_memcpy_h2d_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "vp_wait_uploads": (C.c_int, [C.c_void_p]),
    "vp_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
}


def _share_hip_runtime_with_torch():
    """PyTorch ships its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).  A process must end up on ONE runtime:
    with torch imported first libvp binds to torch's copy by soname; the other way round torch would load a second copy next to
    /opt/rocm's and see no devices.  So when torch is installed but not imported yet, its runtime is loaded first, by path —
    torch itself is not imported."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Loads libvp.so (once).  Raises VpError when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise VpError(f"{LIB_PATH} is missing: build it with `python cuauv-vision-pipeline_amd/build.py` "
                                  "(there is no CPU fallback)")
                _share_hip_runtime_with_torch()
                l = C.CDLL(LIB_PATH)
                for name, (res, args) in _SIGS.items():
                    fn = getattr(l, name)
                    fn.restype = res
                    fn.argtypes = args
                _lib = l
    return _lib


def exported_symbols():
    return sorted(_SIGS)


def check(rc, ctx=None):
    if rc != 0:
        l = lib()
        msg = l.vp_last_error(ctx)
        raise VpError(f"libvp: {l.vp_strerror(rc).decode()} ({rc}): {msg.decode() if msg else ''}")


class Context:
    """One libvp context = one HIP stream on one device (one per camera direction)."""

    def __init__(self, device=0):
        l = lib()
        self.handle = l.vp_create(int(device))
        if not self.handle:
            raise VpError("vp_create failed: " + l.vp_last_error(None).decode())
        self.device = device

    def close(self):
        if self.handle:
            pool = getattr(self, "_pool", None)      # device buffers of images that stayed in HBM (vision/devmat.py)
            if pool is not None:
                pool.drain()
            lib().vp_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle):
        check(lib().vp_set_stream(self.handle, stream_handle), self.handle)

    def set_option(self, option, value):
        check(lib().vp_set_option(self.handle, int(option), int(value)), self.handle)

    def synchronize(self):
        check(lib().vp_synchronize(self.handle), self.handle)

    def timer_start(self):
        check(lib().vp_timer_start(self.handle), self.handle)

    def timer_stop(self):
        ms = C.c_float()
        check(lib().vp_timer_stop(self.handle, C.byref(ms)), self.handle)
        return ms.value

    def profile_begin(self, max_records):
        check(lib().vp_profile_begin(self.handle, int(max_records)), self.handle)

    def profile_end(self):
        """-> {kernel name: (total_ms, launches)}"""
        ms = np.zeros(PROF_KERNELS, np.float64)
        cnt = np.zeros(PROF_KERNELS, np.int32)
        check(lib().vp_profile_end(self.handle, ptr(ms), ptr(cnt)), self.handle)
        return {lib().vp_profile_kernel_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(PROF_KERNELS) if cnt[i]}

    def chain_run(self, desc, bufs, n):
        check(lib().vp_chain_run(self.handle, C.byref(desc), C.byref(bufs), int(n)), self.handle)

    def chain_run_host(self, desc, bufs, n):
        check(lib().vp_chain_run_host(self.handle, C.byref(desc), C.byref(bufs), int(n)), self.handle)

    def chain_run_contours(self, desc, bufs, cdesc, cbufs, n):
        check(lib().vp_chain_run_contours(self.handle, C.byref(desc), C.byref(bufs), C.byref(cdesc), C.byref(cbufs), int(n)), self.handle)

    def chain_run_contours_host(self, desc, bufs, cdesc, cbufs, n):
        check(lib().vp_chain_run_contours_host(self.handle, C.byref(desc), C.byref(bufs), C.byref(cdesc), C.byref(cbufs), int(n)),
              self.handle)


_tls = threading.local()


def default_context(device=0):
    """Per-(thread, device) context: ModuleBase runs process() on a non-main thread (core/base.py:701-703) and starts a new one on
    every FRAMEWORK_DELETED retry.  The contexts hang off thread-local storage, so a thread that ends releases its contexts (streams,
    events, device workspace, staging memory) with it, and a recycled thread id can never inherit another thread's context."""
    ctxs = getattr(_tls, "ctxs", None)
    if ctxs is None:
        ctxs = _tls.ctxs = {}
    ctx = ctxs.get(device)
    if ctx is None or not ctx.handle:
        ctx = ctxs[device] = Context(device)
    return ctx


def pinned_empty(ctx, shape, dtype):
    """numpy array over page-locked host memory (freed when the array is garbage collected)."""
    import weakref
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dtype.itemsize
    p = C.c_void_p()
    check(lib().vp_host_alloc(ctx.handle, max(nbytes, 1), C.byref(p)), ctx.handle)
    buf = (C.c_ubyte * max(nbytes, 1)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    addr = p.value
    weakref.finalize(buf, lambda: lib().vp_host_free(None, addr))   # page-locked memory is not tied to a context, which may be gone by then
    return arr


def ptr(a):
    """Address of a numpy array for a `void*` argument, as an int (`.ctypes.data_as(c_void_p)` costs 3 us a time, this 1; eight of them per
    module call).  The caller keeps the array alive across the call - every call site passes a named local."""
    return a.ctypes.data if a is not None else None


def get_tables():
    gamma = np.zeros(256, np.uint16)
    cbrt = np.zeros(3072, np.uint16)
    sdiv = np.zeros(256, np.int32)
    hdiv = np.zeros(256, np.int32)
    labc = np.zeros(9, np.int32)
    check(lib().vp_get_tables(ptr(gamma), ptr(cbrt), ptr(sdiv), ptr(hdiv), ptr(labc)))
    return gamma, cbrt, sdiv, hdiv, labc


def make_chain_desc(width, height, color_mode, lo, hi, morph=(), ccl=1, numbering=CCL_BLOCK2X2, max_labels=256):
    """morph: sequence of (op, kw, kh[, iterations])."""
    d = ChainDesc()
    d.width, d.height, d.color_mode = int(width), int(height), int(color_mode)
    lo = list(np.broadcast_to(np.asarray(lo), (3,))) if np.ndim(lo) else [int(lo), 0, 0]
    hi = list(np.broadcast_to(np.asarray(hi), (3,))) if np.ndim(hi) else [int(hi), 255, 255]
    for c in range(3):
        d.lo[c] = int(lo[c])
        d.hi[c] = int(hi[c])
    if len(morph) > CHAIN_MAX_MORPH:
        raise ValueError("too many morphology ops")
    d.n_morph = len(morph)
    for i, m in enumerate(morph):
        d.morph_op[i], d.morph_kw[i], d.morph_kh[i] = int(m[0]), int(m[1]), int(m[2])
        d.morph_iter[i] = int(m[3]) if len(m) > 3 else 1
    d.ccl, d.numbering, d.max_labels = int(ccl), int(numbering), int(max_labels)
    return d


def make_contour_desc(source="cleaned", mode=0, method=2, max_contours=64, max_points=8192):
    d = ContourDesc()
    d.source = {"cleaned": 1, "threshed": 2}[source] if isinstance(source, str) else int(source)
    d.mode, d.method, d.max_contours, d.max_points = int(mode), int(method), int(max_contours), int(max_points)
    return d


def contour_arrays(alloc, n, cdesc):
    """(dict of numpy arrays, ContourBuffers) for n frames; alloc(shape, dtype) -> array (plain or pinned)."""
    mc, mp = int(cdesc.max_contours), int(cdesc.max_points)
    arrs = {"info": alloc((n, 2), np.int32), "counts": alloc((n, mc), np.int32), "offsets": alloc((n, mc), np.int32),
            "is_hole": alloc((n, mc), np.uint8), "points": alloc((n, mp, 2), np.int32), "features": alloc((n, mc, 8), np.float64)}
    b = ContourBuffers()
    for k, a in arrs.items():
        setattr(b, k, a.ctypes.data)
    return arrs, b


def contour_lists(arrs, cdesc):
    """Per frame: (tuple of (N,1,2) int32 contours in cv2's order = newest first, uint8 hole flags), or None for a frame whose
    contours did not fit the capacities (arrs["info"][f] then tells the sizes needed)."""
    out = []
    mc, mp = int(cdesc.max_contours), int(cdesc.max_points)
    for f in range(arrs["info"].shape[0]):
        k, npts = int(arrs["info"][f, 0]), int(arrs["info"][f, 1])
        if k > mc or npts > mp:
            out.append(None)
            continue
        cnt, off, pts = arrs["counts"][f], arrs["offsets"][f], arrs["points"][f]
        cs = tuple(pts[off[i]:off[i] + cnt[i]].reshape(-1, 1, 2).copy() for i in range(k - 1, -1, -1))
        out.append((cs, arrs["is_hole"][f, :k][::-1].copy()))
    return out


def contour_features(arrs, cdesc):
    """Per frame: (k, 8) float64 rows {m00, m10, m01, area, x, y, width, height} in the order of contour_lists (cv2's), or None
    where the frame exceeded the capacities."""
    out = []
    mc, mp = int(cdesc.max_contours), int(cdesc.max_points)
    for f in range(arrs["info"].shape[0]):
        k, npts = int(arrs["info"][f, 0]), int(arrs["info"][f, 1])
        out.append(None if (k > mc or npts > mp) else arrs["features"][f, :k][::-1].copy())
    return out
