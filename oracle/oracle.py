"""ctypes wrapper around oracle/liboracle.so — the CPU ORACLE.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from the product package (`vision.*` / libvp.so).  See the header of
vp_oracle.c for what it restates and why parity is "unpinned" (the reference has no tests
and cv2 is absent).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ERODE, DILATE, OPEN, CLOSE, GRADIENT = range(5)
MORPH_RECT, MORPH_CROSS, MORPH_ELLIPSE = 0, 1, 2
MODE_LAB, MODE_HSV, MODE_GRAY = 0, 1, 2

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)


def build(force=False):
    """Builds liboracle.so when it is missing or older than its sources.  Several processes may ask at once (the two-rank test):
    the build runs under a file lock into a temporary name and is renamed into place, so nobody ever loads a half-written file."""
    import fcntl
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("vp_oracle.c", "vp_oracle_balance.c", "vp_oracle_warp.c", "Makefile")]

    def stale():
        return not os.path.exists(so) or any(os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(so) for f in srcs)
    if not (force or stale()):
        return so
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or stale():
            tmp = os.path.join(_HERE, f"liboracle.{os.getpid()}.tmp.so")
            subprocess.check_call(["make", "-C", _HERE, "-B", f"TARGET={os.path.basename(tmp)}", os.path.basename(tmp)], stdout=subprocess.DEVNULL)
            os.replace(tmp, so)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_ccl_u8.restype = C.c_int
        _LIB.orc_chain_u8.restype = C.c_int
        _LIB.orc_morph_u8.restype = C.c_int
        _LIB.orc_structuring_element.restype = C.c_int
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _c(a, dtype=np.uint8):
    return np.ascontiguousarray(a, dtype=dtype)


def tables(variant=None):
    gamma = np.zeros(256, np.uint16)
    cbrt = np.zeros(3072, np.uint16)
    sdiv = np.zeros(256, np.int32)
    hdiv = np.zeros(256, np.int32)
    labc = np.zeros(9, np.int32)
    lib().orc_get_tables(_p(gamma, C.c_void_p), _p(cbrt, C.c_void_p), _p(sdiv, _i32p), _p(hdiv, _i32p), _p(labc, _i32p))
    if variant is not None:
        lib().orc_build_lab_tables(C.c_int(variant), _p(gamma, C.c_void_p), _p(cbrt, C.c_void_p))
    return gamma, cbrt, sdiv, hdiv, labc


def _cvt3(fn, bgr):
    bgr = _c(bgr)
    h, w, _ = bgr.shape
    out = np.empty_like(bgr)
    fn(_p(bgr, _u8p), C.c_size_t(w * 3), w, h, _p(out, _u8p), C.c_size_t(w * 3))
    return out


def bgr2lab(bgr):
    return _cvt3(lib().orc_bgr2lab_u8, bgr)


def bgr2hsv(bgr):
    return _cvt3(lib().orc_bgr2hsv_u8, bgr)


def bgr2ycrcb(bgr):
    return _cvt3(lib().orc_bgr2ycrcb_u8, bgr)


def bgr2hls(bgr):
    return _cvt3(lib().orc_bgr2hls_u8, bgr)


def hsv2bgr(hsv, variant=0):
    """cv2.cvtColor(COLOR_HSV2BGR), 8-bit; variant 0 = vector arithmetic form, 1 = scalar form (see vp_oracle_balance.c)."""
    hsv = _c(hsv)
    h, w, _ = hsv.shape
    out = np.empty_like(hsv)
    lib().orc_hsv2bgr_u8(_p(hsv, _u8p), C.c_size_t(w * 3), w, h, _p(out, _u8p), C.c_size_t(w * 3), int(variant))
    return out


def color_balance(bgr, equalize_rgb=True, rgb_contrast_correct=False, hsv_contrast_correct=True, hsi_contrast_correct=False,
                  rgb_extrema_clipping=True, adaptive_cast_correction=False, horizontal_blocks=1, vertical_blocks=1,
                  mean_mode=0, hsv_variant=0):
    """modules/color_balance.py:93-110 balance() -> process_frame (utils/color_correction/color_balance.cpp:343-780)."""
    out = _c(bgr).copy()
    h, w, _ = out.shape
    L = lib()
    L.orc_color_balance.restype = C.c_int
    rc = L.orc_color_balance(_p(out, _u8p), C.c_size_t(h), C.c_size_t(w), int(equalize_rgb), int(rgb_contrast_correct),
                             int(hsv_contrast_correct), int(hsi_contrast_correct), int(rgb_extrema_clipping),
                             int(adaptive_cast_correction), int(horizontal_blocks), int(vertical_blocks), int(mean_mode), int(hsv_variant))
    if rc != 0:
        raise ValueError(f"orc_color_balance: {rc}")
    return out


def gaussian_kernel_fixed(n, sigma=0.0):
    out = np.zeros(n, np.uint16)
    assert lib().orc_gaussian_kernel_fixed(int(n), C.c_double(sigma), _p(out, C.c_void_p)) == 0
    return out


def gaussian_blur(img, ksize, sigma1=0.0, sigma2=0.0):
    """cv2.GaussianBlur(img, (kw, kh), sigma1, sigma2) on uint8 images (bit-exact fixed-point path)."""
    img = _c(img)
    cn = 1 if img.ndim == 2 else img.shape[2]
    h, w = img.shape[:2]
    out = np.empty_like(img)
    L = lib()
    L.orc_gaussian_blur_u8.restype = C.c_int
    rc = L.orc_gaussian_blur_u8(_p(img, _u8p), w, h, cn, int(ksize[0]), int(ksize[1]), C.c_double(sigma1), C.c_double(sigma2), _p(out, _u8p))
    if rc != 0:
        raise ValueError(f"orc_gaussian_blur_u8: {rc}")
    return out


def rotation_matrix_2d(center, angle, scale=1.0):
    """cv2.getRotationMatrix2D(center, angle, scale) -> (2, 3) float64."""
    M = np.empty(6, np.float64)
    lib().orc_rotation_matrix_2d(C.c_double(center[0]), C.c_double(center[1]), C.c_double(angle), C.c_double(scale), M.ctypes.data_as(C.c_void_p))
    return M.reshape(2, 3)


def warp_affine(img, M, dsize, inverse_map=False, border="constant", value=0):
    """cv2.warpAffine(img, M, dsize, flags=INTER_LINEAR [| WARP_INVERSE_MAP], borderMode=BORDER_CONSTANT | BORDER_REPLICATE)."""
    img = _c(img)
    cn = 1 if img.ndim == 2 else img.shape[2]
    h, w = img.shape[:2]
    dw, dh = int(dsize[0]), int(dsize[1])
    out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.uint8)
    Md = np.ascontiguousarray(np.asarray(M, dtype=np.float64).reshape(6))
    cv = np.zeros(4, np.uint8)
    cv[:] = np.broadcast_to(np.asarray(value, dtype=np.uint8), (4,)) if np.ndim(value) == 0 else np.pad(np.asarray(value, np.uint8), (0, 4 - len(value)))
    L = lib()
    L.orc_warp_affine_u8.restype = C.c_int
    rc = L.orc_warp_affine_u8(_p(img, _u8p), w, h, cn, Md.ctypes.data_as(C.c_void_p), int(bool(inverse_map)), {"constant": 0, "replicate": 1}[border],
                              _p(cv, _u8p), _p(out, _u8p), dw, dh)
    if rc != 0:
        raise ValueError(f"orc_warp_affine_u8: {rc}")
    return out


def canny(img, t1, t2):
    """cv2.Canny(img, t1, t2) with the default 3x3 aperture and L1 gradient, uint8 images with 1..4 channels."""
    img = _c(img)
    cn = 1 if img.ndim == 2 else img.shape[2]
    h, w = img.shape[:2]
    out = np.empty((h, w), np.uint8)
    L = lib()
    L.orc_canny_u8.restype = C.c_int
    rc = L.orc_canny_u8(_p(img, _u8p), w, h, cn, C.c_double(t1), C.c_double(t2), _p(out, _u8p))
    if rc != 0:
        raise ValueError(f"orc_canny_u8: {rc}")
    return out


def adaptive_threshold_mean(img, max_value, inv, block, c):
    """cv2.adaptiveThreshold(img, max_value, ADAPTIVE_THRESH_MEAN_C, THRESH_BINARY_INV if inv else THRESH_BINARY, block, c)."""
    img = _c(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    L = lib()
    L.orc_adaptive_threshold_mean_u8.restype = C.c_int
    rc = L.orc_adaptive_threshold_mean_u8(_p(img, _u8p), w, h, C.c_double(max_value), int(bool(inv)), int(block), C.c_double(c), _p(out, _u8p))
    if rc != 0:
        raise ValueError(f"orc_adaptive_threshold_mean_u8: {rc}")
    return out


def bgr2gray(bgr):
    bgr = _c(bgr)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_bgr2gray_u8(_p(bgr, _u8p), C.c_size_t(w * 3), w, h, _p(out, _u8p), C.c_size_t(w))
    return out


def gray2bgr(gray):
    gray = _c(gray)
    h, w = gray.shape
    out = np.empty((h, w, 3), np.uint8)
    lib().orc_gray2bgr_u8(_p(gray, _u8p), C.c_size_t(w), w, h, _p(out, _u8p), C.c_size_t(w * 3))
    return out


def split3(img):
    img = _c(img)
    h, w, _ = img.shape
    ps = [np.empty((h, w), np.uint8) for _ in range(3)]
    lib().orc_split3_u8(_p(img, _u8p), C.c_size_t(w * 3), w, h, *[_p(p, _u8p) for p in ps])
    return tuple(ps)


def inrange(img, lo, hi):
    if img.dtype == np.float32:
        img = _c(img, np.float32)
        h, w = img.shape
        out = np.empty((h, w), np.uint8)
        lib().orc_inrange_f32(_p(img, _f32p), C.c_size_t(w * 4), w, h, C.c_float(lo), C.c_float(hi), _p(out, _u8p), C.c_size_t(w))
        return out
    img = _c(img)
    cn = 1 if img.ndim == 2 else img.shape[2]
    h, w = img.shape[:2]
    lo = np.ascontiguousarray(np.broadcast_to(np.asarray(lo, np.int64), (cn,)).astype(np.int32))
    hi = np.ascontiguousarray(np.broadcast_to(np.asarray(hi, np.int64), (cn,)).astype(np.int32))
    out = np.empty((h, w), np.uint8)
    lib().orc_inrange_u8(_p(img, _u8p), C.c_size_t(w * cn), w, h, cn, _p(lo, _i32p), _p(hi, _i32p), _p(out, _u8p), C.c_size_t(w))
    return out


def color_distance(planes, color, wts, skipmask=0):
    ps = [_c(p) for p in planes]
    h, w = ps[0].shape
    color = np.asarray(color, np.float32)
    wts = np.asarray(wts, np.float32)
    d2 = np.empty((h, w), np.float32)
    sq = np.empty((h, w), np.uint8)
    lib().orc_color_distance_u8(_p(ps[0], _u8p), _p(ps[1], _u8p), _p(ps[2], _u8p), w, h, _p(color, _f32p), _p(wts, _f32p),
                                int(skipmask), _p(d2, _f32p), _p(sq, _u8p))
    return d2, sq


def structuring_element(shape, kw, kh):
    out = np.empty((kh, kw), np.uint8)
    if lib().orc_structuring_element(shape, kw, kh, _p(out, _u8p)) != 0:
        raise ValueError("bad structuring element size")
    return out


def morph(op, img, kernel, iterations=1, anchor=(-1, -1), fast=False):
    img = _c(img)
    cn = 1 if img.ndim == 2 else img.shape[2]
    h, w = img.shape[:2]
    out = np.empty_like(img)
    if kernel is None:
        kp, kw, kh = None, 0, 0
    else:
        kernel = _c(kernel)
        kh, kw = kernel.shape
        kp = _p(kernel, _u8p)
    rc = lib().orc_morph_u8(op, _p(img, _u8p), w, h, cn, kp, kw, kh, anchor[0], anchor[1], iterations, _p(out, _u8p), int(fast))
    assert rc == 0
    return out


def ccl(mask, block=2, max_k=None, want_labels=True):
    mask = _c(mask)
    h, w = mask.shape
    if max_k is None:
        max_k = (h * w + 1) // 1 + 1
        max_k = min(max_k, ((h + 1) // 2) * ((w + 1) // 2) + 1)
    labels = np.empty((h, w), np.int32) if want_labels else None
    stats = np.empty((max_k, 5), np.int32)
    cent = np.empty((max_k, 2), np.float64)
    n = lib().orc_ccl_u8(_p(mask, _u8p), C.c_size_t(w), w, h, block, _p(labels, _i32p), _p(stats, _i32p), _p(cent, _f64p), max_k)
    k = min(n, max_k)
    return n, labels, stats[:k].copy(), cent[:k].copy()


def chain(bgr, mode, lo, hi, ops, kw=5, kh=5, block=2, max_k=4096, want_labels=True):
    """colour -> inRange -> rect morphology ops -> CCL (+stats).  Returns dict."""
    bgr = _c(bgr)
    h, w, _ = bgr.shape
    lo = np.ascontiguousarray(np.asarray(lo, np.int32))
    hi = np.ascontiguousarray(np.asarray(hi, np.int32))
    ops = np.ascontiguousarray(np.asarray(ops, np.int32))
    th = np.empty((h, w), np.uint8)
    cl = np.empty((h, w), np.uint8)
    labels = np.empty((h, w), np.int32) if (want_labels and block) else None
    stats = np.empty((max_k, 5), np.int32)
    cent = np.empty((max_k, 2), np.float64)
    n = lib().orc_chain_u8(_p(bgr, _u8p), w, h, mode, _p(lo, _i32p), _p(hi, _i32p), _p(ops, _i32p), len(ops), kw, kh, block,
                           _p(th, _u8p), _p(cl, _u8p), _p(labels, _i32p), _p(stats, _i32p), _p(cent, _f64p), max_k)
    k = min(n, max_k)
    return dict(threshed=th, cleaned=cl, labels=labels, nlabels=n, stats=stats[:k].copy(), centroids=cent[:k].copy())


RETR_EXTERNAL, RETR_LIST = 0, 1
CHAIN_APPROX_NONE, CHAIN_APPROX_SIMPLE = 1, 2


def find_contours(mask, mode=RETR_EXTERNAL, method=CHAIN_APPROX_SIMPLE, with_holes=False):
    """-> tuple of (N,1,2) int32 arrays in cv2.findContours order (newest first)."""
    mask = _c(mask)
    h, w = mask.shape
    L = lib()
    L.orc_find_contours.restype = C.c_int
    max_c, max_p = 1024, 1 << 16
    while True:
        pts = np.empty((max_p, 2), np.int32)
        counts = np.empty(max_c, np.int32)
        holes = np.empty(max_c, np.uint8)
        total = C.c_long(0)
        rc = L.orc_find_contours(_p(mask, _u8p), C.c_size_t(w), w, h, int(mode), int(method), _p(pts, _i32p), C.c_long(max_p),
                                 _p(counts, _i32p), _p(holes, _u8p), max_c, C.byref(total))
        if rc >= 0:
            break
        max_c = max(max_c, 2 * abs(rc)) if abs(rc) > max_c else max_c
        max_p = max(max_p, 2 * total.value)
    out, o = [], 0
    for k in range(rc):
        out.append(pts[o:o + counts[k]].reshape(-1, 1, 2).copy())
        o += counts[k]
    return (tuple(out), holes[:rc].copy()) if with_holes else tuple(out)


def contour_moments(contour):
    pts = np.ascontiguousarray(np.asarray(contour, np.int32).reshape(-1, 2))
    m = [C.c_double() for _ in range(4)]
    lib().orc_contour_moments(_p(pts, _i32p), len(pts), *[C.byref(v) for v in m])
    return dict(m00=m[0].value, m10=m[1].value, m01=m[2].value, area=m[3].value)
