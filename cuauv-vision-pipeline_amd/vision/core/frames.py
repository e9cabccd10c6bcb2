"""Frame copies of the runtime in page-locked memory.

The runtime hands every module its own writable copy of each frame (reference: core/base.py:765-768; the arrays read from the
camera_message_framework block view the library's buffer).  The first thing an accelerated module does with that copy is upload it:
from page-locked memory the 6.2 MB of a 1080p frame cross PCIe about twice as fast as from pageable memory (0.12 vs 0.25 ms), so the
copy is made into a pinned buffer.  The result is an ordinary writable numpy array; its buffer returns to a small pool when the last
reference to the array (or a view of it) is dropped, because hipHostMalloc / hipHostFree are far too slow to run per frame.

Without a device context on the calling thread (a CPU box; a module that never calls an accelerated operator; the first frame) or with
VP_PINNED_FRAMES=0 the copy is a plain numpy copy.
"""
import ctypes as C
import os
import threading
import weakref

import numpy as np

_enabled = os.environ.get("VP_PINNED_FRAMES", "1") != "0"
_lock = threading.Lock()
_free = {}            # size class -> [address]
_held = 0
_CAP_BYTES = 1 << 30
_broken = False


def _size_class(nbytes):
    n = max(int(nbytes), 4096)
    return (n + (1 << 20) - 1) & ~((1 << 20) - 1) if n > (1 << 20) else 1 << (n - 1).bit_length()


def _release(addr, cls):
    global _held
    from vision import _vp
    with _lock:
        if _held + cls <= _CAP_BYTES:
            _free.setdefault(cls, []).append(addr)
            _held += cls
            return
    try:
        _vp.lib().vp_host_free(None, addr)
    except Exception:
        pass


def pinned_like(shape, dtype):
    """Uninitialised writable array of the given shape in page-locked memory, or None when no device runtime is usable."""
    global _held, _broken
    if not _enabled or _broken:
        return None
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dtype.itemsize
    cls = _size_class(nbytes)
    addr = None
    with _lock:
        lst = _free.get(cls)
        if lst:
            addr = lst.pop()
            _held -= cls
    if addr is None:
        try:
            from vision import _vp
            # Never the first to touch the device: loading the HIP runtime and creating a context take a second or two when cold, and
            # this runs inside the runtime's frame loop.  Frames become page-locked once the thread has a context, i.e. from the
            # second frame of a module that uses the accelerated operators; a module that never does gets plain copies for ever.
            ctxs = getattr(_vp._tls, "ctxs", None) if _vp._lib is not None else None
            ctx = next((c for c in (ctxs or {}).values() if c.handle), None)
            if ctx is None:
                return None
            p = C.c_void_p()
            _vp.check(_vp.lib().vp_host_alloc(ctx.handle, cls, C.byref(p)), ctx.handle)
            addr = p.value
        except Exception:
            _broken = True                           # allocation failed: plain copies from now on
            return None
    buf = (C.c_ubyte * cls).from_address(addr)
    weakref.finalize(buf, _release, addr, cls)
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def copy_frame(view):
    """Fresh writable C-contiguous copy of `view` (pinned when possible)."""
    view = np.asarray(view)
    out = pinned_like(view.shape, view.dtype)
    if out is None:
        return np.array(view, copy=True)
    np.copyto(out, view)
    return out
