"""Images that stay in HBM between operator calls.

The reference's modules call one thin cv2 wrapper after another on numpy arrays (modules/red_buoy.py:21-38).  A literal
mirror uploads and downloads every image at every call: at 1080p `bgr_to_lab` alone brings 12.4 MB back of which the module
reads nothing.  `DeviceMat` is what the `vision.utils` mirror returns instead: an array-like whose data lives on the device and
reaches host memory only when Python looks at it.

  * the next `vision.utils` / cv2-facade call that receives it reads the device copy in place (no transfer);
  * anything else — numpy functions, indexing, `.copy()`, `np.asarray`, `self.post(...)`, a C extension asking for the buffer —
    materialises a host copy once (one D2H), and from then on the object behaves like the numpy array it wraps: writable, owned
    by the caller, mutated in place by `draw_contours`-style code (modules/red_buoy.py:39);
  * a host copy that may have been written to (handed out writable, indexed for assignment, used as `out=`) invalidates the
    device copy, so a later operator call uploads the host data again — results never depend on which side was used.

  * an operator whose input is such an image and whose result is one (morphology) does not even launch until the result is needed
    (`DeviceMat.deferred`): by the next operator, by the host, or - before the input changes - by a host-side write to the input.
    A module that computes a cleaned mask only to post it (modules/red_buoy.py:26-31) then pays nothing for it when posts are off
    (`--enable-performance`, core/base.py:846-876).  `VP_DEFER=0` launches every operator at its call.

It is not an `np.ndarray` subclass (numpy offers no hook on raw buffer reads, so a subclass could not be lazy); code that
insists on `isinstance(x, np.ndarray)` can call `np.asarray(x)`.  `VP_LAZY=0` (or `set_lazy(False)`) makes the mirror return
plain numpy arrays, as in round 1.
"""
import ctypes as C
import os
import threading
import weakref

import numpy as np

from vision import _vp

_lazy = os.environ.get("VP_LAZY", "1") != "0"
_defer = os.environ.get("VP_DEFER", "1") != "0"


def set_defer(on: bool):
    global _defer
    _defer = bool(on)


def defer_enabled() -> bool:
    return _defer and _lazy


def set_lazy(on: bool):
    global _lazy
    _lazy = bool(on)


def lazy_enabled() -> bool:
    return _lazy


def _prod(shape):
    """Number of elements of a shape tuple (np.prod costs 2.5 us a call, and an image asks a dozen times per operator chain)."""
    n = 1
    for v in shape:
        n *= v
    return n


class _Pool:
    """Free lists of device buffers of one context, by size class (hipMalloc / hipFree synchronise: never on the per-frame path).
    Everything on a context runs on its one stream, so a buffer released by the garbage collector can be handed to the next
    operator at once: the kernels that still read it were enqueued earlier."""
    CAP_BYTES = 2 << 30

    def __init__(self, ctx):
        self.ctx = weakref.ref(ctx)
        self.free = {}
        self.held = 0
        self.lock = threading.Lock()

    @staticmethod
    def size_class(nbytes):
        n = max(int(nbytes), 256)
        return (n + 65535) & ~65535 if n > 65536 else 1 << (n - 1).bit_length()

    def take(self, nbytes):
        cls = self.size_class(nbytes)
        with self.lock:
            lst = self.free.get(cls)
            if lst:
                self.held -= cls
                return lst.pop(), cls
        ctx = self.ctx()
        p = C.c_void_p()
        _vp.check(_vp.lib().vp_dev_alloc(ctx.handle, cls, C.byref(p)), ctx.handle)
        return p.value, cls

    def give(self, ptr, cls):
        ctx = self.ctx()
        if ctx is None or not ctx.handle:
            _vp.lib().vp_dev_free(None, ptr)         # the context is gone; its device memory is not (plain hipMalloc): free it here
            return
        with self.lock:
            if self.held + cls <= self.CAP_BYTES:
                self.free.setdefault(cls, []).append(ptr)
                self.held += cls
                return
        _vp.lib().vp_dev_free(ctx.handle, ptr)

    def drain(self):
        ctx = self.ctx()
        with self.lock:
            ptrs = [p for lst in self.free.values() for p in lst]
            self.free.clear()
            self.held = 0
        if ctx is not None and ctx.handle:
            for p in ptrs:
                _vp.lib().vp_dev_free(ctx.handle, p)


def pool_of(ctx):
    p = getattr(ctx, "_pool", None)
    if p is None:
        p = ctx._pool = _Pool(ctx)
    return p


class _DevBuf:
    """One device allocation; returns to its pool when the last DeviceMat (or view bookkeeping) drops it."""
    __slots__ = ("ptr", "cls", "pool", "__weakref__")

    def __init__(self, ctx, nbytes):
        self.pool = pool_of(ctx)
        self.ptr, self.cls = self.pool.take(nbytes)

    def __del__(self):
        try:
            self.pool.give(self.ptr, self.cls)
        except Exception:
            pass


class DeviceMat:
    """(h, w) or (h, w, c) image, tightly packed, whose authoritative copy may be on the device (`_dev_ok`), on the host
    (`_host` is not None and `_host_ok`), or both."""
    __array_priority__ = 100.0
    __slots__ = ("_ctx", "_buf", "_off", "_shape", "_dtype", "_host", "_dev_ok", "_escaped", "binary", "_pending", "_consumers", "_bits", "__weakref__")

    def __init__(self, ctx, shape, dtype=np.uint8, binary=False):
        self._ctx = ctx
        self._shape = tuple(int(s) for s in shape)
        self._dtype = np.dtype(dtype)
        self._buf = _DevBuf(ctx, _prod(self._shape) * self._dtype.itemsize)
        self._off = 0                       # byte offset of this image inside its device allocation (planes of one frame share one)
        self._host = None
        self._dev_ok = True
        self._escaped = False               # a writable alias of the host copy is out: the device copy can go stale without notice
        self.binary = bool(binary)          # known to hold only 0 / 255 (a mask made by this library)
        self._pending = None
        self._consumers = []                # shared with every reshaped() alias: pending operators that read this image
        self._bits = None                   # a mask's bit-packed form, made with it (range_threshold); gone with the first write

    @classmethod
    def deferred(cls, ctx, shape, dtype, binary, inputs, run):
        """Result of an operator between device images that has not been launched yet: `run(out)` launches it into `out.dev_ptr`
        (the inputs' `dev_ptr` are valid then).  It runs when the result is first needed, at the latest just before one of `inputs` is
        handed to the host for writing - so the result is always the one an immediate launch would have given."""
        m = object.__new__(cls)
        m._ctx, m._shape, m._dtype = ctx, tuple(int(s) for s in shape), np.dtype(dtype)
        m._buf, m._off, m._host, m._dev_ok, m._escaped, m.binary, m._consumers, m._bits = None, 0, None, True, False, bool(binary), [], None
        m._pending = run
        for x in inputs:
            if len(x._consumers) > 8:
                x._consumers[:] = [r for r in x._consumers if (c := r()) is not None and c._pending is not None]
            x._consumers.append(weakref.ref(m))
        return m

    @classmethod
    def over_buffer(cls, ctx, buf, offset, shape, dtype):
        """An image at `offset` bytes inside device allocation `buf` (a _DevBuf the images share): the planes of one frame that was
        moved out of its ring slot with one copy (vision.core.bindings.camera_message_framework.BlockAccessor.read_frame_device)."""
        m = object.__new__(cls)
        m._ctx, m._shape, m._dtype = ctx, tuple(int(s) for s in shape), np.dtype(dtype)
        m._buf, m._off, m._host, m._dev_ok, m._escaped, m.binary, m._consumers, m._pending, m._bits = buf, int(offset), None, True, False, False, [], None, None
        return m

    def _force(self):
        run = self._pending
        if run is not None:
            self._pending = None
            if not self._ctx.handle:
                raise _vp.VpError("the context that owns this image was closed before the image was computed")
            self._buf = _DevBuf(self._ctx, _prod(self._shape) * self._dtype.itemsize)
            run(self)

    def _before_write(self):
        """Pending operators that read this image run before its contents can change (host-side writes and in-place device writes
        alike; the list is shared by every reshaped() alias of the image)."""
        self._bits = None                        # the bit plane describes the contents as they were
        cs = self._consumers
        if cs:
            pending = cs[:]
            del cs[:]
            for r in pending:
                c = r()
                if c is not None:
                    c._force()

    # ---- what the operator wrappers use ------------------------------------------------------------------------------------
    @property
    def dev_ptr(self):
        if self._pending is not None:
            self._force()
        return self._buf.ptr + self._off

    @property
    def host_escaped(self):
        """A writable alias of the host copy has been handed out: writes through it are invisible to this object."""
        return self._escaped

    def device_valid_for(self, ctx):
        return self._dev_ok and ctx is self._ctx and bool(ctx.handle)

    @classmethod
    def from_host(cls, ctx, arr, binary=False, pending=None):
        """Uploads a packed host array; the host array is NOT kept (the caller may go on mutating it).
        pending: a list - the copy is only enqueued and the packed array appended to the list; the caller must `finish_uploads(ctx,
        pending)` before it hands control back to code that could change `arr` (an operator sets up its results and launches in
        between, while the copy crosses PCIe)."""
        arr = np.ascontiguousarray(arr)
        m = cls(ctx, arr.shape, arr.dtype, binary)
        if pending is None:
            _vp.check(_vp.lib().vp_memcpy_h2d(ctx.handle, m._buf.ptr, arr.ctypes.data, arr.nbytes), ctx.handle)
        else:
            _vp.check(_vp.lib().vp_memcpy_h2d_async(ctx.handle, m._buf.ptr, arr.ctypes.data, arr.nbytes), ctx.handle)
            pending.append(arr)                      # (a packed temporary must outlive the copy)
        return m

    def reshaped(self, shape):
        """Same data under another shape (e.g. (h, w, 1) -> (h, w)); shares the device buffer, copies nothing."""
        self._force()
        m = object.__new__(DeviceMat)
        m._ctx, m._buf, m._off, m._dtype, m._dev_ok, m.binary = self._ctx, self._buf, self._off, self._dtype, self._dev_ok, self.binary
        m._pending, m._consumers, m._escaped = None, self._consumers, self._escaped
        m._bits = None                           # (an alias is not told when the image is written to: it does not inherit the bit plane)
        m._shape = tuple(int(x) for x in shape)
        m._host = None if self._host is None else self._host.reshape(m._shape)
        if not self._dev_ok and m._host is None:
            raise RuntimeError("image has neither a valid device nor a host copy")
        return m

    def host(self, writable=True, escape=True):
        """The host copy (one D2H the first time).  Handing it out writable makes it the authoritative copy.  escape=False: the caller
        writes now and keeps no alias (item assignment, in-place operators, `out=`): the next operator uploads once and the device
        copy is trusted again."""
        if self._pending is not None:
            self._force()
        if writable:
            self._before_write()
        if self._host is None:
            out = np.empty(self._shape, self._dtype)
            ctx = self._ctx
            if not ctx.handle:
                raise _vp.VpError("the context that owns this image was closed before the image was read")
            _vp.check(_vp.lib().vp_memcpy_d2h(ctx.handle, out.ctypes.data, self._buf.ptr + self._off, out.nbytes), ctx.handle)
            self._host = out
        if writable:
            # From here on the caller may hold an alias (a view, the array itself) and write through it at any later time without this
            # object noticing: the device copy is never trusted again, every operator re-uploads the host data (refresh_device).
            self._dev_ok = False
            self._escaped = self._escaped or escape
            self.binary = False
            return self._host
        v = self._host.view()
        v.flags.writeable = False
        return v

    def host_copy(self):
        """A fresh host array with the current contents that the caller owns (self.post): straight from the device when the device
        copy is the valid one - nothing is cached and the device copy stays valid."""
        if self._pending is not None:
            self._force()
        if self._dev_ok and self._host is None and self._ctx.handle:
            out = np.empty(self._shape, self._dtype)
            _vp.check(_vp.lib().vp_memcpy_d2h(self._ctx.handle, out.ctypes.data, self._buf.ptr + self._off, out.nbytes), self._ctx.handle)
            return out
        return self.host(writable=False).copy()

    def refresh_device(self, ctx):
        """Device copy of the current contents on `ctx` (re-uploads after host-side writes or a change of context)."""
        if self._pending is not None:
            if ctx is self._ctx:
                return None                      # not launched yet: `dev_ptr` launches it, on this very context
            self._force()
        if self.device_valid_for(ctx):
            return self._buf.ptr + self._off
        h = self.host(writable=False)
        if ctx is not self._ctx:
            self._ctx = ctx
            self._buf = _DevBuf(ctx, h.nbytes)
            self._off = 0
        _vp.check(_vp.lib().vp_memcpy_h2d(ctx.handle, self._buf.ptr + self._off, h.ctypes.data, h.nbytes), ctx.handle)
        self._dev_ok = not self._escaped     # with a writable alias out, the upload is good for this one operator only
        return self._buf.ptr + self._off

    # ---- array protocol -----------------------------------------------------------------------------------------------------------
    shape = property(lambda self: self._shape)
    dtype = property(lambda self: self._dtype)
    ndim = property(lambda self: len(self._shape))
    size = property(lambda self: _prod(self._shape))
    nbytes = property(lambda self: _prod(self._shape) * self._dtype.itemsize)
    itemsize = property(lambda self: self._dtype.itemsize)

    def __len__(self):
        return self._shape[0]

    def __array__(self, dtype=None, copy=None):
        h = self.host(writable=True)
        if dtype is not None and np.dtype(dtype) != h.dtype:
            return h.astype(dtype)
        return h.copy() if copy else h

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        def conv(x, w):
            return x.host(writable=w) if isinstance(x, DeviceMat) else x
        ins = tuple(conv(x, False) for x in inputs)
        if "out" in kwargs:
            kwargs["out"] = tuple((x.host(writable=True, escape=False) if isinstance(x, DeviceMat) else x) for x in kwargs["out"])
        return getattr(ufunc, method)(*ins, **kwargs)

    def __array_function__(self, func, types, args, kwargs):
        def conv(x):
            if isinstance(x, DeviceMat):
                return x.host(writable=True)
            if isinstance(x, (list, tuple)):
                return type(x)(conv(y) for y in x)
            if isinstance(x, dict):
                return {k: conv(v) for k, v in x.items()}
            return x
        return func(*conv(args), **conv(kwargs))

    def __getitem__(self, key):
        return self.host(writable=True)[key]         # views of the host copy can be written through

    def __setitem__(self, key, value):
        if isinstance(value, DeviceMat):
            value = value.host(writable=False)
        self.host(writable=True, escape=False)[key] = value

    def __iter__(self):
        return iter(self.host(writable=True))

    def __getattr__(self, name):                     # everything else an ndarray has: .copy(), .astype(), .sum(), .T, .ctypes, .flags ...
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.host(writable=True), name)

    def __repr__(self):
        where = "not computed yet" if self._pending is not None else "device" if self._host is None else ("host+device" if self._dev_ok else "host")
        return f"DeviceMat(shape={self._shape}, dtype={self._dtype}, on={where})"

    def __bool__(self):
        return bool(self.host(writable=False))

    def __eq__(self, other):
        return np.equal(self, other)

    def __ne__(self, other):
        return np.not_equal(self, other)

    __hash__ = None


def _binop(name, ufunc, swap=False):
    def f(self, other):
        return ufunc(other, self) if swap else ufunc(self, other)
    f.__name__ = name
    return f


def _iop(name, ufunc):
    def f(self, other):
        h = self.host(writable=True, escape=False)
        ufunc(h, other.host(writable=False) if isinstance(other, DeviceMat) else other, out=h)
        return self
    f.__name__ = name
    return f


for _n, _u in (("add", np.add), ("sub", np.subtract), ("mul", np.multiply), ("truediv", np.true_divide), ("floordiv", np.floor_divide),
               ("mod", np.remainder), ("pow", np.power), ("and", np.bitwise_and), ("or", np.bitwise_or), ("xor", np.bitwise_xor),
               ("lshift", np.left_shift), ("rshift", np.right_shift), ("matmul", np.matmul)):
    setattr(DeviceMat, f"__{_n}__", _binop(f"__{_n}__", _u))
    setattr(DeviceMat, f"__r{_n}__", _binop(f"__r{_n}__", _u, swap=True))
    if _n != "matmul":
        setattr(DeviceMat, f"__i{_n}__", _iop(f"__i{_n}__", _u))
for _n, _u in (("lt", np.less), ("le", np.less_equal), ("gt", np.greater), ("ge", np.greater_equal)):
    setattr(DeviceMat, f"__{_n}__", _binop(f"__{_n}__", _u))
DeviceMat.__neg__ = lambda self: np.negative(self)
DeviceMat.__pos__ = lambda self: np.positive(self)
DeviceMat.__abs__ = lambda self: np.absolute(self)
DeviceMat.__invert__ = lambda self: np.invert(self)


def finish_uploads(ctx, pending):
    """Returns once the copies enqueued with from_host(..., pending=pending) have read their host arrays."""
    if pending:
        try:
            _vp.check(_vp.lib().vp_wait_uploads(ctx.handle), ctx.handle)
        finally:
            del pending[:]


def to_host(x):
    """np.ndarray for either kind (the caller may write to it)."""
    return x.host(writable=True) if isinstance(x, DeviceMat) else x


def to_host_readonly(x):
    """Host data for reading only: does not give up the device copy (self.post, drawing sources, comparisons)."""
    return x.host(writable=False) if isinstance(x, DeviceMat) else x
