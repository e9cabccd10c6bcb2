"""Posts: the debug images a module publishes for the GUI, one shared-memory block per name.

Contract (reference core/base.py:846-876 `post`, :832-839 flush): `post(name, image, color_space)` takes the image AS IT IS at the call
(the reference copies it into a fresh uint8 array), a later post under the same name in the same iteration replaces the earlier one,
and after the handlers of an iteration every queued post is written into the block `module_<Module>_post%<idx>%<name>#<CS>` (`idx` =
position in the queue when the block is first made).  `--enable-performance` makes post() a no-op; posting is the default.

What is different inside: an image that lives in HBM (`vision.devmat.DeviceMat`) is not downloaded and memcpy'd - two host passes over
10 MB per red_buoy frame.  post() opens the block's next ring slot (cmf_write_begin: first sequence word bumped, readers keep being
served the last complete frame) and queues ONE copy by the GPU's copy engine from the image straight into the slot, on a stream of
its own beside the module's kernels (vp_post_d2h).  The flush commits the slot (metadata, second sequence word, uid, wake-up:
cmf_write_commit) once the copy's event has passed; a copy that is still on its way is committed by the next flush, or by the next
post() to the same block, so the copy of frame k's last post crosses PCIe under frame k + 1's Python.  No host instruction touches
the pixels.

Snapshot semantics without a snapshot: kernels queued after post() that only READ the image run beside the copy; anything that would
OVERWRITE it (an in-place draw, an upload after host-side writes) passes `DeviceMat._before_write`, where the pending post makes the
module's stream wait for the copy first.  The allocation itself is kept alive by the pending post until the copy is done.

`VP_DMA_POSTS=0` restores the download + write_frame path (what round 3 did for every post, and what host arrays still take).
"""
import os
import time
import weakref
from collections import OrderedDict
from typing import Callable, Dict, Optional

import numpy as np

from vision.devmat import DeviceMat
from vision.utils.helpers import as_mat

VALID_COLOR_SPACES = ("BGR", "RGB", "HSV", "LAB", "HLS", "YCRCB", "LUV", "GRAY")
_DMA_POSTS = os.environ.get("VP_DMA_POSTS", "1") != "0"


def _now_ms() -> int:
    return int(time.monotonic() * 1000)


class _DmaPost:
    """One image on its way into its block's open slot.  Doubles as a 'pending reader' of the image (`_pending` / `_force` is the
    protocol of DeviceMat._consumers): forcing it fences the module's stream behind the copy."""
    __slots__ = ("block", "ticket", "slot", "lane", "event", "ctx", "shape", "keep", "flushed", "_pending", "__weakref__")

    def __init__(self, block, slot, ticket, lane):
        self.block, self.slot, self.ticket, self.lane = block, slot, ticket, lane
        self.event = self.ctx = self.keep = self.shape = self._pending = None
        self.flushed = False

    def copy_from(self, image: DeviceMat):
        """Queues the copy of `image` as it is now into the slot (again, if an earlier post of this iteration is being replaced: the
        post stream runs its copies in order, the later one lands last)."""
        from vision import _vp
        import ctypes as C
        ctx = image._ctx
        src = image.dev_ptr                                     # launches the operator if the image was still deferred
        done = C.c_void_p()
        _vp.check(_vp.lib().vp_post_d2h(ctx.handle, self.lane, self.slot, src, image.nbytes, C.byref(done)), ctx.handle)
        self._release_event()
        self.event, self.ctx, self.shape, self.keep = done.value, ctx, image.shape, image._buf
        self._pending = True
        cs = image._consumers
        if len(cs) > 8:                                         # an image posted over and over (a static overlay): drop what has long been published
            cs[:] = [r for r in cs if (c := r()) is not None and c._pending is not None]
        cs.append(weakref.ref(self))

    def _force(self):
        """Something is about to overwrite the image: the module's stream waits for the copy (DeviceMat._before_write)."""
        if self._pending is not None:
            self._pending = None
            if self.event is not None and self.ctx.handle:
                from vision import _vp
                _vp.check(_vp.lib().vp_post_fence(self.ctx.handle, self.event), self.ctx.handle)

    def done(self) -> bool:
        from vision import _vp
        if self.event is None or not self.ctx.handle:
            return True
        rc = _vp.lib().vp_post_done(self.ctx.handle, self.event)
        if rc < 0:
            _vp.check(rc, self.ctx.handle)
        return rc == 1

    def wait(self):
        from vision import _vp
        if self.event is not None and self.ctx.handle:
            _vp.check(_vp.lib().vp_post_wait(self.ctx.handle, self.event), self.ctx.handle)

    def _release_event(self):
        if self.event is not None:
            from vision import _vp
            _vp.lib().vp_post_free(self.ctx.handle if self.ctx.handle else None, self.event)
            self.event = None
        self._pending = None
        self.keep = None

    def commit(self, stamp_ms: int):
        """The bytes are in the slot: publish it."""
        shape = self.shape
        self._release_event()
        self.block.commit_device_write(self.ticket, stamp_ms, shape, 1)

    def cancel(self):
        """Gives the slot up (the copy is waited for first: nothing may write into a slot that is closed)."""
        try:
            self.wait()
        finally:
            self._release_event()
            self.block.abort_device_write(self.ticket)


class PostQueue:
    """The post side of a module: `post()` during an iteration, `flush()` after its handlers, `drain()` before the blocks go away.

    open_block(block_name, idx, nbytes) -> the BlockAccessor of a post block, created on first use (ModuleManager.post_block);
    write_host(block_name, idx, stamp_ms, array): the reference's path for host arrays (ModuleManager.post)."""

    def __init__(self, open_block: Callable, write_host: Callable, enabled: bool = True):
        self._open_block, self._write_host, self.enabled = open_block, write_host, enabled
        self.queue: "OrderedDict[str, tuple]" = OrderedDict()   # name -> (ndarray | _DmaPost, colour space)
        self._open: Dict[str, _DmaPost] = {}                    # block name -> post whose slot is open (queued or flushed, not committed)
        self._lanes: Dict[str, int] = {}                        # block name -> post stream of its copies (a block keeps its lane)
        self.dma_posts = self.host_posts = 0                    # counters for tools and tests

    def __len__(self):
        return len(self.queue)

    def post(self, name: str, image, color_space: str = "BGR"):
        if not self.enabled:
            return
        if "%" in name:
            raise RuntimeError("Cannot have % in name")
        image = as_mat(image)
        color_space = color_space.upper()
        if color_space not in VALID_COLOR_SPACES:
            color_space = "BGR"
        key = f"{name}#{color_space}"
        entry = None
        if _DMA_POSTS and isinstance(image, DeviceMat) and image.dtype == np.uint8 and 1 <= image.ndim <= 3 and image.nbytes > 0 \
                and image._dev_ok and image._ctx.handle:
            entry = self._post_device(name, key, image)
        if entry is None:
            self._settle(key, cancel_queued=True)               # a host write cannot share the block with an open slot
            if isinstance(image, DeviceMat) and image.dtype == np.uint8:
                entry = image.host_copy()                       # one download into an array of its own; the image stays usable on the device
            else:
                entry = np.array(image, np.uint8, copy=True, order="C", ndmin=1)
        old = self.queue.get(name)
        if old is not None and isinstance(old[0], _DmaPost) and old[0] is not entry:
            self._settle(f"{name}#{old[1]}", cancel_queued=True)    # same name, another colour space: the earlier post is dropped
        self.queue[name] = (entry, color_space)

    def _settle(self, key: str, cancel_queued: bool):
        """Closes whatever slot is open on block `key`: a post already flushed is published (its copy is waited for), one that is
        still queued in this iteration is being replaced and is given up."""
        p = self._open.pop(key, None)
        if p is None:
            return
        if p.flushed:
            p.wait()
            p.commit(_now_ms())
        elif cancel_queued:
            p.cancel()

    def _post_device(self, name: str, key: str, image: DeviceMat) -> Optional[_DmaPost]:
        p = self._open.get(key)
        if p is not None and not p.flushed:                     # replaced within the iteration: same slot, one more copy behind the first
            p.copy_from(image)
            return p
        self._settle(key, cancel_queued=False)                  # last iteration's post to this block, still on its way: publish it first
        idx = list(self.queue).index(name) if name in self.queue else len(self.queue)
        try:
            block = self._open_block(key, idx, image.nbytes)
        except RuntimeError:
            return None                                         # no block to be had now: the host path reports it at the flush, as the reference does
        opened = block.begin_device_write(image._ctx, image.nbytes)
        if opened is None:
            return None
        p = _DmaPost(block, opened[0], opened[1], self._lanes.setdefault(key, len(self._lanes)))
        try:
            p.copy_from(image)
        except Exception:
            block.abort_device_write(p.ticket)
            raise
        self._open[key] = p
        return p

    def flush(self, wait: bool = False):
        """End of an iteration: host posts are written now; posts by DMA are committed if their copy has arrived (or `wait`), the rest
        stay open and are committed by a later flush()."""
        if self.queue:
            for idx, (name, (data, color_space)) in enumerate(self.queue.items()):
                if isinstance(data, _DmaPost):
                    data.flushed = True
                    self.dma_posts += 1
                else:
                    self._write_host(f"{name}#{color_space}", idx, _now_ms(), data)
                    self.host_posts += 1
            self.queue.clear()
        if self._open:
            for key, p in list(self._open.items()):
                if p.flushed and (wait or p.done()):
                    if wait:
                        p.wait()
                    del self._open[key]
                    p.commit(_now_ms())

    def pending(self) -> int:
        """Posts flushed but not yet published (their copies are still crossing)."""
        return sum(1 for p in self._open.values() if p.flushed)

    def drain(self):
        """Publishes everything that was flushed, gives up what was only queued; after this no copy targets any block."""
        for key, p in list(self._open.items()):
            del self._open[key]
            try:
                if p.flushed:
                    p.wait()
                    p.commit(_now_ms())
                else:
                    p.cancel()
            except Exception:
                pass
        self.queue.clear()
