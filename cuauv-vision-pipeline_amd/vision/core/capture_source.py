"""Capture-source harness: generator threads that publish frames into shared-memory blocks.

Contract of the reference core/capture_source.py (FpsLimiter :23-67, CaptureSource :70-238): a capture "UDL" is a generator
`udl(fps_limiter, args)` that yields `(direction, time_ms, image_or_planes[, plane_names])`; the harness creates the block of a
direction on its first frame, sized for that frame; `run_event_loop()` blocks until SIGINT, until a UDL is exhausted or until one
raises - any of which stops every other UDL of the source.  `for t_ms in limiter.rate(fps)` paces a loop and ends with the source.
"""
import signal
import threading
import time
import traceback
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from vision.core.base import auvlog
from vision.core.bindings.camera_message_framework import BlockAccessor


class FpsLimiter:
    """Pacing iterator: every step sleeps whatever is left of 1/fps since the previous step and returns monotonic milliseconds;
    rate(0) / rate(None) never sleeps.  Falling behind is logged once, catching up again once."""

    def __init__(self, name: str, quit_flag: threading.Event):
        self._say = getattr(auvlog.vision.capture_source.fps_limiter, name or "anonymous")
        self._quit_flag = quit_flag
        self._period = 0.0
        self._mark = 0.0
        self._behind = False

    def rate(self, fps: Optional[int]):
        fps = fps or 0
        assert fps >= 0, "given negative fps which is invalid"
        self._period = 1.0 / fps if fps else 0.0
        return self

    def __iter__(self):
        self._mark = time.monotonic()
        return self

    def __next__(self) -> int:
        if self._quit_flag.is_set():
            raise StopIteration
        left = self._period - (time.monotonic() - self._mark)
        if (left <= 0) != self._behind:                  # the state changed: say so, once
            self._behind = left <= 0
            self._say("too slow! dropped frames!" if self._behind else "recovered!", True)
        time.sleep(max(left, 0.0))
        self._mark = time.monotonic()
        return int(self._mark * 1000)


def _planes_of(direction: str, img) -> Tuple[np.ndarray, ...]:
    """What a UDL yielded as the frame: one array, or a non-empty sequence of arrays."""
    if isinstance(img, np.ndarray):
        return (img,)
    if not isinstance(img, Sequence):
        raise TypeError(f"unsupported frame type {type(img)} for direction '{direction}'")
    if not img:
        raise ValueError("capture source yielded an empty frame sequence")
    bad = next((i for i, p in enumerate(img) if not isinstance(p, np.ndarray)), None)
    if bad is not None:
        raise TypeError(f"frame at index {bad} for direction '{direction}' is not an ndarray")
    return tuple(img)


class CaptureSource:
    """Subclass (or instantiate), register UDLs, then run_event_loop()."""

    def __init__(self):
        self._say = getattr(auvlog.vision.capture_source, type(self).__name__)
        self._blocks: Dict[str, BlockAccessor] = {}
        self._workers: List[threading.Thread] = []
        self._quit_flag = threading.Event()

    # -- threads ------------------------------------------------------------------------------------------------------------
    def _spawn(self, label: str, run: Callable[[], None], stop_on_return: bool):
        """A worker that stops the whole source when it raises - and, for a capture UDL (`stop_on_return`), also when it simply
        ends: an exhausted capture stops the source, a logical UDL that returns (one-shot hardware set-up) does not
        (core/capture_source.py:113-127 sets the flag in its `except` only)."""
        def guarded():
            try:
                run()
            except Exception:
                self._say(f"Caught exception in {label} printing stack trace and unwinding ...")
                traceback.print_exc()
                self._quit_flag.set()
            else:
                if stop_on_return:
                    self._quit_flag.set()
        self._workers.append(threading.Thread(target=guarded, name=f"capture-{label}"))

    def register_logical_udl(self, udl: Callable[[FpsLimiter, Tuple[Any, ...]], None], args: Tuple[Any, ...] = ()):
        """A UDL that publishes nothing itself (it steers hardware, watches a flag ...)."""
        self._spawn("logical udl", lambda: udl(FpsLimiter("", self._quit_flag), args), stop_on_return=False)

    def register_capture_udl(self, name: str, udl, args: Tuple[Any, ...] = ()):
        def pump():
            self._say(f"starting capture udl '{name}'", True)
            for item in udl(FpsLimiter(name, self._quit_flag), args):
                if not isinstance(item, tuple):
                    raise RuntimeError(f"capture UDL '{name}' yielded unsupported type {type(item)}")
                if not 3 <= len(item) <= 4:
                    raise RuntimeError(f"capture UDL '{name}' yielded tuple of unexpected length {len(item)}")
                self._send(*item)
            told_to = self._quit_flag.is_set()
            self._say(f"capture udl '{name}' stopped as a result of another stop signal" if told_to else f"capture udl '{name}' exhausted", True)
        self._spawn(name, pump, stop_on_return=True)

    def run_event_loop(self):
        if threading.current_thread() is threading.main_thread():
            def on_sigint(*_):
                print("\n\nCtrl-C Caught")
                self._quit_flag.set()
            signal.signal(signal.SIGINT, on_sigint)
        for w in self._workers:
            w.start()
        while not self._quit_flag.wait(0.1):
            pass
        for w in self._workers:
            w.join()
        self._say("graceful shut down", True)

    # -- publishing -----------------------------------------------------------------------------------------------------------
    def _send(self, direction: str, acquisition_time: int, img, names: Optional[Sequence[str]] = None):
        planes = _planes_of(direction, img)
        size = sum(int(p.nbytes) for p in planes)
        if size <= 0:
            raise ValueError(f"total serialized size for direction '{direction}' must be positive")
        if names is not None and len(names) != len(planes):
            raise ValueError(f"direction '{direction}' provided {len(planes)} planes but {len(names)} names")
        block = self._blocks.get(direction)
        if block is None:                                  # first frame of the direction: the block is made to its size
            block = self._blocks[direction] = BlockAccessor(direction, max_entry_size_bytes=size).__enter__()
        if names is not None:
            block.write_frame(acquisition_time, tuple(zip(names, planes)))
        else:
            block.write_frame(acquisition_time, planes[0] if len(planes) == 1 else planes)

    def close(self):
        """Deletes the blocks this source created (readers see FRAMEWORK_DELETED)."""
        blocks, self._blocks = self._blocks, {}
        for block in blocks.values():
            block.__exit__(None, None, None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
