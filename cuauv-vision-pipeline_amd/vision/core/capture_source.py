"""Capture-source harness: user-defined generator threads that publish frames into CMF blocks.

Mirror of the reference core/capture_source.py (FpsLimiter :23-67, CaptureSource :70-238): same method names
and yield protocol — a capture UDL yields (direction, time_ms, image_or_planes[, plane_names]) and the harness
creates the block on first use, sized for that first frame."""
import signal
import threading
import time
import traceback
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

from numpy import ndarray

from vision.core.base import auvlog
from vision.core.bindings.camera_message_framework import BlockAccessor


class FpsLimiter:
    """Iterator that paces a loop: `for t_ms in limiter.rate(30): ...` yields monotonic milliseconds and stops when
    the quit flag is set.  rate(0 / None) does not sleep."""

    def __init__(self, name: str, quit_flag: threading.Event):
        self._logger = getattr(auvlog.vision.capture_source.fps_limiter, name or "anonymous")
        self._quit_flag = quit_flag
        self._slow = False
        self._fps, self._target, self._last_time = 0, 0.0, 0.0

    def rate(self, fps: Optional[int]):
        fps = fps if fps else 0
        assert fps >= 0, "given negative fps which is invalid"
        self._fps = fps
        self._target = 1.0 / fps if fps > 0 else 0.0
        self._last_time = 0.0
        return self

    def __iter__(self):
        self._last_time = time.monotonic()
        return self

    def __next__(self):
        if self._quit_flag.is_set():
            raise StopIteration
        elapsed = time.monotonic() - self._last_time
        pause = 0.0
        if elapsed < self._target:
            if self._slow:
                self._slow = False
                self._logger("recovered!", True)
            pause = self._target - elapsed
        elif not self._slow:
            self._slow = True
            self._logger("too slow! dropped frames!", True)
        time.sleep(pause)
        self._last_time = time.monotonic()
        return int(self._last_time * 1000)


class CaptureSource:
    """Subclass, register UDLs, then run_event_loop() (blocks until SIGINT or until a UDL ends / raises)."""

    def __init__(self):
        self._logger = getattr(auvlog.vision.capture_source, self.__class__.__name__)
        self._frameworks: Dict[str, BlockAccessor] = {}
        self._threads: List[threading.Thread] = []
        self._quit_flag = threading.Event()

    def run_event_loop(self):
        if threading.current_thread() is threading.main_thread():
            signal.signal(signal.SIGINT, lambda sig, frame: (print("\n\nCtrl-C Caught"), self._quit_flag.set()))
        for t in self._threads:
            t.start()
        while not self._quit_flag.is_set():
            time.sleep(0.1)
        for t in self._threads:
            t.join()
        self._logger("graceful shut down", True)

    def register_logical_udl(self, udl: Callable[[FpsLimiter, Tuple[Any, ...]], None], args: Tuple[Any, ...] = ()):
        def body():
            try:
                udl(FpsLimiter("", self._quit_flag), args)
            except Exception:
                self._logger("Caught exception printing stack trace and unwinding ...")
                traceback.print_exc()
                self._quit_flag.set()
        self._threads.append(threading.Thread(target=body))

    def register_capture_udl(self, name: str, udl, args: Tuple[Any, ...] = ()):
        def body():
            self._logger(f"starting capture udl '{name}'", True)
            try:
                for item in udl(FpsLimiter(name, self._quit_flag), args):
                    if not isinstance(item, tuple):
                        raise RuntimeError(f"capture UDL '{name}' yielded unsupported type {type(item)}")
                    if len(item) not in (3, 4):
                        raise RuntimeError(f"capture UDL '{name}' yielded tuple of unexpected length {len(item)}")
                    self._send(item[0], item[1], item[2], item[3] if len(item) == 4 else None)
            except Exception:
                self._logger(f"Caught exception in {name} printing stack trace and unwinding ...")
                traceback.print_exc()
                self._quit_flag.set()
            first_to_stop = not self._quit_flag.is_set()
            self._quit_flag.set()
            self._logger(f"capture udl '{name}' exhausted" if first_to_stop
                         else f"capture udl '{name}' stopped as a result of another stop signal", True)
        self._threads.append(threading.Thread(target=body))

    def _send(self, direction: str, acquisition_time: int, img, names: Optional[Sequence[str]] = None):
        if isinstance(img, ndarray):
            planes: Tuple[ndarray, ...] = (img,)
        elif isinstance(img, Sequence):
            if len(img) == 0:
                raise ValueError("capture source yielded an empty frame sequence")
            for idx, plane in enumerate(img):
                if not isinstance(plane, ndarray):
                    raise TypeError(f"frame at index {idx} for direction '{direction}' is not an ndarray")
            planes = tuple(img)
        else:
            raise TypeError(f"unsupported frame type {type(img)} for direction '{direction}'")
        total_bytes = sum(int(p.size * p.itemsize) for p in planes)
        if total_bytes <= 0:
            raise ValueError(f"total serialized size for direction '{direction}' must be positive")
        accessor = self._frameworks.get(direction)
        if accessor is None:
            accessor = BlockAccessor(direction, max_entry_size_bytes=total_bytes)
            accessor.__enter__()
            self._frameworks[direction] = accessor
        if names is not None:
            if len(names) != len(planes):
                raise ValueError(f"direction '{direction}' provided {len(planes)} planes but {len(names)} names")
            payload = tuple(zip(names, planes))
        else:
            payload = planes[0] if len(planes) == 1 else planes
        accessor.write_frame(acquisition_time, payload)

    def close(self):
        for accessor in self._frameworks.values():
            accessor.__exit__(None, None, None)
        self._frameworks.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
