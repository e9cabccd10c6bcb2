"""Module runtime: source parsing, tuner/post blocks, the polling loop and multi-source dispatch.

Mirror of the reference core/base.py (VideoSource :36-120, sources :123-149, ModuleManager :161-322,
ModuleReader :325-510, VideoSourceMetadata :517-574, ModuleBase :577-942): same class and method names, CLI
flags, block naming (`module_<Class>-on-<src>_post%<idx>%<name>#<CS>`, `_tune%<idx>%<TunerClass>_<name>`), frame
copy / dispatch rules and retry-on-FRAMEWORK_DELETED contract, so module files written against the reference
run here unchanged.  The image operators those modules call (vision.utils.*) run on the GPU via libvp.

Deliberate differences: logging falls back to a stdout logger when `auvlog` is absent; every
VideoSourceMetadata owns its latency window (the reference shares one deque between all instances through a
dataclass default); there is no cv2 dependency (`UMat` inputs to post() are unwrapped by duck typing).
"""
import argparse
import contextlib
import glob
import signal
import threading
import time
from collections import OrderedDict, deque
from dataclasses import dataclass, field
from typing import Any, Callable, Deque, Dict, List, Optional, Tuple, Union

import numpy as np

from vision.core.bindings.camera_message_framework import BLOCK_STUB, BlockAccessor, ReadStatus
from vision.core.tuners import BoolTuner, DoubleTuner, IntTuner, TunerBase
from vision.core.frames import copy_frame
from vision.devmat import DeviceMat
from vision.utils.helpers import as_mat

try:  # the CUAUV logging daemon client, when the monorepo is around
    from auvlog.client import log as auvlog  # type: ignore
except Exception:  # pragma: no cover - exercised wherever auvlog is absent
    class _StdoutLogger:
        """auvlog.client.log look-alike: attribute access narrows the channel, calling logs."""

        def __init__(self, path=()):
            self._path = path

        def __getattr__(self, item):
            if item.startswith("__") and item.endswith("__"):
                raise AttributeError(item)
            return _StdoutLogger(self._path + (item,))

        def __call__(self, message, copy_to_stdout=False):
            if copy_to_stdout:
                print(f"[{'.'.join(self._path)}] {message}", flush=True)

    auvlog = _StdoutLogger()

_TYPE_CODES = {1: {"u8": np.uint8, "i8": np.int8}, 4: {"u32": np.uint32, "i32": np.int32, "f32": np.float32},
               8: {"u64": np.uint64, "i64": np.int64, "f64": np.float64}}
_TYPE_DEFAULT = {1: np.uint8, 4: np.float32, 8: np.float64}
VALID_COLOR_SPACES = ("BGR", "RGB", "HSV", "LAB", "HLS", "YCRCB", "LUV", "GRAY")


def _now_ms() -> int:
    return int(time.monotonic() * 1000)


@dataclass
class VideoSource:
    """How to decode one direction: `name[alias,...]:<t1>:<t4>:<t8>` (e.g. "zed[forward,depth]:f32")."""
    name: str
    byte_type: type = np.uint8
    short_type: type = np.float32
    long_type: type = np.float64
    plane_aliases: Tuple[str, ...] = ()

    @classmethod
    def _parse_name_and_aliases(cls, source: str) -> Tuple[str, Tuple[str, ...]]:
        if "[" not in source:
            return source, tuple()
        name, rest = source.split("[", maxsplit=1)
        inner = rest.rsplit("]", maxsplit=1)[0]
        return name, tuple(a.strip() for a in inner.split(",") if a.strip())

    @classmethod
    def create(cls, source_str: Union[str, "VideoSource"]) -> "VideoSource":
        if isinstance(source_str, VideoSource):
            return source_str
        name_part, _, types = source_str.partition(":")
        name, aliases = cls._parse_name_and_aliases(name_part)
        chosen = []
        for width in (1, 4, 8):
            # substring test in declaration order, like the reference ("u8" wins over "i8", ...)
            chosen.append(next((t for code, t in _TYPE_CODES[width].items() if code in types), _TYPE_DEFAULT[width]))
        return VideoSource(name.strip(), chosen[0], chosen[1], chosen[2], aliases)

    @classmethod
    def into_accessor(cls, instn: "VideoSource"):
        return BlockAccessor(instn.name, byte_type=instn.byte_type, short_type=instn.short_type, long_type=instn.long_type)


def sources(*source_specs: str):
    """Binds a method to an ordered list of aliases; the loop calls it with one image per alias once all are
    cached and at least one is new.  "zed[forward]" names the alias `forward`; a bare "downward" is itself."""
    def alias_of(spec: str) -> str:
        spec = spec.strip()
        if "[" in spec and "]" in spec:
            return spec.split("[", 1)[1].rsplit("]", 1)[0].strip()
        return spec

    def decorate(fn: Callable):
        fn._sources_aliases = tuple(alias_of(s) for s in source_specs)
        return fn
    return decorate


@dataclass
class VideoMessage:
    source: VideoSource
    status: ReadStatus
    data: Optional[Union[np.ndarray, Tuple[np.ndarray, ...]]]
    acquisition_time: int
    plane_names: Tuple[str, ...] = tuple()
    private: bool = False          # data already is the module's own writable copy (read straight into page-locked memory)


class ModuleManager:
    """The module's end of its blocks: reads video directions and tuner updates, creates post blocks lazily."""

    def __init__(self, module_name: str, video_sources: List[VideoSource], tuner_sources: List[TunerBase]):
        self._module_name = "module_" + module_name
        self._post_name = self._module_name + "_post"
        self._tune_name = self._module_name + "_tune"
        self._first = True
        self._video_sources: Dict[str, VideoSource] = {vs.name: vs for vs in video_sources}
        self._tuner_sources: Dict[str, TunerBase] = {ts.name: ts for ts in tuner_sources}
        if len(self._video_sources) != len(video_sources):
            raise RuntimeError("cannot have multiple video sources of the same name")
        if len(self._tuner_sources) != len(tuner_sources):
            raise RuntimeError("cannot have multiple tuner types of the same name")
        self._video_accessor: Dict[str, BlockAccessor] = {vs.name: VideoSource.into_accessor(vs) for vs in video_sources}
        # the index in the block name tells the GUI how to order the tuners
        self._tuner_accessor: Dict[str, BlockAccessor] = {
            ts.name: BlockAccessor(f"{self._tune_name}%{idx}%{ts}", max_entry_size_bytes=ts.byte_size())
            for idx, ts in enumerate(tuner_sources)}
        self._post_accessor: Dict[str, BlockAccessor] = {}
        self._exit_stack = contextlib.ExitStack()
        self._inside_ctx = False

    def _require_ctx(self):
        if not self._inside_ctx:
            raise RuntimeError("attempted to access ModuleManager while not in a context manager")

    def post(self, name: str, idx: int, acquisition_time: int, data: np.ndarray):
        self._require_ctx()
        accessor = self._post_accessor.get(name)
        if accessor is None:
            accessor = BlockAccessor(f"{self._post_name}%{idx}%{name}", data.nbytes)
            self._exit_stack.enter_context(accessor)
            self._post_accessor[name] = accessor
        accessor.write_frame(acquisition_time, data)

    def read_messages(self) -> List[VideoMessage]:
        self._require_ctx()
        for name, accessor in self._tuner_accessor.items():
            status, frame, _ = accessor.read_frame()
            if status == ReadStatus.FRAMEWORK_DELETED:
                raise RuntimeError("Unexpected deleted Tuner")
            if frame is not None:
                self._tuner_sources[name].deserialize(frame.tobytes("C"))
        messages: List[VideoMessage] = []
        for name, accessor in self._video_accessor.items():
            status, data, acquisition_time, private = accessor.read_frame_device()
            if status == ReadStatus.FRAMEWORK_DELETED:
                raise RuntimeError(f"{accessor.direction} was marked for deletion")
            if data is not None:
                messages.append(VideoMessage(self._video_sources[name], status, data, acquisition_time, accessor.last_plane_names(), private))
        return messages

    def __getitem__(self, key: str) -> Any:
        return self._tuner_sources[key].value

    def __str__(self) -> str:
        return f"ModuleManager(name={self._module_name}, video_sources={self._video_sources}, tuner_sources={self._tuner_sources})"

    def __enter__(self):
        if self._inside_ctx:
            raise RuntimeError("double dipped in context manager for ModuleManager")
        self._inside_ctx = True
        self._exit_stack.__enter__()
        try:
            for accessor in list(self._video_accessor.values()) + list(self._tuner_accessor.values()):
                self._exit_stack.enter_context(accessor)
            if self._first:   # publish the defaults once so that the GUI can render the tuners
                self._first = False
                for ts in self._tuner_sources.values():
                    self._tuner_accessor[ts.name].write_frame(_now_ms(), np.frombuffer(ts.serialize(), dtype=np.uint8))
        except BaseException:
            self._exit_stack.close()
            self._inside_ctx = False
            raise
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self._exit_stack.__exit__(exc_type, exc_value, traceback)
        self._exit_stack = contextlib.ExitStack()
        self._post_accessor.clear()
        self._inside_ctx = False


class ModuleReader:
    """The GUI's end: discovers a running module's post / tuner blocks under /dev/shm and polls them."""

    def __init__(self, module_name: str):
        if module_name not in ModuleReader.get_active_modules():
            raise RuntimeError("Module name is not active")
        self._base_module_name = module_name
        self._module_name = f"module_{module_name}"
        self._post_name = f"{self._module_name}_post%"
        self._tune_name = f"{self._module_name}_tune%"
        self._quit_flag = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self._post_udls: List[Callable[[str, str, int, np.ndarray, str], None]] = []
        self._tuner_udls: List[Callable[[str, str, int, TunerBase], None]] = []
        self._tuner_guard = False
        self._framework_deleted = False
        self._all_posts: Dict[str, Tuple[int, BlockAccessor, str]] = {}
        for block in self.active_posts:
            idx, name, color_space = self.parse_post_name(block)
            self._all_posts[name] = (idx, BlockAccessor(block), color_space)
        self._all_tuners: Dict[str, Tuple[int, BlockAccessor, TunerBase]] = {}
        for block in self.active_tuners:
            idx, tuner, name = self.parse_tune_name(block)
            self._all_tuners[name] = (idx, BlockAccessor(block), tuner)

    @classmethod
    def get_active_modules(cls):
        # /dev/shm/auv_visiond_module_<Name>_... -> <Name>
        return list({path.split("_")[3] for path in glob.glob(f"{BLOCK_STUB}module_*")})

    def _blocks_with_prefix(self, prefix: str) -> List[str]:
        return [path[len(BLOCK_STUB):] for path in glob.glob(BLOCK_STUB + prefix + "*")]

    @property
    def active_posts(self) -> List[str]:
        return self._blocks_with_prefix(self._post_name)

    @property
    def active_tuners(self):
        return self._blocks_with_prefix(self._tune_name)

    @property
    def framework_deleted(self):
        return self._framework_deleted

    def parse_post_name(self, s: str) -> Tuple[int, str, str]:
        _, idx, tail = s.split("%")
        name, sep, color_space = tail.partition("#")
        return int(idx), name, (color_space if sep else "BGR")

    def parse_tune_name(self, s: str) -> Tuple[int, TunerBase, str]:
        _, idx, tail = s.split("%")
        kind, name = tail.split("_", maxsplit=1)
        tuner = IntTuner(name, 0) if kind == "IntTuner" else DoubleTuner(name, 0) if kind == "DoubleTuner" else BoolTuner(name, False)
        return int(idx), tuner, name

    def register_post_udl(self, udl):
        self._post_udls.append(udl)

    def register_tuner_udl(self, udl):
        self._tuner_udls.append(udl)

    def allow_resend_tuners_once(self):
        self._tuner_guard = True

    def update_tuner_value(self, name: str, value: Any):
        _, accessor, tuner = self._all_tuners[name]
        tuner._current_value = value
        accessor.write_frame(_now_ms(), np.frombuffer(tuner.serialize(), dtype=np.uint8))

    def run_forever(self, fps: int = 60):
        if self._thread is not None:
            raise RuntimeError("cannot run already running module reader")
        self._quit_flag = threading.Event()
        self._thread = threading.Thread(target=self._loop, args=(fps,))
        self._thread.start()

    def _on_deleted(self):
        print(f"ModuleReader: {self._base_module_name} framework deleted")
        self._framework_deleted = True
        self._quit_flag.set()

    def _loop(self, fps: int):
        period = 1.0 / fps
        with contextlib.ExitStack() as stack:
            for _, accessor, _ in list(self._all_posts.values()) + list(self._all_tuners.values()):
                stack.enter_context(accessor)
            while not self._quit_flag.is_set():
                tick = time.monotonic()
                for name, (idx, accessor, color_space) in self._all_posts.items():
                    status, data, _ = accessor.read_frame()
                    if status == ReadStatus.SUCCESS and data is not None:
                        for cb in self._post_udls:
                            cb(self._base_module_name, name, idx, data, color_space)
                    elif status == ReadStatus.FRAMEWORK_DELETED:
                        self._on_deleted()
                resent = False
                for name, (idx, accessor, tuner) in self._all_tuners.items():
                    status, data, _ = accessor.read_frame()
                    if (self._tuner_guard or status == ReadStatus.SUCCESS) and data is not None:
                        resent = resent or self._tuner_guard
                        tuner.deserialize(data.tobytes("C"))
                        for cb in self._tuner_udls:
                            cb(self._base_module_name, name, idx, tuner)
                    elif status == ReadStatus.FRAMEWORK_DELETED:
                        self._on_deleted()
                if resent:
                    self._tuner_guard = False
                time.sleep(max(0.0, period - (time.monotonic() - tick)))

    def unblock(self):
        if self._thread is None:
            print(f"[WARNING]: {self._module_name} was already terminated")
            return
        self._quit_flag.set()
        self._thread.join()
        self._thread = None

    def __del__(self):
        if getattr(self, "_thread", None) is not None:
            print("[WARNING]: object garbage collected without freeing underlying resources")
            self._quit_flag.set()
            self._thread.join()


@dataclass
class VideoSourceMetadata:
    """Per-direction bookkeeping: last frame shape (for normalisation), latency window, liveness."""
    _frames_read: int = 0
    _shape: Tuple[int, int] = (1, 1)
    _acquisition_times: Deque[int] = field(default_factory=lambda: deque(maxlen=30))
    _dead_counter: int = 0

    def update(self, mat: Union[np.ndarray, Tuple[np.ndarray, ...]], acquisition_time: int):
        self._acquisition_times.append(_now_ms() - acquisition_time)
        if isinstance(mat, tuple):
            if len(mat) == 0:
                return
            mat = mat[0]
        self._shape = (mat.shape[0], mat.shape[1])
        self._frames_read += 1
        self._dead_counter = max(0, self._dead_counter - 1)

    def mark_as_dead(self):
        """-> True when the source had been healthy until now."""
        was_alive = self._dead_counter == 0
        self._dead_counter = 3
        return was_alive

    def get_latency(self) -> int:
        return int(sum(self._acquisition_times) / len(self._acquisition_times))

    def normalize_axis(self, coord: float, axis: int) -> float:
        """(coord - dim/2) / width; axis 0 = x, 1 = y.  Both axes are scaled by the WIDTH (core/base.py:553-563)."""
        return (coord - self._shape[1 - axis] / 2) / self._shape[1]

    def normalize_coord(self, coord: Tuple[float, float]) -> Tuple[float, float]:
        """(y, x) -> normalised (y, x)."""
        return self.normalize_axis(coord[0], 1), self.normalize_axis(coord[1], 0)


class ModuleBase:
    """Base class of a vision module: `Module(sources, tuners)()` polls the sources at `fps` and calls
    process(direction, image) — or the @sources-decorated handlers — on a worker thread, one frame at a time."""

    def __init__(self, video_sources: List[Union[VideoSource, str]] = [], tuners: List[TunerBase] = [], fps: int = 10, **kwargs):
        parser = argparse.ArgumentParser(f"{__file__}", description="runs this vision module against its camera directions",
                                         formatter_class=argparse.RawTextHelpFormatter)
        parser.add_argument("-f", "--fps", type=int, default=fps,
                            help="upper bound on loop iterations per second (the sources set the real rate)")
        parser.add_argument("--verbose", action="store_true", help="log what the loop is doing")
        parser.add_argument("--enable-performance", action="store_true",
                            help="post() becomes a no-op: nothing is copied or published for the GUI")
        parser.add_argument("sources", nargs="*", type=str,
                            help="directions to read, each `name[alias,...]:<1-byte type>:<4-byte type>:<8-byte type>` with types from u8 i8 / u32 i32 f32 /\n"
                                 "u64 i64 f64 (e.g. forward, zed[forward,depth]:f32); none given: the module's own list")
        args = parser.parse_args()
        if "_" in self.__class__.__name__:
            raise RuntimeError(f"Class name '{self.__class__.__name__}'cannot have an underscore")
        src = [VideoSource.create(s) for s in (args.sources if args.sources else video_sources)]
        self._name = self.__class__.__name__ + "-on-" + "-".join(s.name for s in src)
        self._fps: int = args.fps if args.fps else fps
        self._verbose: bool = args.verbose
        self._module_manager = ModuleManager(self._name, src, tuners)
        self._post_queue: "OrderedDict[str, np.ndarray]" = OrderedDict()
        self._post_color_spaces: Dict[str, str] = {}
        self._performance_enabled = args.enable_performance
        self._retry = True
        self._video_metadata: Dict[str, VideoSourceMetadata] = {}
        for source in src:
            self._video_metadata[source.name] = VideoSourceMetadata()
            for alias in source.plane_aliases:
                self._video_metadata.setdefault(alias, VideoSourceMetadata())
        self._current_direction = ""
        self._quit_flag: Optional[threading.Event] = None

    def stop(self):
        """Ends a running __call__() from another thread (what SIGINT does); not part of the reference API."""
        if self._quit_flag is not None:
            self._quit_flag.set()

    @property
    def tuners(self):
        return self._module_manager

    def __call__(self):
        logger = getattr(auvlog, self._name)
        logger(f"Running {self._name}", True)
        if self._performance_enabled:
            logger("Module running in performance mode", True)
        previous_handler = signal.getsignal(signal.SIGINT)
        quit_flag = threading.Event()
        self._quit_flag = quit_flag

        def on_sigint(*sig):
            logger(f"Caught signal: {sig[0]}. It may take up to 2 seconds to clean up.", self._verbose)
            quit_flag.set()

        logger(f"Target FPS = {self._fps}", self._verbose)
        while self._retry:   # a deleted source re-enters the manager and waits for the source to come back
            self._retry = False
            quit_flag.clear()
            with self._module_manager:
                if threading.current_thread() is threading.main_thread():
                    signal.signal(signal.SIGINT, on_sigint)
                    logger("Registered SIGINT handler", self._verbose)
                logger(f"Initialized module manager {self._module_manager}", self._verbose)
                worker = threading.Thread(target=self._loop, args=(quit_flag, logger))
                worker.start()
                worker.join()
            if self._retry and threading.current_thread() is threading.main_thread():
                signal.signal(signal.SIGINT, previous_handler)
                logger("Unregistered SIGINT handler", self._verbose)
        logger(f"Cleaning {self.__class__.__name__}", True)

    # -- loop ---------------------------------------------------------------------------------------
    def _discover_handlers(self, logger) -> List[Tuple[Callable[..., None], Tuple[str, ...]]]:
        found = []
        for attr in dir(self):
            try:
                member = getattr(self, attr)
            except Exception:
                continue
            aliases = getattr(member, "_sources_aliases", None)
            if aliases is None and hasattr(member, "__func__"):
                aliases = getattr(member.__func__, "_sources_aliases", None)
            if aliases:
                found.append((member, tuple(aliases)))
                logger(f"Registered multi-source handler {attr} with aliases: {aliases}", True)
        return found

    def _plane_aliases(self, message: VideoMessage, count: int) -> Tuple[str, ...]:
        # names stored in the block win, then the [alias] list of the source string, then name[i]
        if message.plane_names and len(message.plane_names) == count and all(len(str(n)) > 0 for n in message.plane_names):
            return tuple(message.plane_names)
        if message.source.plane_aliases and len(message.source.plane_aliases) == count:
            return message.source.plane_aliases
        return tuple(f"{message.source.name}[{i}]" for i in range(count))

    def _loop(self, quit_flag: threading.Event, logger):
        frame_cache: Dict[str, Tuple[np.ndarray, int]] = {}
        handlers = self._discover_handlers(logger)
        covered = {alias for _, aliases in handlers for alias in aliases}
        complained = set()
        while not quit_flag.is_set():
            tick = time.monotonic()
            try:
                messages = self._module_manager.read_messages()
            except RuntimeError as e:
                logger(f"Error: {e}", True)
                quit_flag.set()
                self._retry = True
                break
            fresh = set()
            for message in messages:
                source, image, acq_time = message.source, message.data, message.acquisition_time
                if message.status == ReadStatus.SUCCESS and image is not None:
                    # module code gets writable arrays of its own: either the library already read the frame into page-locked memory
                    # that is now ours (message.private), or the arrays view its read buffer and are copied here
                    if not message.private:
                        image = tuple(copy_frame(p) for p in image) if isinstance(image, tuple) else copy_frame(image)
                    self._update_metadata_for_direction(source.name, image, acq_time)
                    self._current_direction = source.name
                    if isinstance(image, tuple):
                        for alias, plane in zip(self._plane_aliases(message, len(image)), image):
                            frame_cache[alias] = (plane, acq_time)
                            fresh.add(alias)
                            self._update_metadata_for_direction(alias, plane, acq_time)
                            if alias not in covered:
                                self._current_direction = alias
                                self.process(alias, plane)
                    else:
                        frame_cache[source.name] = (image, acq_time)
                        fresh.add(source.name)
                        if source.name not in covered:
                            self.process(source.name, image)
                elif message.status == ReadStatus.NO_NEW_FRAME:
                    if self._video_metadata[source.name].mark_as_dead():
                        logger(f"{source.name} appears to be slow or dead!", self._verbose)
            for handler, aliases in handlers:
                missing = [a for a in aliases if a not in frame_cache]
                if missing:
                    if handler not in complained:
                        complained.add(handler)
                        logger(f"Handler {handler.__name__} waiting for aliases: {missing}. Available: {list(frame_cache.keys())}", True)
                    continue
                if any(a in fresh for a in aliases):
                    handler(*[frame_cache[a][0] for a in aliases])
            for idx, (name, data) in enumerate(self._post_queue.items()):
                color_space = self._post_color_spaces.get(name, "BGR")
                self._module_manager.post(f"{name}#{color_space}", idx, _now_ms(), data)
            self._post_queue.clear()
            self._post_color_spaces.clear()
            time.sleep(max((1 / self._fps) - (time.monotonic() - tick), 0))

    # -- services used by module code ---------------------------------------------------------------
    def post(self, name: str, image, color_space: str = "BGR"):
        """Queues a uint8 copy of `image` for the GUI; no-op under --enable-performance."""
        if self._performance_enabled:
            return
        if "%" in name:
            raise RuntimeError("Cannot have % in name")
        image = as_mat(image)
        if isinstance(image, DeviceMat) and image.dtype == np.uint8:
            image = image.host_copy()                # one download into an array of its own; the image stays usable on the device
        else:
            image = np.array(image, np.uint8, copy=True, order="C", ndmin=1)
        color_space = color_space.upper()
        self._post_queue[name] = image
        self._post_color_spaces[name] = color_space if color_space in VALID_COLOR_SPACES else "BGR"

    def get_latency(self) -> int:
        return self._video_metadata[self._current_direction].get_latency()

    def normalize(self, coordinate: Tuple[float, float]) -> Tuple[float, float]:
        """(y, x) in pixels of the current direction -> ((y - h/2)/w, (x - w/2)/w)."""
        return self._video_metadata[self._current_direction].normalize_coord(coordinate)

    def normalize_axis(self, coordinate: float, axis: int) -> float:
        return self._video_metadata[self._current_direction].normalize_axis(coordinate, axis)

    def _update_metadata_for_direction(self, direction: str, frame, acquisition_time: int):
        self._video_metadata.setdefault(direction, VideoSourceMetadata()).update(frame, acquisition_time)

    def process_bundle(self, direction: str, frames: Tuple[np.ndarray, ...], aliases: Tuple[str, ...], acquisition_time: int):
        if aliases and len(aliases) != len(frames):
            raise RuntimeError(f"direction '{direction}' provided {len(frames)} planes but {len(aliases)} aliases")
        if not aliases:
            aliases = tuple(f"{direction}[{idx}]" for idx in range(len(frames)))
        for alias, frame in zip(aliases, frames):
            self._update_metadata_for_direction(alias, frame, acquisition_time)
            self._current_direction = alias
            self.process(alias, frame)

    def process(self, direction: str, image: np.ndarray):
        """Per-alias hook; modules that only use @sources handlers need not override it."""
        return None
