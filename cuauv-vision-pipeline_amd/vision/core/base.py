"""Module runtime: what a vision module inherits from and what feeds it.

The contract is the reference's core/base.py (the names modules and GUIs use: `VideoSource`, `sources`, `ModuleManager`,
`ModuleReader`, `VideoSourceMetadata`, `ModuleBase`; the launch flags; the block names `module_<Class>-on-<src>_post%<idx>%<name>#<CS>`
and `_tune%<idx>%<TunerClass>_<name>`; which frames reach `process()` and which reach an `@sources` handler; retry after a source
was deleted) - SURVEY.md section 3.1 / 8b lists the behaviours with their lines.  The insides are this repository's own:

  * a module's blocks are `_Port` records kept in one table per kind, opened and closed through one ExitStack;
  * a frame does not pass through host memory on its way to a module: `BlockAccessor.read_frame_device` moves it from its ring
    slot into HBM with one DMA and hands out device images (`vision.devmat.DeviceMat`, array-likes that materialise on the host
    only if Python touches them).  On a box without a device the frames are private host arrays, as in the reference;
  * the loop body is split into `_deliver`, `_fire_handlers` and `_flush_posts`;
  * posts of device images go into their block's ring slot by DMA and are committed when the copy has arrived
    (`vision.core.posts`): no host pass over the pixels of a post.

Differences a module can observe: logging falls back to stdout when `auvlog` is absent; every VideoSourceMetadata owns its latency
window (the reference shares one deque between all instances through a dataclass default); `UMat` inputs to post() are unwrapped by
duck typing (no cv2 dependency); `ModuleBase.stop()` ends a running module from another thread.
"""
import argparse
import contextlib
import glob
import re
import signal
import threading
import time
from collections import OrderedDict, deque, namedtuple
from typing import Any, Callable, Dict, List, NamedTuple, Optional, Tuple, Union

import numpy as np

from vision.core.bindings.camera_message_framework import BLOCK_STUB, BlockAccessor, ReadStatus
from vision.core.tuners import BoolTuner, DoubleTuner, IntTuner, TunerBase
from vision.core.frames import copy_frame
from vision.core.posts import VALID_COLOR_SPACES, PostQueue

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

# item size -> ((code in a source string, dtype) in the order the codes are looked for, default dtype)
_PLANE_TYPES = {1: ((("u8", np.uint8), ("i8", np.int8)), np.uint8),
                4: ((("u32", np.uint32), ("i32", np.int32), ("f32", np.float32)), np.float32),
                8: ((("u64", np.uint64), ("i64", np.int64), ("f64", np.float64)), np.float64)}
_SOURCE_SPEC = re.compile(r"^\s*(?P<name>[^\[\]:]*?)\s*(?:\[(?P<aliases>[^\]]*)\])?\s*(?::(?P<types>.*))?$")


def _now_ms() -> int:
    return int(time.monotonic() * 1000)


def _alias_list(text: Optional[str]) -> Tuple[str, ...]:
    return tuple(a.strip() for a in (text or "").split(",") if a.strip())


class VideoSource(namedtuple("VideoSource", "name byte_type short_type long_type plane_aliases",
                             defaults=(np.uint8, np.float32, np.float64, ()))):
    """How to decode one direction: `name[alias,...]:<t1>:<t4>:<t8>` (e.g. "zed[forward,depth]:f32"): the dtypes planes of item size
    1 / 4 / 8 are read as, and names for the planes when the block stores none."""
    __slots__ = ()

    @classmethod
    def create(cls, source_str: Union[str, "VideoSource"]) -> "VideoSource":
        if isinstance(source_str, cls):
            return source_str                            # already parsed
        m = _SOURCE_SPEC.match(source_str)
        if m is None:                                    # brackets out of place: everything before the first ':' is the name
            name, _, types = source_str.partition(":")
            m_name, aliases = name.strip(), ()
        else:
            m_name, aliases, types = m.group("name"), _alias_list(m.group("aliases")), m.group("types") or ""
        picked = []
        for width in (1, 4, 8):
            codes, default = _PLANE_TYPES[width]
            # a substring test in the table's order, as the reference does it: "u8" wins over "i8" when both are named
            picked.append(next((dtype for code, dtype in codes if code in types), default))
        return cls(m_name, picked[0], picked[1], picked[2], aliases)

    @classmethod
    def into_accessor(cls, instn: "VideoSource"):
        return BlockAccessor(instn.name, byte_type=instn.byte_type, short_type=instn.short_type, long_type=instn.long_type)


def sources(*source_specs: str):
    """`@sources("zed[forward]", "downward")` binds a method to an ordered list of aliases: the loop calls it with one image per alias
    once every alias has been seen and at least one of them is new.  "zed[forward]" names the alias `forward`; a bare name is its own."""
    def alias_of(spec: str) -> str:
        m = _SOURCE_SPEC.match(spec)
        inner = _alias_list(m.group("aliases")) if m is not None and m.group("aliases") is not None else ()
        return inner[0] if inner else spec.strip()
    aliases = tuple(alias_of(s) for s in source_specs)

    def bind(fn: Callable):
        fn._sources_aliases = aliases
        return fn
    return bind


class VideoMessage(NamedTuple):
    """One read of one direction, as ModuleManager.read_messages reports it."""
    source: VideoSource
    status: ReadStatus
    data: Any                       # image, tuple of images (planes), or None
    acquisition_time: int
    plane_names: Tuple[str, ...] = ()
    private: bool = False           # data already belongs to the module (device images / page-locked arrays of its own)


class _Port(NamedTuple):
    """A block the module holds open, with what it carries."""
    key: str
    accessor: BlockAccessor
    payload: Any = None             # VideoSource or TunerBase


def _table(ports: List[_Port], what: str) -> "OrderedDict[str, _Port]":
    out: "OrderedDict[str, _Port]" = OrderedDict((p.key, p) for p in ports)
    if len(out) != len(ports):
        raise RuntimeError(f"cannot have multiple {what} of the same name")
    return out


class ModuleManager:
    """The module's end of its blocks: the directions it reads, one one-frame block per tuner (created here; the index in the block
    name orders the GUI's controls), post blocks created on first use.  Usable inside `with` only."""

    def __init__(self, module_name: str, video_sources: List[VideoSource], tuner_sources: List[TunerBase]):
        self._module_name = "module_" + module_name
        self._videos = _table([_Port(v.name, VideoSource.into_accessor(v), v) for v in video_sources], "video sources")
        self._tuners = _table([_Port(t.name, BlockAccessor(f"{self._module_name}_tune%{i}%{t}", max_entry_size_bytes=t.byte_size()), t)
                               for i, t in enumerate(tuner_sources)], "tuner types")
        self._posts: Dict[str, BlockAccessor] = {}
        self._defaults_published = False
        self._open: Optional[contextlib.ExitStack] = None

    def video_accessor(self, name: str) -> BlockAccessor:
        """The accessor of a direction, for tools that want its counters (copies dropped as lapped, ...)."""
        return self._videos[name].accessor

    def _stack(self) -> contextlib.ExitStack:
        if self._open is None:
            raise RuntimeError("attempted to access ModuleManager while not in a context manager")
        return self._open

    def post_block(self, name: str, idx: int, nbytes: int) -> BlockAccessor:
        """The block of post `name` (`<post name>#<colour space>`), created on first use with the size of that first image."""
        block = self._posts.get(name)
        if block is None:
            block = self._posts[name] = self._stack().enter_context(BlockAccessor(f"{self._module_name}_post%{idx}%{name}", nbytes))
        return block

    def post(self, name: str, idx: int, acquisition_time: int, data: np.ndarray):
        self._stack()
        self.post_block(name, idx, data.nbytes).write_frame(acquisition_time, data)

    def read_messages(self) -> List[VideoMessage]:
        self._stack()
        for port in self._tuners.values():
            status, raw, _ = port.accessor.read_frame()
            if status == ReadStatus.FRAMEWORK_DELETED:
                raise RuntimeError("Unexpected deleted Tuner")
            if raw is not None:
                port.payload.deserialize(raw.tobytes("C"))
        out = []
        for port in self._videos.values():
            status, data, stamp, private = port.accessor.read_frame_device()
            if status == ReadStatus.FRAMEWORK_DELETED:
                raise RuntimeError(f"{port.accessor.direction} was marked for deletion")
            if data is not None:
                out.append(VideoMessage(port.payload, status, data, stamp, port.accessor.last_plane_names(), private))
        return out

    def __getitem__(self, key: str) -> Any:
        return self._tuners[key].payload.value

    def __str__(self) -> str:
        return (f"ModuleManager(name={self._module_name}, video_sources={[p.payload for p in self._videos.values()]}, "
                f"tuner_sources={[p.payload for p in self._tuners.values()]})")

    def __enter__(self):
        if self._open is not None:
            raise RuntimeError("double dipped in context manager for ModuleManager")
        with contextlib.ExitStack() as stack:
            for port in list(self._videos.values()) + list(self._tuners.values()):
                stack.enter_context(port.accessor)
            if not self._defaults_published:      # once: the GUI renders a tuner from the first frame of its block
                for port in self._tuners.values():
                    port.accessor.write_frame(_now_ms(), np.frombuffer(port.payload.serialize(), dtype=np.uint8))
                self._defaults_published = True
            self._open = stack.pop_all()          # everything opened: keep it (an exception above closes what was opened)
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        stack, self._open = self._open, None
        self._posts.clear()
        return stack.__exit__(exc_type, exc_value, traceback) if stack is not None else None


class _Watched(NamedTuple):
    """A post or tuner block of a running module as the GUI side sees it."""
    index: int
    accessor: BlockAccessor
    extra: Any                      # colour space of a post / TunerBase of a tuner


_TUNER_KINDS = {"IntTuner": lambda n: IntTuner(n, 0), "DoubleTuner": lambda n: DoubleTuner(n, 0), "BoolTuner": lambda n: BoolTuner(n, False)}


class ModuleReader:
    """The GUI's end of a running module: finds its post and tuner blocks under /dev/shm, polls them on a thread and calls the
    registered callbacks; `update_tuner_value` writes an edit back into the tuner's block."""

    def __init__(self, module_name: str):
        if module_name not in self.get_active_modules():
            raise RuntimeError("Module name is not active")
        self._module, self._stem = module_name, "module_" + module_name
        self._halt, self._poller = threading.Event(), None
        self._on_post: List[Callable[[str, str, int, np.ndarray, str], None]] = []      # (module, post name, index, image, colour space)
        self._on_tuner: List[Callable[[str, str, int, TunerBase], None]] = []           # (module, tuner name, index, tuner)
        self._resend_once = self._deleted = False
        self._all_posts: Dict[str, _Watched] = {}
        for block in self.active_posts:
            index, name, color_space = self.parse_post_name(block)
            self._all_posts[name] = _Watched(index, BlockAccessor(block), color_space)
        self._all_tuners: Dict[str, _Watched] = {}
        for block in self.active_tuners:
            index, tuner, name = self.parse_tune_name(block)
            self._all_tuners[name] = _Watched(index, BlockAccessor(block), tuner)

    @classmethod
    def get_active_modules(cls):
        # /dev/shm/auv_visiond_module_<Name>_... -> <Name> (hence no underscore in class or direction names)
        return sorted({path.split("_")[3] for path in glob.glob(f"{BLOCK_STUB}module_*")})

    def _blocks(self, kind: str) -> List[str]:
        return [path[len(BLOCK_STUB):] for path in glob.glob(f"{BLOCK_STUB}{self._stem}_{kind}%*")]

    active_posts = property(lambda self: self._blocks("post"))
    active_tuners = property(lambda self: self._blocks("tune"))
    framework_deleted = property(lambda self: self._deleted)

    def parse_post_name(self, s: str) -> Tuple[int, str, str]:
        _, index, rest = s.split("%")
        name, marked, color_space = rest.partition("#")
        return int(index), name, color_space if marked else "BGR"

    def parse_tune_name(self, s: str) -> Tuple[int, TunerBase, str]:
        _, index, rest = s.split("%")
        kind, name = rest.split("_", maxsplit=1)
        return int(index), _TUNER_KINDS.get(kind, _TUNER_KINDS["BoolTuner"])(name), name

    def register_post_udl(self, udl):
        self._on_post.append(udl)

    def register_tuner_udl(self, udl):
        self._on_tuner.append(udl)

    def allow_resend_tuners_once(self):
        self._resend_once = True

    def update_tuner_value(self, name: str, value: Any):
        watched = self._all_tuners[name]
        watched.extra._current_value = value
        watched.accessor.write_frame(_now_ms(), np.frombuffer(watched.extra.serialize(), dtype=np.uint8))

    def run_forever(self, fps: int = 60):
        if self._poller is not None:
            raise RuntimeError("cannot run already running module reader")
        self._halt = threading.Event()
        self._poller = poller = threading.Thread(target=self._poll, args=(1.0 / fps,))
        poller.start()

    def _gone(self):
        print(f"ModuleReader: {self._module} framework deleted")
        self._deleted = True
        self._halt.set()                                # the poller leaves; unblock() still joins it

    def _poll(self, period: float):
        with contextlib.ExitStack() as stack:
            for watched in list(self._all_posts.values()) + list(self._all_tuners.values()):
                stack.enter_context(watched.accessor)
            while not self._halt.is_set():
                began = time.monotonic()
                for name, post in self._all_posts.items():
                    status, data, _ = post.accessor.read_frame()
                    if status == ReadStatus.FRAMEWORK_DELETED:
                        self._gone()
                    elif status == ReadStatus.SUCCESS and data is not None:
                        for udl in self._on_post:
                            udl(self._module, name, post.index, data, post.extra)
                resend, self._resend_once = self._resend_once, False
                for name, tune in self._all_tuners.items():
                    status, data, _ = tune.accessor.read_frame()
                    if status == ReadStatus.FRAMEWORK_DELETED:
                        self._gone()
                    elif data is not None and (resend or status == ReadStatus.SUCCESS):
                        tune.extra.deserialize(data.tobytes("C"))
                        for udl in self._on_tuner:
                            udl(self._module, name, tune.index, tune.extra)
                time.sleep(max(0.0, period - (time.monotonic() - began)))

    def _join(self) -> bool:
        poller, self._poller = getattr(self, "_poller", None), None
        if poller is None:
            return False
        self._halt.set()
        poller.join()
        return True

    def unblock(self):
        if not self._join():
            print(f"[WARNING]: {self._stem} was already terminated")

    def __del__(self):
        if self._join():
            print("[WARNING]: object garbage collected without freeing underlying resources")


class VideoSourceMetadata:
    """Per-direction bookkeeping: shape of the last frame (for normalisation), age of the last 30 frames, liveness."""
    WINDOW = 30

    def __init__(self):
        self._frames_read = 0
        self._shape: Tuple[int, int] = (1, 1)
        self._acquisition_times = deque(maxlen=self.WINDOW)
        self._dead_counter = 0

    def update(self, mat, acquisition_time: int):
        self._acquisition_times.append(_now_ms() - acquisition_time)
        if isinstance(mat, tuple):
            if not mat:
                return                                  # an empty plane tuple: nothing to measure
            mat = mat[0]
        self._shape = (mat.shape[0], mat.shape[1])
        self._frames_read += 1
        self._dead_counter = max(0, self._dead_counter - 1)

    def mark_as_dead(self) -> bool:
        """-> True when the source had been healthy until now."""
        healthy, self._dead_counter = self._dead_counter == 0, 3
        return healthy

    def get_latency(self) -> int:
        return int(sum(self._acquisition_times) / len(self._acquisition_times))

    def normalize_axis(self, coord: float, axis: int) -> float:
        """(coord - dim/2) / width; axis 0 = x, 1 = y.  BOTH axes are scaled by the width (core/base.py:553-563)."""
        height, width = self._shape
        return (coord - (width if axis == 0 else height) / 2) / width

    def normalize_coord(self, coord: Tuple[float, float]) -> Tuple[float, float]:
        """(y, x) in pixels -> normalised (y, x)."""
        return self.normalize_axis(coord[0], 1), self.normalize_axis(coord[1], 0)


_CLI = (  # the launch line of a module (core/base.py:599-635): flags, then what add_argument gets
    (("-f", "--fps"), dict(type=int, help="upper bound on loop iterations per second (the sources set the real rate)")),
    (("--verbose",), dict(action="store_true", help="log what the loop is doing")),
    (("--enable-performance",), dict(action="store_true", help="post() becomes a no-op: nothing is copied or published for the GUI")),
    (("sources",), dict(nargs="*", type=str,
                        help="directions to read, each `name[alias,...]:<1-byte type>:<4-byte type>:<8-byte type>` with types from u8 i8 / u32 i32 f32 /\n"
                             "u64 i64 f64 (e.g. forward, zed[forward,depth]:f32); none given: the module's own list")),
)


class ModuleBase:
    """Base class of a vision module: `Module(sources, tuners)()` polls the sources at `fps` and calls process(direction, image) -
    or the @sources-decorated handlers - on a worker thread, one frame at a time."""

    def __init__(self, video_sources: List[Union[VideoSource, str]] = [], tuners: List[TunerBase] = [], fps: int = 10, **kwargs):
        cli = argparse.ArgumentParser(f"{__file__}", description="runs this vision module against its camera directions",
                                      formatter_class=argparse.RawTextHelpFormatter)
        for flags, spec in _CLI:
            cli.add_argument(*flags, **spec)
        cli.set_defaults(fps=fps)
        args = cli.parse_args()
        cls_name = type(self).__name__
        if "_" in cls_name:                               # block names are split on "_" by the GUI side
            raise RuntimeError(f"Class name '{cls_name}'cannot have an underscore")
        chosen = [VideoSource.create(s) for s in (args.sources or video_sources)]
        self._name = f"{cls_name}-on-" + "-".join(s.name for s in chosen)
        self._fps: int = args.fps or fps
        self._chatty = bool(args.verbose)
        self._performance_enabled: bool = args.enable_performance
        self._module_manager = ModuleManager(self._name, chosen, tuners)
        self._posts = PostQueue(self._module_manager.post_block, self._module_manager.post, enabled=not self._performance_enabled)
        self._post_queue = self._posts.queue               # name -> (what is queued, colour space); emptied by every flush
        self._retry = True                                # __call__ keeps (re-)entering the manager while this is set
        self._directions: Dict[str, VideoSourceMetadata] = {}     # per direction and per named plane
        for s in chosen:
            for key in (s.name,) + tuple(s.plane_aliases):
                self._directions.setdefault(key, VideoSourceMetadata())
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
        say = getattr(auvlog, self._name)
        say(f"Running {self._name}", True)
        if self._performance_enabled:
            say("Module running in performance mode", True)
        quit_flag = self._quit_flag = threading.Event()
        on_main = threading.current_thread() is threading.main_thread()
        old_handler = signal.getsignal(signal.SIGINT) if on_main else None

        def interrupted(signum, *_):
            say(f"Caught signal: {signum}. It may take up to 2 seconds to clean up.", self._chatty)
            quit_flag.set()

        say(f"Target FPS = {self._fps}", self._chatty)
        while self._retry:                                # a deleted source: leave the manager, enter it again, wait for the source
            self._retry = False
            quit_flag.clear()
            with self._module_manager:
                if on_main:
                    signal.signal(signal.SIGINT, interrupted)
                    say("Registered SIGINT handler", self._chatty)
                say(f"Initialized module manager {self._module_manager}", self._chatty)
                worker = threading.Thread(target=self._loop, args=(quit_flag, say))
                worker.start()
                worker.join()
            if self._retry and on_main:
                signal.signal(signal.SIGINT, old_handler)
                say("Unregistered SIGINT handler", self._chatty)
        say(f"Cleaning {type(self).__name__}", True)

    # -- the loop -----------------------------------------------------------------------------------------------------------
    def _bound_handlers(self, say) -> List[Tuple[Callable[..., None], Tuple[str, ...]]]:
        """Methods decorated with @sources, with their alias lists."""
        found = []
        for attr in dir(self):
            try:
                member = getattr(self, attr)
            except Exception:
                continue
            aliases = getattr(member, "_sources_aliases", None) or getattr(getattr(member, "__func__", None), "_sources_aliases", None)
            if aliases:
                found.append((member, tuple(aliases)))
                say(f"Registered multi-source handler {attr} with aliases: {aliases}", True)
        return found

    @staticmethod
    def _names_of_planes(message: VideoMessage, count: int) -> Tuple[str, ...]:
        """Names stored in the block win, then the [alias] list of the source string, then name[i]."""
        stored = message.plane_names
        if stored and len(stored) == count and all(str(n) for n in stored):
            return tuple(stored)
        listed = message.source.plane_aliases
        if listed and len(listed) == count:
            return listed
        return tuple(f"{message.source.name}[{i}]" for i in range(count))

    def _deliver(self, message: VideoMessage, cache: dict, fresh: set, covered: set):
        """One successfully read direction: its frame(s) become the module's own, enter the alias cache, and reach process() unless
        an @sources handler covers the alias."""
        source, image, stamp = message.source, message.data, message.acquisition_time
        if not message.private:                           # views of the library's buffer (a box without a device): copy, as the reference does
            image = tuple(copy_frame(p) for p in image) if isinstance(image, tuple) else copy_frame(image)
        self._note_frame(source.name, image, stamp)
        self._current_direction = source.name
        if isinstance(image, tuple):
            for alias, plane in zip(self._names_of_planes(message, len(image)), image):
                cache[alias] = plane
                fresh.add(alias)
                self._note_frame(alias, plane, stamp)
                if alias not in covered:
                    self._current_direction = alias
                    self.process(alias, plane)
            return
        cache[source.name] = image
        fresh.add(source.name)
        if source.name not in covered:
            self.process(source.name, image)

    def _fire_handlers(self, handlers, cache: dict, fresh: set, told: set, say):
        for handler, aliases in handlers:
            waiting = [a for a in aliases if a not in cache]
            if waiting:
                if handler not in told:
                    told.add(handler)
                    say(f"Handler {handler.__name__} waiting for aliases: {waiting}. Available: {list(cache)}", True)
            elif fresh.intersection(aliases):
                handler(*(cache[a] for a in aliases))

    def _flush_posts(self, wait: bool = False):
        """Publishes the iteration's posts.  Device images are already on their way into their blocks' slots (post() queued the
        copies); those that have arrived are committed, the others by the next call - `wait` blocks until every one is out."""
        self._posts.flush(wait)

    def _loop(self, quit_flag: threading.Event, say):
        cache: Dict[str, Any] = {}
        handlers = self._bound_handlers(say)
        covered = {alias for _, aliases in handlers for alias in aliases}
        told: set = set()
        period = 1.0 / self._fps
        try:
            while not quit_flag.is_set():
                began = time.monotonic()
                try:
                    messages = self._module_manager.read_messages()
                except RuntimeError as problem:               # a source went away: __call__ re-enters the manager
                    say(f"Error: {problem}", True)
                    self._retry = True
                    quit_flag.set()
                    break
                fresh: set = set()
                for message in messages:
                    if message.status == ReadStatus.SUCCESS:
                        self._deliver(message, cache, fresh, covered)
                    elif message.status == ReadStatus.NO_NEW_FRAME and self._directions[message.source.name].mark_as_dead():
                        say(f"{message.source.name} appears to be slow or dead!", self._chatty)
                self._fire_handlers(handlers, cache, fresh, told, say)
                self._flush_posts()
                slack = period - (time.monotonic() - began)
                if slack > 0:
                    # the loop is about to idle: posts whose copies are still crossing are published now rather than one period later
                    # (a saturated loop leaves them to the next iteration, where they overlap with its work)
                    if self._posts.pending():
                        self._flush_posts(wait=True)
                        slack = period - (time.monotonic() - began)
                    time.sleep(max(slack, 0))
        finally:
            self._posts.drain()                               # no copy may target a block once the manager closes it

    # -- services used by module code -------------------------------------------------------------------------------------------
    def post(self, name: str, image, color_space: str = "BGR"):
        """Queues `image` as it is now for the GUI (published after the handlers of this iteration); no-op under --enable-performance.
        A host array is copied (uint8); a device image is copied by the GPU straight into its block (vision.core.posts)."""
        if self._performance_enabled:
            return
        self._posts.post(name, image, color_space)

    def get_latency(self) -> int:
        return self._directions[self._current_direction].get_latency()

    def normalize(self, coordinate: Tuple[float, float]) -> Tuple[float, float]:
        """(y, x) in pixels of the current direction -> ((y - h/2)/w, (x - w/2)/w)."""
        return self._directions[self._current_direction].normalize_coord(coordinate)

    def normalize_axis(self, coordinate: float, axis: int) -> float:
        return self._directions[self._current_direction].normalize_axis(coordinate, axis)

    def _note_frame(self, direction: str, frame, acquisition_time: int):
        self._directions.setdefault(direction, VideoSourceMetadata()).update(frame, acquisition_time)

    def process_bundle(self, direction: str, frames: Tuple[np.ndarray, ...], aliases: Tuple[str, ...], acquisition_time: int):
        """Feeds the planes of one frame to process() one by one (a helper for callers outside the loop)."""
        if aliases and len(aliases) != len(frames):
            raise RuntimeError(f"direction '{direction}' provided {len(frames)} planes but {len(aliases)} aliases")
        for alias, frame in zip(aliases or tuple(f"{direction}[{i}]" for i in range(len(frames))), frames):
            self._note_frame(alias, frame, acquisition_time)
            self._current_direction = alias
            self.process(alias, frame)

    def process(self, direction: str, image: np.ndarray):
        """Per-alias hook; modules that only use @sources handlers need not override it."""
        return None
