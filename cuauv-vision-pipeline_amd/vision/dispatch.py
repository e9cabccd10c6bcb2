"""Host-fed multi-device dispatch of frame batches (BASELINE config 4: "video capture source, 4K frames batched 32-deep, full
chain sharded across 8 MI355X").

The reference fans one decoded frame out to its directions on one host thread (capture_sources/video.py:9-29) and every module
processes its frames alone.  Frames are independent, so a batch shards with no exchange step: frames [lo, hi) of every
batch go to shard g (contiguous split, `shard_of` - config 4: frames [4g, 4g + 4) -> GPU g), and what limits the rate is the host feed.
Hence, per device:

  * `ring` feeder threads, each with its own context (= its own HIP stream) and its own page-locked staging buffers
    (vision.utils.chain.ChainRunner): while one slot's kernels run, the next slot's frames are copied into pinned memory and cross
    PCIe, and the previous slot's results come back;
  * every feeder thread is bound to the CPUs of the NUMA node its GPU hangs off (/sys/bus/pci/devices/<pci address>/numa_node ->
    /sys/devices/system/node/node<k>/cpulist), so that the staging copy runs on cores next to the memory the DMA engine reads;
  * results are collected in submission order, frames in batch order.

One process can drive several devices (threads per device), or one process per device (torch.distributed.run, as bench.py is
launched): `rank` / `world` then place this process's devices in the global split.  The runner is injectable, so the CPU suite drives
this very code with a stand-in for the device (tests/test_dispatch.py), world_size 2 over gloo included.
"""
import collections
import os
import queue
import threading

import numpy as np


def shard_of(n_items, shard, n_shards):
    """Contiguous slice of n_items owned by `shard` (the same split bench.py uses for ranks)."""
    base, extra = divmod(n_items, n_shards)
    lo = shard * base + min(shard, extra)
    return lo, lo + base + (1 if shard < extra else 0)


# ---- NUMA placement -----------------------------------------------------------------------------------------------------------

def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def numa_node_of_pci(pci_address, sysfs="/sys"):
    """NUMA node of a PCI device (-1 / None when the platform does not say)."""
    try:
        with open(os.path.join(sysfs, "bus", "pci", "devices", pci_address, "numa_node")) as f:
            node = int(f.read().strip())
        return node if node >= 0 else None
    except (OSError, ValueError):
        return None


def cpus_of_numa_node(node, sysfs="/sys"):
    try:
        with open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")) as f:
            return _parse_cpulist(f.read())
    except (OSError, ValueError):
        return set()


def cpus_near_device(device, sysfs="/sys", pci_lookup=None):
    """CPUs of the NUMA node next to HIP device `device`, restricted to what this process may run on; empty set = unknown."""
    if pci_lookup is None:
        def pci_lookup(dev):
            import ctypes as C
            from vision import _vp
            buf = C.create_string_buffer(32)
            if _vp.lib().vp_device_pci_bus_id(int(dev), buf, 32) != 0:
                return None
            return buf.value.decode()
    addr = pci_lookup(device)
    if not addr:
        return set()
    node = numa_node_of_pci(addr, sysfs)
    if node is None:
        return set()
    cpus = cpus_of_numa_node(node, sysfs)
    try:
        cpus &= os.sched_getaffinity(0)
    except (AttributeError, OSError):
        pass
    return cpus


def bind_current_thread(cpus):
    """sched_setaffinity for the calling thread only (Linux: pid 0 = the calling thread).  Returns the set actually applied."""
    if not cpus:
        return set()
    try:
        os.sched_setaffinity(0, cpus)
        return set(cpus)
    except (AttributeError, OSError):
        return set()


# ---- the dispatcher -----------------------------------------------------------------------------------------------------------

def _default_runner_factory(chain):
    def make(device, n_frames, height, width):
        from vision.utils.chain import ChainRunner
        return ChainRunner(n_frames, height, width, device=device, **chain)
    return make


class BatchDispatcher:
    """submit(frames) -> later collect() gives, per batch and in submission order, the results of this process's shards.

    devices      HIP device indices driven by this process
    batch        frames per batch (the whole batch, e.g. 32); every shard owns a fixed slice of it
    chain        keyword arguments for vision.utils.chain.ChainRunner (color_mode, lo, hi, morph, ccl, numbering, max_labels, want)
    rank, world  this process's place when several processes share the batch: shard index of local device i is
                 rank * len(devices) + i of world * len(devices)
    ring         feeder threads (= contexts, staging slots) per device
    make_runner  (device, n_frames, height, width) -> object with `.input` ((n, h, w, 3) uint8 array to fill) and `.run()` -> dict of
                 arrays valid until the next run; default: ChainRunner on the device
    """

    def __init__(self, devices, batch, height, width, chain=None, rank=0, world=1, ring=2, bind_numa=True, make_runner=None,
                 sysfs="/sys", pci_lookup=None, copy_threads=4):
        self.devices = list(devices)
        self.batch, self.h, self.w = int(batch), int(height), int(width)
        n_shards = int(world) * len(self.devices)
        self.slices = [shard_of(self.batch, int(rank) * len(self.devices) + i, n_shards) for i in range(len(self.devices))]
        make = make_runner or _default_runner_factory(chain or {})
        self._in = [queue.Queue(maxsize=max(1, ring)) for _ in self.devices]     # bounded: submit() blocks when a device falls behind
        self._out = queue.Queue()
        self._pending = collections.OrderedDict()
        self._next_id = 0
        self._errors = []
        self.copy_threads = max(1, int(copy_threads))                             # staging copy of a slot: this many threads per feeder
        self.bound_cpus = {}                                                      # device -> CPUs its feeders run on (empty: not bound)
        self._threads = []
        started = threading.Barrier(len(self.devices) * max(1, ring) + 1)
        for di, dev in enumerate(self.devices):
            lo, hi = self.slices[di]
            for slot in range(max(1, ring)):
                t = threading.Thread(target=self._feeder, name=f"vp-feed-d{dev}-s{slot}", daemon=True,
                                     args=(di, dev, lo, hi, make, bind_numa, sysfs, pci_lookup, started))
                t.start()
                self._threads.append(t)
        started.wait()
        if self._errors:
            self.close()
            raise self._errors[0]

    def _feeder(self, di, dev, lo, hi, make, bind_numa, sysfs, pci_lookup, started):
        runner = None
        try:
            if bind_numa:
                self.bound_cpus[dev] = bind_current_thread(cpus_near_device(dev, sysfs, pci_lookup))   # before the pinned buffers are made
            else:
                self.bound_cpus.setdefault(dev, set())
            if hi > lo:
                runner = make(dev, hi - lo, self.h, self.w)
        except BaseException as e:                     # noqa: BLE001 - reported to the submitting thread
            self._errors.append(e)
        started.wait()
        if self._errors:
            return
        # The staging copy (frames -> page-locked slot) is the one touch of the frames on the host and what bounds the host-fed rate:
        # one thread moves about 10 GB/s, the link takes 50, so a slot's frames are copied in slices by a few threads of the feeder's
        # own (numpy releases the interpreter for the copy; the threads inherit the feeder's CPU binding).
        pool = None
        if self.copy_threads > 1 and hi - lo > 1:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(min(self.copy_threads, hi - lo), thread_name_prefix=f"vp-copy-d{dev}")
        try:
            self._feed(di, lo, hi, runner, pool)
        finally:
            if pool is not None:
                pool.shutdown(wait=False)

    def _feed(self, di, lo, hi, runner, pool):
        def stage(frames):
            if pool is None:
                np.copyto(runner.input, frames[lo:hi])
                return
            k = pool._max_workers
            cuts = [lo + (hi - lo) * i // k for i in range(k + 1)]
            list(pool.map(lambda ab: np.copyto(runner.input[ab[0] - lo:ab[1] - lo], frames[ab[0]:ab[1]]), zip(cuts, cuts[1:])))
        while True:
            item = self._in[di].get()
            if item is None:
                return
            bid, frames = item
            try:
                if runner is None:
                    res = {}
                else:
                    stage(frames)
                    out = runner.run()
                    res = {k: np.array(v, copy=True) for k, v in out.items() if isinstance(v, np.ndarray)}
                    for k, v in out.items():
                        if not isinstance(v, np.ndarray):
                            res[k] = v
                self._out.put((bid, di, res, None))
            except BaseException as e:                 # noqa: BLE001
                self._out.put((bid, di, None, e))

    def submit(self, frames):
        """frames: (batch, h, w, 3) uint8 array-like (a numpy array, a memory map of a frame stack ...).  The caller must not change it
        until the batch has been collected.  Returns the batch id."""
        if frames.shape[0] != self.batch or tuple(frames.shape[1:3]) != (self.h, self.w):
            raise ValueError(f"expected a ({self.batch}, {self.h}, {self.w}, 3) batch")
        bid = self._next_id
        self._next_id += 1
        self._pending[bid] = {}
        for di in range(len(self.devices)):
            self._in[di].put((bid, frames))            # device di's two feeders take turns (whichever is free)
        return bid

    def collect(self):
        """Blocks until the oldest submitted batch is complete: (batch id, [(lo, hi, results dict) per local device, in frame order])."""
        if not self._pending:
            raise RuntimeError("nothing submitted")
        bid = next(iter(self._pending))
        while len(self._pending[bid]) < len(self.devices):
            b, di, res, err = self._out.get()
            self._pending[b][di] = (res, err)          # a failure is an answer too: the batch completes, then raises
        got = self._pending.pop(bid)
        failed = [err for (_, err) in got.values() if err is not None]
        if failed:
            raise failed[0]                            # the batch is gone from the books: the next collect() serves the next batch
        return bid, [(self.slices[di][0], self.slices[di][1], got[di][0]) for di in range(len(self.devices))]

    def close(self):
        per_dev = len(self._threads) // max(1, len(self.devices))
        for di in range(len(self.devices)):
            while True:                                # batches nobody will collect any more: make room for the stop marks
                try:
                    self._in[di].get_nowait()
                except queue.Empty:
                    break
            for _ in range(per_dev):
                try:
                    self._in[di].put(None, timeout=2)  # (a feeder stuck in its runner never makes room: do not wait for ever)
                except queue.Full:
                    break
        for t in self._threads:
            t.join(timeout=10)
        self._threads = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def batches_of(path, batch, loop=False):
    """(n, h, w, 3) frame stack on disk (the `.npy` "video" of vision.capture_sources.video) cut into consecutive batches of
    `batch` frames, memory-mapped: a 4K clip does not have to fit in RAM and every frame is read once, by the feeder that stages it."""
    frames = np.load(path, mmap_mode="r")
    if frames.ndim != 4:
        raise RuntimeError("expected an (n, h, w, c) frame stack")
    while True:
        for i in range(0, len(frames) - batch + 1, batch):
            yield frames[i:i + batch]
        if not loop:
            return
