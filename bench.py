#!/usr/bin/env python3
"""bench.py — frames/sec of the fused colour -> inRange -> OPEN/CLOSE 5x5 -> CCL chain at 1080p.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  A "step" is one vp_chain_run over one batch of
synthetic 1080p frames already resident in HBM (BASELINE.json configs[1]: preprocessor+red_buoy
chain, S1 frames, LAB-a in [150,255], OPEN then CLOSE with a 5x5 rect, 8-connected CCL + stats).
Frames are independent, so ranks shard them with no data-path collective ("weak" scaling: every
rank processes its own batch); torch.distributed is used only for the barrier and the max over
ranks of the timed region.

Prints ONE JSON line on rank 0: metric / value / ... plus
  roofline:     dominant kernel's algorithmic bytes per launch / its HIP-event duration vs 8 TB/s
  cpu_baseline: the CPU oracle (scalar C port, 1 thread) timed on a bounded sample of the same frames
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "cuauv-vision-pipeline_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

W, H = 1920, 1080
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic bytes per pixel of each kernel
KERNEL_BYTES_PER_PX = {
    "k_color_thresh": 3 + 1 + 1.0 / 8,      # read BGR, write 0/255 mask + bit-packed mask
    "k_morph_bits": 1.0 / 8 + 1 + 1.0 / 8,  # read bits, write cleaned mask + cleaned bits
    "k_ccl_write": 1.0 / 8 + 4,             # read bits, write int32 labels
}
# the same kernels by SURVEY 8d's own count (the images of the reference's pipeline only: 3 R + 1 W + 1 W + 4 W = 9 B/px; the bit planes
# are this implementation's intermediates and are not in it) - reported beside the figure above as frac_8d
KERNEL_BYTES_PER_PX_8D = {"k_color_thresh": 3 + 1, "k_morph_bits": 1, "k_ccl_write": 4}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="1080p frames per step per GPU")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic frames (tiled to the batch)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the chain + contours side measurement")
    ap.add_argument("--max-labels", type=int, default=256)
    ap.add_argument("--regions", type=int, default=5,
                    help="timed regions of exactly --steps steps each (every one bracketed by barrier + synchronize, MAX over ranks); "
                         "`value` comes from the median region (SURVEY 8d: median of 5)")
    ap.add_argument("--traffic-file", default=os.path.join("profiles", "r04", "traffic.json"),
                    help="rocprofv3 --pmc summary (tools/pmc_traffic.py) to take roofline.traffic from; used only when its recorded "
                         "source digest equals the running build and it holds the exact kernel instantiation that was timed")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development only: run the N-rank code path with every rank on cuda:0 and a gloo group (a one-GPU box cannot host "
                         "an RCCL group); the printed value is then not a scaling number")
    return ap.parse_args()


def init_distributed(backend):
    """One process per GPU (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  The process group is
    used only for barriers and the MAX over ranks of the timed region: frames are independent, no data-path collective."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist_mod.init_process_group(backend)
        dist = dist_mod
    return rank, local_rank, world, dist


def shard_of(n_items, rank, world):
    """Contiguous slice of n_items owned by `rank` (config 4: frames [lo, hi) of every batch go to GPU `rank`)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def timed_steps(step_fn, sync_fn, steps, dist, device=None):
    """Exactly `steps` calls of step_fn between barrier + device sync on both sides; returns the MAX over ranks (s)."""
    import torch
    if dist is not None:
        dist.barrier()
    sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def host_fed_leg(dist, rank, world, device, width, height, batch=32, batches=8, ring=4, make_runner=None, chain=None, bind_numa=True):
    """The host-fed form of the path on N ranks (BASELINE config 4; reference capture_sources/video.py:9-29 fans decoded frames out):
    after a barrier EVERY rank feeds its slice `shard_of(batch, rank, world)` of each batch from host memory through pinned staging into
    its own GPU (vision.dispatch.BatchDispatcher, feeder threads bound to the CPUs of that GPU's NUMA node) - the form in which host
    memory bandwidth, PCIe and NUMA placement, not HBM, decide how N GPUs scale.  -> this rank's record; rank 0 also gets every
    rank's record and the aggregate under "ranks" / "aggregate_frames_per_s".  `make_runner` replaces the device chain (CPU tests)."""
    from vision import dispatch as D
    frames = np.empty((batch, height, width, 3), np.uint8)
    frames[:] = (np.arange(batch, dtype=np.uint8) * 7 + 1)[:, None, None, None]
    chain = chain or {}
    with D.BatchDispatcher([device], batch, height, width, chain=chain, rank=rank, world=world, ring=ring, make_runner=make_runner,
                           bind_numa=bind_numa) as d:
        lo, hi = d.slices[0]
        d.submit(frames); d.collect()                     # contexts, staging, first launch
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        inflight = 0
        for _ in range(batches):
            d.submit(frames); inflight += 1
            if inflight > ring:
                d.collect(); inflight -= 1
        while inflight:
            d.collect(); inflight -= 1
        dt = time.perf_counter() - t0
        cpus = sorted(d.bound_cpus.get(device, ()))
    node = None
    try:
        import ctypes as C_
        from vision import _vp
        buf = C_.create_string_buffer(32)
        if make_runner is None and _vp.lib().vp_device_pci_bus_id(int(device), buf, 32) == 0:
            node = D.numa_node_of_pci(buf.value.decode())
    except Exception:
        node = None
    mine = {"rank": rank, "device": device, "frames_of_each_batch": [lo, hi], "frames_per_s": round(batches * (hi - lo) / dt, 1),
            "pcie_GBps": round(batches * (hi - lo) * width * height * 3 / dt / 1e9, 2), "seconds": round(dt, 3), "numa_node": node,
            "feeder_cpus_bound": len(cpus), "first_cpus": cpus[:4]}
    if dist is None:
        return dict(mine, ranks=[mine], aggregate_frames_per_s=mine["frames_per_s"], batches_per_s=round(batches / dt, 2))
    every = [None] * world
    dist.all_gather_object(every, mine)
    if rank == 0:
        slow = max(r["seconds"] for r in every)           # a batch is done when its slowest slice is
        return dict(mine, ranks=every, aggregate_frames_per_s=round(batches * batch / slow, 1), batches_per_s=round(batches / slow, 2),
                    aggregate_pcie_GBps=round(sum(r["pcie_GBps"] for r in every), 2))
    return mine


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        print(f"warning: WORLD_SIZE={world} != --gpus {args.gpus}", file=sys.stderr)
    n_gpus = max(world, 1)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libvp has no CPU path")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    _, _, _, dist = init_distributed("gloo" if args.rehearse_on_one_gpu else "nccl")

    import frames as F
    from vision import _vp
    ctx = _vp.Context(local_rank)
    L = _vp.lib()

    # ---- synthetic input, resident in HBM before the timed region -------------------------------
    B = args.batch
    distinct = [F.s1_buoy(rank * 1000 + i, W, H) for i in range(min(args.distinct, B))]
    host = np.stack([distinct[i % len(distinct)] for i in range(B)])
    d_bgr = torch.from_numpy(host).cuda()
    d_thr = torch.empty((B, H, W), dtype=torch.uint8, device="cuda")
    d_cln = torch.empty((B, H, W), dtype=torch.uint8, device="cuda")
    d_lab = torch.empty((B, H, W), dtype=torch.int32, device="cuda")
    d_stats = torch.zeros((B, args.max_labels, 5), dtype=torch.int32, device="cuda")
    d_cent = torch.zeros((B, args.max_labels, 2), dtype=torch.float64, device="cuda")
    d_nl = torch.zeros((B,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    morph = [(_vp.MORPH_OPEN, 5, 5), (_vp.MORPH_CLOSE, 5, 5)]
    desc = _vp.make_chain_desc(W, H, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), morph, ccl=1,
                               numbering=_vp.CCL_BLOCK2X2, max_labels=args.max_labels)
    bufs = _vp.ChainBuffers()
    bufs.bgr, bufs.threshed, bufs.cleaned = d_bgr.data_ptr(), d_thr.data_ptr(), d_cln.data_ptr()
    bufs.labels, bufs.stats, bufs.centroids, bufs.nlabels = d_lab.data_ptr(), d_stats.data_ptr(), d_cent.data_ptr(), d_nl.data_ptr()
    alg_bytes_step = int(L.vp_chain_algorithmic_bytes(C.byref(desc), C.byref(bufs), B))

    for _ in range(args.warmup):
        ctx.chain_run(desc, bufs, B)
    ctx.synchronize()

    # ---- timed region: exactly K steps, barrier + synchronize on both sides, MAX over ranks ------------
    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
    region_s = [timed_steps(lambda: ctx.chain_run(desc, bufs, B), sync, args.steps, dist, device="cpu" if args.rehearse_on_one_gpu else "cuda")
                for _ in range(max(1, args.regions))]
    elapsed = sorted(region_s)[len(region_s) // 2]          # median region (every region is exactly K steps)

    # ---- per-kernel attribution with HIP events on the launch stream (same K steps again) ----------
    ctx.profile_begin(args.steps * 16)
    for _ in range(args.steps):
        ctx.chain_run(desc, bufs, B)
    prof = ctx.profile_end()
    n_labels_seen = d_nl.cpu().numpy()

    fps = n_gpus * B * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- extra (not `value`): the same chain followed by cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) on the cleaned
    # mask of every frame (modules/red_buoy.py:36-38), batch resident in HBM, rank-local rate
    extras = {}
    if not args.no_extras:
        cdesc = _vp.make_contour_desc("cleaned", _vp.RETR_EXTERNAL, _vp.CHAIN_APPROX_SIMPLE, 64, 8192)
        carr = {"info": torch.zeros((B, 2), dtype=torch.int32, device="cuda"), "counts": torch.zeros((B, 64), dtype=torch.int32, device="cuda"),
                "offsets": torch.zeros((B, 64), dtype=torch.int32, device="cuda"), "is_hole": torch.zeros((B, 64), dtype=torch.uint8, device="cuda"),
                "points": torch.zeros((B, 8192, 2), dtype=torch.int32, device="cuda")}
        cb = _vp.ContourBuffers()
        for k_, v_ in carr.items():
            setattr(cb, k_, v_.data_ptr())
        ksteps = max(1, min(args.steps, 10))
        ctx.chain_run_contours(desc, bufs, cdesc, cb, B)
        sync()
        t0 = time.perf_counter()
        for _ in range(ksteps):
            ctx.chain_run_contours(desc, bufs, cdesc, cb, B)
        sync()
        dt = time.perf_counter() - t0
        info = carr["info"].cpu().numpy()
        extras["chain_plus_outer_contours"] = {"frames_per_s_per_gpu": round(B * ksteps / dt, 1), "ms_per_step": round(1e3 * dt / ksteps, 4),
                                               "steps": ksteps, "contours_per_frame": [int(info[:, 0].min()), int(info[:, 0].max())],
                                               "points_per_frame": [int(info[:, 1].min()), int(info[:, 1].max())]}

        # and the chain when the caller asks for the statistics only (no mask images, no label image): what a module that steers by
        # blob centroids needs; the masks stay bit-packed in HBM
        b2 = _vp.ChainBuffers()
        b2.bgr, b2.stats, b2.centroids, b2.nlabels = bufs.bgr, bufs.stats, bufs.centroids, bufs.nlabels
        ctx.chain_run(desc, b2, B)
        sync()
        t0 = time.perf_counter()
        for _ in range(ksteps):
            ctx.chain_run(desc, b2, B)
        sync()
        dt = time.perf_counter() - t0
        extras["chain_stats_only"] = {"frames_per_s_per_gpu": round(B * ksteps / dt, 1), "ms_per_step": round(1e3 * dt / ksteps, 4), "steps": ksteps,
                                      "outputs": "stats, centroids, nlabels"}

        # the other frame families of SURVEY 8d through the same entry point (not `value`): S2 = the bins chain
        # (modules/bins.py:13-27: BGR2HSV, inRange on three channels, OPEN 5x5, labelling) and S3 = uniform noise, once through the
        # bench chain and once straight into the labelling (grey >= 128, no morphology: about half the pixels set, tens of thousands
        # of components per frame, max_labels sized for them)
        def side_run(tag, gen, mode, lo, hi, mops, ml, what):
            nb = B if ml <= 4096 else min(B, 32)      # the noise-labelling case needs 36 B of tables per label and frame
            fr = [gen(rank * 1000 + i, W, H) for i in range(4)]
            d_in = torch.from_numpy(np.stack([fr[i % 4] for i in range(nb)])).cuda()
            st = torch.zeros((nb, ml, 5), dtype=torch.int32, device="cuda")
            ce = torch.zeros((nb, ml, 2), dtype=torch.float64, device="cuda")
            nl = torch.zeros((nb,), dtype=torch.int32, device="cuda")
            dsc = _vp.make_chain_desc(W, H, mode, lo, hi, mops, ccl=1, numbering=_vp.CCL_BLOCK2X2, max_labels=ml)
            bb = _vp.ChainBuffers()
            bb.bgr, bb.threshed, bb.cleaned, bb.labels = d_in.data_ptr(), d_thr.data_ptr(), d_cln.data_ptr(), d_lab.data_ptr()
            bb.stats, bb.centroids, bb.nlabels = st.data_ptr(), ce.data_ptr(), nl.data_ptr()
            ctx.chain_run(dsc, bb, nb)
            sync()
            t0 = time.perf_counter()
            for _ in range(ksteps):
                ctx.chain_run(dsc, bb, nb)
            sync()
            dt = time.perf_counter() - t0
            nls = nl.cpu().numpy()
            extras[tag] = {"frames_per_s_per_gpu": round(nb * ksteps / dt, 1), "ms_per_step": round(1e3 * dt / ksteps, 4), "steps": ksteps,
                           "frames_per_step": nb, "workload": what, "max_labels": ml, "labels_per_frame_seen": [int(nls.min()), int(nls.max())]}
        side_run("s2_bins_chain", F.s2_bins, _vp.BGR2HSV, (10, 20, 60), (30, 100, 255), [(_vp.MORPH_OPEN, 5, 5)], args.max_labels,
                 "S2 1080p: BGR->HSV inRange[10,20,60]-[30,100,255] -> OPEN 5x5 -> CCL + stats, all outputs")
        side_run("s3_noise_chain", F.s3_noise, _vp.BGR2LAB, (0, 150, 0), (255, 255, 255), morph, args.max_labels,
                 "S3 1080p uniform noise through the bench chain (the OPEN removes nearly every pixel)")
        side_run("s3_noise_labelling", F.s3_noise, _vp.BGR2GRAY, (128, 0, 0), (255, 255, 255), [], 131072,
                 "S3 1080p uniform noise: grey >= 128 -> CCL + stats with no morphology (worst-case component count)")

    # ---- the rates a module sees (not `value`; PCIe and Python inclusive): module bodies through the per-operator mirror, a module on
    # the runtime end to end, host-fed batches through the dispatcher - each with the bound it runs against, measured in this run
    if not args.no_extras and rank == 0 and n_gpus == 1:
        import contextlib
        import module_harness as MH
        try:
            with contextlib.redirect_stdout(sys.stderr):   # the modules announce themselves on stdout; stdout is for the one JSON line
                extras["process_body_red_buoy"] = MH.body_rates("buoy")
                extras["process_body_bins"] = MH.body_rates("bins")
                extras["runtime_e2e_red_buoy"] = MH.runtime_rate("buoy", seconds=3.0)
                # the reference's default mode: every post() published for the GUI (core/base.py:846-876; --enable-performance is opt-in)
                extras["runtime_e2e_red_buoy_posts_on"] = MH.runtime_rate("buoy", seconds=3.0, flags=())
                extras["runtime_e2e_bins"] = MH.runtime_rate("bins", seconds=2.0)
                # findContours of ONE image, the call modules/red_buoy.py:38 makes on the un-cleaned mask: clean and speckled masks
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import exp_contours_single
                extras["contours_single_image"] = dict(exp_contours_single.measure(100), what="cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) of one "
                                                       "1080p device image, contour list out: ms per call through vision.utils.feature.outer_contours, and of the C-ABI call alone")
                up = MH.pcie_upload_rate(ctx)
                for tag, (fw, fh, nb) in (("host_fed_1080p", (1920, 1080, 10)), ("host_fed_4k", (3840, 2160, 4))):
                    r = MH.host_fed_rate(fw, fh, batch=32, batches=nb, ring=4)
                    r["pcie_upload_GBps_measured_now"] = round(up, 2)
                    r["frac_of_pcie"] = round(r["pcie_GBps"] / up, 3)
                    extras[tag] = r
        except Exception as e:   # a side measurement must not take the headline down with it; the failure is reported as such
            extras["runtime_rates_error"] = repr(e)

    # N > 1: the host-fed leg on every rank at once (its own GPU, feeders on that GPU's NUMA node) - where scaling would bend first.
    # `value` stays the resident-batch number.
    if not args.no_extras and n_gpus > 1:
        try:
            chain = dict(color_mode=_vp.BGR2LAB, lo=(0, 150, 0), hi=(255, 255, 255), morph=morph, ccl=1, max_labels=args.max_labels, want=("stats",))
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):
                leg = host_fed_leg(dist, rank, n_gpus, local_rank, 3840, 2160, batch=32, batches=6, ring=4, chain=chain)
            if rank == 0:
                extras["host_fed_4k_all_ranks"] = dict(leg, what="config 4: 32 4K frames per batch from host memory, frames [lo, hi) of every batch "
                                                       "to rank r's GPU, statistics back; every rank at once after a barrier")
        except Exception as e:
            extras["host_fed_all_ranks_error"] = repr(e)

    # HBM-side bytes per launch (rocprofv3 PMC passes of THIS command with --no-extras --no-cpu-baseline: FETCH_SIZE and WRITE_SIZE in
    # separate runs, corrected as MI355X_MICROARCH.md prescribes, tools/pmc_traffic.py).  The file is named on the command line and is
    # used only if (a) it was collected from the very kernel sources this build was made from and (b) it holds exactly one
    # instantiation of the kernel that was timed - otherwise the field is null rather than a number of unknown provenance.
    traffic_tab, traffic_src = {}, None
    tpath = args.traffic_file if os.path.isabs(args.traffic_file) else os.path.join(ROOT, args.traffic_file)
    if args.batch == 128 and os.path.exists(tpath):
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location("vp_build", os.path.join(ROOT, "cuauv-vision-pipeline_amd", "build.py"))
            vb = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(vb)
            tab = json.load(open(tpath))
            if tab.get("_meta", {}).get("csrc_sha256") == vb.source_digest():
                traffic_tab, traffic_src = tab, os.path.relpath(tpath, ROOT)
            else:
                traffic_src = os.path.relpath(tpath, ROOT) + " (stale: collected from other kernel sources, ignored)"
        except Exception as e:   # evidence file unreadable: report no traffic
            traffic_src = f"{args.traffic_file}: {e}"

    def traffic_of(prof_name):
        # profile names are the kernels' base names; the file is keyed by full instantiation
        base = {"k_color_thresh": "k_color_thresh_flat", "k_morph_bits": "k_morph_bits"}.get(prof_name, prof_name)
        hits = [k for k in traffic_tab if k != "_meta" and (k == base or k.startswith(base + "<") or k.startswith(base + "_"))]
        return (hits[0], traffic_tab[hits[0]]["traffic_bytes_per_launch"]) if len(hits) == 1 else (None, None)

    roof = None
    kernels = {}
    for name, (ms, cnt) in prof.items():
        kernels[name] = {"avg_us": 1e3 * ms / cnt, "launches_per_step": cnt / args.steps}
    if prof:
        dom = max((k for k in prof if k in KERNEL_BYTES_PER_PX), key=lambda k: prof[k][0] / prof[k][1])
        avg_s = 1e-3 * prof[dom][0] / prof[dom][1]
        kb = KERNEL_BYTES_PER_PX[dom] * W * H * B
        achieved = kb / avg_s / 1e9
        inst, traffic = traffic_of(dom)
        kb8 = KERNEL_BYTES_PER_PX_8D[dom] * W * H * B
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "achieved_8d": round(kb8 / avg_s / 1e9, 1), "frac_8d": round(kb8 / avg_s / 1e9 / HBM_PEAK_GBS, 4),
                "bytes_per_px": KERNEL_BYTES_PER_PX[dom], "bytes_per_px_8d": KERNEL_BYTES_PER_PX_8D[dom],
                "traffic": traffic, "traffic_kernel": inst, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": int(kb), "avg_launch_us": round(1e6 * avg_s, 2),
                "chain_achieved_GBps": round(alg_bytes_step / (1e-3 * ms_per_step) / 1e9, 1),
                "chain_frac": round(alg_bytes_step / (1e-3 * ms_per_step) / 1e9 / HBM_PEAK_GBS, 4)}
        for k in kernels:
            inst_k, t_k = traffic_of(k)
            if t_k is not None:
                kernels[k]["measured_traffic_bytes_per_launch"] = t_k
                kernels[k]["instantiation"] = inst_k

    cpu = None
    real_cv2 = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        try:   # a real OpenCV (third-party, not reference code) gives the reference's own CPU path; never the facade of this repo
            import cv2 as _cv2
            if "vision" not in (getattr(_cv2, "__file__", "") or "") and hasattr(_cv2, "connectedComponentsWithStats"):
                real_cv2 = _cv2
        except Exception:
            real_cv2 = None
    if real_cv2 is not None:
        cv2 = real_cv2
        cv2.setNumThreads(1)
        k5 = cv2.getStructuringElement(cv2.MORPH_RECT, (5, 5))
        done, t_start = 0, time.perf_counter()
        while True:
            f = distinct[done % len(distinct)]
            lab = cv2.cvtColor(f, cv2.COLOR_BGR2LAB)                      # modules/red_buoy.py:21 via utils/color.py:21-22
            _, a, _ = cv2.split(lab)
            th = cv2.inRange(a, 150, 255)                                 # utils/color.py:121
            cl = cv2.morphologyEx(cv2.morphologyEx(th, cv2.MORPH_OPEN, k5), cv2.MORPH_CLOSE, k5)   # utils/transform.py:129,146
            cv2.connectedComponentsWithStats(cl, connectivity=8, ltype=cv2.CV_32S)
            done += 1
            if time.perf_counter() - t_start >= args.cpu_seconds:
                break
        dt = time.perf_counter() - t_start
        cpu = {"value": round(done / dt, 2), "unit": "frames/s", "cores": 1, "kind": "reference",
               "sample": f"{done} S1 1080p frames through cv2 {cv2.__version__} (cvtColor, split, inRange, morphologyEx x2, "
                         f"connectedComponentsWithStats; setNumThreads(1), {dt:.1f} s); host has {os.cpu_count()} cores"}
    elif rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        orc.lib()
        done, t_start = 0, time.perf_counter()
        while True:
            f = distinct[done % len(distinct)]
            orc.chain(f, orc.MODE_LAB, (0, 150, 0), (255, 255, 255), [orc.OPEN, orc.CLOSE], 5, 5, 2, args.max_labels)
            done += 1
            if time.perf_counter() - t_start >= args.cpu_seconds:
                break
        dt = time.perf_counter() - t_start
        cpu = {"value": round(done / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": f"{done} S1 1080p frames through oracle/vp_oracle.c orc_chain_u8 (scalar C, 1 thread, {dt:.1f} s); "
                         f"host has {os.cpu_count()} cores; cv2 absent so the reference's own cv2 path cannot be timed"}
        # the same port, frame-parallel on this GPU's share of the host (ctypes releases the GIL): SURVEY 8d asks for both figures
        from concurrent.futures import ThreadPoolExecutor
        nthr = max(1, min(16, os.cpu_count() or 1))
        per = max(2, int(args.cpu_seconds / 2 * cpu["value"]))

        def _work(k):
            for j in range(per):
                orc.chain(distinct[(k + j) % len(distinct)], orc.MODE_LAB, (0, 150, 0), (255, 255, 255), [orc.OPEN, orc.CLOSE], 5, 5, 2, args.max_labels)
            return per
        t_start = time.perf_counter()
        with ThreadPoolExecutor(nthr) as ex:
            total = sum(ex.map(_work, range(nthr)))
        dt = time.perf_counter() - t_start
        extras["cpu_baseline_frame_parallel"] = {"value": round(total / dt, 2), "unit": "frames/s", "cores": nthr, "kind": "port",
                                                 "sample": f"{total} frames, {nthr} threads x {per} frames each, {dt:.1f} s"}

    if rank == 0:
        out = {
            "metric": "frames/sec for color->threshold->morph->CCL at 1080p",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "regions": len(region_s), "region_ms_per_step": [round(1e3 * r / args.steps, 4) for r in region_s], "value_from": "median region",
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[1]: preprocessor+red_buoy fused chain, S1 synthetic 1080p frames resident in HBM: "
                                   "BGR->LAB(a) inRange[150,255] -> OPEN 5x5 -> CLOSE 5x5 -> 8-conn CCL + stats",
                       "width": W, "height": H, "frames_per_step_per_gpu": B, "sharding": "independent frames per rank, no collective",
                       "outputs": "threshold mask u8, cleaned mask u8, labels i32, stats/centroids", "max_labels": args.max_labels,
                       "labels_per_frame_seen": [int(n_labels_seen.min()), int(n_labels_seen.max())]},
            "algorithmic_bytes_per_frame": alg_bytes_step // B,
            "roofline": roof, "cpu_baseline": cpu, "kernels": kernels, "extras": extras,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
