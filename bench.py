#!/usr/bin/env python3
"""bench.py -- headline benchmark of the perfect-tandem-repeat scan on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hg38|chr22|chr22-real|chr1|random|hg38-random]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one complete scan of the workload for every motif size in [kmin,kmax]: two back-to-back launches on one stream
(the fused scan+verify kernel: rows sorted per 65536-position tile; the row gather: one compact array sorted by
(contig, start, end) + counters posted to the host), row count back on the host -- on a genome that is already packed and
resident in HBM when the timed region starts.  At N = 1 two scans are in flight
(prf_scan_genome_async / prf_scan_wait): scan i+1 is enqueued before scan i is collected; all K scans are collected --
row count on the host, checked -- inside the timed region (--no-pipeline: one at a time).

Workload (default = the configuration BASELINE.json's metric is quoted on): hg38 -- 25 contigs with the hg38
primary-assembly lengths (3 088 286 401 bp), motif sizes 1-50, min_repeats 3, min_span 9.  No genome FASTA exists offline, so
the contigs follow the stand-in recipe of colab-repeat-finder_amd/synth.py::standin2 (uniform background, N blocks at both
ends and a centromere-like gap, one planted perfect tandem repeat per 588 positions: ~1.8 k rows / Mbp like the reference's
golden chr22 BED), generated ON the device (prf_genome_standin) so nothing crosses PCIe.  Other workloads:
chr22 (BASELINE configs[1], 50 818 468 bp), chr1 (configs[2], 248 956 422 bp), random (configs[4]: ONE contig of 10^10 bp
uniform ACGT, seed 2026, motif 1-100), hg38-random (the hg38 lengths, uniform ACGT).

N > 1 is STRONG scaling on the same workload: one process per GPU, every rank holds the whole genome and scans its share
of it (multi_gpu.plan_parts: position ranges cut at tile multiples, balanced by tile cost; a row belongs to the share that
holds its first position, so no halo and no exchange), then the rows -- packed to 8 bytes each on the device
(prf_last_hits_packed_to_device) -- are concatenated on rank 0 with one padded RCCL gather per step, inside the timed region
(a communication thread issues the gathers from a ring of send buffers while the main thread scans on; every gather has
completed when the timed region ends).

Rank 0 prints ONE JSON line.  `value` is whole-job Gbp/s from the wall clock (max over ranks); the `roofline` object prices
the dominant kernel (prf_vscan_kernel) with the HIP events recorded around each of its launches in the timed region on the
library's stream (read after the loop, prf_scan_timings_split); `cpu_baseline` is the CPU oracle (a C restatement of the
reference, oracle/prf_oracle.c) timed on a bounded sample of the same workload on this host: one thread, and all host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "colab-repeat-finder_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

HG38_LENS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
             133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285,
             58617616, 64444167, 46709983, 50818468, 156040895, 57227415, 16569]   # chr1-22, X, Y, M

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves


def pmc_record(workload, kmin, kmax):
    """The committed rocprofv3 PMC measurement of this workload (profiles/pmc_traffic.json): HBM bytes per launch of the
    scan kernel (and of the row gather) -- FETCH_SIZE and WRITE_SIZE collected in separate runs, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950 -- and the share of the VALU issue cycles the scan kernel uses.  A constant
    looked up in a committed file, not a live counter (rocprofv3 cannot run inside the timed process); None if no
    committed measurement matches this workload."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            for rec in json.load(f):
                if rec.get("workload") == workload and rec["kmin"] == kmin and rec["kmax"] == kmax:
                    return rec
    except (ValueError, KeyError):
        pass
    return None


def _oracle_chunk(job):
    from oracle import prf_oracle
    seq, kmin, kmax, r, span = job
    t0 = time.perf_counter()
    n = len(prf_oracle.detect_rows(seq, kmin, kmax, r, span))
    return n, time.perf_counter() - t0


def _usable_cpus():
    """The cpus this process may run on, cut to the cgroup's CPU quota if there is one (a one-GPU share of a box shows all 256
    hardware threads in its affinity mask but is granted 16 cpus' worth of time: 256 workers would time-share them)."""
    cpus = sorted(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, -(-int(quota) // int(period)))
            phys = _physical_cores(cpus)
            cpus = (phys if len(phys) >= n else cpus)[:n]
    except (OSError, ValueError):
        pass
    return cpus


def _quota_text():
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            return f.read().strip()
    except OSError:
        return "n/a"


def _physical_cores(cpus):
    """one hardware thread per physical core among `cpus` (thread_siblings_list of sysfs; all of them if that is unreadable)"""
    seen, out = set(), []
    for c in cpus:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                sib = f.read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            out.append(c)
    return out


def _many_cores(sample, cpus, settings, rounds=3):
    """Every cpu of `cpus` scans its own window of the sample at the same time, `rounds` times; the median wall time of a
    round.  Reproducible: one worker process per cpu, pinned to it (os.sched_setaffinity), started ONCE;
    the windows are cut before the fork; a round runs from a barrier every worker has reached to a barrier every worker has
    reached again, so that starting the processes and their scheduling jitter are not part of the rate."""
    import multiprocessing as mp
    n = len(cpus)
    win = max(1, len(sample) // 8)                                   # an eighth of the sample: ~0.7 s of work
    starts = [(len(sample) - win) * i // max(1, n - 1) for i in range(n)]
    jobs = [(sample[st:st + win],) + settings for st in starts]     # windows spread evenly over the sample, overlapping
    mpc = mp.get_context("fork")
    gate_in, gate_out = mpc.Barrier(n + 1), mpc.Barrier(n + 1)

    def worker(i):
        try:
            os.sched_setaffinity(0, {cpus[i]})
        except OSError:
            pass
        for _ in range(rounds):
            gate_in.wait()
            _oracle_chunk(jobs[i])
            gate_out.wait()

    procs = [mpc.Process(target=worker, args=(i,)) for i in range(n)]
    for p in procs:
        p.start()
    walls = []
    try:
        for _ in range(rounds):
            gate_in.wait(timeout=300)
            t0 = time.perf_counter()
            gate_out.wait(timeout=600)
            walls.append(time.perf_counter() - t0)
    finally:
        for p in procs:
            p.join(timeout=60)
    if len(walls) != rounds or any(p.exitcode != 0 for p in procs):
        return None
    wall = sorted(walls)[rounds // 2]
    return {"value": n * win / wall / 1e9, "unit": "Gbp/s", "cores": n,
            "note": f"{n} pinned processes of the oracle, one {win} bp window of the sample each, median of {rounds} rounds "
                    f"({', '.join(f'{w:.2f}' for w in walls)} s), a round = barrier to barrier"}


def cpu_baseline(sample, kmin, kmax, min_repeats, min_span, what):
    """Oracle (kind 'port': C restatement of the reference's state machine) on a bounded sample: one thread on the
    whole sample, then many cores on windows of it at the same time -- one process per hardware thread, and one per physical
    core (the reference's own parallel strategy is independent interval jobs, one CPU each:
    hail_batch_pipeline/run_hail_batch_pipeline.py:101)."""
    settings = (kmin, kmax, min_repeats, min_span)
    n_rows, dt = _oracle_chunk((sample,) + settings)
    cpus = _usable_cpus()
    out = {"value": len(sample) / dt / 1e9, "unit": "Gbp/s", "cores": 1, "kind": "port",
           "sample": f"{what}, motif {kmin}-{kmax}, {n_rows} rows, {dt:.1f} s single-thread C oracle (the pure-Python "
                     f"reference runs ~0.018 Mbp/s at motif 1-50, BASELINE.md)"}
    if len(cpus) > 1:
        res = _many_cores(sample, cpus, settings)
        if res:
            out["all_cores"] = res
            res["note"] += f" (affinity mask {len(os.sched_getaffinity(0))} cpus, cgroup quota {_quota_text()})"
        phys = _physical_cores(cpus)
        if 1 < len(phys) < len(cpus):
            res = _many_cores(sample, phys, settings)
            if res:
                out["physical_cores"] = res
    return out


def build_workload(args, ctx, synth):
    """-> (genome, lens, name, description, host_sample(fn or None))"""
    w = args.workload
    if w == "hg38":
        lens = HG38_LENS
        seeds = [1000 + i for i in range(len(lens))]
        g = ctx.standin(lens, seeds, args.kmax)
        desc = (f"hg38-shaped genome: 25 contigs with the hg38 primary-assembly lengths ({sum(lens)} bp), stand-in recipe "
                "(synth.standin2: uniform background, N blocks, one planted repeat per 588 positions) generated on the device")
        sample = lambda n: synth.standin2(lens[0], seeds[0], 10_000, n).tobytes()
        return g, lens, desc, sample, "first %d bp behind the leading N block of contig 0"
    if w == "chr1":
        lens = [synth.CHR1_LEN if not args.length else args.length]
        g = ctx.standin(lens, [1], args.kmax)
        desc = f"hg38 chr1-sized contig ({lens[0]} bp), stand-in recipe (synth.standin2) generated on the device"
        return g, lens, desc, (lambda n: synth.standin2(lens[0], 1, 10_000, n).tobytes()), "first %d bp behind the leading N block"
    if w == "chr22":
        length = args.length or synth.CHR22_LEN
        n_head = 10_510_000 if length >= 20_000_000 else length // 10
        seq = synth.chr_standin(length=length, seed=22, n_head=n_head, n_tail=min(10_000, length // 100)).tobytes()
        g = ctx.load([seq], args.kmax)
        desc = f"chr22-sized synthetic stand-in contig ({length} bp; hg38-like N blocks, planted repeats; synth.chr_standin)"
        return g, [length], desc, (lambda n: seq[n_head:n_head + n]), "first %d non-N bp"
    if w == "chr22-real":
        arr, planted = synth.chr22_real()
        seq = arr.tobytes()
        g = ctx.load([seq], args.kmax)
        desc = (f"chr22 stand-in ({len(seq)} bp) with every cluster of the reference's golden chr22 BED planted at its real coordinate "
                f"(synth.chr22_real: {len(planted)} golden rows, up to 759 rows in one 65536-position tile)")
        return g, [len(seq)], desc, (lambda n: seq[10_510_000:10_510_000 + n]), "first %d non-N bp"
    if w == "random":
        lens = [args.length or 10_000_000_000]
        g = ctx.synth(lens, [2026], args.kmax)                     # generated in HBM, nothing crosses PCIe
        desc = f"ONE contig of {lens[0]} bp uniform random ACGT (seed 2026), generated on the device"
        from oracle import prf_oracle
        return g, lens, desc, (lambda n: prf_oracle.synth(n, 2026)), "first %d bp"
    lens = HG38_LENS                                                # hg38-random
    seeds = [1000 + i for i in range(len(lens))]
    g = ctx.synth(lens, seeds, args.kmax)
    desc = f"25 contigs with the hg38 primary-assembly lengths ({sum(lens)} bp) of uniform random ACGT generated on the device"
    from oracle import prf_oracle
    return g, lens, desc, (lambda n: prf_oracle.synth(n, seeds[0])), "first %d bp of contig 0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["hg38", "chr22", "chr22-real", "chr1", "random", "hg38-random"], default="hg38")
    ap.add_argument("--length", type=int, default=0, help="contig length for chr22 / chr1 / random (default: the config's own)")
    ap.add_argument("--kmin", type=int, default=1)
    ap.add_argument("--kmax", type=int, default=0, help="default 50 (100 for --workload random: BASELINE configs[4])")
    ap.add_argument("--min-repeats", type=int, default=3)
    ap.add_argument("--min-span", type=int, default=9)
    ap.add_argument("--cpu-sample-bp", type=int, default=12_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--generic", action="store_true", help="force the generic kernel")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="N=1: wait for every scan before the next is enqueued (default: two scans in flight)")
    ap.add_argument("--preroll-ms", type=float, default=300.0,
                    help="untimed back-to-back scans in front of the W warmup steps, until this much wall time has passed: the chip "
                         "raises its clock over the first tens of milliseconds of sustained load (DVFS) and the scan is VALU-bound -- "
                         "per-step kernel time falls from 0.66 to 0.53 ms over 40 steps from an idle GPU.  0: none")
    ap.add_argument("--gather-every", type=int, default=1, help="N>1: gather the rows to rank 0 every this many steps (0: never)")
    args = ap.parse_args()
    if not args.kmax:
        args.kmax = 100 if args.workload == "random" else 50

    import numpy as np
    import torch
    import prf_native
    import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # Rehearsal knobs (never used by the driver): PRF_BENCH_BACKEND=gloo + PRF_BENCH_ONE_GPU=1 run N ranks on ONE GPU
    # with CPU tensors for the collectives, to exercise the N>1 code path on a single-GPU box.
    backend = os.environ.get("PRF_BENCH_BACKEND", "nccl")
    one_gpu = os.environ.get("PRF_BENCH_ONE_GPU") == "1"
    dev_index = 0 if one_gpu else local_rank
    tdev = "cpu" if backend == "gloo" else "cuda"
    dist = None
    torch.cuda.set_device(dev_index)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    ctx = prf_native.Context(dev_index)
    genome, lens, desc, sample_fn, sample_what = build_workload(args, ctx, synth)
    total_bp = sum(lens)
    flags = prf_native.SCAN_FORCE_GENERIC if args.generic else prf_native.SCAN_DEFER_TIMING
    scan = lambda fetch: genome.scan(args.kmin, args.kmax, args.min_repeats, args.min_span, flags=flags, fetch=fetch)

    rows_whole = None
    shares = None
    if world > 1:
        import multi_gpu
        if rank == 0:                                   # untimed: the whole scan on one GPU, to check the gathered rows against
            rows_whole, _ = scan(True)
        tile = prf_native.tile_positions()
        classes = [genome.tile_classes(c) for c in range(len(lens))]
        shares = multi_gpu.plan_parts(lens, world, tile, classes)
        genome.select(shares[rank])

    # one untimed full scan of this rank's share: sizes the scratch buffers, gives the row count used to size the gather
    rows, st0 = scan(True)
    n_rows_local = len(rows)
    import hashlib
    rows_sha256 = hashlib.sha256(np.ascontiguousarray(rows).tobytes()).hexdigest()   # pinned by tests/test_gpu_parity.py for the default workload
    my_bp = int(st0.positions)
    gather_cap = None
    if world > 1:
        import queue
        import threading
        cap = torch.tensor([n_rows_local], device=tdev, dtype=torch.int64)
        dist.all_reduce(cap, op=dist.ReduceOp.MAX)
        gather_cap = int(cap.item())
        SIDE = 1024                                                          # rows longer than 65 534 bp travel whole
        n_words = gather_cap + 1 + 3 * SIDE                                  # 8-byte wire rows + count word + side list
        bases = genome.contig_bases()
        # The gather of step i runs on a communication thread while the main thread scans step i+1, i+2, ...:
        # NBUF send buffers cycle between the two (the scan and the collectives release the GIL).
        NBUF = 4
        send_devs = [torch.zeros(n_words, dtype=torch.int64, device="cuda") for _ in range(NBUF)]
        sends = send_devs if tdev == "cuda" else [torch.zeros(n_words, dtype=torch.int64) for _ in range(NBUF)]
        recvs = [[torch.zeros_like(sends[0]) for _ in range(world)] if rank == 0 else None for _ in range(NBUF)]
        free_q, work_q = queue.Queue(), queue.Queue()
        for b in range(NBUF):
            free_q.put(b)
        comm_state = {"last": None, "error": None, "gather_s": [], "n": 0}

        comm_stream = torch.cuda.Stream()      # the send buffers are handed to it on the device (prf_stream_wait_for): no host wait

        def comm_loop():
            try:
                torch.cuda.set_device(dev_index)
                with torch.cuda.stream(comm_stream):
                    while True:
                        item = work_q.get()
                        if item is None:
                            return
                        b, do_gather = item
                        if do_gather:
                            t0 = time.perf_counter()
                            if sends[b] is not send_devs[b]:
                                sends[b].copy_(send_devs[b])                 # gloo rehearsal only
                            dist.gather(sends[b], recvs[b], dst=0)
                            comm_stream.synchronize()                        # the buffer may be refilled now
                            comm_state["gather_s"].append(time.perf_counter() - t0)
                            comm_state["last"] = b
                        free_q.put(b)
            except BaseException as exc:                                     # surfaces in fence()
                comm_state["error"] = exc
                free_q.put(-1)

        comm_thread = threading.Thread(target=comm_loop, name="prf-gather", daemon=True)
        comm_thread.start()

    def take_buffer():
        b = free_q.get()
        if b < 0:
            raise RuntimeError("gather thread failed") from comm_state["error"]
        return b

    step_no = [0]

    def step():
        """one synchronous step (--no-pipeline, the generic kernel): the host waits for the scan and, at N > 1, for the pack"""
        _, st = scan(False)
        if world > 1:
            step_no[0] += 1
            if args.gather_every > 0 and step_no[0] % args.gather_every == 0:
                b = take_buffer()
                ctx.last_hits_packed_to_device(genome, send_devs[b].data_ptr(), gather_cap, SIDE)
                work_q.put((b, True))
        return st

    def enqueue_sharded():
        """N > 1: one scan of this rank's share enqueued, its rows packed to 8-byte wire rows on the library's stream behind it
        (no host step in between), the send buffer handed to the communication stream by an event: the host neither waits for
        the scan nor for the pack.  Returns the scan's serial number (collected one step later, like at N = 1)."""
        step_no[0] += 1
        do_gather = args.gather_every > 0 and step_no[0] % args.gather_every == 0
        if not do_gather:
            return genome.scan_async(args.kmin, args.kmax, args.min_repeats, args.min_span)
        b = take_buffer()                                                    # blocks only if all NBUF gathers are pending
        seq = genome.scan_async_packed(args.kmin, args.kmax, args.min_repeats, args.min_span, send_devs[b].data_ptr(), gather_cap, SIDE)
        ctx.stream_wait_for(comm_stream.cuda_stream)
        work_q.put((b, True))
        return seq

    def fence():
        if world > 1:
            held = [take_buffer() for _ in range(NBUF)]                      # every gather issued so far has completed
            for b in held:
                free_q.put(b)
            dist.barrier()
        torch.cuda.synchronize()

    pipelined = st0.path == 1 and not args.generic and not args.no_pipeline and (world == 1 or args.kmax <= 511)

    def run_pipelined(n):
        """n steps with two scans in flight: scan i+1 is enqueued before scan i is collected, so its launch and the
        host's share overlap the kernels of scan i.  Every scan is collected (row count checked) inside the call."""
        out, pending = [], None
        for _ in range(n):
            s = enqueue_sharded() if world > 1 else genome.scan_async(args.kmin, args.kmax, args.min_repeats, args.min_span)
            if pending is not None:
                out.append(ctx.scan_wait(pending))
            pending = s
        if pending is not None:
            out.append(ctx.scan_wait(pending))
        assert all(int(st.n_hits) == n_rows_local for st in out)
        return out

    # clock ramp (see --preroll-ms): the same steps as the timed ones, untimed, until the GPU has been busy for a while
    preroll_steps = 0
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.preroll_ms:
        if pipelined:
            run_pipelined(8)
        else:
            for _ in range(8):
                step()
        preroll_steps += 8
    preroll_ms = (time.perf_counter() - t_pre) * 1e3
    if pipelined:
        run_pipelined(args.warmup)
    else:
        for _ in range(args.warmup):
            step()
    fence()
    if world > 1:
        comm_state["gather_s"].clear()
    seqs = []
    p1_ms, p2_ms = [], []
    t0 = time.perf_counter()
    if pipelined:
        seqs = [st.seq for st in run_pipelined(args.steps)]
    else:
        for _ in range(args.steps):
            st = step()
            p1_ms.append(st.phase1_ms)
            p2_ms.append(st.phase2_ms)
            seqs.append(st.seq)
    fence()
    elapsed = time.perf_counter() - t0
    scan_ms = gather_kernel_ms = None
    if st0.path == 1 and st0.n_launches:
        # fused path: kernel durations from the HIP events recorded around each launch of the timed region (the
        # last TIMING_RING steps if there were more)
        tail = seqs[-prf_native.TIMING_RING:]
        scan_ms, gather_kernel_ms = ctx.scan_timings_split(tail[0], len(tail))
        p1 = float(np.mean(scan_ms))
    else:
        p1 = float(np.mean(p1_ms)) if p1_ms else 0.0
    kernel_ms_ranks = [p1]
    both_ms = p1 + (float(np.mean(gather_kernel_ms)) if gather_kernel_ms is not None else 0.0)   # scan + row gather kernels of this rank
    both_ms_max = both_ms
    if world > 1:
        t = torch.tensor([elapsed], device=tdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([n_rows_local], device=tdev, dtype=torch.int64)
        dist.all_reduce(tot)
        n_rows_total = int(tot.item())
        km = [torch.zeros(1, device=tdev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(km, torch.tensor([p1], device=tdev, dtype=torch.float64))
        kernel_ms_ranks = [float(x.item()) for x in km]
        bm = torch.tensor([both_ms], device=tdev, dtype=torch.float64)
        dist.all_reduce(bm, op=dist.ReduceOp.MAX)
        both_ms_max = float(bm.item())
    else:
        n_rows_total = n_rows_local

    gathered_ok = None
    if world > 1 and rank == 0 and comm_state["last"] is not None:
        # the last gather must hold every rank's rows; the shares are in genome order, so their concatenation IS the
        # whole scan's sorted row array
        recv = recvs[comm_state["last"]]
        parts = [multi_gpu.unpack_rows(r.cpu().numpy(), gather_cap, SIDE, bases, prf_native.tile_positions()) for r in recv]
        got = np.concatenate(parts)
        gathered_ok = bool(len(got) == n_rows_total and np.array_equal(got, rows_whole))
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_bp / (elapsed / args.steps) / 1e9
        # algorithmic bytes per launch of the dominant kernel: the 2-bit input once for all k,
        # plus the 24-byte rows -- of this rank's share
        bytes_alg = (my_bp + 3) // 4 + 24 * n_rows_local
        # t_scan of SURVEY 8(d): first scan-kernel start -> compacted rows ready = scan kernel + row gather (HIP events on the
        # library's stream around every launch of the timed region); the scan kernel alone is the labelled secondary
        t_scan = both_ms if st0.path == 1 else p1
        achieved = bytes_alg / (t_scan * 1e-3) / 1e9 if t_scan else 0.0
        achieved_scan_only = bytes_alg / (p1 * 1e-3) / 1e9 if p1 else 0.0
        hbm_meas = ctx.measure_hbm_read(1 << 30, 5)
        pmc = pmc_record(args.workload, args.kmin, args.kmax) if (st0.path == 1 and world == 1) else None
        out = {
            "metric": f"Gbp/s scanned (motif {args.kmin}-{args.kmax})", "value": round(value, 4), "unit": "Gbp/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32 bit planes (2-bit bases)",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}; motif {args.kmin}-{args.kmax}, min_repeats {args.min_repeats}, "
                                   f"min_span {args.min_span}; genome packed + resident in HBM before the timed region",
                       "positions_total": total_bp, "positions_rank0": my_bp,
                       "kernel_path": "generic" if st0.path == 0 else "vertical",
                       "rows_rank0": n_rows_local, "rows_total": n_rows_total,
                       "rows_sorted_on_device": bool(st0.sorted_on_device),
                       "rows_sha256_rank0": rows_sha256, "launches_of_the_untimed_scan": int(st0.n_launches),
                       "candidate_records_rank0": int(st0.n_candidates),
                       "steps_in_flight": 2 if pipelined else 1,
                       "clock_ramp": {"untimed_steps_before_warmup": preroll_steps, "ms": round(preroll_ms, 1),
                                      "why": "DVFS: from an idle GPU the per-step kernel time falls for the first ~40 back-to-back steps "
                                             "(roofline.kernel_ms_per_step shows what is left of that); --preroll-ms 0 turns it off"},
                       # what the host adds to a step beyond the slowest rank's two kernels (launches, polling, hand-off to the gather)
                       "host_overhead_ms_per_step": round(ms_per_step - both_ms_max, 5),
                       "multi_gpu": ({"sharding": "every rank holds the genome and scans its share of the tiles (prf_genome_select); "
                                                  "no data-path collective",
                                      "gather": f"one padded RCCL gather of 8-byte wire rows (prf_last_hits_packed_to_device) to rank 0 every "
                                                f"{args.gather_every} step(s), overlapped with the following scans",
                                      "gather_verified": gathered_ok,
                                      "gather_ms_mean": round(float(np.mean(comm_state["gather_s"])) * 1e3, 4) if comm_state["gather_s"] else None,
                                      "gather_rows_per_rank_max": gather_cap, "gather_bytes_per_rank": 8 * n_words,
                                      "scan_kernel_ms_per_rank": [round(x, 5) for x in kernel_ms_ranks]} if world > 1 else "n/a")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5),
                         "traffic": (pmc["hbm_bytes_per_launch"] + pmc.get("gather_hbm_bytes_per_launch", 0)) if pmc else None,
                         "traffic_source": ("profiles/pmc_traffic.json: " + pmc.get("note", "")) if pmc else None,
                         "valu_busy": pmc.get("valu_busy") if pmc else None,
                         "kernel": ("prf_vscan_kernel (fused scan + verify + per-tile sorted rows) + prf_vgather_kernel (slabs -> one sorted "
                                    "array): t_scan of SURVEY 8(d)") if st0.path == 1 else "prf_scan_generic_kernel",
                         "t_scan_ms": round(t_scan, 5),
                         "kernel_ms": round(p1, 5),
                         "kernel_ms_min_median": [round(float(np.min(scan_ms)), 5), round(float(np.median(scan_ms)), 5)] if scan_ms is not None else None,
                         "kernel_ms_per_step": [round(float(x), 4) for x in scan_ms] if scan_ms is not None else None,
                         "gather_kernel_ms": round(float(np.mean(gather_kernel_ms)), 5) if gather_kernel_ms is not None else None,
                         "scan_kernel_only": {"achieved": round(achieved_scan_only, 2), "frac": round(achieved_scan_only / HBM_PEAK_GBPS, 5)},
                         "algorithmic_bytes_per_launch": bytes_alg,
                         # the same counting only the tiles that are launched (tiles of nothing but N are skipped)
                         "achieved_nonN": round(((int(st0.tiles_launched) * 65536) // 4 + 24 * n_rows_local) / (t_scan * 1e-3) / 1e9, 2) if t_scan else None,
                         "measured_hbm_read_GBps": round(hbm_meas, 1),
                         "frac_of_measured_read": round(achieved / hbm_meas, 5)},
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is timed on rank 0 at N=1 only
            n = min(args.cpu_sample_bp, lens[0] - 20_000) if lens[0] > 40_000 else lens[0]
            out["cpu_baseline"] = cpu_baseline(sample_fn(n), args.kmin, args.kmax, args.min_repeats, args.min_span,
                                               (sample_what % n) + " of the workload")
        print(json.dumps(out), flush=True)
    genome.free()
    ctx.close()
    if world > 1:
        work_q.put(None)
        comm_thread.join()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
