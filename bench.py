#!/usr/bin/env python3
"""bench.py -- headline benchmark of the perfect-tandem-repeat scan on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one complete scan of the workload for every motif size in [kmin,kmax]: ONE kernel launch (scan +
verify + compaction of the rows into one array in HBM + counters), row count back on the host -- on a genome
that is already packed and resident in HBM when the timed region starts.  At N = 1 two scans are in flight (prf_scan_genome_async / prf_scan_wait): scan i+1 is
enqueued before scan i is collected, so its launch and the host's share overlap the kernel of scan i; all K scans
are collected -- row count on the host, checked -- inside the timed region (--no-pipeline: one at a time).
Workload (BASELINE.json configs[1]): a
chr22-sized contig (50 818 468 bp), motif sizes 1-50, min_repeats 3, min_span 9.  No genome FASTA exists
offline, so the contig is the synthetic stand-in of colab-repeat-finder_amd/synth.py (hg38-like N blocks,
~1.8 k planted repeats per Mbp, uniform ACGT elsewhere).

N > 1: one process per GPU; every rank scans its own chr22-sized contig (seed 22 + rank; weak scaling, no
data-path collective) and the rows are then concatenated on rank 0 with one padded RCCL gather, inside
the timed region (a communication thread issues the gathers from a ring of 4 send buffers while the main
thread scans on: the gather of step i overlaps the scans of the following steps; every gather has completed
when the timed region ends).

Rank 0 prints ONE JSON line.  `value` is whole-job Gbp/s from the wall clock (max over ranks); the
`roofline` object prices the dominant kernel with the HIP events recorded around each of its launches in
the timed region on the library's stream (read after the loop, prf_scan_timings); `cpu_baseline` is the CPU oracle (a C restatement of the reference, oracle/prf_oracle.c) timed on
a bounded sample of the same workload on this host.
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


def pmc_traffic(length, kmin, kmax):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950);
    None if no committed measurement matches this workload."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            for rec in json.load(f):
                if rec["length"] == length and rec["kmin"] == kmin and rec["kmax"] == kmax:
                    return rec["hbm_bytes_per_launch"]
    except (ValueError, KeyError):
        pass
    return None


def cpu_baseline(seq_bytes, kmin, kmax, min_repeats, min_span, sample_bp):
    """Oracle (kind 'port': C restatement of the reference's state machine), 1 thread, bounded sample."""
    from oracle import prf_oracle
    n_lead = len(seq_bytes) - len(seq_bytes.lstrip(b"N"))
    sample = seq_bytes[n_lead:n_lead + sample_bp]
    t0 = time.perf_counter()
    rows = prf_oracle.detect_rows(sample, kmin, kmax, min_repeats, min_span)
    dt = time.perf_counter() - t0
    return {"value": len(sample) / dt / 1e9, "unit": "Gbp/s", "cores": 1, "kind": "port",
            "sample": f"first {len(sample)} non-N bp of the workload, motif {kmin}-{kmax}, {len(rows)} rows, {dt:.1f} s "
                      f"single-thread C oracle (the pure-Python reference runs ~0.018 Mbp/s at motif 1-50, BASELINE.md)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--length", type=int, default=0, help="contig length per GPU (default: chr22, 50 818 468)")
    ap.add_argument("--kmin", type=int, default=1)
    ap.add_argument("--kmax", type=int, default=50)
    ap.add_argument("--min-repeats", type=int, default=3)
    ap.add_argument("--min-span", type=int, default=9)
    ap.add_argument("--cpu-sample-bp", type=int, default=12_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--generic", action="store_true", help="force the generic kernel")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="N=1: wait for every scan before the next is enqueued (default: two scans in flight, "
                         "prf_scan_genome_async / prf_scan_wait)")
    ap.add_argument("--workload", choices=["chr22", "random", "hg38"], default="chr22",
                    help="chr22: the default stand-in contig (BASELINE config C2, the headline); random: uniform ACGT "
                         "generated on the device (config C5 is --workload random --length 1250000000 --kmax 100); "
                         "hg38: 25 contigs with the hg38 primary-assembly lengths (3.09 Gbp of random ACGT generated "
                         "on the device), dealt to the ranks longest first (config C4's shape; strong scaling)")
    args = ap.parse_args()

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

    length = args.length or synth.CHR22_LEN
    n_head = 10_510_000 if length >= 20_000_000 else length // 10
    ctx = prf_native.Context(dev_index)
    total_bp = length * world
    if args.workload == "hg38":
        import multi_gpu
        mine = multi_gpu.plan_contig_shards(HG38_LENS, world)[rank]          # what scan_contigs_sharded() does
        genome = ctx.synth([HG38_LENS[i] for i in mine], [1000 + i for i in mine], args.kmax)
        seq = None
        length = sum(HG38_LENS[i] for i in mine)                             # this rank's positions
        total_bp = sum(HG38_LENS)
    elif args.workload == "random":
        genome = ctx.synth([length], [22 + rank], args.kmax)        # generated in HBM, nothing crosses PCIe
        seq = None
    else:
        seq = synth.chr_standin(length=length, seed=22 + rank, n_head=n_head, n_tail=min(10_000, length // 100)).tobytes()
        genome = ctx.load([seq], args.kmax)
    # the HIP events of every scan are read after the timed loop (prf_scan_timings), not waited for inside it
    flags = prf_native.SCAN_FORCE_GENERIC if args.generic else prf_native.SCAN_DEFER_TIMING
    scan = lambda fetch: genome.scan(args.kmin, args.kmax, args.min_repeats, args.min_span, flags=flags, fetch=fetch)

    # one untimed full scan: sizes the scratch buffers, gives the row count used to size the gather
    rows, st0 = scan(True)
    n_rows_local = len(rows)
    gather_cap = None
    if world > 1:
        import queue
        import threading
        cap = torch.tensor([n_rows_local], device=tdev, dtype=torch.int64)
        dist.all_reduce(cap, op=dist.ReduceOp.MAX)
        gather_cap = int(cap.item()) + 1                                     # 24-byte rows + one count record
        # The gather of step i runs on a communication thread while the main thread scans step i+1, i+2, ...:
        # NBUF send buffers cycle between the two (the scan and the collectives release the GIL).
        NBUF = 4
        send_devs = [torch.zeros((gather_cap, 3), dtype=torch.int64, device="cuda") for _ in range(NBUF)]
        sends = send_devs if tdev == "cuda" else [torch.zeros((gather_cap, 3), dtype=torch.int64) for _ in range(NBUF)]
        recvs = [[torch.zeros_like(sends[0]) for _ in range(world)] if rank == 0 else None for _ in range(NBUF)]
        free_q, work_q = queue.Queue(), queue.Queue()
        for b in range(NBUF):
            free_q.put(b)
        comm_state = {"last": None, "error": None}

        def comm_loop():
            try:
                torch.cuda.set_device(dev_index)
                comm_stream = torch.cuda.Stream()
                with torch.cuda.stream(comm_stream):
                    while True:
                        b = work_q.get()
                        if b is None:
                            return
                        if sends[b] is not send_devs[b]:
                            sends[b].copy_(send_devs[b])                     # gloo rehearsal only
                        dist.gather(sends[b], recvs[b], dst=0)
                        comm_stream.synchronize()                            # the buffer may be refilled now
                        comm_state["last"] = b
                        free_q.put(b)
            except BaseException as exc:                                     # surfaces in fence()
                comm_state["error"] = exc
                free_q.put(-1)

        comm_thread = threading.Thread(target=comm_loop, name="prf-gather", daemon=True)
        comm_thread.start()
        in_flight = [0]

    def take_buffer():
        b = free_q.get()
        if b < 0:
            raise RuntimeError("gather thread failed") from comm_state["error"]
        return b

    def step():
        if world > 1:
            b = take_buffer()                                                # blocks only if all NBUF gathers are pending
            ctx.set_row_sink(send_devs[b].data_ptr(), gather_cap - 1)       # the kernel compacts the rows straight into the
            _, st = scan(False)                                             # send buffer and writes the count record
            work_q.put(b)
        else:
            _, st = scan(False)
        return st

    def fence():
        if world > 1:
            held = [take_buffer() for _ in range(NBUF)]                      # every gather issued so far has completed
            for b in held:
                free_q.put(b)
            dist.barrier()
        torch.cuda.synchronize()

    pipelined = world == 1 and st0.path == 1 and not args.generic and not args.no_pipeline

    def run_pipelined(n):
        """n steps with two scans in flight: scan i+1 is enqueued before scan i is collected, so its launch and the
        host's share overlap the kernel of scan i.  Every scan is collected (row count checked) inside the call."""
        out, pending = [], None
        for _ in range(n):
            s = genome.scan_async(args.kmin, args.kmax, args.min_repeats, args.min_span)
            if pending is not None:
                out.append(ctx.scan_wait(pending))
            pending = s
        if pending is not None:
            out.append(ctx.scan_wait(pending))
        assert all(int(st.n_hits) == n_rows_local for st in out)
        return out

    if pipelined:
        run_pipelined(args.warmup)
    else:
        for _ in range(args.warmup):
            step()
    fence()
    p1_ms, p2_ms, seqs = [], [], []
    t0 = time.perf_counter()
    if pipelined:
        for st in run_pipelined(args.steps):
            p1_ms.append(0.0)
            p2_ms.append(0.0)
            seqs.append(st.seq)
    else:
        for _ in range(args.steps):
            st = step()
            p1_ms.append(st.phase1_ms)
            p2_ms.append(st.phase2_ms)
            seqs.append(st.seq)
    fence()
    elapsed = time.perf_counter() - t0
    if st0.path == 1:
        # fused path: kernel durations from the HIP events recorded around each launch of the timed region (the
        # last TIMING_RING steps if there were more)
        tail = seqs[-prf_native.TIMING_RING:]
        p1_ms = ctx.scan_timings(tail[0], len(tail))
        p2_ms = [0.0]
    if world > 1:
        t = torch.tensor([elapsed], device=tdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([n_rows_local], device=tdev, dtype=torch.int64)
        dist.all_reduce(tot)
        n_rows_total = int(tot.item())
    else:
        n_rows_total = n_rows_local

    gathered_ok = None
    if world > 1 and rank == 0:
        # the last gather must hold every rank's rows: counts add up and rank 0's part equals its own fetched rows
        recv = recvs[comm_state["last"]]
        counts = [int(r[gather_cap - 1, 0].item()) for r in recv]
        mine = recv[0][:counts[0]].cpu().numpy().view(np.uint8).reshape(-1, 24)
        ref = np.ascontiguousarray(rows).view(np.uint8).reshape(-1, 24)
        order = lambda a: a[np.lexsort(a.T[::-1])]
        gathered_ok = bool(sum(counts) == n_rows_total and counts[0] == n_rows_local and np.array_equal(order(mine), order(ref)))
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_bp / (elapsed / args.steps) / 1e9
        p1 = float(np.mean(p1_ms))
        # algorithmic bytes per launch of the dominant kernel: the 2-bit input once for all k,
        # plus the 24-byte rows
        bytes_alg = (length + 3) // 4 + 24 * n_rows_local
        achieved = bytes_alg / (p1 * 1e-3) / 1e9
        hbm_meas = ctx.measure_hbm_read(1 << 30, 5)
        out = {
            "metric": f"Gbp/s scanned (motif {args.kmin}-{args.kmax})", "value": round(value, 4), "unit": "Gbp/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong" if args.workload == "hg38" else "weak", "vs_baseline": None, "dtype": "u64 bitplanes (2-bit bases)",
            "data": "synthetic",
            "config": {"workload": (f"chr22-sized synthetic stand-in contig per GPU ({length} bp; hg38-like N blocks, "
                                    "planted repeats)" if args.workload == "chr22" else
                                    f"uniform random ACGT contig per GPU ({length} bp, generated on the device)"
                                    if args.workload == "random" else
                                    f"25 contigs with the hg38 primary-assembly lengths ({total_bp} bp of random ACGT "
                                    f"generated on the device) dealt to {world} rank(s) longest first, rank 0 holds {length} bp") +
                                   f", motif {args.kmin}-{args.kmax}, min_repeats {args.min_repeats}, "
                                   f"min_span {args.min_span}; genome packed + resident in HBM before the timed region",
                       "kernel_path": "generic" if st0.path == 0 else "vertical",
                       "rows_per_gpu": n_rows_local, "rows_total": n_rows_total,
                       "candidates_per_gpu": int(st0.n_candidates),
                       "steps_in_flight": 2 if pipelined else 1,
                       "multi_gpu": ("one contig per rank, no data-path collective; one padded RCCL gather of rows to rank 0"
                                     f" (gather verified: {gathered_ok})") if world > 1 else "n/a"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": pmc_traffic(length, args.kmin, args.kmax) if (st0.path == 1 and args.workload == "chr22") else None,
                         "kernel": "prf_vscan_kernel (fused scan + verify + row compaction: the only launch of a step)" if st0.path == 1 else "prf_scan_generic_kernel",
                         "kernel_ms": round(p1, 5),
                         "algorithmic_bytes_per_launch": bytes_alg,
                         "measured_hbm_read_GBps": round(hbm_meas, 1),
                         "frac_of_measured_read": round(achieved / hbm_meas, 5)},
            "device_ms": ({"fused_scan_verify_compact_kernel": round(p1, 5)} if st0.path == 1 else
                          {"scan_kernel": round(p1, 5), "verify_kernel": round(float(np.mean(p2_ms)), 5)}),
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is timed on rank 0 at N=1 only
            if seq is None:
                from oracle import prf_oracle
                seq = prf_oracle.synth(min(HG38_LENS[0] if args.workload == 'hg38' else length, args.cpu_sample_bp),
                                       1000 if args.workload == 'hg38' else 22 + rank)
            out["cpu_baseline"] = cpu_baseline(seq, args.kmin, args.kmax, args.min_repeats, args.min_span,
                                               args.cpu_sample_bp)
        print(json.dumps(out), flush=True)
    if world > 1:
        ctx.set_row_sink(None, 0)
    genome.free()
    ctx.close()
    if world > 1:
        work_q.put(None)
        comm_thread.join()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
