"""CPU: the N>1 paths with world_size 2 over gloo -- genome shares (multi_gpu.plan_parts: position ranges cut at tile
multiples, a row belongs to the share that holds its first position), the padded row gather, rank-local FASTA reading and
part files of the command line, and the propagation of a rank-local failure.  The GPU scan is replaced by the oracle (tests
may call it), so what is checked is the sharding/gather/merge logic: the result must equal the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _contigs():
    import synth
    out = []
    for i, n in enumerate([200_000, 3_000, 0, 25_000, 12_345, 800]):
        out.append(synth.chr_standin(length=n, seed=100 + i, n_head=n // 20, n_tail=n // 50, repeats_per_mbp=4000).tobytes())
    return out


def _oracle_scan(seqs, settings):
    from oracle import prf_oracle
    rows = []
    for ci, s in enumerate(seqs):
        for a, b, _ml, k in prf_oracle.detect_rows(s, *settings):
            rows.append((a, b, k, ci))
    return rows


def _worker(rank, world, port, out_path):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import multi_gpu
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows = multi_gpu.scan_contigs_sharded(_contigs(), (1, 20, 3, 9), _oracle_scan, dist, torch, "cpu")
        if rank == 0:
            np.save(out_path, rows)
        else:
            assert rows is None
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_plan_contig_shards_balances_and_covers():
    import multi_gpu
    lengths = [248, 242, 198, 190, 181, 170, 159, 145, 138, 133, 135, 133, 114, 107, 101, 90, 83, 80, 58, 64, 46, 50, 156, 57]
    for world in (1, 2, 4, 8):
        shards = multi_gpu.plan_contig_shards(lengths, world)
        assert sorted(i for s in shards for i in s) == list(range(len(lengths)))
        loads = [sum(lengths[i] for i in s) for s in shards]
        assert max(loads) <= 1.25 * sum(lengths) / world + max(lengths) * (world > 4)


def test_two_rank_gather_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    import multi_gpu
    out = str(tmp_path / "rows.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    want = np.array(_oracle_scan(_contigs(), (1, 20, 3, 9)), dtype=multi_gpu.ROW_DTYPE)
    want = want[np.lexsort((want["end"], want["start"], want["contig"]))]
    assert len(want) > 100
    assert got.dtype == want.dtype and np.array_equal(got, want)


def test_plan_parts_covers_the_genome_in_order_at_tile_multiples():
    import multi_gpu
    tile = 65_536
    lens = [248_956_422, 50_818_468, 0, 16_569, 65_536, 131_073]
    for world in (1, 2, 3, 8, 64):
        shares = multi_gpu.plan_parts(lens, world, tile)
        assert len(shares) == world
        flat = [p for share in shares for p in share]
        assert flat == sorted(flat)                                        # genome order, rank after rank
        for c, n in enumerate(lens):                                       # every contig covered exactly once, cut at tile multiples
            mine = [(b, e) for cc, b, e in flat if cc == c]
            assert sum(e - b for b, e in mine) == n
            pos = 0
            for b, e in mine:
                assert b == pos and b % tile == 0 and (e % tile == 0 or e == n)
                pos = e
        sizes = [sum(e - b for _c, b, e in share) for share in shares]
        assert max(sizes) - min(sizes) <= (2 + len(lens)) * tile            # shares are balanced in tiles; a contig's last tile is partial
    # per-tile cost classes: tiles that are never scanned (class 2) cost nothing, so the rest is what gets balanced
    cls = [np.r_[np.full(100, 2), np.zeros(100)].astype(np.uint8)]
    a, b = multi_gpu.plan_parts([200 * tile], 2, tile, cls)
    assert a == [(0, 0, 150 * tile)] and b == [(0, 150 * tile, 200 * tile)]


def _failing_worker(rank, world, port):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import multi_gpu
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def scan(seqs, settings):
        if rank == 1:
            raise ValueError("unsupported symbol at contig 0 position 4")
        return _oracle_scan(seqs, settings)
    try:
        with pytest.raises(multi_gpu.ShardError) as info:           # on BOTH ranks, with rank 1's message, before any row collective
            multi_gpu.scan_contigs_sharded(_contigs(), (1, 20, 3, 9), scan, dist, torch, "cpu")
        assert info.value.rank == 1 and info.value.kind == "ValueError" and "position 4" in info.value.message
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_a_failure_on_one_rank_is_raised_on_every_rank():
    import torch.multiprocessing as mp
    mp.spawn(_failing_worker, args=(2, _free_port()), nprocs=2, join=True)


def _cli_worker(rank, world, port, workdir, fasta_path, out_prefix, extra=()):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), PRF_DIST_BACKEND="gloo")
    os.chdir(workdir)
    import ctypes
    import perfect_repeat_finder as prf
    from oracle import prf_oracle

    def oracle_parts(entries, settings, parts):   # stands in for the GPU scan of this rank's share
        rows = []
        for ci, e in enumerate(entries):
            mine = [(b, en) for c, b, en in parts if c == ci]
            for a, b, _ml, k in prf_oracle.detect_rows(ctypes.string_at(e.addr, e.length), *settings):
                if any(lo <= a < hi for lo, hi in mine):             # a row belongs to the part that holds its first position
                    rows.append((a, b, k, ci))
        return rows
    prf._gpu_scan_parts = oracle_parts

    def oracle_whole(entries, settings, parts):   # stands in for prf_scan on this rank's whole contigs (min_repeats == 1)
        assert all(b == 0 and en == entries[c].length for c, b, en in parts)
        return [(a, b, ml, ci) for ci, e in enumerate(entries)
                for a, b, ml, _k in prf_oracle.detect_rows(ctypes.string_at(e.addr, e.length), *settings)]
    prf._gpu_scan_whole_contigs = oracle_whole
    prf.main(["-min", "1", "-max", "20", "-o", out_prefix, fasta_path] + list(extra))


def test_command_line_under_two_ranks_writes_the_whole_genome_bed(tmp_path, capfd):
    """`torch.distributed.run`-style launch of the drop-in CLI (WORLD_SIZE=2, gloo): the genome cut into two shares (the
    first contig is split between the ranks), every rank reads only the contigs of its share -- by seeking, once a .fai
    index lies next to the file -- and writes its piece; rank 0 concatenates: byte-identical to the single-process result."""
    import argparse
    import torch.multiprocessing as mp
    from oracle import prf_oracle
    contigs = {f"ctg{i}": c.decode() for i, c in enumerate(_contigs())}
    fasta = tmp_path / "toy.fa"
    with open(fasta, "wt") as f:
        for name, seq in contigs.items():
            f.write(f">{name}\n")
            for i in range(0, len(seq), 70):
                f.write(seq[i:i + 70] + "\n")
    fs = argparse.Namespace(min_motif_size=1, max_motif_size=20, min_repeats=3, min_span=9)
    want = "".join(f"{name}\t{s}\t{e}\t{m}\n" for name, seq in contigs.items() for s, e, m in prf_oracle.detect_repeats(seq, fs))
    assert want.count("\n") > 300
    mp.spawn(_cli_worker, args=(2, _free_port(), str(tmp_path), str(fasta), "parsed"), nprocs=2, join=True)
    assert open(tmp_path / "parsed.bed").read() == want
    out = capfd.readouterr().out
    assert out.count("Wrote results to parsed.bed") == 1 and out.count("Processing ctg") == len(contigs)
    # with a samtools-style index: (name, length, offset, bases per line, bytes per line)
    off = 0
    with open(str(fasta) + ".fai", "wt") as f:
        for name, seq in contigs.items():
            off += len(name) + 2
            f.write(f"{name}\t{len(seq)}\t{off}\t70\t71\n")
            off += len(seq) + -(-len(seq) // 70)
    mp.spawn(_cli_worker, args=(2, _free_port(), str(tmp_path), str(fasta), "indexed"), nprocs=2, join=True)
    assert open(tmp_path / "indexed.bed").read() == want
    assert not [p for p in os.listdir(tmp_path) if ".part" in p]
    # min_repeats == 1 (the literal lane): whole contigs dealt to the ranks in genome order
    fs1 = argparse.Namespace(min_motif_size=1, max_motif_size=20, min_repeats=1, min_span=12)
    want1 = "".join(f"{name}\t{s}\t{e}\t{m}\n" for name, seq in contigs.items() for s, e, m in prf_oracle.detect_repeats(seq, fs1))
    assert want1.count("\n") > 300 and want1 != want
    mp.spawn(_cli_worker, args=(2, _free_port(), str(tmp_path), str(fasta), "one", ("--min-repeats", "1", "--min-span", "12")),
             nprocs=2, join=True)
    assert open(tmp_path / "one.bed").read() == want1
