"""CPU: the N>1 paths with world_size 2 over gloo -- genome shares (multi_gpu.plan_parts: position ranges cut at tile
multiples, a row belongs to the share that holds its first position), the padded gather of 8-byte wire rows (the product's
data path: pack_rows -> gather_packed -> unpack_rows), rank-local FASTA reading and part files of the command line, and the
propagation of a rank-local failure.  The GPU scan is replaced by the oracle (tests
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


TILE = 65_536
SIDE = 4


def _genome():
    """(contigs, contig bases in the library's global position space): contigs start on tile multiples behind a guard gap"""
    contigs = _contigs() + [b"ACGT" * 10 + b"A" * 70_000 + b"C"]       # the last one: a row whose span does not fit 16 bits
    bases, cur = [], 0
    for c in contigs:
        bases.append(cur)
        cur = -(-(cur + len(c) + 20 + 64) // TILE) * TILE
    return contigs, bases


def _share_rows(contigs, parts, settings):
    """what a rank's GPU scan of its share returns: the rows whose first position lies in one of the share's parts"""
    import multi_gpu
    rows = [r for r in _oracle_scan(contigs, settings) if any(c == r[3] and lo <= r[0] < hi for c, lo, hi in parts)]
    return np.array(rows, dtype=multi_gpu.ROW_DTYPE)


def _worker(rank, world, port, out_path):
    """The product's N > 1 data path (bench.py) with the oracle in place of the GPU scan: plan_parts shares ->
    rows of the share -> 8-byte wire rows (the host encoder, checked against the device's in a GPU test) -> ONE padded
    gather -> decoded and concatenated on rank 0."""
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
        contigs, bases = _genome()
        shares = multi_gpu.plan_parts([len(c) for c in contigs], world, TILE)
        rows, error = np.zeros(0, dtype=multi_gpu.ROW_DTYPE), None
        try:
            rows = _share_rows(contigs, shares[rank], (1, 20, 3, 9))
        except Exception as exc:                                        # noqa: BLE001
            error = exc
        multi_gpu.agree_or_raise(error, dist, torch, "cpu")
        cap = torch.tensor([len(rows)], dtype=torch.int64)
        dist.all_reduce(cap, op=dist.ReduceOp.MAX)
        cap = int(cap.item())
        send = torch.from_numpy(multi_gpu.pack_rows(rows, cap, SIDE, bases, TILE).view(np.int64).copy())
        recv = multi_gpu.gather_packed(send, dist, torch)
        if rank == 0:
            got = np.concatenate([multi_gpu.unpack_rows(r.numpy(), cap, SIDE, bases, TILE) for r in recv])
            np.save(out_path, got)
        else:
            assert recv is None
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_gather_of_wire_rows_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    import multi_gpu
    out = str(tmp_path / "rows.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    contigs, _bases = _genome()
    want = np.array(_oracle_scan(contigs, (1, 20, 3, 9)), dtype=multi_gpu.ROW_DTYPE)
    want = want[np.lexsort((want["end"], want["start"], want["contig"]))]
    assert len(want) > 100 and (want["end"] - want["start"]).max() > 65_535
    # no sort on rank 0: the shares are in genome order, their concatenation IS the sorted row array
    assert got.dtype == want.dtype and np.array_equal(got, want)


def test_wire_rows_round_trip_and_limits():
    import multi_gpu
    bases = [0, 3 * TILE]
    rows = np.array([(5, 17, 3, 0), (TILE - 1, TILE + 20, 7, 0), (40, 70_040, 1, 1), (100, 1_123, 511, 1)], dtype=multi_gpu.ROW_DTYPE)
    words = multi_gpu.pack_rows(rows, 6, 2, bases, TILE)
    assert len(words) == 6 + 1 + 3 * 2 and int(words[6]) == 4 | (1 << 40)
    assert np.array_equal(multi_gpu.unpack_rows(words, 6, 2, bases, TILE), rows)
    with pytest.raises(ValueError):
        multi_gpu.pack_rows(rows, 3, 2, bases, TILE)                    # more rows than the buffer holds
    with pytest.raises(ValueError):
        multi_gpu.pack_rows(rows, 6, 0, bases, TILE)                    # no room for the long row
    big = rows.copy()
    big["k"][0] = 512
    with pytest.raises(ValueError):
        multi_gpu.pack_rows(big, 6, 2, bases, TILE)                     # motif sizes above 511 do not fit the wire row


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
    try:
        error = ValueError("unsupported symbol at contig 0 position 4") if rank == 1 else None
        with pytest.raises(multi_gpu.ShardError) as info:           # on BOTH ranks, with rank 1's message, before any row collective
            multi_gpu.agree_or_raise(error, dist, torch, "cpu")
        assert info.value.rank == 1 and info.value.kind == "ValueError" and "position 4" in info.value.message
        multi_gpu.agree_or_raise(None, dist, torch, "cpu")          # nobody failed: returns on every rank
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


def _stale_cli_worker(rank, world, port, workdir, fasta_path):
    import pytest as _pytest
    with _pytest.raises(Exception) as info:
        _cli_worker(rank, world, port, workdir, fasta_path, "stale")
    assert "index says" in str(info.value) and "ctg3" in str(info.value)            # on BOTH ranks, whoever read the contig


def test_a_stale_index_is_an_error_on_every_rank_not_a_short_bed(tmp_path):
    """the shares of the N-rank command line are planned from the .fai; an index whose lengths do not match the
    file must not produce a BED that silently differs from the single-process one."""
    import torch.multiprocessing as mp
    contigs = {f"ctg{i}": c.decode() for i, c in enumerate(_contigs())}
    fasta = tmp_path / "toy.fa"
    off = 0
    lines = []
    with open(fasta, "wt") as f:
        for name, seq in contigs.items():
            f.write(f">{name}\n")
            off += len(name) + 2
            lines.append([name, len(seq), off, 70, 71])
            for i in range(0, len(seq), 70):
                f.write(seq[i:i + 70] + "\n")
            off += len(seq) + -(-len(seq) // 70)
    lines[3][1] -= 1_000                                                                 # ctg3 (25 000 bp): the index says 24 000
    with open(str(fasta) + ".fai", "wt") as f:
        f.writelines("\t".join(map(str, ln)) + "\n" for ln in lines)
    mp.spawn(_stale_cli_worker, args=(2, _free_port(), str(tmp_path), str(fasta)), nprocs=2, join=True)
    assert not os.path.exists(tmp_path / "stale.bed") and not [p for p in os.listdir(tmp_path) if ".part" in p]
