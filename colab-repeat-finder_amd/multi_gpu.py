"""Multi-GPU sharding of the scan: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests), contigs as the unit of work, no data-path collective.

The reference's only parallel strategy is file-level interval sharding in a cloud batch pipeline whose outputs
are concatenated with `cat | sort | uniq` and repaired by merge_loci (reference
hail_batch_pipeline/run_hail_batch_pipeline.py:76-77, :151-153, :170-175).  Here contigs are dealt to the ranks
(longest first, to the least loaded rank), every rank scans its own contigs with its own libprf context, and the
rows -- tiny compared with the input -- are concatenated on rank 0 with ONE padded gather.  Because per-contig
scans are independent the result is exactly the single-GPU result; nothing has to be repaired.
"""
import numpy as np

ROW_DTYPE = np.dtype([("start", "<u8"), ("end", "<u8"), ("k", "<u4"), ("contig", "<u4")])


def plan_contig_shards(lengths, world):
    """Longest-processing-time-first: returns, per rank, the sorted list of contig indices it scans."""
    shards = [[] for _ in range(world)]
    load = [0] * world
    for idx in sorted(range(len(lengths)), key=lambda i: (-lengths[i], i)):
        r = min(range(world), key=lambda j: (load[j], j))
        shards[r].append(idx)
        load[r] += lengths[idx]
    return [sorted(s) for s in shards]


TILE_COST = (1.0, 1.3, 0.0, 12.0)   # ordinary tile; tile with N in reach (three planes); all-N tile (never scanned); tile with a
                                   # symbol outside ACGTN in reach (generic kernels: an order of magnitude slower)


def plan_whole_contigs(lengths, world):
    """Whole contigs dealt to `world` ranks in genome order (contiguous runs of contigs, balanced by length): the same shape
    as plan_parts() returns, every part a whole contig.  For the literal lane (min_repeats == 1), whose rows depend on where
    a sequence begins and ends, so a sequence cannot be cut."""
    total = float(sum(lengths)) or 1.0
    shares = [[] for _ in range(world)]
    acc = 0
    for c, n in enumerate(lengths):
        r = min(world - 1, int((acc + n / 2.0) * world / total))
        shares[r].append((c, 0, int(n)))
        acc += n
    return shares


def plan_parts(lengths, world, tile, classes=None):
    """Cut ONE genome into `world` shares of equal cost: returns, per rank, a list of (contig, begin, end) position ranges
    cut at multiples of `tile` (prf_native.tile_positions()), in genome order -- what Genome.select() takes.  A contig
    longer than a share is split; a row belongs to the part that holds its first position, so the shares' row sets are
    disjoint and their union is the whole scan (include/prf.h, prf_genome_select).  classes: optional per-contig arrays of
    per-tile cost classes (Genome.tile_classes); without them every tile costs the same."""
    costs, owner, index = [], [], []
    for c, n in enumerate(lengths):
        nt = -(-n // tile)
        if classes is not None:
            cls = np.asarray(classes[c], dtype=np.int64)
            assert len(cls) == nt
            w = np.asarray(TILE_COST)[cls]
        else:
            w = np.ones(nt)
        costs.append(w)
        owner.append(np.full(nt, c, dtype=np.int64))
        index.append(np.arange(nt, dtype=np.int64))
    if not costs:
        return [[] for _ in range(world)]
    costs, owner, index = np.concatenate(costs), np.concatenate(owner), np.concatenate(index)
    cum = np.concatenate(([0.0], np.cumsum(costs)))
    cuts = [int(np.searchsorted(cum, cum[-1] * r / world, side="left")) for r in range(world)] + [len(costs)]
    shares = []
    for r in range(world):
        lo, hi = cuts[r], max(cuts[r], cuts[r + 1])
        parts = []
        i = lo
        while i < hi:
            c = int(owner[i])
            j = i
            while j < hi and owner[j] == c:
                j += 1
            begin, end = int(index[i]) * tile, min(int(index[j - 1] + 1) * tile, lengths[c])
            if end > begin:
                parts.append((c, begin, end))
            i = j
        shares.append(parts)
    return shares


def unpack_rows(words, capacity, side_capacity, bases, tile):
    """Decode the 8-byte wire format of prf_last_hits_packed_to_device (include/prf.h): `words` = capacity packed rows, one
    count word, 3 * side_capacity words of long rows.  bases: Genome.contig_bases().  Returns a ROW_DTYPE array."""
    words = np.asarray(words).view(np.uint64).reshape(-1)
    count = int(words[capacity])
    n, n_side = count & ((1 << 40) - 1), count >> 40
    w = words[:n]
    gpos = (w >> np.uint64(41)) * np.uint64(tile) + ((w >> np.uint64(25)) & np.uint64(0xFFFF))
    span = (w >> np.uint64(9)) & np.uint64(0xFFFF)
    rows = np.zeros(n, dtype=ROW_DTYPE)
    bases = np.asarray(bases, dtype=np.uint64)
    contig = np.searchsorted(bases, gpos, side="right") - 1
    rows["contig"] = contig
    rows["start"] = gpos - bases[contig]
    rows["end"] = rows["start"] + span
    rows["k"] = (w & np.uint64(511)).astype(np.uint32)
    if n_side:
        side = words[capacity + 1:capacity + 1 + 3 * n_side].reshape(n_side, 3)
        full = {(int(c), int(s), int(k)): int(e) for s, e, kc in side.tolist() for k, c in [(kc & 0xFFFFFFFF, kc >> 32)]}
        for i in np.nonzero(span == np.uint64(0xFFFF))[0].tolist():
            rows["end"][i] = full[(int(rows["contig"][i]), int(rows["start"][i]), int(rows["k"][i]))]
    return rows


def rows_to_tensor(rows, capacity, torch, device):
    """(capacity+1, 3) int64 tensor: 24-byte rows as three int64 words, the row count in the last row."""
    t = torch.zeros((capacity + 1, 3), dtype=torch.int64, device=device)
    n = len(rows)
    if n:
        flat = np.ascontiguousarray(rows).view(np.int64).reshape(n, 3)
        t[:n] = torch.from_numpy(flat.copy()).to(device)
    t[capacity, 0] = n
    return t


def tensor_to_rows(t):
    a = t.cpu().numpy()
    n = int(a[-1, 0])
    return np.ascontiguousarray(a[:n]).view(ROW_DTYPE).reshape(n)


def gather_rows(local_rows, dist, torch, device):
    """One padded gather of every rank's rows to rank 0.  Returns the concatenated rows sorted by
    (contig, start, end) on rank 0, None elsewhere."""
    rank, world = dist.get_rank(), dist.get_world_size()
    cap = torch.tensor([len(local_rows)], dtype=torch.int64, device=device)
    dist.all_reduce(cap, op=dist.ReduceOp.MAX)
    capacity = int(cap.item())
    send = rows_to_tensor(local_rows, capacity, torch, device)
    recv = [torch.zeros_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, recv, dst=0)
    if rank != 0:
        return None
    parts = [tensor_to_rows(t) for t in recv]
    rows = np.concatenate(parts) if parts else np.zeros(0, dtype=ROW_DTYPE)
    order = np.lexsort((rows["end"], rows["start"], rows["contig"]))
    return rows[order]


class ShardError(RuntimeError):
    """A rank's scan failed; raised on EVERY rank with the failing rank's message, before any row collective."""

    def __init__(self, rank, kind, message, code=None):
        super().__init__(f"rank {rank}: {kind}: {message}")
        self.rank, self.kind, self.message, self.code = rank, kind, message, code


def agree_or_raise(error, dist, torch, device):
    """Collective: every rank passes its local exception (or None).  If any rank failed, every rank raises the
    ShardError of the lowest failing rank -- nobody is left waiting in the row gather for a peer that has died."""
    rank, world = dist.get_rank(), dist.get_world_size()
    flag = torch.tensor([1 if error is not None else 0], dtype=torch.int64, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()) == 0:
        return
    msgs = [None] * world
    dist.all_gather_object(msgs, None if error is None else (type(error).__name__, str(getattr(error, "message", error)),
                                                             getattr(error, "code", None)))
    bad = next(i for i, m in enumerate(msgs) if m is not None)
    raise ShardError(bad, *msgs[bad])


def scan_contigs_sharded(contigs, settings, scan_fn, dist, torch, device):
    """contigs: list of bytes, identical on every rank.  scan_fn(list_of_bytes, settings) -> rows with
    contig indices local to the list it was given.  Returns the rows of ALL contigs (global contig indices,
    sorted) on rank 0, None elsewhere.  A failure of scan_fn on one rank is raised on all of them (ShardError)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = plan_contig_shards([len(c) for c in contigs], world)[rank]
    rows, error = np.zeros(0, dtype=ROW_DTYPE), None
    try:
        if mine:
            rows = np.array(scan_fn([contigs[i] for i in mine], settings), dtype=ROW_DTYPE)
            if len(rows):
                rows = rows.copy()
                rows["contig"] = np.asarray(mine, dtype=np.uint32)[rows["contig"]]
    except Exception as exc:          # noqa: BLE001 -- whatever it is, the peers must hear about it
        error = exc
    agree_or_raise(error, dist, torch, device)
    return gather_rows(rows, dist, torch, device)
