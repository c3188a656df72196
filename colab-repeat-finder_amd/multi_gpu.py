"""Multi-GPU sharding of the scan: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests), SHARES OF ONE GENOME as the unit of work, no data-path collective.

The reference's only parallel strategy is file-level interval sharding in a cloud batch pipeline whose outputs
are concatenated with `cat | sort | uniq` and repaired by merge_loci (reference
hail_batch_pipeline/run_hail_batch_pipeline.py:76-77, :151-153, :170-175).  Here every rank holds the genome and scans a
share of it: position ranges cut at multiples of the 65536-position tile, balanced by tile cost (plan_parts).  A row
belongs to the share that holds its first position, so the shares' row sets are disjoint, their union is exactly the
single-GPU result and -- the shares being in genome order -- their concatenation is the sorted row array; nothing has to be
repaired.  The rows, tiny compared with the input, reach rank 0 with ONE padded gather of 8-byte wire rows
(pack_rows / prf_last_hits_packed_to_device -> gather_packed -> unpack_rows).  With min_repeats == 1 a sequence cannot be
cut (plan_whole_contigs).
"""
import numpy as np

ROW_DTYPE = np.dtype([("start", "<u8"), ("end", "<u8"), ("k", "<u4"), ("contig", "<u4")])


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
    if count == 0xFFFFFFFFFFFFFFFF:
        raise ValueError("the packed row buffer was too small for the scan's rows (poisoned count word)")
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


WIRE_K_MAX = 511   # motif sizes the 8-byte wire row holds (9 bits); larger ones are refused by the library (PRF_EUNSUPPORTED)


def pack_rows(rows, capacity, side_capacity, bases, tile):
    """Host reference encoder of the 8-byte wire format (what prf_pack_rows_kernel, csrc/verify.hip, writes on the device;
    a GPU test checks the two against each other): `capacity` packed rows, one count word (rows | long rows << 40),
    3 * side_capacity words of rows whose span does not fit 16 bits.  Returns a uint64 array."""
    rows = np.asarray(rows, dtype=ROW_DTYPE)
    n = len(rows)
    if n > capacity:
        raise ValueError(f"the packed row buffer holds {capacity} rows, there are {n}")
    if n and int(rows["k"].max()) > WIRE_K_MAX:
        raise ValueError(f"motif sizes above {WIRE_K_MAX} do not fit the 8-byte wire row")
    words = np.zeros(capacity + 1 + 3 * side_capacity, dtype=np.uint64)
    bases = np.asarray(bases, dtype=np.uint64)
    gpos = bases[rows["contig"]] + rows["start"]
    span = rows["end"] - rows["start"]
    s16 = np.minimum(span, np.uint64(65535))
    words[:n] = ((gpos // np.uint64(tile)) << np.uint64(41)) | ((gpos % np.uint64(tile)) << np.uint64(25)) | (s16 << np.uint64(9)) \
        | rows["k"].astype(np.uint64)
    long_rows = rows[span >= np.uint64(65535)]
    if len(long_rows) > side_capacity:
        raise ValueError(f"the side list holds {side_capacity} rows, {len(long_rows)} rows are longer than 65534")
    for j, r in enumerate(long_rows):
        words[capacity + 1 + 3 * j: capacity + 4 + 3 * j] = (int(r["start"]), int(r["end"]), int(r["k"]) | (int(r["contig"]) << 32))
    words[capacity] = n | (len(long_rows) << 40)
    return words


def gather_packed(send, dist, torch, dst=0):
    """The one collective of the data path: every rank's packed rows (an int64 tensor of the same length on every rank,
    pack_rows' layout) gathered on rank `dst`.  Returns the list of tensors there, None elsewhere."""
    rank, world = dist.get_rank(), dist.get_world_size()
    recv = [torch.zeros_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst)
    return recv


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
