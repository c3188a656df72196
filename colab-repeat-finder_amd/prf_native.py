"""ctypes binding of libprf.so (C ABI in include/prf.h) -- the only door to the GPU.

There is no CPU fallback behind this module: if the shared library is missing, or no gfx950
device is usable, the calls raise.  PyTorch is not involved; the library owns its device
memory and its HIP stream.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRF_LIB", os.path.join(_HERE, "libprf.so"))  # PRF_LIB: diagnostic builds only

PRF_OK = 0
PRF_EINVAL = -1
PRF_ENODEV = -2
PRF_EHIP = -3
PRF_ENOMEM = -4
PRF_EUNSUPPORTED = -5
PRF_ESYMBOL = -6
PRF_EINDEX = -7

SCAN_DEFAULT = 0
SCAN_FORCE_GENERIC = 1
SCAN_NO_FETCH = 2
SCAN_DEFER_TIMING = 4
TIMING_RING = 128


class PrfError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libprf error {code}: {message}")
        self.code = code
        self.message = message


class _Contig(ctypes.Structure):
    _fields_ = [("ascii", ctypes.c_void_p), ("len", ctypes.c_uint64)]


class _Part(ctypes.Structure):
    _fields_ = [("contig", ctypes.c_uint32), ("begin", ctypes.c_uint64), ("end", ctypes.c_uint64)]


class _Hit(ctypes.Structure):
    _fields_ = [("start", ctypes.c_uint64), ("end", ctypes.c_uint64), ("k", ctypes.c_uint32), ("contig", ctypes.c_uint32)]


class _Hits(ctypes.Structure):
    _fields_ = [("rows", ctypes.POINTER(_Hit)), ("n", ctypes.c_uint64)]


class ScanStats(ctypes.Structure):
    _fields_ = [("scan_ms", ctypes.c_double), ("phase1_ms", ctypes.c_double), ("phase2_ms", ctypes.c_double),
                ("positions", ctypes.c_uint64), ("packed_bytes", ctypes.c_uint64), ("n_candidates", ctypes.c_uint64),
                ("n_hits", ctypes.c_uint64), ("n_launches", ctypes.c_uint32), ("path", ctypes.c_uint32),
                ("seq", ctypes.c_uint64), ("sorted_on_device", ctypes.c_uint32), ("tiles_launched", ctypes.c_uint32)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


EXPORTS = ["prf_abi_version", "prf_device_count", "prf_last_error", "prf_open", "prf_close", "prf_genome_load",
           "prf_genome_free", "prf_genome_positions", "prf_scan_genome", "prf_scan", "prf_free_hits",
           "prf_measure_hbm_read", "prf_last_hits_to_device", "prf_plan_describe", "prf_fasta_open", "prf_fasta_count",
           "prf_fasta_entry", "prf_fasta_close", "prf_write_bed", "prf_write_tsv", "prf_genome_synth", "prf_scan_timings", "prf_set_row_sink", "prf_fasta_open_contig", "prf_scan_genome_async",
           "prf_scan_wait", "prf_genome_standin", "prf_genome_select", "prf_genome_tile_classes", "prf_tile_positions", "prf_scan_timings_split", "prf_last_hits_packed_to_device",
           "prf_genome_contig_bases", "prf_scan_literal", "prf_scan_genome_async_packed", "prf_stream_wait_for",
           "prf_genome_footprint"]

_lib = None
_lib_lock = threading.Lock()


class _HostOnly:
    """A library that exports only the host-side entry points (FASTA reader, BED/TSV writers, planner): the prototypes of the
    others are accepted and dropped, calling one raises.  Test infrastructure (tests/test_asan_host.py), selected by
    PRF_LIB_HOST_ONLY=1 together with PRF_LIB."""

    class _Absent:
        def __init__(self, name):
            self._name = name

        def __call__(self, *args):
            raise ImportError(f"{self._name} is not part of the host-only build")

    def __init__(self, cdll):
        self._cdll = cdll

    def __getattr__(self, name):
        try:
            f = getattr(self._cdll, name)
        except AttributeError:
            f = _HostOnly._Absent(name)
        self.__dict__[name] = f
        return f


def load_library():
    """dlopen libprf.so and declare the prototypes.  Raises if the library was not built."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C colab-repeat-finder_amd/csrc).  There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        if os.environ.get("PRF_LIB_HOST_ONLY") == "1":
            lib = _HostOnly(lib)      # the sanitizer build of the host-only parts (make asan): the GPU entry points are absent
        vp = ctypes.c_void_p
        lib.prf_abi_version.restype = ctypes.c_int
        lib.prf_device_count.restype = ctypes.c_int
        lib.prf_last_error.restype = ctypes.c_char_p
        lib.prf_open.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
        lib.prf_close.argtypes = [vp]
        lib.prf_close.restype = None
        lib.prf_genome_load.argtypes = [vp, ctypes.POINTER(_Contig), ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(vp)]
        lib.prf_genome_synth.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.c_int,
                                         ctypes.c_uint32, ctypes.POINTER(vp)]
        lib.prf_genome_standin.argtypes = lib.prf_genome_synth.argtypes
        lib.prf_tile_positions.restype = ctypes.c_uint64
        lib.prf_genome_select.argtypes = [vp, ctypes.POINTER(_Part), ctypes.c_int]
        lib.prf_genome_tile_classes.argtypes = [vp, ctypes.c_uint32, vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
        lib.prf_genome_free.argtypes = [vp]
        lib.prf_genome_free.restype = None
        lib.prf_genome_positions.argtypes = [vp]
        lib.prf_genome_positions.restype = ctypes.c_uint64
        lib.prf_scan_genome.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                        ctypes.c_uint32, ctypes.POINTER(_Hits), ctypes.POINTER(ScanStats)]
        lib.prf_scan.argtypes = [vp, ctypes.POINTER(_Contig), ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32,
                                 ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(_Hits),
                                 ctypes.POINTER(ScanStats)]
        lib.prf_scan_literal.argtypes = [vp, ctypes.POINTER(_Contig), ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                         ctypes.c_uint32, ctypes.c_uint64, ctypes.POINTER(_Hits), ctypes.POINTER(ScanStats)]
        lib.prf_free_hits.argtypes = [ctypes.POINTER(_Hits)]
        lib.prf_free_hits.restype = None
        lib.prf_measure_hbm_read.argtypes = [vp, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        lib.prf_last_hits_to_device.argtypes = [vp, vp, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
        lib.prf_set_row_sink.argtypes = [vp, vp, ctypes.c_uint64]
        lib.prf_last_hits_packed_to_device.argtypes = [vp, vp, vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
        lib.prf_genome_contig_bases.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
        lib.prf_genome_footprint.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
        lib.prf_scan_genome_async.argtypes = [vp, vp] + [ctypes.c_uint32] * 4 + [ctypes.POINTER(ctypes.c_uint64)]
        lib.prf_scan_genome_async_packed.argtypes = [vp, vp] + [ctypes.c_uint32] * 4 + [vp, ctypes.c_uint64, ctypes.c_uint64,
                                                                                         ctypes.POINTER(ctypes.c_uint64)]
        lib.prf_stream_wait_for.argtypes = [vp, vp]
        lib.prf_scan_wait.argtypes = [vp, ctypes.c_uint64, ctypes.POINTER(ScanStats)]
        lib.prf_scan_timings.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_float)]
        lib.prf_scan_timings_split.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_float),
                                               ctypes.POINTER(ctypes.c_float)]
        lib.prf_plan_describe.argtypes = [ctypes.c_uint32] * 4 + [ctypes.c_char_p, ctypes.c_uint64]
        lib.prf_fasta_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
        lib.prf_fasta_open_contig.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(vp)]
        lib.prf_fasta_count.argtypes = [vp]
        lib.prf_fasta_entry.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(vp),
                                        ctypes.POINTER(ctypes.c_uint64)]
        lib.prf_fasta_close.argtypes = [vp]
        lib.prf_fasta_close.restype = None
        lib.prf_write_bed.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(_Contig),
                                      ctypes.c_int, ctypes.POINTER(_Hits), ctypes.POINTER(ctypes.c_uint64)]
        lib.prf_write_tsv.argtypes = [ctypes.c_char_p, ctypes.POINTER(_Contig), ctypes.POINTER(_Hits),
                                      ctypes.POINTER(ctypes.c_uint64)]
        _lib = lib
        return lib


def _check(lib, rc):
    if rc != PRF_OK:
        raise PrfError(rc, lib.prf_last_error().decode("utf-8", "replace"))


def _contig_array(seqs):
    """(array, keep-alive list).  seqs: bytes objects, or (address, length) pairs of memory that outlives the call."""
    arr = (_Contig * max(1, len(seqs)))()
    keep = []
    for i, s in enumerate(seqs):
        if isinstance(s, tuple):
            arr[i].ascii, arr[i].len = s
            continue
        if isinstance(s, bytearray):
            s = bytes(s)
        if not isinstance(s, bytes):
            raise TypeError("contigs must be bytes")
        keep.append(s)
        arr[i].ascii = ctypes.cast(ctypes.c_char_p(s), ctypes.c_void_p)
        arr[i].len = len(s)
    return arr, keep


def _rows(hits):
    n = hits.n
    if n == 0:
        return []
    import numpy as np
    buf = np.ctypeslib.as_array(ctypes.cast(hits.rows, ctypes.POINTER(ctypes.c_uint8)), shape=(n * ctypes.sizeof(_Hit),))
    rec = buf.view(np.dtype([("start", "<u8"), ("end", "<u8"), ("k", "<u4"), ("contig", "<u4")])).copy()
    return rec


class Genome:
    """Contigs packed and resident in HBM (prf_genome)."""

    def __init__(self, ctx, handle, n_contigs):
        self.ctx = ctx
        self._h = handle
        self.n_contigs = n_contigs

    @property
    def positions(self):
        return self.ctx.lib.prf_genome_positions(self._h)

    def scan(self, kmin, kmax, min_repeats, min_span, flags=SCAN_DEFAULT, fetch=True):
        """Returns (rows, stats): rows is a numpy record array (start, end, k, contig) sorted by
        (contig, start, end), or None when fetch=False."""
        lib = self.ctx.lib
        hits = _Hits()
        stats = ScanStats()
        f = flags | (0 if fetch else SCAN_NO_FETCH)
        _check(lib, lib.prf_scan_genome(self.ctx._h, self._h, kmin, kmax, min_repeats, min_span, f,
                                        ctypes.byref(hits), ctypes.byref(stats)))
        if not fetch:
            return None, stats
        try:
            return _rows(hits), stats
        finally:
            lib.prf_free_hits(ctypes.byref(hits))

    def select(self, parts):
        """Restrict the following scans of this genome to `parts`: (contig, begin, end) position ranges cut at multiples
        of tile_positions() (multi_gpu.plan_parts makes them).  None / []: the whole genome again."""
        parts = list(parts or [])
        arr = (_Part * max(1, len(parts)))()
        for i, (c, b, e) in enumerate(parts):
            arr[i].contig, arr[i].begin, arr[i].end = c, b, e
        _check(self.ctx.lib, self.ctx.lib.prf_genome_select(self._h, arr, len(parts)))

    def tile_classes(self, contig):
        """numpy uint8 array, one cost class per tile of the contig (0 ordinary, 1 not-ACGT in reach, 2 never scanned)."""
        import numpy as np
        n = ctypes.c_uint64(0)
        _check(self.ctx.lib, self.ctx.lib.prf_genome_tile_classes(self._h, contig, None, 0, ctypes.byref(n)))
        out = np.zeros(max(1, n.value), dtype=np.uint8)
        _check(self.ctx.lib, self.ctx.lib.prf_genome_tile_classes(self._h, contig, out.ctypes.data_as(ctypes.c_void_p), n.value,
                                                                  ctypes.byref(n)))
        return out[:n.value]

    def contig_bases(self):
        """First position of every contig in the genome's coordinate space (numpy uint64)."""
        import numpy as np
        n = ctypes.c_uint64(0)
        _check(self.ctx.lib, self.ctx.lib.prf_genome_contig_bases(self._h, None, 0, ctypes.byref(n)))
        out = (ctypes.c_uint64 * max(1, n.value))()
        _check(self.ctx.lib, self.ctx.lib.prf_genome_contig_bases(self._h, out, n.value, ctypes.byref(n)))
        return np.array(out[:n.value], dtype=np.uint64)

    def footprint(self):
        """(device bytes the resident genome holds, positions of its coordinate space): prf_genome_footprint."""
        b, n = ctypes.c_uint64(0), ctypes.c_uint64(0)
        _check(self.ctx.lib, self.ctx.lib.prf_genome_footprint(self._h, ctypes.byref(b), ctypes.byref(n)))
        return b.value, n.value

    def scan_async(self, kmin, kmax, min_repeats, min_span):
        """Enqueue a scan (at most two in flight); returns its serial number for Context.scan_wait()."""
        seq = ctypes.c_uint64(0)
        _check(self.ctx.lib, self.ctx.lib.prf_scan_genome_async(self.ctx._h, self._h, kmin, kmax, min_repeats, min_span,
                                                                 ctypes.byref(seq)))
        return seq.value

    def scan_async_packed(self, kmin, kmax, min_repeats, min_span, dst_ptr, capacity_rows, side_capacity):
        """scan_async() whose rows also leave as 8-byte wire words in caller-owned device memory (capacity_rows + 1 +
        3 * side_capacity words), packed on the library's stream behind the scan: no host step in between."""
        seq = ctypes.c_uint64(0)
        _check(self.ctx.lib, self.ctx.lib.prf_scan_genome_async_packed(self.ctx._h, self._h, kmin, kmax, min_repeats, min_span,
                                                                        ctypes.c_void_p(dst_ptr), capacity_rows, side_capacity,
                                                                        ctypes.byref(seq)))
        return seq.value

    def free(self):
        if self._h is not None:
            self.ctx.lib.prf_genome_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One GPU, one HIP stream (prf_ctx)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.prf_open(device, ctypes.byref(h)))
        self._h = h
        self.device = device

    def load(self, seqs, kmax_hint):
        arr, _keep = _contig_array(seqs)
        g = ctypes.c_void_p()
        _check(self.lib, self.lib.prf_genome_load(self._h, arr, len(seqs), kmax_hint, ctypes.byref(g)))
        return Genome(self, g, len(seqs))

    def synth(self, lens, seeds, kmax_hint):
        """Contigs generated on the device (SURVEY 8(d) generator): nothing crosses PCIe."""
        n = len(lens)
        la = (ctypes.c_uint64 * max(1, n))(*lens)
        sa = (ctypes.c_uint64 * max(1, n))(*seeds)
        g = ctypes.c_void_p()
        _check(self.lib, self.lib.prf_genome_synth(self._h, la, sa, n, kmax_hint, ctypes.byref(g)))
        return Genome(self, g, n)

    def standin(self, lens, seeds, kmax_hint):
        """Contigs of the stand-in recipe 2 (synth.standin2) generated on the device."""
        n = len(lens)
        la = (ctypes.c_uint64 * max(1, n))(*lens)
        sa = (ctypes.c_uint64 * max(1, n))(*seeds)
        g = ctypes.c_void_p()
        _check(self.lib, self.lib.prf_genome_standin(self._h, la, sa, n, kmax_hint, ctypes.byref(g)))
        return Genome(self, g, n)

    def scan(self, seqs, kmin, kmax, min_repeats, min_span, flags=SCAN_DEFAULT):
        arr, _keep = _contig_array(seqs)
        hits = _Hits()
        stats = ScanStats()
        _check(self.lib, self.lib.prf_scan(self._h, arr, len(seqs), kmin, kmax, min_repeats, min_span, flags,
                                           ctypes.byref(hits), ctypes.byref(stats)))
        try:
            return _rows(hits), stats
        finally:
            self.lib.prf_free_hits(ctypes.byref(hits))

    def scan_literal(self, seq, kmin, kmax, min_repeats, min_span, stop=None):
        """The literal lane on one sequence (prf_scan_literal); stop: lock-step iterations performed, default all."""
        arr, _keep = _contig_array([seq])
        hits = _Hits()
        stats = ScanStats()
        _check(self.lib, self.lib.prf_scan_literal(self._h, arr, kmin, kmax, min_repeats, min_span,
                                                   arr[0].len if stop is None else stop, ctypes.byref(hits), ctypes.byref(stats)))
        try:
            return _rows(hits), stats
        finally:
            self.lib.prf_free_hits(ctypes.byref(hits))

    def last_hits_to_device(self, dst_ptr, capacity_rows, count_row=False):
        """D2D copy of the last scan's rows into caller-owned device memory; returns the row count.
        count_row: record number capacity_rows of the buffer receives (rows copied, 0, 0)."""
        n = ctypes.c_uint64(0)
        _check(self.lib, self.lib.prf_last_hits_to_device(self._h, ctypes.c_void_p(dst_ptr), capacity_rows,
                                                          1 if count_row else 0, ctypes.byref(n)))
        return n.value

    def last_hits_packed_to_device(self, genome, dst_ptr, capacity_rows, side_capacity):
        """The last scan's rows as 8-byte wire words (+ count word + side list of long rows) in caller-owned device memory of
        capacity_rows + 1 + 3 * side_capacity words; returns the row count.  multi_gpu.unpack_rows() decodes."""
        n = ctypes.c_uint64(0)
        _check(self.lib, self.lib.prf_last_hits_packed_to_device(self._h, genome._h, ctypes.c_void_p(dst_ptr), capacity_rows,
                                                                 side_capacity, ctypes.byref(n)))
        return n.value

    def stream_wait_for(self, other_stream):
        """What has been enqueued on the library's stream happens before what `other_stream` (a raw hipStream_t, e.g.
        torch.cuda.Stream().cuda_stream) runs from now on."""
        _check(self.lib, self.lib.prf_stream_wait_for(self._h, ctypes.c_void_p(other_stream)))

    def scan_wait(self, seq):
        """Collect a scan enqueued with Genome.scan_async(); returns its ScanStats (row and candidate counts)."""
        stats = ScanStats()
        _check(self.lib, self.lib.prf_scan_wait(self._h, seq, ctypes.byref(stats)))
        return stats

    def set_row_sink(self, dst_ptr, capacity_rows):
        """Following scans compact their rows into caller-owned device memory (capacity_rows + 1 records, the last
        one receives the row count); dst_ptr None/0 switches back to the library's own array."""
        _check(self.lib, self.lib.prf_set_row_sink(self._h, ctypes.c_void_p(dst_ptr or None), capacity_rows))

    def scan_timings(self, first_seq, n):
        """HIP-event kernel times (ms) of n fused scans from serial number first_seq (ScanStats.seq) on."""
        out = (ctypes.c_float * max(1, n))()
        _check(self.lib, self.lib.prf_scan_timings(self._h, first_seq, n, out))
        return [float(out[i]) for i in range(n)]

    def scan_timings_split(self, first_seq, n):
        """(scan kernel ms, gather kernel ms) lists of n fused scans from serial number first_seq on."""
        a = (ctypes.c_float * max(1, n))()
        b = (ctypes.c_float * max(1, n))()
        _check(self.lib, self.lib.prf_scan_timings_split(self._h, first_seq, n, a, b))
        return [float(a[i]) for i in range(n)], [float(b[i]) for i in range(n)]

    def measure_hbm_read(self, nbytes=1 << 30, iters=5):
        out = ctypes.c_double(0)
        _check(self.lib, self.lib.prf_measure_hbm_read(self._h, nbytes, iters, ctypes.byref(out)))
        return out.value

    def close(self):
        if self._h is not None:
            self.lib.prf_close(self._h)
            self._h = None


class Fasta:
    """FASTA file read by libprf (plain or gzip): entries in file order; the sequence bytes stay in native memory."""

    class Entry:
        def __init__(self, name, addr, length):
            self.name, self.addr, self.length = name, addr, length

        def __len__(self):
            return self.length

        @property
        def seq(self):
            return ctypes.string_at(self.addr, self.length).decode("ascii", "replace") if self.length else ""

    def __init__(self, path, only=None):
        """only: read just that record (by seeking, if `path`.fai exists and the file is not compressed)."""
        self.lib = load_library()
        h = ctypes.c_void_p()
        if only is None:
            _check(self.lib, self.lib.prf_fasta_open(os.fsencode(path), ctypes.byref(h)))
        else:
            _check(self.lib, self.lib.prf_fasta_open_contig(os.fsencode(path), only.encode(), ctypes.byref(h)))
        self._h = h
        self.entries = []
        for i in range(self.lib.prf_fasta_count(h)):
            name, seq, n = ctypes.c_char_p(), ctypes.c_void_p(), ctypes.c_uint64()
            _check(self.lib, self.lib.prf_fasta_entry(h, i, ctypes.byref(name), ctypes.byref(seq), ctypes.byref(n)))
            self.entries.append(Fasta.Entry(name.value.decode(), seq.value or 0, n.value))
        self._by_name = {}
        for e in self.entries:
            self._by_name.setdefault(e.name, e)

    def __iter__(self):
        return iter(self.entries)

    def __len__(self):
        return len(self.entries)

    def __contains__(self, name):
        return name in self._by_name

    def __getitem__(self, key):
        return self.entries[key] if isinstance(key, int) else self._by_name[key]

    def close(self):
        if self._h is not None:
            self.lib.prf_fasta_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fasta_index(path):
    """([(name, length), ...] in file order, whole).  Read from the samtools-style index `path`.fai if it lies next to an
    uncompressed file (then `whole` is None and single records are read by seeking, Fasta(path, only=name)); otherwise the
    file is parsed once and `whole` is the Fasta object that holds every record."""
    fai = os.fspath(path) + ".fai"
    if os.path.exists(fai) and not os.fspath(path).endswith(".gz"):
        index = []
        with open(fai) as f:
            for line in f:
                cols = line.rstrip("\n").split("\t")
                if len(cols) >= 5:
                    index.append((cols[0], int(cols[1])))
        if index:
            return index, None
    whole = Fasta(path)
    return [(e.name, len(e)) for e in whole], whole


def scan_fasta_to_bed(ctx, fasta, bed_path, kmin, kmax, min_repeats, min_span, on_contig=None):
    """All contigs of a FASTA in ONE resident genome and one scan; BED written by libprf.  Returns rows per contig.
    on_contig(entry, n_rows) is called per contig in file order (the CLI prints the reference's lines there)."""
    lib = ctx.lib
    entries = list(fasta)
    arr, _keep = _contig_array([(e.addr, e.length) for e in entries])
    hits, stats = _Hits(), ScanStats()
    _check(lib, lib.prf_scan(ctx._h, arr, len(entries), kmin, kmax, min_repeats, min_span, SCAN_DEFAULT, ctypes.byref(hits),
                             ctypes.byref(stats)))
    try:
        names = (ctypes.c_char_p * max(1, len(entries)))(*[e.name.encode() for e in entries])
        written = ctypes.c_uint64(0)
        _check(lib, lib.prf_write_bed(os.fsencode(bed_path), 0, names, arr, len(entries), ctypes.byref(hits), ctypes.byref(written)))
        import numpy as np
        counts = np.bincount(_rows(hits)["contig"], minlength=len(entries)) if hits.n else np.zeros(len(entries), dtype=np.int64)
    finally:
        lib.prf_free_hits(ctypes.byref(hits))
    if on_contig:
        for e, c in zip(entries, counts):
            on_contig(e, int(c))
    return [int(c) for c in counts], stats


def write_bed(bed_path, entries, rows):
    """BED through libprf's writer (host code) from a numpy row array (start, end, k, contig: 24-byte records, contig =
    index into entries, sorted as the file should be); returns rows per contig."""
    import numpy as np
    lib = load_library()
    rows = np.ascontiguousarray(rows)
    assert rows.dtype.itemsize == ctypes.sizeof(_Hit)
    arr, _keep = _contig_array([(e.addr, e.length) for e in entries])
    hits = _Hits()
    hits.rows = ctypes.cast(rows.ctypes.data, ctypes.POINTER(_Hit))
    hits.n = len(rows)
    names = (ctypes.c_char_p * max(1, len(entries)))(*[e.name.encode() for e in entries])
    written = ctypes.c_uint64(0)
    _check(lib, lib.prf_write_bed(os.fsencode(bed_path), 0, names, arr, len(entries), ctypes.byref(hits), ctypes.byref(written)))
    counts = np.bincount(rows["contig"], minlength=len(entries)) if len(rows) else np.zeros(len(entries), dtype=np.int64)
    return [int(c) for c in counts]


def tile_positions():
    return int(load_library().prf_tile_positions())


def plan_describe(kmin, kmax, min_repeats, min_span):
    """Host-only: the fused kernel's work plan for these parameters, as a dict (no GPU needed)."""
    import json
    lib = load_library()
    buf = ctypes.create_string_buffer(1 << 16)
    rc = lib.prf_plan_describe(kmin, kmax, min_repeats, min_span, buf, len(buf))
    if rc < 0:
        raise PrfError(rc, lib.prf_last_error().decode("utf-8", "replace"))
    return json.loads(buf.value.decode())


_default_ctx = {}


def default_context(device=None):
    """Process-wide context for detect_repeats(); device from PRF_DEVICE / LOCAL_RANK / 0."""
    if device is None:
        device = int(os.environ.get("PRF_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
