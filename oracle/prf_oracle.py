"""ctypes front end of the CPU oracle (oracle/prf_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of prf_oracle.c.  Only tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this
module.  It mirrors the call signature and the error behaviour of the
reference's ``detect_repeats`` (reference perfect_repeat_finder.py:10-81) so a
parity test can swap one for the other.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("PRF_ORACLE_LIB") or os.path.join(_HERE, "libprf_oracle.so")   # PRF_ORACLE_LIB: the sanitizer build


class _Row(ctypes.Structure):
    _fields_ = [("start", ctypes.c_int64), ("end", ctypes.c_int64),
                ("motif_len", ctypes.c_int32), ("k", ctypes.c_int32)]


def build(force=False):
    """Compile the C restatement with gcc (no HIP, no GPU)."""
    src = os.path.join(_HERE, "prf_oracle.c")
    if os.environ.get("PRF_ORACLE_LIB"):
        return _LIB_PATH                      # built by whoever set the variable (tests/test_asan_host.py)
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-Wall", "-o", _LIB_PATH, src])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.prf_oracle_detect.restype = ctypes.c_int
        _lib.prf_oracle_detect.argtypes = [
            ctypes.c_char_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
            ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
            ctypes.POINTER(ctypes.POINTER(_Row)), ctypes.POINTER(ctypes.c_int64)]
        _lib.prf_oracle_free.argtypes = [ctypes.POINTER(_Row)]
        _lib.prf_oracle_free.restype = None
        _lib.prf_oracle_synth.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_uint64]
        _lib.prf_oracle_synth.restype = None
    return _lib


def _validate(fs):
    # reference perfect_repeat_finder.py:23-30 (getattr without default -> AttributeError)
    if not getattr(fs, "min_motif_size") or fs.min_motif_size < 1:
        raise ValueError(f"min_motif_size is set to {fs.min_motif_size}. It must be at least 1.")
    if not getattr(fs, "max_motif_size") or fs.max_motif_size < fs.min_motif_size:
        raise ValueError(f"max_motif_size is set to {fs.max_motif_size}. It must be at least min_motif_size.")
    if not getattr(fs, "min_repeats") or fs.min_repeats < 1:
        raise ValueError(f"min_repeats is set to {fs.min_repeats}. It must be at least 1.")
    if not getattr(fs, "min_span") or fs.min_span < 1:
        raise ValueError(f"min_span is set to {fs.min_span}. It must be at least 1.")


def detect_rows(seq_bytes, kmin, kmax, min_repeats, min_span, interval=None):
    """Raw rows [(start, end, motif_len, k)] for an ASCII byte string."""
    rows = ctypes.POINTER(_Row)()
    n = ctypes.c_int64(0)
    has_iv = 0 if interval is None else 1
    a, b = (0, 0) if interval is None else interval
    rc = lib().prf_oracle_detect(seq_bytes, len(seq_bytes), kmin, kmax, min_repeats, min_span,
                                 has_iv, a, b, ctypes.byref(rows), ctypes.byref(n))
    if rc == 1:
        raise AssertionError("RepeatTracker did not reach end of the sequence")
    if rc == 2:
        raise IndexError("string index out of range")
    if rc != 0:
        raise MemoryError("prf_oracle_detect failed")
    try:
        return [(rows[i].start, rows[i].end, rows[i].motif_len, rows[i].k) for i in range(n.value)]
    finally:
        lib().prf_oracle_free(rows)


def detect_repeats(input_sequence, filter_settings, verbose=False, show_progress_bar=False, debug=False):
    """Oracle twin of the reference's detect_repeats(): list of (start_0based, end, motif)."""
    _validate(filter_settings)
    raw = input_sequence.encode("ascii")
    interval = None
    if hasattr(filter_settings, "interval_start_0based") or hasattr(filter_settings, "interval_end"):
        interval = (getattr(filter_settings, "interval_start_0based", 0),
                    getattr(filter_settings, "interval_end", len(raw)))
    rows = detect_rows(raw, filter_settings.min_motif_size, filter_settings.max_motif_size,
                       filter_settings.min_repeats, filter_settings.min_span, interval)
    up = input_sequence.upper()
    return [(s, e, up[s:s + ml]) for (s, e, ml, _k) in rows]


def synth(n, seed, start=0):
    """SURVEY 8(d) counter-based generator, C copy; returns bytes."""
    buf = ctypes.create_string_buffer(n)
    lib().prf_oracle_synth(buf, start, n, seed)
    return buf.raw
