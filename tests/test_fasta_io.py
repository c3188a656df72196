"""CPU: the host-side data formats of libprf -- FASTA reader (plain / gzip, CRLF, long lines, empty records, lower
case, text before the first header) and the BED / TSV row writers -- against plain Python and the oracle."""
import ctypes
import gzip
import os

import numpy as np

import prf_native
from oracle import prf_oracle


def make_fasta(path, records, gz=False, crlf=False, width=60):
    nl = "\r\n" if crlf else "\n"
    text = "; a comment line before the first record" + nl
    for name, desc, seq in records:
        text += f">{name}{desc}{nl}"
        for i in range(0, len(seq), width):
            text += seq[i:i + width] + nl
        if not seq:
            text += nl
    opener = gzip.open if gz else open
    with opener(path, "wt", newline="") as f:
        f.write(text)


RECORDS = [("chr1", " first record", "ACGTNacgtn" * 37 + "CAG" * 40),
           ("empty", "", ""),
           ("chr_2", "\tdescription with tab", "TTTTTTTTTTTTGACCA" * 500),
           ("scaffold|3", " x", "N" * 200 + "AC" * 300 + "g" * 33)]


def test_reader_matches_python(tmp_path):
    for gz, crlf, width in ((False, False, 60), (True, False, 70), (False, True, 61), (True, True, 100000)):
        path = str(tmp_path / f"t_{gz}_{crlf}_{width}.fa")
        make_fasta(path, RECORDS, gz=gz, crlf=crlf, width=width)
        fa = prf_native.Fasta(path)
        assert [e.name for e in fa] == [r[0] for r in RECORDS]
        assert [e.seq for e in fa] == [r[2] for r in RECORDS]
        assert "chr_2" in fa and "nope" not in fa and fa["scaffold|3"].length == len(RECORDS[3][2])
        fa.close()
    # a header that is the last line, without newline
    path = str(tmp_path / "tail.fa")
    open(path, "w").write(">a\nACGT\n>b")
    fa = prf_native.Fasta(path)
    assert [(e.name, e.seq) for e in fa] == [("a", "ACGT"), ("b", "")]


def test_missing_file_is_an_error():
    try:
        prf_native.Fasta("/nonexistent/file.fa")
    except prf_native.PrfError as exc:
        assert exc.code == prf_native.PRF_EINVAL
    else:
        raise AssertionError("expected PrfError")


def test_bed_and_tsv_writers_match_the_reference_format(tmp_path):
    lib = prf_native.load_library()
    seqs = [r[2].encode() for r in RECORDS]
    names = [r[0] for r in RECORDS]
    rows = []
    want = ""
    for ci, s in enumerate(seqs):
        for a, b, _ml, k in prf_oracle.detect_rows(s, 1, 20, 3, 9):
            rows.append((a, b, k, ci))
            want += f"{names[ci]}\t{a}\t{b}\t{s[a:a + k].decode().upper()}\n"
    assert len(rows) > 5
    arr = (prf_native._Hit * len(rows))(*[prf_native._Hit(*r) for r in rows])
    hits = prf_native._Hits(arr, len(rows))
    contigs, keep = prf_native._contig_array(seqs)
    cnames = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
    n = ctypes.c_uint64()
    bed = str(tmp_path / "out.bed")
    assert lib.prf_write_bed(bed.encode(), 0, cnames, contigs, len(seqs), ctypes.byref(hits), ctypes.byref(n)) == 0
    assert n.value == len(rows) and open(bed).read() == want
    assert lib.prf_write_bed(bed.encode(), 1, cnames, contigs, len(seqs), ctypes.byref(hits), None) == 0
    assert open(bed).read() == want + want
    # TSV of one contig
    r0 = [r for r in rows if r[3] == 0]
    arr0 = (prf_native._Hit * len(r0))(*[prf_native._Hit(a, b, k, 0) for a, b, k, _c in r0])
    hits0 = prf_native._Hits(arr0, len(r0))
    tsv = str(tmp_path / "out.tsv")
    assert lib.prf_write_tsv(tsv.encode(), contigs, ctypes.byref(hits0), None) == 0
    assert open(tsv).read() == "start_0based\tend\tmotif\n" + "".join(
        f"{a}\t{b}\t{seqs[0][a:a + k].decode().upper()}\n" for a, b, k, _c in r0)
    # a row that does not fit its contig is refused
    bad = (prf_native._Hit * 1)(prf_native._Hit(10, 20, 5, 1))   # contig 1 is empty
    assert lib.prf_write_bed(bed.encode(), 0, cnames, contigs, len(seqs), ctypes.byref(prf_native._Hits(bad, 1)), None) == prf_native.PRF_EINVAL


def test_single_record_access_with_and_without_an_index(tmp_path):
    """prf_fasta_open_contig: by seeking through a samtools-style .fai, by parsing without one, from gzip, with a
    stale index, and an absent name."""
    import gzip
    import prf_native
    recs = {"chr1": "ACGT" * 50 + "N" * 7, "chrEmptyLine": "acgtn" * 31, "chr2 desc": "G" * 1, "chr3": "TTAGGG" * 40}
    path = tmp_path / "g.fa"
    fai = []
    with open(path, "wb") as f:
        for header, seq in recs.items():
            f.write(f">{header}\n".encode())
            off = f.tell()
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60].encode() + b"\n")
            fai.append(f"{header.split()[0]}\t{len(seq)}\t{off}\t60\t61\n")
    names = {h.split()[0]: s for h, s in recs.items()}

    def fetch(p, name):
        fa = prf_native.Fasta(str(p), only=name)
        return [(e.name, e.seq) for e in fa]
    for name, seq in names.items():                      # no index: parse and filter
        assert fetch(path, name) == [(name, seq)]
    assert fetch(path, "chrNope") == []
    with open(str(path) + ".fai", "wt") as f:
        f.writelines(fai)
    for name, seq in names.items():                      # with the index: seek
        assert fetch(path, name) == [(name, seq)]
    assert fetch(path, "chrNope") == []
    with open(str(path) + ".fai", "wt") as f:            # stale index (wrong offsets): detected, falls back to parsing
        f.writelines(line.replace("\t60\t61", "\t50\t51") for line in fai)
    for name, seq in names.items():
        assert fetch(path, name) == [(name, seq)]
    # hostile / damaged indexes: never trusted further than the file goes, never the wrong record
    twin = tmp_path / "twin.fa"                           # two records of the same length: an index whose offsets are swapped
    with open(twin, "wb") as f:
        f.write(b">a\n" + b"ACGT" * 5 + b"\n>b\n" + b"TTTT" * 5 + b"\n")
    with open(str(twin) + ".fai", "wt") as f:
        f.write("a\t20\t27\t20\t21\nb\t20\t3\t20\t21\n")
    assert fetch(twin, "a") == [("a", "ACGT" * 5)] and fetch(twin, "b") == [("b", "TTTT" * 5)]
    for bad in ("a\t20\t99999999999\t20\t21\n", "a\t18446744073709551615\t3\t20\t21\n", "a\t20\t3\t1\t4000000000\n",
                "a\t20\t3\t0\t0\n", "a\tx\ty\n"):
        with open(str(twin) + ".fai", "wt") as f:
            f.write(bad)
        assert fetch(twin, "a") == [("a", "ACGT" * 5)], bad
    gz = tmp_path / "g.fa.gz"
    with gzip.open(gz, "wb") as f:
        f.write(open(path, "rb").read())
    assert fetch(gz, "chr3") == [("chr3", names["chr3"])]
    # an index whose LENGTH is stale (shorter than the record, same line geometry): the seek path must not hand over a
    # truncated record -- it notices that the record goes on and parses the file instead
    with open(str(path) + ".fai", "wt") as f:
        f.writelines("\t".join([ln.split("\t")[0], str(int(ln.split("\t")[1]) - 7)] + ln.split("\t")[2:]) if ln.startswith("chr3") else ln
                     for ln in fai)
    assert fetch(path, "chr3") == [("chr3", names["chr3"])]
    assert dict(prf_native.fasta_index(str(path))[0])["chr3"] == len(names["chr3"]) - 7      # (the index itself is what it is)


def test_malformed_inputs_never_crash_the_reader(tmp_path):
    """Untrusted files: random bytes, a header only, control characters, a truncated gzip stream, a directory, an index of
    garbage -- every call returns (an error, or whatever records the bytes spell), none reads outside its buffers.  Run
    under AddressSanitizer / UBSan by tests/test_asan_host.py."""
    rng = np.random.default_rng(11)
    blobs = [b"", b">", b">\n", b">a", b"\n\n\n", b"ACGT", b">a\n>b\n>c", b">x\r\nAC\rGT\r\n\r\n", b"\x00" * 100, b">n\n" + b"\xff" * 50,
             b">" + b"A" * 10000 + b"\n" + b"C" * 3, b">a\n" + b"ACGT\n" * 1000 + b">a\n" + b"T" * 7]
    blobs += [rng.integers(0, 256, size=int(n), dtype=np.uint8).tobytes() for n in (1, 17, 4096, 70000)]
    blobs += [b">r%d\n" % i + rng.choice(list(b"ACGTNacgtn\n>\r \t"), size=300).astype(np.uint8).tobytes() for i in range(20)]
    for i, blob in enumerate(blobs):
        path = str(tmp_path / f"m{i}.fa")
        open(path, "wb").write(blob)
        for only in (None, "a", "r3"):
            try:
                fa = prf_native.Fasta(path, only=only)
                for e in fa:
                    assert len(e.seq) == e.length
                fa.close()
            except prf_native.PrfError as exc:
                assert exc.code in (prf_native.PRF_EINVAL, prf_native.PRF_ENOMEM)
        # the same bytes behind an index of garbage
        open(path + ".fai", "wb").write(rng.integers(0, 256, size=64, dtype=np.uint8).tobytes() + b"\na\t5\t3\t5\t6\nr3\t300\t4\t300\t301\n")
        for only in ("a", "r3"):
            try:
                prf_native.Fasta(path, only=only).close()
            except prf_native.PrfError:
                pass
        try:
            prf_native.fasta_index(path)
        except (prf_native.PrfError, ValueError):
            pass
    whole = gzip.compress(b">g\n" + b"ACGT" * 5000 + b"\n")
    for cut in (0, 1, 10, len(whole) // 2, len(whole) - 1):
        path = str(tmp_path / f"cut{cut}.fa.gz")
        open(path, "wb").write(whole[:cut])
        try:
            prf_native.Fasta(path).close()
        except prf_native.PrfError as exc:
            assert exc.code == prf_native.PRF_EINVAL
    try:
        prf_native.Fasta(str(tmp_path))
    except prf_native.PrfError as exc:
        assert exc.code == prf_native.PRF_EINVAL
