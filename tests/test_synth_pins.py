"""CPU: the synthetic inputs of bench.py and of the parity tests are pinned by length + SHA-256: the
numpy generator, the oracle's C copy of it and the committed golden fixtures must all describe the same bytes, round
after round; the device copy (prf_genome_synth) is pinned against the oracle's in the GPU tests."""
import hashlib

from conftest import load_json


def test_counter_based_generator_is_pinned():
    import synth
    from oracle import prf_oracle
    a = synth.synth_bases(1_000_000, 22).tobytes()
    assert hashlib.sha256(a).hexdigest() == "212cc50d259d7c6f05cbf0499c40b496cd6bee316fb1ca17ffd1dbd0f5583f95"
    assert prf_oracle.synth(1_000_000, 22) == a
    assert prf_oracle.synth(1000, 22, start=777_000) == a[777_000:778_000]
    assert a[:64].decode() == load_json("synth_n1000000_seed22_k1-50.json")["head"]


def test_chr22_standin_of_the_headline_benchmark_is_pinned():
    """BASELINE config C2's stand-in (bench.py default workload): 50 818 468 bp; 76 379 rows at motif 1-50 (GPU tests)."""
    import synth
    s = synth.chr_standin().tobytes()
    assert len(s) == synth.CHR22_LEN == 50_818_468
    assert hashlib.sha256(s).hexdigest() == "dab186cc2a38ee8c4d025758598234bf3d6386e985f9467ab42588ce2bc0bc7a"
    assert s[:10_510_000] == b"N" * 10_510_000 and s[-10_000:] == b"N" * 10_000


def test_chr22_real_is_pinned():
    """The chr22 stand-in with every cluster of the reference's golden BED at its real coordinate (bench.py --workload
    chr22-real; the GPU test scans it against the oracle): 67 538 of the BED's 67 638 rows are planted (100 belong to clusters
    whose flank base collides with a neighbour's), in 582 tiles of 65536 positions, the densest with 748 rows."""
    import collections
    import synth
    seq, rows = synth.chr22_real()
    assert len(seq) == synth.CHR22_LEN and len(rows) == 67_538
    assert hashlib.sha256(seq.tobytes()).hexdigest() == "9f475c4ad96a6e4c32a9150718622d7abfd1dcf47f984b8bf5d007eb5609a778"
    assert hashlib.sha256(rows.tobytes()).hexdigest() == "06ab3d2b05042deb42cb35840b66660d407f528445495ae1c57716b4c7557dd3"
    per_tile = collections.Counter((rows["start"] // 65_536).tolist())
    assert len(per_tile) == 582 and max(per_tile.values()) == 748
