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
