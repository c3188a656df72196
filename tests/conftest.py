import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "colab-repeat-finder_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_jsonl_gz(name):
    with gzip.open(os.path.join(GOLDEN, name), "rt") as f:
        return [json.loads(line) for line in f]


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def load_clusters():
    out = []
    with gzip.open(os.path.join(GOLDEN, "chr22_clusters.tsv.gz"), "rt") as f:
        for line in f:
            if line.startswith("#"):
                continue
            pos, seq, rows = line.rstrip("\n").split("\t")
            want = []
            for r in rows.split(","):
                a, b, m = r.split(":")
                want.append((int(a), int(b), m))
            out.append((int(pos), seq, want))
    return out


@pytest.fixture(scope="session")
def golden_fuzz():
    return load_jsonl_gz("fuzz_small.jsonl.gz")


@pytest.fixture(scope="session")
def golden_adversarial():
    return load_jsonl_gz("adversarial.jsonl.gz")


@pytest.fixture(scope="session")
def golden_min_repeats_one():
    return load_jsonl_gz("min_repeats_one.jsonl.gz")


@pytest.fixture(scope="session")
def golden_odd_intervals():
    return load_jsonl_gz("odd_intervals.jsonl.gz")


@pytest.fixture(scope="session")
def golden_iupac():
    return load_jsonl_gz("iupac.jsonl.gz")


@pytest.fixture(scope="session")
def golden_unit():
    return load_json("ref_unit_tests.json")["cases"]


@pytest.fixture(scope="session")
def golden_clusters():
    return load_clusters()
