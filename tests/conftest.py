import gzip
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def gz_text(path):
    with gzip.open(path, "rt") as f:
        return f.read()


@pytest.fixture(scope="session")
def oracle_build():
    """Build the CPU restatement (test infrastructure) once per session."""
    d = os.path.join(REPO, "oracle")
    subprocess.check_call(["make", "-s", "-C", d])
    return d


@pytest.fixture(scope="session")
def c1_inputs():
    """Regenerate BASELINE.json configs[0] inputs from the committed seeds; verified by sha256."""
    import hashlib
    import json
    from phaser_amd import synth
    meta = json.load(open(os.path.join(GOLD, "c1", "meta.json")))
    g = meta["gen"]
    v, gs, ge, w = synth.make_variants(g["region"][0], g["region"][1], g["region"][2], g["n_snps"], g["vseed"],
                                       n_genes=g["n_genes"])
    rb = synth.make_reads(v, gs, ge, w, g["n_pairs"], g["rseed"])
    rf = rb.select(synth.samtools_keep(rb, g["mapq"]))
    sam = "\n".join(synth.sam_lines(rf, [("chr22", 50818468)])) + "\n"
    vcf = "\n".join(synth.vcf_lines([v])) + "\n"
    assert hashlib.sha256(sam.encode()).hexdigest() == meta["sam_sha256"], "synthetic generator drifted from the golden inputs"
    assert hashlib.sha256(vcf.encode()).hexdigest() == meta["vcf_sha256"]
    return {"meta": meta, "variants": v, "reads": rf, "sam": sam, "vcf": vcf}
