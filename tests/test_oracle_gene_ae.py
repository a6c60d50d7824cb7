"""Pins oracle/gene_ae_oracle.py against what the reference's own phaser_gene_ae.py wrote (tests/golden/gene_ae/*)."""
import json
import os
import sys

import pytest

from conftest import GOLD, REPO, gz_text

CASES = sorted(os.listdir(os.path.join(GOLD, "gene_ae")))


def case_inputs(name):
    d = os.path.join(GOLD, "gene_ae", name)
    meta = json.load(open(os.path.join(d, "case.json")))
    kw = {}
    a = meta["args"]
    for i in range(0, len(a), 2):
        k = a[i].lstrip("-")
        kw[k] = int(a[i + 1]) if k == "min_cov" else float(a[i + 1])
    return gz_text(os.path.join(GOLD, meta["haplotypic_counts"])), open(os.path.join(d, "features.bed")).read(), kw, \
        gz_text(os.path.join(d, "out.gene_ae.txt.gz"))


@pytest.mark.parametrize("name", CASES)
def test_gene_ae_oracle_matches_reference(name):
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import gene_ae_oracle as go
    hc, bed, kw, want = case_inputs(name)
    got = go.gene_ae(hc, bed, **kw)
    assert go.canonical(got) == go.canonical(want)
