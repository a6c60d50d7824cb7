"""phaser_expr_matrix drop-in vs what the reference's script wrote (tests/golden/expr_matrix, both directory orders)."""
import gzip
import os
import shutil

import pytest

from conftest import GOLD, gz_text


@pytest.mark.parametrize("order", ["sorted", "reversed"])
def test_expr_matrix_matches_reference(tmp_path, order):
    from phaser_amd import expr_matrix
    d = os.path.join(GOLD, "expr_matrix")
    gdir = tmp_path / "in"; gdir.mkdir()
    for fn in os.listdir(os.path.join(d, "gene_ae")):
        (gdir / fn[:-3]).write_text(gz_text(os.path.join(d, "gene_ae", fn)))
    a, g, log = expr_matrix.expr_matrix(str(gdir), os.path.join(d, "features.bed"), order)
    assert a == gz_text(os.path.join(d, "out.%s.bed.gz" % order))
    assert g == gz_text(os.path.join(d, "out.%s.gw_phased.bed.gz" % order))
    ref_log = gz_text(os.path.join(d, "out.%s.log.txt.gz" % order))
    assert len(log) == 2
    for l in log:
        assert l.split(":", 1)[1] in ref_log          # same error line (the path prefix differs)


def test_expr_matrix_cli_writes_bgzf(tmp_path):
    from phaser_amd import _lib, expr_matrix
    _lib.build()
    d = os.path.join(GOLD, "expr_matrix")
    gdir = tmp_path / "in"; gdir.mkdir()
    for fn in os.listdir(os.path.join(d, "gene_ae")):
        (gdir / fn[:-3]).write_text(gz_text(os.path.join(d, "gene_ae", fn)))
    assert expr_matrix.main(["--gene_ae_dir", str(gdir), "--features", os.path.join(d, "features.bed"), "--o", str(tmp_path / "m"), "--t", "2"]) == 0
    assert gzip.open(str(tmp_path / "m.bed.gz"), "rt").read() == gz_text(os.path.join(d, "out.sorted.bed.gz"))
