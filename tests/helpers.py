"""Shared helpers for the tests (text renderings that mirror the reference's file formats)."""
from phaser_amd import synth


def variant_table_text(v, maf="None"):
    """The mapper's variant table as generate_mapping_table writes it (phaser.py:1402-1404)."""
    rows = []
    pos = v.pos.tolist(); ref = v.ref.tolist(); alt = v.alt.tolist()
    for i in range(len(v)):
        r, a = synth.BASES[ref[i]], synth.BASES[alt[i]]
        uid = "%s_%d_%s_%s" % (v.chrom, pos[i], r, a)
        rows.append("\t".join([v.chrom, str(pos[i]), uid, v.rsid[i], r + "," + a, "1", v.gt[i], maf]))
    return "\n".join(rows) + "\n"
