"""Phased VCF output -- phaser/phaser.py:1661-1855 (`write_vcf`), SURVEY.md 8(f) next-2.

Input is the sample's VCF cut to columns 1-9 + sample (what `gunzip -c | cut -f 1-9,S` hands the reference) and
the per-variant block lookup built while the blocks were written (engine.merge_fragments).  Output text is what the
reference writes to <o>.vcf before compressing it; we compress it as BGZF ourselves (bamio._bgzf_block) and do
not write a tabix index.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

from . import bamio

TAGS = ['PG', 'PB', 'PI', 'PW', 'PC', 'PM']


def phased_vcf_text(cut_lines: List[str], lookup: Dict[str, tuple], id_separator: str = "_", chromosome_of_interest: str = "",
                    gw_phase_vcf: int = 0, min_confidence: float = 0.9) -> Tuple[str, int, int]:
    """-> (vcf text, unphased_phased, phase_corrections)"""
    out: List[str] = []
    format_text = ""
    corrections = unphased_phased = 0
    for line in cut_lines:
        c = line.split("\t")
        if "##FORMAT" in line:
            format_text += line + "\n"
            out.append(line + "\n")
        elif line.startswith("#CHROM"):
            for tag, desc in [("PG", "phASER Local Genotype"), ("PB", "phASER Local Block"),
                              ("PI", "phASER Local Block Index (unique for each block)"), ("PM", "phASER Local Block Maximum Variant MAF"),
                              ("PW", "phASER Genome Wide Genotype"), ("PC", "phASER Genome Wide Confidence")]:
                if "##FORMAT=<ID=%s," % tag not in format_text:
                    out.append("##FORMAT=<ID=%s,Number=1,Type=String,Description=\"%s\">\n" % (tag, desc))
            if gw_phase_vcf == 2 and "##FORMAT=<ID=PS," not in format_text:
                out.append("##FORMAT=<ID=PS,Number=1,Type=String,Description=\"Phase Set\">\n")
            out.append("\t".join(c[0:9] + [c[9]]) + "\n")
        elif line[0:1] == "#":
            out.append(line + "\n")
        else:
            chrom = c[0]; pos = int(c[1])
            if not (chromosome_of_interest == "" or chrom == chromosome_of_interest):
                continue
            if "GT" in c[8]:
                gt_index = c[8].split(":").index("GT")
                genotype = list(c[9].split(":")[gt_index])
                if "|" in genotype:
                    genotype.remove("|")
                if "/" in genotype:
                    genotype.remove("/")
                all_alleles = [c[3]] + c[4].split(",")
                n_fields = len(c[8].split(":"))
                for i in range(9, len(c)):
                    have = len(c[i].split(":"))
                    if have != n_fields:
                        c[i] += ":" * (n_fields - have)
                fmt = c[8].split(":")
                for tag in TAGS:
                    if tag not in fmt:
                        fmt.append(tag)
                c[8] = ":".join(fmt)
                # rebuilt WITHOUT --chr_prefix, as the reference does (phaser.py:1763): with a prefix nothing matches there either
                uid = chrom + id_separator + str(pos) + id_separator + id_separator.join(all_alleles)
                hit = lookup.get(uid)
                if hit is not None:
                    v, i, block_index = hit
                    alleles_out = []; gw_out = ["", ""]
                    for a in v["hap"][i].split("|"):
                        base = v["alleles"][i][int(a)]
                        vidx = all_alleles.index(base)
                        g = v["gw"][i][int(a)]
                        if g is not None:
                            gw_out[g] = str(vidx)
                        alleles_out.append(str(vidx))
                    names = [r.replace(":", "_") for r in v["rsids"]]
                    stat = v["stat"]
                    if "-" not in gw_out:
                        x = c[9].split(":")
                        new_phase = "|".join(gw_out)
                        if stat >= min_confidence:
                            if "|" in x[gt_index] and x[gt_index] != new_phase:
                                corrections += 1
                            if "/" in x[gt_index] and x[gt_index] != "./." and x[gt_index] != new_phase:
                                unphased_phased += 1
                            if gw_phase_vcf in (1, 2):
                                x[gt_index] = new_phase
                                c[9] = ":".join(x)
                        if gw_phase_vcf == 2 and stat < min_confidence:
                            x[gt_index] = "|".join(alleles_out)
                            c[9] = ":".join(x)
                    sf = c[9].split(":")
                    sf += [''] * (len(fmt) - len(sf))
                    sf[fmt.index('PG')] = "|".join(alleles_out)
                    sf[fmt.index('PB')] = ",".join(names)
                    sf[fmt.index('PI')] = str(block_index)
                    sf[fmt.index('PM')] = v["max_maf_txt"]
                    sf[fmt.index('PW')] = "|".join(gw_out)
                    sf[fmt.index('PC')] = v["stat_txt"]
                    if gw_phase_vcf == 2 and stat < min_confidence:
                        if 'PS' not in fmt:
                            c[8] += ":PS"; fmt.append("PS"); sf.append('')
                        sf[fmt.index('PS')] = str(block_index)
                    c[9] = ":".join(sf)
                else:
                    sf = c[9].split(":")
                    sf += [''] * (len(fmt) - len(sf))
                    sf[fmt.index('PG')] = "/".join(sorted(genotype))
                    sf[fmt.index('PB')] = '.'; sf[fmt.index('PI')] = '.'; sf[fmt.index('PM')] = '.'
                    sf[fmt.index('PW')] = c[9].split(":")[gt_index]
                    sf[fmt.index('PC')] = '.'
                    c[9] = ":".join(sf)
            out.append("\t".join(c[0:9] + [c[9]]) + "\n")
    return "".join(out), unphased_phased, corrections


def write_bgzf(path: str, text, threads: int = 0):
    """BGZF-compress the phased VCF text (what the reference gets from `bgzip`, phaser.py:1851) with the native parallel writer."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    data = text.encode() if isinstance(text, str) else bytes(text)
    st = lib.phz_bgzf_write(path.encode(), C.cast(C.c_char_p(data), C.c_void_p), len(data), int(threads), 6)
    if st != _lib.PHZ_OK:
        raise _lib.PhzError(st, "phz_bgzf_write(%s) failed" % path)
