"""SAM text -> read shards (the text path of the drop-in; BAM decoding lives in bamio).

Applies what the mapper applies to a record before mapping (read_variant_map.py:33-64): |TLEN| <= isize
when an insert-size cutoff is given, AS = last AS: tag.  QNAMEs are interned per chromosome so mates and
the same template in several BAMs share one id (the reference keys reads by QNAME string, phaser.py:1305).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from . import soa


class QnameInterner:
    def __init__(self):
        self.ids: Dict[str, int] = {}
        self.names: List[str] = []

    def __call__(self, name: str) -> int:
        i = self.ids.get(name)
        if i is None:
            i = self.ids[name] = len(self.names)
            self.names.append(name)
        return i

    def __len__(self):
        return len(self.names)


def shards_from_sam(sam_text: str, interners: Dict[str, QnameInterner], isize_cutoff: float = 0.0
                    ) -> Dict[str, soa.ReadShard]:
    """-> {chrom: ReadShard with qid / aln_score / has_as}, records in input order."""
    by: Dict[str, list] = {}
    for line in sam_text.split("\n"):
        if not line or line[0] == "@":
            continue
        c = line.rstrip().split("\t")
        if len(c) < 11:
            continue
        if not (isize_cutoff == 0 or abs(int(c[8])) <= isize_cutoff):
            continue
        a = None
        for i in range(11, len(c)):
            if c[i].startswith("AS:"):
                a = int(c[i].split(":")[2])
        by.setdefault(c[2], []).append((c[0], int(c[3]), c[5], c[9], c[10], a))
    out = {}
    for chrom, recs in by.items():
        it = interners.setdefault(chrom, QnameInterner())
        sh = soa.pack_sam([(r[1], r[2], r[3], r[4]) for r in recs])
        sh.qid = torch.tensor([it(r[0]) for r in recs], dtype=torch.int32)
        sh.aln_score = torch.tensor([0 if r[5] is None else r[5] for r in recs], dtype=torch.int32)
        sh.has_as = torch.tensor([0 if r[5] is None else 1 for r in recs], dtype=torch.uint8)
        out[chrom] = sh
    return out
