#!/usr/bin/env python3
"""One worker of the all-cores CPU figure of bench.py's phasing baseline: oracle/phasing_oracle.py (the CPU restatement of process_vcf's
stages T1-O2, test infrastructure) on the call file of ONE chromosome -- the unit the reference's `parallelize` hands to a pool worker
(phaser/phaser.py:2077-2094; BASELINE.md section 3: one process per chromosome).  Prints "<phased variants> <seconds> <sha256 of the five
files in canonical form>"."""
import hashlib
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle")); sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import phasing_oracle as po
    from helpers import OUTPUTS, canonical
    paths, baseq = sys.argv[1].split(","), int(sys.argv[2])               # one call file per BAM (comma-separated), the chromosome's lines of that BAM
    out_dir = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None          # optional: the five files in canonical form are written there
    # optional: the run's global AS cutoffs (one per BAM, comma-separated) and noise level (the chromosome is one shard of a whole-genome run)
    cutoffs = [float(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 and sys.argv[4] != "-" else None
    noise = float.fromhex(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5] != "-" else None
    names = sys.argv[6].split(",") if len(sys.argv) > 6 else (["bench"] if len(paths) == 1 else ["bam%d" % b for b in range(len(paths))])
    texts = [open(p).read() for p in paths]
    t0 = time.perf_counter()
    ph = po.Phaser(names, baseq=baseq, global_as_cutoffs=cutoffs, global_noise=noise)
    for text in texts:
        ph.add_bam([text])
    out = ph.finish()
    dt = time.perf_counter() - t0
    h = hashlib.sha256()
    for n in OUTPUTS:
        c = canonical(n, out[n])
        h.update(c.encode())
        if out_dir:
            open(os.path.join(out_dir, n + ".txt"), "w").write(c)
    print(ph.phased, "%.3f" % dt, h.hexdigest())


if __name__ == "__main__":
    main()
