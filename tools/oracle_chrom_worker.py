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
    path, baseq = sys.argv[1], int(sys.argv[2])
    out_dir = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None          # optional: the five files in canonical form are written there
    cutoff = float(sys.argv[4]) if len(sys.argv) > 4 else None           # optional: the run's global AS cutoff and noise level (the chromosome is
    noise = float.fromhex(sys.argv[5]) if len(sys.argv) > 5 else None    # one shard of a whole-genome run)
    text = open(path).read()
    t0 = time.perf_counter()
    ph = po.Phaser(["bench"], baseq=baseq, global_as_cutoffs=None if cutoff is None else [cutoff], global_noise=noise)
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
