#!/usr/bin/env python3
"""GPU-box helper: the CLI on the files tools/run_cli_scale.py left in /tmp (run that first in the same gpurun command), once per environment
setting given on the command line ("VAR=val VAR2=val2" per argument; "" = defaults), total and BAM-stage seconds per run.
usage: tools/cli_env_sweep.py "" "PHZ_BAM_NCOPY=12" ..."""
import io, os, re, sys, time, contextlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from phaser_amd import phaser
os.environ["PHZ_TIMING"] = "1"
for setting in sys.argv[1:] or [""]:
    keys = []
    for kv in setting.split():
        k, v = kv.split("=", 1); os.environ[k] = v; keys.append(k)
    best = None
    for rep in range(2):
        err = io.StringIO()
        t0 = time.perf_counter()
        with contextlib.redirect_stderr(err), contextlib.redirect_stdout(io.StringIO()):
            phaser.main(["--vcf", "/tmp/cli_scale.vcf.gz", "--bam", "/tmp/cli_scale.bam", "--sample", "S1", "--mapq", "255", "--baseq", "10", "--paired_end", "1",
                         "--o", "/tmp/cli_sweep_out", "--threads", "64", "--write_vcf", "1"])
        dt = time.perf_counter() - t0
        m = re.search(r"H2D \+ K_inflate \(\+ free of the compressed copy\)\s+([0-9.]+) ms", err.getvalue())
        b = re.search(r"bam decode \+ filters \+ qname interning\s+([0-9.]+) s", err.getvalue())
        row = (dt, float(m.group(1)) if m else -1.0, float(b.group(1)) if b else -1.0)
        if best is None or row[0] < best[0]:
            best = row
    print("%-44s total %.3f s   H2D + inflate %.0f ms   BAM stage %.2f s" % (setting or "(default)", best[0], best[1], best[2]), flush=True)
    for k in keys:
        del os.environ[k]
