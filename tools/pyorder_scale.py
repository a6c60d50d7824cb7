#!/usr/bin/env python3
"""GPU-box tool: the raw-byte tier (--py_hash_order 1) at scale.  Inputs as tools/run_cli_scale.py writes them (/tmp/cli_scale.bam, /tmp/cli_scale.vcf.gz must
exist: run that tool first at the same scale).  Runs the CLI three ways as fresh processes and reports wall time and the stage line of the finish stage:
  fast path (canonical order), --py_hash_order 1 native (phz_pyorder_replay), and -- when asked (argv[1] == "twin") -- the pure-Python twin under
  PYTHONHASHSEED=0, whose five files must equal the native tier's byte for byte (sha256 of every file printed)."""
import hashlib, os, subprocess, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
twin = len(sys.argv) > 1 and sys.argv[1] == "twin"
NAMES = ("allelic_counts", "variant_connections", "haplotypes", "haplotypic_counts", "allele_config")


def run(tag, extra, env_extra):
    out = "/tmp/pyorder_" + tag
    env = dict(os.environ, PHZ_TIMING="1", PYTHONPATH=REPO, **env_extra)
    t0 = time.perf_counter()
    pr = subprocess.run([sys.executable, "-m", "phaser_amd.phaser", "--vcf", "/tmp/cli_scale.vcf.gz", "--bam", "/tmp/cli_scale.bam", "--sample", "S1", "--mapq", "255", "--baseq", "10",
                         "--paired_end", "1", "--o", out, "--threads", "32", "--write_vcf", "0"] + extra, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    dt = time.perf_counter() - t0
    lines = [l for l in pr.stdout.split("\n") if l.startswith("[phz timing] tally") or l.startswith("[phz timing] total") or "HOT PATH" in l or "PHASED" in l]
    sha = {n: hashlib.sha256(open(out + "." + n + ".txt", "rb").read()).hexdigest()[:16] for n in NAMES} if pr.returncode == 0 else {}
    print("=== %s: rc %d, process wall %.2f s\n%s\n    sha256/16: %s" % (tag, pr.returncode, dt, "\n".join(lines), sha), flush=True)
    if pr.returncode != 0:
        print(pr.stdout[-3000:])
    return sha


a = run("fast", [], {})
b = run("native", ["--py_hash_order", "1"], {"PYTHONHASHSEED": "31337"})
assert a["allelic_counts"] == b["allelic_counts"] and a["allele_config"] == b["allele_config"], "the two byte-stable files differ"
if twin:
    c = run("python_twin", ["--py_hash_order", "1"], {"PYTHONHASHSEED": "0", "PHZ_PYORDER_PYTHON": "1"})
    print("native tier == pure-Python twin (real CPython sets), all five files:", b == c)
    assert b == c
