#!/usr/bin/env python3
"""Runs ON THE GPU BOX.  Full-size parity of BASELINE.json configs[2] (22 autosomes, ~80M records, ~1.5M het SNPs, one BAM):
  1. K_map: the call list of EVERY record of EVERY chromosome shard against the C mapper oracle (oracle/rvm_oracle.c on all host cores);
  2. the five files of the WHOLE genome (device row stage) against oracle/phasing_oracle.py run as one process per chromosome on the full
     call lists -- the reference's own decomposition (`parallelize` over contigs, phaser.py:2077-2094) -- compared in canonical form
     (rows sorted, read labels renumbered by first appearance).
Prints one line per chromosome and a verdict; the log is kept under profiles/."""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch
from helpers import OUTPUTS, call_text, canonical, oracle_map_readbatch_threads, oracle_lib
from phaser_amd import dist as pdist, synth, vcf as pvcf, workloads
from phaser_amd.engine import Config, Engine
from phaser_amd.mapper import Mapper


def oracle_all_records(oracle_dir, rb, vpos, baseq, n_threads):
    """(read_idx, var_idx, code) of all records through the C oracle, record ranges on n_threads Python threads (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    lib = oracle_lib(oracle_dir)
    n = len(rb)
    pos = np.ascontiguousarray(rb.pos.numpy().astype(np.int32)); coff = np.ascontiguousarray(rb.cigar_off.numpy().astype(np.int64))
    cig = np.ascontiguousarray(rb.cigar.numpy().astype(np.uint32)); seq = np.ascontiguousarray(rb.seq.numpy()); qual = np.ascontiguousarray(rb.qual.numpy())
    vp = np.ascontiguousarray(np.asarray(vpos, dtype=np.int32)); rl = np.ones(len(vp), dtype=np.uint8)
    L = rb.L
    bounds = [n * t // n_threads for t in range(n_threads + 1)]

    def work(t):
        lo, hi = bounds[t], bounds[t + 1]
        m = hi - lo
        if m == 0:
            return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.uint8)
        cap = m + 4096
        while True:
            o_r = np.zeros(cap, np.int32); o_v = np.zeros(cap, np.int32); o_c = np.zeros(cap, np.uint8)
            k = lib.rvm_oracle_map_soa(m, pos.ctypes.data + 4 * lo, coff.ctypes.data + 8 * lo, cig.ctypes.data, seq.ctypes.data + L * lo, qual.ctypes.data + L * lo, L,
                                       baseq, len(vp), vp.ctypes.data, rl.ctypes.data, cap, o_r.ctypes.data, o_v.ctypes.data, o_c.ctypes.data, None)
            if k <= cap:
                return o_r[:k] + lo, o_v[:k], o_c[:k]
            cap = k + 16
    with ThreadPoolExecutor(n_threads) as ex:
        parts = list(ex.map(work, range(n_threads)))
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts])


def main():
    baseq = 10
    scale = float(os.environ.get("PHZ_PARITY_SCALE", "1.0"))
    plan = workloads.genome_plan(scale=scale)
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle")])
    mapper = Mapper(0)
    cores = max(1, pdist.effective_cpus())
    tmp = tempfile.mkdtemp(prefix="phz_full_parity_")
    t_all = time.perf_counter()
    try:
        vsets = {}; shards = {}; calls = {}
        n_rec = n_calls = 0
        workers = []
        for chrom, ln, n_snps, nr, seed in plan:
            v, sh, smp = workloads.make_shard(chrom, ln, n_snps, nr, seed, "cuda:0", keep_sample=nr)
            c = mapper.map(sh, v.pos, baseq)
            t0 = time.perf_counter()
            o_r, o_v, o_c = oracle_all_records(os.path.join(REPO, "oracle"), smp, v.pos.numpy(), baseq, cores)
            dt = time.perf_counter() - t0
            ok = c.n == len(o_r) and np.array_equal(c.read_idx.cpu().numpy(), o_r) and np.array_equal(c.var_idx.cpu().numpy(), o_v) and np.array_equal(c.code.cpu().numpy(), o_c)
            print("K_map %-6s %9d records %8d calls  identical to the C oracle: %s  (oracle %.2f s on %d threads)" % (chrom, len(smp), c.n, ok, dt, cores), flush=True)
            assert ok, chrom
            n_rec += len(smp); n_calls += c.n
            vsets[chrom] = v; shards[chrom] = sh; calls[chrom] = c
            d = os.path.join(tmp, chrom); os.makedirs(d)
            open(os.path.join(d, "calls.tsv"), "w").write(call_text(v, sh, c))
            del smp
        print("K_map: all %d records of %d shards, %d calls identical" % (n_rec, len(plan), n_calls), flush=True)
        # ---- product: the whole genome in one pass (device row stage)
        vs = pvcf.load_variants("\n".join(synth.vcf_lines([vsets[p[0]] for p in plan])))
        eng = Engine(vs, ["bench"], Config(baseq=baseq, want_vcf=False), mapper=mapper)
        for p in plan:
            eng.add_mapped(0, p[0], shards[p[0]], calls[p[0]], int(shards[p[0]].qid.max()) + 1)
        eng.close_bam(0)
        t0 = time.perf_counter()
        got = eng.finish()
        eng.resolve_cutoffs()          # (the percentile was taken on the device: bring the value in)
        cutoff = float(next(sh for sh in eng.shards[plan[0][0]] if sh is not None).cutoff)
        print("product: stages T1-O2 of the whole genome in %.3f s (first pass), rows on the %s, %d phased variants" % (time.perf_counter() - t0, eng.rows_path, eng.phased), flush=True)
        # ---- phasing oracle: one process per chromosome, `cores` at a time, largest first; every process gets the two scalars the reference computes over
        # ALL chromosomes (AS cutoff :545-553, noise level :610-632) from the product's run -- they are pinned by the log lines of the fixtures
        t0 = time.perf_counter()
        order = sorted(plan, key=lambda p: -p[3])
        running = []; results = {}
        def reap(block):
            for item in list(running):
                ch, pr = item
                if block or pr.poll() is not None:
                    out = pr.communicate()[0].split()
                    assert pr.returncode == 0, ch
                    results[ch] = (int(out[0]), float(out[1]), out[2]); running.remove(item)
                    if block:
                        return
        for p in order:
            while len(running) >= cores:
                reap(False); time.sleep(0.2)
            d = os.path.join(tmp, p[0])
            running.append((p[0], subprocess.Popen([sys.executable, os.path.join(REPO, "tools", "oracle_chrom_worker.py"), os.path.join(d, "calls.tsv"), str(baseq), d, repr(cutoff), float(eng.noise).hex()],
                                                   stdout=subprocess.PIPE, text=True)))
        while running:
            reap(True)
        t_or = time.perf_counter() - t0
        print("phasing oracle: %d processes (<= %d at a time), %.1f s wall, %.1f CPU-seconds, %d phased variants" %
              (len(plan), cores, t_or, sum(r[1] for r in results.values()), sum(r[0] for r in results.values())), flush=True)
        assert eng.phased == sum(r[0] for r in results.values())
        bad = 0
        for name in OUTPUTS:
            mine = canonical(name, got[name]).split("\n")
            head = mine[0]; rows = mine[1:-1]
            want = []
            for p in plan:
                lines = open(os.path.join(tmp, p[0], name + ".txt")).read().split("\n")
                assert lines[0] == head
                want += lines[1:-1]
            if name in ("allelic_counts", "allele_config"):
                same = sorted(rows) == sorted(want) and len(rows) == len(want)          # byte-stable files: the row multiset (the global order across chromosomes is checked by the fixtures)
            else:
                same = rows == sorted(want)
            print("%-20s %9d rows  whole genome identical to the per-chromosome oracle runs (canonical form): %s   sha256 %s" %
                  (name, len(rows), same, hashlib.sha256("\n".join(rows).encode()).hexdigest()[:16]), flush=True)
            bad += 0 if same else 1
        print("VERDICT: %s (%.0f s)" % ("IDENTICAL" if bad == 0 else "%d FILES DIFFER" % bad, time.perf_counter() - t_all))
        return 1 if bad else 0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
