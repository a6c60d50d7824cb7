#!/usr/bin/env python3
"""Headline benchmark: het-SNP allele calls/s of the read-backed phasing hot path on MI355X.

python bench.py --gpus N --steps K --warmup W       (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one resident shard: BASELINE.json configs[1]
(chr1, 40k het SNPs, 50M records of 76 bp) already packed as structure-of-arrays in HBM.  With N GPUs every
rank owns its own shard of that shape (chromosomes / BAMs are independent shards):
weak scaling, no data-path collective; value = calls of all ranks / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0     # MI355X spec (MI355X_MICROARCH.md); measured copy peak is ~6290 GB/s
CALL_BYTES = 17           # read_idx 4 + var_idx 4 + code 1 + aux0 4 + aux1 4


def pmc_traffic():
    """HBM bytes per k_map launch from the committed PMC passes (profiles/<round>/pmc_kmap_*/{fetch,write}.csv, collected
    with tools/prof_pmc.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc runs).  Units are KiB; FETCH_SIZE is
    doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B).  None when no PMC summary is present."""
    import csv, glob
    dirs = sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "pmc_kmap_*")))
    if not dirs:
        return None
    vals = {}
    for name in ("fetch", "write"):
        f = os.path.join(dirs[-1], name + ".csv")
        if not os.path.exists(f):
            return None
        rows = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_map" in r["Kernel_Name"]]
        if not rows:
            return None
        vals[name] = sum(rows) / len(rows)
    return {"bytes_per_launch": 2 * vals["fetch"] * 1024 + vals["write"] * 1024, "fetch_raw_kib": vals["fetch"], "write_kib": vals["write"],
            "source": os.path.relpath(dirs[-1], REPO), "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE; same command as bench (configs[1] shard)"}


def cpu_baseline(sample, vpos, baseq):
    """Oracle (CPU restatement, kind 'port') timed on one host core over a bounded sample."""
    import subprocess
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from helpers import oracle_map_readbatch
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle")])
    from helpers import oracle_map_readbatch_threads
    t0 = time.perf_counter()
    o_r, o_v, o_c, _ = oracle_map_readbatch(os.path.join(REPO, "oracle"), sample, vpos, baseq, with_text=False)
    dt = time.perf_counter() - t0
    # all host cores (the reference's own fan-out is one process per chromosome, phaser.py:2077-2094)
    cores = max(1, min(64, os.cpu_count() or 1))
    t0 = time.perf_counter()
    m_all = oracle_map_readbatch_threads(os.path.join(REPO, "oracle"), sample, vpos, baseq, cores)
    dt_all = time.perf_counter() - t0
    assert m_all == len(o_r)
    return (o_r, o_v, o_c), dt, (cores, dt_all)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--records", type=int, default=50_000_000)
    ap.add_argument("--snps", type=int, default=40_000)
    ap.add_argument("--baseq", type=int, default=10)
    ap.add_argument("--cpu-sample", type=int, default=24_000_000)
    ap.add_argument("--no-phasing", action="store_true", help="skip the (untimed-for-value) phasing-stage measurement")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # one rank per GPU over RCCL ("nccl"); PHZ_BENCH_BACKEND=gloo lets several ranks share a GPU (used only to exercise the
    # multi-rank path on a 1-GPU box)
    backend = os.environ.get("PHZ_BENCH_BACKEND", "nccl")
    local = local % max(1, torch.cuda.device_count())
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    red_dev = dev if backend == "nccl" else "cpu"

    from phaser_amd import workloads
    from phaser_amd.mapper import Mapper
    from phaser_amd import _lib
    import ctypes as C

    t_gen = time.perf_counter()
    v, shard, sample = workloads.make_shard("chr1", workloads.CHR1_LEN, a.snps, a.records, 20240807 + 17 * rank, dev,
                                            keep_sample=a.cpu_sample if (rank == 0 and world == 1) else 0)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    mapper = Mapper(local)
    vpos = v.pos.to(dev)

    # first (untimed) pass sizes the output buffers; later passes reuse them through the raw ABI
    calls = mapper.map(shard, vpos, a.baseq)
    n_calls = calls.n
    cap = n_calls + 16
    bufs = [torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev),
            torch.empty(cap, dtype=torch.uint8, device=dev), torch.empty(cap, dtype=torch.int32, device=dev),
            torch.empty(cap, dtype=torch.int32, device=dev)]
    p = lambda t: C.c_void_p(t.data_ptr())
    r = _lib.phz_reads(shard.n, int(shard.cigar.numel()), int(shard.seq2.numel()), p(shard.pos), p(shard.cigar_off),
                       p(shard.cigar), p(shard.seq_off), p(shard.seq2), p(shard.qual))
    vv = _lib.phz_variants(int(vpos.numel()), p(vpos), None)
    cc = _lib.phz_calls(cap, *[p(b) for b in bufs])
    n_out = C.c_int64(0)

    def step():
        mapper.ctx.check(mapper.ctx.lib.phz_map_reads(mapper.ctx.h, C.byref(r), C.byref(vv), a.baseq, C.byref(cc),
                                                      C.byref(n_out), _lib.PHZ_DEVICE))
        assert n_out.value == n_calls

    for _ in range(a.warmup):
        step()
    mapper.ctx.reset_timing()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    _, k_total_ms, k_n = mapper.ctx.timing(_lib.PHZ_T_MAP)

    # idempotence: the timed passes reproduce the first pass bit for bit
    same = all(bool(torch.equal(b[:n_calls], c)) for b, c in zip(bufs, (calls.read_idx, calls.var_idx, calls.code, calls.aux0, calls.aux1)))
    assert same, "repeated passes differ"
    # sortedness: mapper order == (record, variant) lexicographic
    key = bufs[0][:n_calls].to(torch.int64) * (int(vpos.numel()) + 1) + bufs[1][:n_calls].to(torch.int64)
    assert bool((key[1:] > key[:-1]).all()), "call list not in mapper order"

    tot_calls = torch.tensor([float(n_calls)], device=red_dev, dtype=torch.float64); tmax = torch.tensor([dt], device=red_dev, dtype=torch.float64)
    tot_recs = torch.tensor([float(shard.n)], device=red_dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tot_calls); dist.all_reduce(tot_recs); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # ---- second rate of the metric: phased variants/s over stages T1-O2 (AS cutoff, K_tally, pair test, components,
    #      native block phasing + row writer) on the same resident shard; two passes, outside the timed K_map region
    phasing = None
    if not a.no_phasing:
        from phaser_amd import synth, vcf as pvcf
        from phaser_amd.engine import Engine, Config
        vs = pvcf.load_variants("\n".join(synth.vcf_lines([v])))
        host_threads = max(1, min(32, (os.cpu_count() or 1) // max(1, world)))
        runs = []
        for rep in range(2):            # two passes over the same shard (fresh Engine each time); the faster one is reported, both are listed
            eng = Engine(vs, ["bench"], Config(baseq=a.baseq, host_threads=host_threads, want_vcf=False), mapper=mapper)
            eng.add_shard(0, "chr1", shard, int(shard.qid.max()) + 1)
            torch.cuda.synchronize()
            tp0 = time.perf_counter()
            eng.close_bam(0)
            tp1 = time.perf_counter()
            counts = eng.tally_all()
            tp2 = time.perf_counter()
            noise = eng.noise_from_counts(*counts)
            frag = eng.chrom_fragment("chr1", noise, 0)
            tp3 = time.perf_counter()
            runs.append({"value": frag["phased"] / (tp3 - tp0), "unit": "phased variants/s", "phased_variants": frag["phased"],
                         "call_lines_kept": frag["lines"], "blocks": frag["n_blocks"],
                         "seconds": {"as_cutoff": tp1 - tp0, "k_tally_incl_copies": tp2 - tp1, "host_assembly": tp3 - tp2,
                                     "ordering_pairtest_components": eng.stats.get("prepare_s"), "block_phasing_and_rows": eng.stats.get("rows_s")},
                         "host_threads": host_threads, "k_tally_kernel_ms": eng.ctx.timing(_lib.PHZ_T_TALLY)[0]})
            del eng, frag
        phasing = dict(max(runs, key=lambda r: r["value"]))
        phasing["passes"] = [round(r["value"]) for r in runs]

    if rank == 0:
        alg_bytes = shard.nbytes_map_inputs() + int(vpos.numel()) * 4 + CALL_BYTES * n_calls
        k_avg_s = k_total_ms / max(1, k_n) / 1e3
        achieved = alg_bytes / k_avg_s / 1e9
        out = {
            "metric": "het-SNP allele calls/sec (value) + phased variants/sec (phasing.value), RNA-seq shape, per-GPU shards",
            "value": float(tot_calls.item()) * a.steps / dt, "unit": "allele calls/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int32", "data": "synthetic",
            "config": {"workload": "configs[1]: chr1 full, %d het SNPs, %d records x 76bp, one shard per GPU" % (a.snps, a.records),
                       "records_per_gpu": shard.n, "het_snps": int(vpos.numel()), "calls_per_gpu": n_calls,
                       "records_per_s": float(tot_recs.item()) * a.steps / dt, "gen_seconds": round(t_gen, 1)},
            "roofline": {"bound": "hbm", "kernel": "k_map", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(),
                         "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_record": alg_bytes / shard.n,
                         "kernel_ms_avg": k_avg_s * 1e3, "launches": k_n},
        }
        if phasing is not None:
            out["phasing"] = phasing
        if world == 1 and sample is not None:
            (o_r, o_v, o_c), cpu_dt, (cpu_cores, cpu_dt_all) = cpu_baseline(sample, v.pos.numpy(), a.baseq)
            m = len(o_r)
            # at-scale parity: the GPU call list restricted to the sampled records equals the oracle's
            assert np.array_equal(bufs[0][:m].cpu().numpy(), o_r) and np.array_equal(bufs[1][:m].cpu().numpy(), o_v) \
                and np.array_equal(bufs[2][:m].cpu().numpy(), o_c), "GPU != oracle on the sampled prefix"
            assert n_calls == m or int(bufs[0][m]) >= len(sample)
            out["cpu_baseline"] = {"value": m / cpu_dt, "unit": "allele calls/s", "cores": 1, "kind": "port",
                                   "sample": "first %d records of the same shard through oracle/rvm_oracle.c (array front end, "
                                             "no SAM text parsing), %.1f s; %.0f records/s" % (len(sample), cpu_dt, len(sample) / cpu_dt),
                                   "parity_on_sample": "bit-exact (%d calls)" % m,
                                   "all_cores": {"value": m / cpu_dt_all, "unit": "allele calls/s", "cores": cpu_cores,
                                                 "records_per_s": len(sample) / cpu_dt_all},
                                   "reference_note": "the reference's own Cython mapper ran 1.03e5 records/s/core in the build container "
                                                     "(BASELINE.md); this port is the faster, parity-locked stand-in on the GPU box"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
