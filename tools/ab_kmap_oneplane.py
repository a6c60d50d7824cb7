#!/usr/bin/env python3
"""GPU-box A/B: K_map reading base + quality of a call from ONE byte (phz_reads.bq: base << 6 | min(phred, 62), 63 = escape) against the two planes
(2-bit seq2 + 1-byte qual) it reads in production -- the layout change the round-4 and round-5 verdicts asked to be measured, not argued
(read_variant_map.py:165-258 is what both compute).  Both sides run the PROFILING instantiation of the kernel (PHZ_MAP_DBG != 0: the production
instantiation has no switch to flip), on the configs[2] shards of bench.py, alternating, with HIP-event kernel times; the call lists of both sides
must equal the production kernel's.       usage: tools/ab_kmap_oneplane.py [rounds=6] [steps=30]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
from phaser_amd import workloads, _lib
from phaser_amd.mapper import Mapper
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = "cuda:0"
plan = workloads.genome_plan(80_000_000, 1_500_000)
vsets = {}; shards = {}
for chrom, ln, n_snps, n_rec, seed in plan:
    v, shard, _ = workloads.make_shard(chrom, ln, n_snps, n_rec, seed, dev)
    vsets[chrom] = v; shards[chrom] = shard
chroms = [p[0] for p in plan]
sh_list = [shards[c] for c in chroms]; vp_list = [vsets[c].pos for c in chroms]
from phaser_amd import soa
os.environ["PHZ_MAP_ONE_PLANE"] = "1"          # (the plane is off by default)
t0 = time.perf_counter()
esc = 0; nb = 0
for sh in sh_list:          # the one-byte plane (soa.bq_plane): 2-bit base << 6 | min(phred, 62); escape where the quality byte carries the non-ACGT flag
    bq = soa.bq_plane(sh)
    esc += int(((bq & 63) == 63).sum()); nb += bq.numel()
torch.cuda.synchronize()
print("one-byte plane built in %.2f s: %d bases, %d escapes (%.4f %%)" % (time.perf_counter() - t0, nb, esc, 100.0 * esc / nb), flush=True)
mapper = Mapper(0)
os.environ.pop("PHZ_MAP_DBG", None)
os.environ["PHZ_MAP_TWO_PLANES"] = "1"          # the reference call lists come from the two-plane production kernel of rounds 1-5
ref = mapper.map_batch(sh_list, vp_list, 10, aux=False)
os.environ.pop("PHZ_MAP_TWO_PLANES")
n_calls = [c.n for c in ref]
call, bufs, N = mapper.prepare_batch(sh_list, vp_list, 10, [n + 16 for n in n_calls], aux=False)


def run(dbg, label):
    if dbg is None:
        os.environ.pop("PHZ_MAP_DBG", None)
    else:
        os.environ["PHZ_MAP_DBG"] = str(dbg)
    for _ in range(3):
        mapper.ctx.check(call())
    mapper.ctx.reset_timing()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps):
        mapper.ctx.check(call())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps * 1e3
    _, tot, n = mapper.ctx.timing(_lib.PHZ_T_MAP)
    ok = all(bool(torch.equal(bufs[i][k][:n_calls[i]], (ref[i].read_idx, ref[i].var_idx, ref[i].code)[k])) for i in range(len(chroms)) for k in range(3)) and \
        [int(N[i]) for i in range(len(chroms))] == n_calls
    print("%-46s k_map %.4f ms   step %.4f ms   calls identical to production: %s" % (label, tot / n, dt, ok), flush=True)
    assert ok
    return tot / n


os.environ["PHZ_MAP_TWO_PLANES"] = "1"
run(None, "PRODUCTION instantiation, two planes")
os.environ.pop("PHZ_MAP_TWO_PLANES")
run(None, "PRODUCTION instantiation, ONE byte per base")
pa = []; pb = []
for r in range(rounds):
    os.environ["PHZ_MAP_TWO_PLANES"] = "1"
    pa.append(run(None, "PRODUCTION instantiation, two planes"))
    os.environ.pop("PHZ_MAP_TWO_PLANES")
    pb.append(run(None, "PRODUCTION instantiation, ONE byte per base"))
ma = sorted(pa)[len(pa) // 2]; mb = sorted(pb)[len(pb) // 2]
print("PRODUCTION, median of %d alternating rounds: two planes %.4f ms, one plane %.4f ms -> %+.2f %%" % (rounds, ma, mb, 100.0 * (mb - ma) / ma))
a = []; b = []
for r in range(rounds):
    a.append(run(16384, "profiling instantiation, two planes"))
    b.append(run(16384 | 4096, "profiling instantiation, ONE byte per base"))
ma = sorted(a)[len(a) // 2]; mb = sorted(b)[len(b) // 2]
print("median of %d alternating rounds: two planes %.4f ms, one plane %.4f ms -> %+.2f %%" % (rounds, ma, mb, 100.0 * (mb - ma) / ma))

