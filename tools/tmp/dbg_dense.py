import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "./tests")
import numpy as np, torch
from phaser_amd import soa, synth
from phaser_amd.mapper import Mapper
from helpers import oracle_map_readbatch
v, gs, ge, w = synth.make_variants("chr1", 1, 2_000_000, 7800, 11, n_genes=6)
rb = synth.make_reads(v, gs, ge, w, 60000, 12, n_rate=0.002)
rb = rb.select(synth.samtools_keep(rb, 255))
o_r, o_v, o_c, o_t = oracle_map_readbatch("./oracle", rb, v.pos.numpy(), 10)
m = Mapper(0)
calls = m.map(soa.pack_readbatch(rb).to("cuda"), v.pos, 10).cpu()
n = len(rb)
co = np.bincount(o_r, minlength=n); cp = np.bincount(calls.read_idx.numpy(), minlength=n)
bad = np.nonzero(co != cp)[0]
print("records", n, "oracle calls", len(o_r), "product", calls.n, "records differing", len(bad))
print("oracle count of differing records: pct", np.percentile(co[bad], [0, 5, 50, 95, 100]))
print("product count of differing records: pct", np.percentile(cp[bad], [0, 5, 50, 95, 100]))
print("max oracle count over all", co.max(), "records with >32:", (co > 32).sum(), ">8:", (co > 8).sum())
ok = np.nonzero(co == cp)[0]
print("oracle count of matching records: pct", np.percentile(co[ok], [0, 50, 95, 100]))
nops = (rb.cigar_off[1:] - rb.cigar_off[:-1]).numpy()
print("n_ops of differing:", np.bincount(nops[bad])[:8], " of all:", np.bincount(nops)[:8])
print("first differing records:", bad[:10], "tiles", (bad[:10] // 256))
tiles = np.unique(bad // 256); print("tiles with differences", len(tiles), "of", (n + 255) // 256)
b0 = bad[0]
print("rec", b0, "pos", int(rb.pos[b0]), "oracle vars", o_v[o_r == b0][:40], "product vars", calls.var_idx.numpy()[calls.read_idx.numpy() == b0][:40])
pos = rb.pos.numpy().astype(np.int64); vp = v.pos.numpy().astype(np.int64)
T = 256
nt = (n + T - 1) // T
tot_o = np.add.reduceat(co, np.arange(0, n, T)); tot_p = np.add.reduceat(cp, np.arange(0, n, T))
first = pos[::T]; last = pos[np.minimum(np.arange(T - 1, n + T - 1, T), n - 1)]
wl = np.searchsorted(vp, last + 65536) - np.searchsorted(vp, first) + 8
cig_words = np.add.reduceat(nops, np.arange(0, n, T))
for t in range(nt):
    flag = "BAD" if tot_o[t] != tot_p[t] else "ok "
    if flag == "BAD" or t < 12:
        print(flag, "tile", t, "oracle", tot_o[t], "product", tot_p[t], "wlen", wl[t], "cigar words", cig_words[t], "max per record", co[t*T:(t+1)*T].max())
print("max tile total ok tiles", tot_o[tot_o == tot_p].max(), "min bad", tot_o[tot_o != tot_p].min())
