import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from phaser_amd import workloads
from phaser_amd.mapper import Mapper
v, shard, _ = workloads.make_shard("chr1", workloads.CHR1_LEN, 40_000, 50_000_000, 20240807, "cuda:0")
m = Mapper(0); vpos = v.pos.to("cuda:0")
calls = m.map(shard, vpos, 10); cap = 2 * calls.n
for dbg in ["0", "1", "16", "17", "8", "2", "0"]:
    os.environ["PHZ_MAP_DBG"] = dbg
    c2 = m.map(shard, vpos, 10, cap=cap)
    m.ctx.reset_timing()
    for _ in range(6):
        m.map(shard, vpos, 10, cap=cap)
    print("dbg=%s k_map avg %.3f ms calls %d" % (dbg, m.ctx.timing()[1] / 6, c2.n), flush=True)
