"""Multi-GPU layer: chromosomes are independent shards; one process per GPU.

Only three things cross ranks, all tiny and latency-bound (single-shot collectives, no ring tuning):
  1. the AS histogram of each BAM (phaser.py:545-553 takes the quantile over ALL chromosomes)  -> all_reduce(SUM)
  2. the two noise counters (phaser.py:610-632 is global over variants)                        -> all_reduce(SUM)
  3. per-chromosome output tables (counts, byte ranges), gathered to rank 0 which assembles the files in the reference's
     global order (engine.merge_fragments); the row TEXT itself (GBs at whole-genome scale) and the block arrays of write_vcf
     never enter a collective: ranks spool them to files and rank 0 splices byte ranges     -> all_gather of an int64 table (KBs)
Backend "nccl" (= RCCL over xGMI) on GPUs; the same code runs on "gloo" for the CPU tests.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def collectives_live() -> bool:
    """True when results have to travel through the process group: more than one rank -- or ONE rank with PHZ_DIST_FORCE_COLLECTIVES=1, which
    runs the whole multi-rank path (the all-reduces, the all-gathers of the fragment tables, the broadcasts, the spool files, the barrier) on a
    single process.  That is how the backend "nccl" (= RCCL) branch of every function below is executed on a one-GPU box."""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("PHZ_DIST_FORCE_COLLECTIVES") == "1"


def assign_chromosomes(weights: Dict[str, float], world_size: int) -> Dict[str, int]:
    """Longest-processing-time assignment of chromosomes to ranks by record count (deterministic)."""
    load = [0.0] * world_size
    owner: Dict[str, int] = {}
    for chrom, w in sorted(weights.items(), key=lambda kv: (-kv[1], kv[0])):
        r = min(range(world_size), key=lambda i: (load[i], i))
        owner[chrom] = r
        load[r] += w
    return owner


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """In-place SUM over ranks; tensors must live where the backend wants them (cuda for nccl, cpu for gloo)."""
    if collectives_live():
        backend = dist.get_backend()
        if backend == "nccl" and t.device.type != "cuda":
            tmp = t.cuda()
            dist.all_reduce(tmp)
            t.copy_(tmp.cpu())
        elif backend == "gloo" and t.device.type != "cpu":
            tmp = t.cpu()
            dist.all_reduce(tmp)
            t.copy_(tmp.to(t.device))
        else:
            dist.all_reduce(t)
    return t


def allreduce_counts(match: int, mism: int):
    if not collectives_live():
        return match, mism
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([match, mism], dtype=torch.int64, device=dev)
    dist.all_reduce(t)
    return int(t[0]), int(t[1])


TEXT_FIELDS = ("conn", "hap", "ase", "cfg", "allelic", "single_ase", "single_hap")


class FileSpan:
    """A byte range of a spool file: what a rank hands to rank 0 instead of its row text (sliceable like bytes, so that
    merge_fragments can cut the per-BAM segments; copied into the output file without passing through Python objects)."""
    __slots__ = ("path", "off", "n")

    def __init__(self, path: str, off: int, n: int):
        self.path = path; self.off = off; self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, sl):
        a, b, _ = sl.indices(self.n)
        return FileSpan(self.path, self.off + a, max(0, b - a))

    def read(self) -> bytes:
        if self.n == 0:
            return b""
        with open(self.path, "rb") as f:
            f.seek(self.off)
            return f.read(self.n)


def as_bytes(x) -> bytes:
    return x.read() if isinstance(x, FileSpan) else bytes(x)


def write_chunks(f, chunks):
    """Write buffers / FileSpans to the binary file object f in order; spans go kernel-side (sendfile) when possible."""
    import os
    for c in chunks:
        if isinstance(c, FileSpan):
            if c.n == 0:
                continue
            f.flush()
            with open(c.path, "rb") as src:
                off = c.off; left = c.n
                try:
                    while left > 0:
                        k = os.sendfile(f.fileno(), src.fileno(), off, left)
                        if k == 0:
                            break
                        off += k; left -= k
                except OSError:
                    pass
                if left > 0:            # sendfile unavailable on this filesystem pair: plain copy of the rest
                    src.seek(off)
                    while left > 0:
                        buf = src.read(min(left, 1 << 24))
                        if not buf:
                            raise IOError("spool file %s is shorter than recorded" % c.path)
                        f.write(buf); left -= len(buf)
                    f.flush()
                else:
                    f.seek(os.lseek(f.fileno(), 0, os.SEEK_CUR))      # where sendfile left the descriptor (not the end: the file may be an older, longer one being overwritten)
        else:
            f.write(c)


VCF_FIELDS = (("size", "int32"), ("var", "int32"), ("hap", "uint8"), ("cor", "int8"), ("stat", "float64"), ("stat_int", "uint8"), ("maxmaf", "int32"))
COUNT_FIELDS = ("lines", "dropped", "phased", "allelic_rows", "n_blocks", "first_bam1")


def _coll_device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def _all_gather_i64(vec: List[int]) -> List[List[int]]:
    """Variable-length int64 vectors of all ranks, through two fixed-layout tensor collectives (lengths, then padded payloads): the same
    calls on nccl (= RCCL over xGMI) and gloo, no pickling."""
    r, w = world()
    dev = _coll_device()
    n = torch.tensor([len(vec)], dtype=torch.int64, device=dev)
    lens = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(w)]
    dist.all_gather(lens, n)
    lens = [int(x.item()) for x in lens]
    m = max(1, max(lens))
    mine = torch.zeros(m, dtype=torch.int64, device=dev)
    if vec:
        mine[:len(vec)] = torch.tensor(vec, dtype=torch.int64, device=dev)
    parts = [torch.zeros(m, dtype=torch.int64, device=dev) for _ in range(w)]
    dist.all_gather(parts, mine)
    return [p[:k].cpu().tolist() for p, k in zip(parts, lens)]


def gather_fragments(local: Dict[str, dict], spool_dir: Optional[str] = None, chrom_order: Optional[List[str]] = None) -> Optional[Dict[str, dict]]:
    """-> on rank 0 the union of all ranks' {chrom: fragment}; None elsewhere (the "final gather" of SURVEY.md 8(e)).
    What crosses the collective is a fixed-layout int64 table per rank (counts, byte ranges): every rank writes its row text -- ~1 GB at
    whole-genome scale -- and the per-block arrays of write_vcf to ONE spool file in `spool_dir` (a directory all ranks of the node share;
    default the system temp directory) and rank 0 receives (offset, length) ranges, which it splices into the output files in the
    reference's global order.  A rank whose spool file rank 0 cannot see (no shared filesystem) sends the bytes through the process group
    instead.  chrom_order: all chromosomes in VCF order (identical on every rank); chromosomes travel as indices into it."""
    r, w = world()
    if not collectives_live():
        return dict(local)
    import os
    import tempfile
    import numpy as np
    dev = _coll_device()
    tok = torch.zeros(2, dtype=torch.int64, device=dev)
    if r == 0:
        tok = torch.tensor([int.from_bytes(os.urandom(7), "little"), os.getpid()], dtype=torch.int64, device=dev)
    dist.broadcast(tok, src=0)
    token = "%014x_%d" % (int(tok[0]), int(tok[1]))
    d = spool_dir or os.environ.get("PHZ_SPOOL_DIR") or tempfile.gettempdir()
    path_of = lambda rank: os.path.join(d, "phz_spool_%s_rank%d.bin" % (token, rank))
    if chrom_order is None:
        raise ValueError("gather_fragments needs chrom_order (all chromosomes, same order on every rank) with more than one rank")
    order = list(chrom_order)
    index = {c: i for i, c in enumerate(order)}
    # ---- spool file + table: [n_chroms, then per chromosome: index, counts..., per text field: n_spans, (bam, off, len)..., has_vcf, per vcf field (off, count)]
    table: List[int] = [len(local)]
    path = path_of(r)
    SPOOL_FILES.append(path)
    with open(path, "wb") as f:
        off = 0
        for c, frag in local.items():
            table.append(index[c])
            table += [int(frag.get(k, 0)) for k in COUNT_FIELDS]
            for k in TEXT_FIELDS:
                keys = frag.get(k + "_bam")
                spans = []                     # (bam key or -1, offset, length); adjacent buffers with the same key become one span
                for i, b in enumerate(frag[k]):
                    n = len(b)
                    if n == 0:
                        continue
                    f.write(b)
                    kb = int(keys[i]) if keys is not None else -1
                    if spans and spans[-1][0] == kb:
                        spans[-1][2] += n
                    else:
                        spans.append([kb, off, n])
                    off += n
                table.append(len(spans))
                for sp in spans:
                    table += sp
            v = frag.get("vcf")
            table.append(1 if v is not None else 0)
            if v is not None:
                for name, dt in VCF_FIELDS:
                    a = np.ascontiguousarray(v[name], dtype=dt)
                    f.write(a.tobytes())
                    table += [off, int(a.size)]
                    off += a.nbytes
        total = off
    table.append(total)
    tables = _all_gather_i64(table)
    # ---- can rank 0 see every spool file?  (all ranks learn the answer: the senders must take part in the transfer)
    seen = torch.ones(w, dtype=torch.int64, device=dev)
    if r == 0:
        for k in range(1, w):
            try:
                ok = os.path.getsize(path_of(k)) == tables[k][-1]
            except OSError:
                ok = False
            seen[k] = 1 if ok else 0
    dist.broadcast(seen, src=0)
    seen = seen.cpu().tolist()
    for k in range(1, w):
        if seen[k]:
            continue
        nbytes = tables[k][-1]
        if r == k:
            buf = torch.from_numpy(np.fromfile(path, dtype=np.uint8)) if nbytes else torch.zeros(0, dtype=torch.uint8)
            for lo in range(0, nbytes, 1 << 28):
                dist.send(buf[lo:lo + (1 << 28)].to(dev), dst=0)
        elif r == 0:
            lp = os.path.join(d, "phz_spool_%s_rank%d.recv.bin" % (token, k))
            SPOOL_FILES.append(lp)
            with open(lp, "wb") as f:
                for lo in range(0, nbytes, 1 << 28):
                    t = torch.empty(min(1 << 28, nbytes - lo), dtype=torch.uint8, device=dev)
                    dist.recv(t, src=k)
                    f.write(t.cpu().numpy().tobytes())
            path_of = (lambda rank, _p=path_of, _k=k, _lp=lp: _lp if rank == _k else _p(rank))
    if r != 0:
        return None
    merged: Dict[str, dict] = {}
    for k in range(w):
        t = tables[k]; p = 0
        nchr = t[p]; p += 1
        src = path_of(k)
        for _ in range(nchr):
            c = order[t[p]]; p += 1
            g = {"chrom": c}
            for name in COUNT_FIELDS:
                g[name] = t[p]; p += 1
            for name in TEXT_FIELDS:
                ns = t[p]; p += 1
                spans = []; skeys = []
                for _s in range(ns):
                    kb, o, n = t[p], t[p + 1], t[p + 2]; p += 3
                    spans.append(FileSpan(src, o, n)); skeys.append(kb)
                g[name] = spans
                if name in ("allelic", "single_ase", "single_hap"):
                    g[name + "_bam"] = skeys
            has_vcf = t[p]; p += 1
            g["vcf"] = None
            if has_vcf:
                v = {}
                with open(src, "rb") as f:
                    for name, dt in VCF_FIELDS:
                        o, cnt = t[p], t[p + 1]; p += 2
                        f.seek(o)
                        v[name] = np.frombuffer(f.read(cnt * np.dtype(dt).itemsize), dtype=dt)
                g["vcf"] = v
            merged[c] = g
    return merged


SPOOL_FILES: List[str] = []


def cleanup_spool():
    """Remove this rank's spool files (call after rank 0 has written the outputs; a barrier separates the two)."""
    import os
    if collectives_live():
        try:
            dist.barrier()
        except Exception:              # a rank that failed must still remove its files
            pass
    while SPOOL_FILES:
        try:
            os.remove(SPOOL_FILES.pop())
        except OSError:
            pass


def write_files(paths_and_chunks, threads: int = 8):
    """Write several (path, chunks) files, one worker per file (writes to ONE file serialise on its inode lock, so splitting a file
    across workers does not help; the page-cache copy of a write() is a single thread at ~8 GB/s)."""
    from concurrent.futures import ThreadPoolExecutor
    items = list(paths_and_chunks)

    def put(item):
        # No O_TRUNC: a file left by an earlier run under the same prefix is overwritten IN PLACE and cut to the new length at the end.  Truncating
        # it first hands its page-cache pages back only to allocate as many again (0.09 s for the 770 MB of a genome's five files, measured).
        # A write that fails half way (ENOSPC, a short spool file) must not leave the new head followed by the old tail -- a file that looks
        # complete: it is cut to the bytes written before the error travels on.
        import os
        path, chunks = item
        with os.fdopen(os.open(path, os.O_WRONLY | os.O_CREAT, 0o666), "wb") as f:
            try:
                write_chunks(f, chunks)
            finally:
                try:
                    f.flush()
                except OSError:
                    pass
                f.truncate(os.lseek(f.fileno(), 0, os.SEEK_CUR))
    if len(items) <= 1 or threads <= 1:
        for it in items:
            put(it)
        return
    with ThreadPoolExecutor(max_workers=min(len(items), max(1, int(threads)))) as ex:
        list(ex.map(put, items))


def effective_cpus() -> int:
    """CPUs this process can really use: the affinity mask, cut down by the cgroup CPU quota when there is one (a container with
    256 visible cores and `cpu.max` = 16 CPUs gets throttled for most of every scheduling period if it runs 64 busy threads)."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return max(1, n)

