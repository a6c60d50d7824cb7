"""Multi-GPU layer: chromosomes are independent shards; one process per GPU.

Only three things cross ranks, all tiny and latency-bound (single-shot collectives, no ring tuning):
  1. the AS histogram of each BAM (phaser.py:545-553 takes the quantile over ALL chromosomes)  -> all_reduce(SUM)
  2. the two noise counters (phaser.py:610-632 is global over variants)                        -> all_reduce(SUM)
  3. per-chromosome output tables (counts, segment offsets, block arrays), gathered to rank 0 which assembles the files in
     the reference's global order (engine.merge_fragments); the row TEXT itself (GBs at whole-genome scale) never enters a
     collective: ranks spool it to files and rank 0 splices byte ranges                      -> gather_object of KBs
Backend "nccl" (= RCCL over xGMI) on GPUs; the same code runs on "gloo" for the CPU tests.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


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
    r, w = world()
    if w > 1:
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
    r, w = world()
    if w == 1:
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
                    f.seek(0, 2)
        else:
            f.write(c)


def gather_fragments(local: Dict[str, dict], spool_dir: Optional[str] = None) -> Optional[Dict[str, dict]]:
    """-> on rank 0 the union of all ranks' {chrom: fragment}; None elsewhere.  Only the small tables travel through the
    collective (counts, segment offsets, block arrays): every rank writes its row text to spool files on the node's filesystem
    (spool_dir, default the system temp directory) and rank 0 receives (path, offset, length) spans, which it splices into the
    output files in the reference's global order.  At whole-genome scale the text is ~1 GB (allele_config alone 680 MB) while the
    tables are KBs -- the "final gather" of SURVEY.md 8(e)."""
    r, w = world()
    if w == 1:
        return dict(local)
    import os
    import tempfile
    import uuid
    token = [uuid.uuid4().hex if r == 0 else None]
    dist.broadcast_object_list(token, src=0)
    d = spool_dir or os.environ.get("PHZ_SPOOL_DIR") or tempfile.gettempdir()
    path = os.path.join(d, "phz_spool_%s_rank%d.bin" % (token[0], r))
    small: Dict[str, dict] = {}
    with open(path, "wb") as f:
        off = 0
        for c, frag in local.items():
            g = dict(frag)
            for k in TEXT_FIELDS:
                # a field is a list of buffers; the BAM-keyed ones (k + "_bam") become one span per run of equal keys
                keys = frag.get(k + "_bam")
                spans = []; span_keys = []
                for i, b in enumerate(frag[k]):
                    n = len(b)
                    if n == 0:
                        continue
                    f.write(b)
                    kb = keys[i] if keys is not None else None
                    if spans and (keys is None or span_keys[-1] == kb):
                        spans[-1] = FileSpan(path, spans[-1].off, spans[-1].n + n)
                    else:
                        spans.append(FileSpan(path, off, n)); span_keys.append(kb)
                    off += n
                g[k] = spans
                if keys is not None:
                    g[k + "_bam"] = span_keys
            small[c] = g
    bucket: List[Optional[dict]] = [None] * w if r == 0 else None
    dist.gather_object((path, small), bucket, dst=0)
    SPOOL_FILES.append(path)
    if r != 0:
        return None
    merged: Dict[str, dict] = {}
    for p_, part in bucket:
        merged.update(part)
    return merged


SPOOL_FILES: List[str] = []


def cleanup_spool():
    """Remove this rank's spool files (call after rank 0 has written the outputs; a barrier separates the two)."""
    import os
    r, w = world()
    if w > 1:
        dist.barrier()
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
        path, chunks = item
        with open(path, "wb") as f:
            write_chunks(f, chunks)
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

