"""Shared helpers for the tests (text renderings that mirror the reference's file formats)."""
from phaser_amd import synth


def variant_table_text(v, maf="None"):
    """The mapper's variant table as generate_mapping_table writes it (phaser.py:1402-1404)."""
    rows = []
    pos = v.pos.tolist(); ref = v.ref.tolist(); alt = v.alt.tolist()
    for i in range(len(v)):
        r, a = synth.BASES[ref[i]], synth.BASES[alt[i]]
        uid = "%s_%d_%s_%s" % (v.chrom, pos[i], r, a)
        rows.append("\t".join([v.chrom, str(pos[i]), uid, v.rsid[i], r + "," + a, "1", v.gt[i], maf]))
    return "\n".join(rows) + "\n"


# ------------------------------------------------------------------ oracle (CPU restatement) access
import ctypes
import os

import numpy as np


def oracle_lib(oracle_dir):
    lib = ctypes.CDLL(os.path.join(oracle_dir, "librvm_oracle.so"))
    lib.rvm_oracle_map_soa.restype = ctypes.c_long
    lib.rvm_oracle_map_soa.argtypes = [ctypes.c_long] + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_long,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long] + [ctypes.c_void_p] * 4
    return lib


def oracle_map_readbatch(oracle_dir, rb, vpos, baseq, with_text=True):
    """Run the C restatement on a synth.ReadBatch (CPU tensors). Returns (read_idx, var_idx, code, text list)."""
    lib = oracle_lib(oracle_dir)
    n = len(rb)
    pos = np.ascontiguousarray(rb.pos.numpy().astype(np.int32))
    coff = np.ascontiguousarray(rb.cigar_off.numpy().astype(np.int64))
    cig = np.ascontiguousarray(rb.cigar.numpy().astype(np.uint32))
    seq = np.ascontiguousarray(rb.seq.numpy()); qual = np.ascontiguousarray(rb.qual.numpy())
    vp = np.ascontiguousarray(np.asarray(vpos, dtype=np.int32)); rl = np.ones(len(vp), dtype=np.uint8)
    cap = n * 2 + 1024
    while True:
        o_r = np.zeros(cap, np.int32); o_v = np.zeros(cap, np.int32); o_c = np.zeros(cap, np.uint8)
        o_s = np.zeros(cap * 32, np.uint8) if with_text else None
        m = lib.rvm_oracle_map_soa(n, pos.ctypes.data, coff.ctypes.data, cig.ctypes.data, seq.ctypes.data, qual.ctypes.data,
                                   rb.L, baseq, len(vp), vp.ctypes.data, rl.ctypes.data, cap, o_r.ctypes.data,
                                   o_v.ctypes.data, o_c.ctypes.data, o_s.ctypes.data if with_text else None)
        if m <= cap:
            break
        cap = m + 16
    text = None
    if with_text:
        text = [bytes(o_s[i * 32:(i + 1) * 32]).split(b"\0")[0].decode() for i in range(m)]
    return o_r[:m], o_v[:m], o_c[:m], text


# ------------------------------------------------------------------ canonical forms
def _relabel(field):
    """aReads / bReads: indices into list(set(reads)) are hash-order labels; relabel by first appearance."""
    if field == "":
        return field
    m = {}
    out = []
    for part in field.split(";"):
        if part == "":
            out.append("")
            continue
        out.append(",".join(str(m.setdefault(x, len(m))) for x in part.split(",")))
    return ";".join(out)


def canonical(name, text):
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines = lines[:-1]
    head, rows = lines[0], lines[1:]
    if name in ("allelic_counts", "allele_config"):
        return text                       # byte-stable files
    if name == "haplotypic_counts":
        fixed = []
        for r in rows:
            f = r.split("\t")
            if len(f) == 20:       # --output_read_ids 1: two QNAME lists (set order in the reference) sit before max_haplo_maf
                f[14] = ",".join(sorted(f[14].split(","))); f[15] = ",".join(sorted(f[15].split(",")))
            f[-2] = _relabel(f[-2]); f[-1] = _relabel(f[-1])
            f[5] = ",".join(sorted(f[5].split(","))) if f[5] else f[5]
            fixed.append("\t".join(f))
        rows = fixed
    return "\n".join([head] + sorted(rows)) + "\n"


OUTPUTS = ["allelic_counts", "variant_connections", "haplotypes", "haplotypic_counts", "allele_config"]


def option_case_kwargs(name, case, blacklist):
    """tests/golden/pipe_opts/cases.json entry -> (vcf loader kwargs, engine Config kwargs, mapper baseq, isize)."""
    load = {}; cfg = {}; baseq = 10; isize = 0.0
    for k, v in case.items():
        if k == "gw_phase_method":
            load[k] = v; cfg[k] = v
        elif k == "id_separator":
            load[k] = v; cfg[k] = v
        elif k == "haplo_count_bam_exclude":
            cfg[k] = [int(x) - 1 for x in v.split(",")]
        elif k == "baseq":
            baseq = v; cfg[k] = v
        elif k == "isize":
            isize = float(v)
        else:
            cfg[k] = v
    if name == "blacklist":
        cfg["haplo_blacklist"] = frozenset(blacklist)
    return load, cfg, baseq, isize


def oracle_map_readbatch_threads(oracle_dir, rb, vpos, baseq, n_threads):
    """Same work split into n_threads record ranges on Python threads (ctypes releases the GIL while the C code runs).
    Returns the total number of calls; used by bench.py's all-cores CPU figure."""
    from concurrent.futures import ThreadPoolExecutor
    lib = oracle_lib(oracle_dir)
    n = len(rb)
    pos = np.ascontiguousarray(rb.pos.numpy().astype(np.int32))
    coff = np.ascontiguousarray(rb.cigar_off.numpy().astype(np.int64))
    cig = np.ascontiguousarray(rb.cigar.numpy().astype(np.uint32))
    seq = np.ascontiguousarray(rb.seq.numpy()); qual = np.ascontiguousarray(rb.qual.numpy())
    vp = np.ascontiguousarray(np.asarray(vpos, dtype=np.int32)); rl = np.ones(len(vp), dtype=np.uint8)
    L = rb.L
    bounds = [n * t // n_threads for t in range(n_threads + 1)]

    def work(t):
        lo, hi = bounds[t], bounds[t + 1]
        m = hi - lo
        if m == 0:
            return 0
        cap = m + 1024
        o_r = np.zeros(cap, np.int32); o_v = np.zeros(cap, np.int32); o_c = np.zeros(cap, np.uint8)
        return lib.rvm_oracle_map_soa(m, pos.ctypes.data + 4 * lo, coff.ctypes.data + 8 * lo, cig.ctypes.data, seq.ctypes.data + L * lo,
                                      qual.ctypes.data + L * lo, L, baseq, len(vp), vp.ctypes.data, rl.ctypes.data, cap, o_r.ctypes.data,
                                      o_v.ctypes.data, o_c.ctypes.data, None)
    with ThreadPoolExecutor(n_threads) as ex:
        return sum(ex.map(work, range(n_threads)))


def call_text(v, shard, calls, qname_prefix="q"):
    """Mapper TSV (read_variant_map.py:117) of one chromosome from a GPU call list + the shard's per-record fields, as input
    for the phasing oracle.  Composite allele texts (code 4: inserted bases / IUPAC symbols) are written as '<other>': any
    text outside the individual's alleles lands in the same class downstream (phaser.py:1312-1324)."""
    ri = calls.read_idx.cpu().numpy(); vi = calls.var_idx.cpu().numpy(); cd = calls.code.cpu().numpy()
    qid = shard.qid.cpu().numpy()[ri]; asc = shard.aln_score.cpu().numpy()[ri]
    pos = v.pos.tolist(); ref = v.ref.tolist(); alt = v.alt.tolist()
    uid = ["%s_%d_%s_%s" % (v.chrom, pos[i], synth.BASES[ref[i]], synth.BASES[alt[i]]) for i in range(len(v))]
    names = ["A", "C", "G", "T", "<other>"]
    rs = v.rsid; gt = v.gt
    return "".join("%s%d\t%s\t%s\t%s\t%d\t%s\tNone\n" % (qname_prefix, q, uid[j], rs[j], names[c if c < 4 else 4], a, gt[j])
                   for q, j, c, a in zip(qid.tolist(), vi.tolist(), cd.tolist(), asc.tolist()))


# ------------------------------------------------------------------ GPU stage results from fixtures (CPU-only host-stage tests)
def genome_from_saved(saved, chrom_list, n_bams):
    """tests/golden/tally/*.pkl.gz hold, per chromosome, what K_tally produced on an MI355X (per-variant counters, edges with their
    nine cells, first-appearance ranks) plus the kept call lines.  Builds what Engine._tally_genome() returns for `chrom_list`:
    the chromosomes' arrays concatenated into joint variant / line spaces and the per-(variant, allele, BAM) read lists as a CSR
    (a stable grouping of the saved lines: what the device sort does)."""
    nb = n_bams
    var_base = {}; line_base = {}
    NV = 0; L = 0
    parts = {k: [] for k in ("var_count", "var_first", "var_distinct", "var_rank", "ea", "eb", "cells", "linked")}
    keys = []; vals = []
    for c in chrom_list:
        R = saved["tally"][c]; nv = R["nv"]
        var_base[c] = NV
        n_lines = len(R["line_cls"])
        for b, base, n in R["bam_offsets"]:
            line_base[(c, b)] = (L + base, n)
        parts["var_count"].append(R["var_count"].reshape(nv, 3)); parts["var_distinct"].append(R["var_distinct"].reshape(nv, 3))
        vf = R["var_first"].astype(np.int64).copy(); vf[vf >= 0] += L
        parts["var_first"].append(vf)
        vr = R["var_rank"].astype(np.uint64).copy()
        ok = vr != np.uint64(0xFFFFFFFFFFFFFFFF)
        vr[ok] += np.uint64((L << 32) + L)
        parts["var_rank"].append(vr)
        parts["ea"].append(R["ea"].astype(np.int32) + NV); parts["eb"].append(R["eb"].astype(np.int32) + NV)
        parts["cells"].append(R["cells"].reshape(-1, 9)); parts["linked"].append(R["linked"].astype(np.uint8))
        cls = R["line_cls"]; sel = np.nonzero(cls < 2)[0]
        keys.append(((R["line_var"][sel].astype(np.int64) + NV) * 2 + cls[sel]) * nb + R["line_bam"][sel])
        vals.append(R["line_qid"][sel].astype(np.int32))
        NV += nv; L += n_lines
    cat = lambda xs, dt, shape=None: (np.concatenate(xs).astype(dt) if xs else np.zeros((0,) + (shape or ()), dt))
    key = cat(keys, np.int64); val = cat(vals, np.int32)
    order = np.argsort(key, kind="stable")
    rl_start = np.zeros(NV * 2 * nb + 1, dtype=np.uint32)
    np.cumsum(np.bincount(key, minlength=NV * 2 * nb), out=rl_start[1:])
    vc = cat(parts["var_count"], np.int32).reshape(NV, 3)
    cto = _cto(cat(parts["cells"], np.int32).reshape(-1, 9))
    return {"stats": _stats(cto), "noise": _noise(vc), "nv": NV, "nb": nb, "var_base": var_base, "line_base": line_base, "n_lines": L, "n_kept": int(vc.sum()), "var_count": vc,
            "var_first": cat(parts["var_first"], np.int64), "var_distinct": cat(parts["var_distinct"], np.int32).reshape(NV, 3),
            "var_rank": cat(parts["var_rank"], np.uint64), "ea": cat(parts["ea"], np.int32), "eb": cat(parts["eb"], np.int32),
            "cto": cto, "linked": cat(parts["linked"], np.uint8), "rl_start": rl_start,
            "rl_qid": val[order], "resident": False}


def _cto(cells):
    """the three sums of test_variant_connection per pair (what k_edge_final hands out): same configuration, opposite, other"""
    c = cells.astype(np.int64)
    return np.stack([c[:, 0] + c[:, 4], c[:, 3] + c[:, 1], c[:, 6] + c[:, 7] + c[:, 2] + c[:, 5] + c[:, 8]], 1).astype(np.int32)


def _stats(cto):
    """the five derived planes k_edge_final hands out (phaser.py:1637-1649): same, opposite, supporting, total, configuration"""
    cis = cto[:, 0].astype(np.int32); trans = cto[:, 1].astype(np.int32); oth = cto[:, 2].astype(np.int32)
    cfg = np.where(cis > trans, 0, np.where(cis < trans, 1, -1)).astype(np.int32)
    return np.ascontiguousarray(np.stack([cis, trans, np.maximum(cis, trans), cis + trans + oth, cfg], 0))


def _noise(vc):
    """the two noise counters k_noise hands out (phaser.py:610-632)"""
    vc = vc.astype(np.int64)
    m = vc[:, 0] + vc[:, 1]; mm = vc[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        ok = (m > 0) & ((mm.astype(np.float64) / (mm + m).astype(np.float64)) < 0.05)
    return int(m[ok].sum()), int(mm[ok].sum())


def component_labels_cpu(G, keep_all):
    """label[v] = smallest variant of v's connected component over the kept edges (what phz_components returns)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    nv = G["nv"]
    k = np.nonzero(keep_all)[0]
    g = coo_matrix((np.ones(len(k), np.int8), (G["ea"][k], G["eb"][k])), shape=(nv, nv))
    _, comp = connected_components(g, directed=False)
    first = np.full(comp.max() + 1 if nv else 0, nv, dtype=np.int64)
    np.minimum.at(first, comp, np.arange(nv))
    return first[comp].astype(np.int32)


def stub_gpu_stages(eng, saved):
    """Engine whose GPU stages (K_tally, components) answer from a fixture: the host stages run unchanged on CPU."""
    eng._tally_genome = lambda: genome_from_saved(saved, eng.chrom_list, len(eng.bam_names))
    eng._component_labels = lambda keep_all: component_labels_cpu(eng.G, keep_all)


# ------------------------------------------------------------------ kernel logic under the host-side HIP emulation (tests/hipemu)
def emu_library(tally_tile=0, row_wave_min=None, stat_n=None):
    """ctypes handle of tests/hipemu/_build/libphz_emu.so (built on first use): the translation units of libphz without gfx950
    intrinsics, compiled by g++ against tests/hipemu/hipemu.h.  TEST INFRASTRUCTURE: the product never loads it.
    tally_tile = 256 / 512: the variant whose K_tally works on tiles of that many lines; row_wave_min = 0: the variant whose row stage
    formats every block row by a wave; stat_n: the variant whose row stage treats blocks of more than stat_n variants as "beyond the tables"."""
    import importlib.util
    from phaser_amd import _lib
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu")
    spec = importlib.util.spec_from_file_location("build_emu", os.path.join(here, "build_emu.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    lib = ctypes.CDLL(mod.build(tally_tile=tally_tile, row_wave_min=row_wave_min, stat_n=stat_n))
    for name, (res, args) in _lib.SYMBOLS.items():
        if hasattr(lib, name):
            fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    return lib


class EmuContext:
    """A phz_ctx of the emulation library with the interface of phaser_amd._lib.Context."""

    def __init__(self, lib=None):
        from phaser_amd import _lib
        self.lib = lib or emu_library()
        h = ctypes.c_void_p()
        assert self.lib.phz_ctx_create(0, ctypes.byref(h)) == 0
        self.h = h; self.device = 0; self._lib = _lib

    def check(self, st, allow=()):
        if st != 0 and st not in allow:
            raise self._lib.PhzError(st, (self.lib.phz_last_error(self.h) or b"").decode() or self.lib.phz_strerror(st).decode())
        return st

    def __del__(self):
        try:
            self.lib.phz_ctx_destroy(self.h)
        except Exception:
            pass


def stub_emu_stages(eng, saved):
    """Engine on an EmuContext: the tally results come from a fixture and are adopted as the ctx's resident tally (phz_tally_import), so
    that the DEVICE row stage (phz_rowsdev_*) runs on them under the emulation; the host stage stays available as its fallback."""
    from phaser_amd import _lib

    def tally():
        G = genome_from_saved(saved, eng.chrom_list, len(eng.bam_names))
        nv = G["nv"]; nb = G["nb"]
        rl_list = np.repeat(np.arange(nv * 2 * nb, dtype=np.uint32), np.diff(G["rl_start"].astype(np.int64))).astype(np.uint32)
        sz = _lib.phz_tally_sizes(G["n_lines"], G["n_kept"], len(G["ea"]), len(G["rl_qid"]), 0, 0, G["noise"][0], G["noise"][1])
        keep = [np.ascontiguousarray(G[k]) for k in ("var_count", "var_first", "var_distinct", "var_rank", "ea", "eb", "linked", "cto", "rl_start", "rl_qid", "stats")]
        vp = lambda a: ctypes.c_void_p(a.ctypes.data) if a.size else None
        out = _lib.phz_tally_out(vp(keep[0]), vp(keep[1]), vp(keep[2]), vp(keep[3]), None, vp(keep[4]), vp(keep[5]), None, vp(keep[6]), vp(keep[7]),
                                 vp(keep[8]), vp(keep[9]), vp(keep[10]))
        eng.ctx.check(eng.lib.phz_tally_import(eng.ctx.h, nv, nb, ctypes.byref(sz), ctypes.byref(out), vp(rl_list), _lib.PHZ_HOST))
        G["resident"] = True; G["fetched"] = True; G["n_edges"] = len(G["ea"]); G["n_read_list"] = len(G["rl_qid"])
        return G
    eng._tally_genome = tally
    eng._component_labels = lambda keep_all: component_labels_cpu(eng.G, keep_all)
