"""phASER command line, MI355X hot path underneath.

Same flags, defaults and output files as phaser/phaser.py:26-178 (`main`), :182-321 (`parse_sample`) and
:378-1263 (`process_vcf`).  What differs is below the CLI: no samtools / bedtools / tabix subprocesses (BAM,
BED and VCF are read in-process) and the seven multiprocessing stages are one `Engine` driving libphz.so.
The phased VCF (`write_vcf`) is written as BGZF `<o>.vcf.gz` with its tabix index `<o>.vcf.gz.tbi`.  Not supported:
`--process_slow`, `--output_network`.

    python -m phaser_amd.phaser --vcf S.vcf.gz --bam a.bam,b.bam --sample S1 --mapq 255 --baseq 10 --paired_end 1 --o out
"""
from __future__ import annotations

import argparse
import datetime
import os
import sys
import time
from typing import Dict, List

import torch
import torch.distributed as dist

from . import bamio, samio, vcf
from . import dist as pdist
from .engine import Config, Engine

VERSION = "1.2.0"


def out(text=""):
    print(text)
    sys.stdout.flush()


def fatal_error(text):
    out("     FATAL ERROR: " + text)
    sys.exit(1)


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--bam", required=False, default='')
    p.add_argument("--vcf", required=True, default='')
    p.add_argument("--sample", required=False, default='')
    p.add_argument("--mapq", required=True)
    p.add_argument("--baseq", type=int, required=True)
    p.add_argument("--paired_end", required=True)
    p.add_argument("--o", required=True)
    p.add_argument("--python_string", default="python3")
    p.add_argument("--haplo_count_bam_exclude", default="")
    p.add_argument("--haplo_count_blacklist", default="")
    p.add_argument("--cc_threshold", type=float, default=0.01)
    p.add_argument("--isize", default="0")
    p.add_argument("--as_q_cutoff", type=float, default=0.05)
    p.add_argument("--blacklist", default="")
    p.add_argument("--write_vcf", type=int, default=1)
    p.add_argument("--include_indels", type=int, default=0)
    p.add_argument("--output_read_ids", type=int, default=0)
    p.add_argument("--remove_dups", type=int, default=1)
    p.add_argument("--pass_only", type=int, default=1)
    p.add_argument("--unphased_vars", type=int, default=1)
    p.add_argument("--chr_prefix", type=str, default="")
    p.add_argument("--gw_phase_method", type=int, default=0)
    p.add_argument("--gw_af_field", default="AF")
    p.add_argument("--gw_phase_vcf", type=int, default=0)
    p.add_argument("--gw_phase_vcf_min_confidence", type=float, default=0.90)
    p.add_argument("--threads", type=int, default=1)
    p.add_argument("--max_block_size", type=int, default=15)
    p.add_argument("--temp_dir", default="")
    p.add_argument("--max_items_per_thread", type=int, default=100000)
    p.add_argument("--show_warning", type=int, default=0)
    p.add_argument("--debug", type=int, default=0)
    p.add_argument("--chr", default="")
    p.add_argument("--unique_ids", type=int, default=0)
    p.add_argument("--py_hash_order", type=int, default=0,
                   help="1: rows and read labels of variant_connections / haplotypes / haplotypic_counts in the order CPython 3.10 gives the reference's sets "
                        "run under PYTHONHASHSEED=0, i.e. files byte-identical to the reference's (the str hash and the set of that interpreter are restated "
                        "in libphz: works under any Python, costs seconds at whole-genome scale; PHZ_PYORDER_PYTHON=1 selects the pure-Python twin)")
    p.add_argument("--id_separator", default="_")
    p.add_argument("--output_network", default="")
    p.add_argument("--process_slow", type=int, default=0, required=False)
    return p


def load_bed(path: str) -> List[tuple]:
    """BED file -> [(chrom, start, end)] (0-based, half-open); header / track lines skipped."""
    iv = []
    with open(path) as f:
        for line in f:
            c = line.rstrip("\n").split("\t")
            if len(c) >= 3 and not line.startswith(("#", "track", "browser")):
                iv.append((c[0], int(c[1]), int(c[2])))
    return iv


LAST_STAGE_SECONDS = {}          # stage name -> seconds of the last main() of this process (rank 0): read by bench.py's end_to_end_files entry


def main(argv=None):
    """The CLI.  Whatever way it ends (result, fatal_error, exception), a BAM prefetch still running on its helper thread is waited for:
    the interpreter must not start tearing down with GPU work of ours in flight."""
    state = {}
    try:
        return _main(argv, state)
    finally:
        for key in ("prefetch", "early"):
            t = state.get(key)
            if t is not None and t.is_alive():
                t.join()


def _main(argv, state):
    args = build_parser().parse_args(argv)
    rank, world = pdist.world()
    # torch's intra-op pool sized by the container's CPU quota, not by the host's core count (see dist.effective_cpus)
    torch.set_num_threads(max(1, min(torch.get_num_threads(), pdist.effective_cpus() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))))
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, n_dev)
    if n_dev:
        torch.cuda.set_device(local)          # before any collective: NCCL / barrier use the current device
    force_pg = os.environ.get("PHZ_DIST_FORCE_COLLECTIVES") == "1"        # one rank, yet every collective of the multi-rank path runs (dist.collectives_live)
    if world == 1 and (int(os.environ.get("WORLD_SIZE", "1")) > 1 or (force_pg and not dist.is_initialized())):
        # one rank per GPU over RCCL ("nccl"); PHZ_DIST_BACKEND=gloo lets several ranks share a GPU (tests on a 1-GPU box)
        backend = os.environ.get("PHZ_DIST_BACKEND", "nccl" if n_dev else "gloo")
        if force_pg and "WORLD_SIZE" not in os.environ:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        rank, world = pdist.world()
    say = out if rank == 0 else (lambda *_: None)
    say("")
    say("##################################################")
    say("              Welcome to phASER v%s" % VERSION)
    say("  Author: Stephane Castel (stephanecastel@gmail.com)")
    say("  Updated by: Bishwa K. Giri (bkgiri@uncg.edu)")
    say("  MI355X hot path: phaser_amd (K_map / K_tally on gfx950)")
    say("##################################################")
    say("")
    if args.id_separator == ":" or args.id_separator == "":
        fatal_error("ID separator must not be ':' or blank. Please choose another separator that is not found in the contig names.")
    if args.process_slow != 0:
        fatal_error("--process_slow is not supported by this build (the GPU path holds all chromosomes; SURVEY.md section 2).")
    if args.output_network != "":
        fatal_error("--output_network is not supported by this build.")
    if args.py_hash_order and world > 1:
        # every rank sees the same arguments and stops HERE, before the first collective: nobody is left waiting in a barrier
        fatal_error("--py_hash_order 1 needs all chromosomes on one rank (run it without torch.distributed).")
    if not os.path.isfile(args.vcf):
        fatal_error("VCF file does not exist.")
    for xfile in [args.blacklist, args.haplo_count_blacklist]:
        if xfile != "" and not os.path.isfile(xfile):
            fatal_error("File: %s not found." % xfile)
    bam_list = args.bam.split(",")
    for xfile in bam_list:
        if xfile != "" and not os.path.isfile(xfile):
            fatal_error("File: %s not found." % xfile)
    start = time.time()
    marks = [("start", time.perf_counter())]
    # scipy's binomial machinery initialises lazily on the first call (~80 ms): pay for it on a side thread while the BAM is being
    # inflated and decoded on the GPU instead of in the middle of the pair tests
    import threading
    import numpy as _np
    from .engine import binom_cdf_dedup
    threading.Thread(target=lambda: binom_cdf_dedup(_np.array([1]), _np.array([3]), 0.99), daemon=True).start()

    # The device context (HIP runtime start-up, the library's code objects, the K_map stream: 0.1-0.2 s in a fresh process) is created on a
    # helper thread from the first moment on -- the VCF is being read and gunzipped meanwhile; the BAM prefetch below takes it over
    early = None
    if (world == 1 and args.chr == "" and n_dev and os.environ.get("PHZ_BAM_HOST") != "1" and os.environ.get("PHZ_BAM_PREFETCH", "1") == "1"
            and not any(b.endswith(".sam") for b in bam_list)):
        early = {}

        def _early():
            t0 = time.perf_counter()
            try:
                from .mapper import Mapper
                early["mapper"] = Mapper(local)
            except BaseException as e:
                early["err"] = e
            early["seconds"] = time.perf_counter() - t0

        early["thread"] = threading.Thread(target=_early, daemon=True, name="phz-ctx")
        state["early"] = early["thread"]
        early["thread"].start()

    def mark(name):          # PHZ_TIMING=1: stage timings on stderr at the end (not part of the reference's output)
        marks.append((name, time.perf_counter()))
    say('STARTED "Read backed phasing and ASE/haplotype analyses" ... ')
    say("    DATE, TIME : %s" % (datetime.datetime.now().strftime('%Y-%m-%d, %H:%M:%S')))
    say("#1. Loading heterozygous variants into intervals...")
    # the host region that will receive the row text is page-locked on a helper thread while the VCF is read (~0.1 s per GB, independent of
    # everything else); sized from the BAMs, grown later if it turns out too small
    arena = None
    if not any(b.endswith(".sam") for b in args.bam.split(",")) and os.environ.get("PHZ_EARLY_ARENA", "1") == "1":
        import threading

        def _arena():
            try:
                import torch
                if torch.cuda.is_available():
                    from . import rowsdev
                    total = sum(os.path.getsize(b) for b in args.bam.split(",") if os.path.exists(b))
                    rowsdev.prepare_arena(int(min(4 << 30, max(64 << 20, total // 4))))
            except Exception:
                pass

        arena = threading.Thread(target=_arena, daemon=True)
        arena.start()
    # The first BAM is inflated / decoded / filtered on the GPU (PCIe- and GPU-bound) WHILE the host reads and parses the VCF: the stages
    # share nothing but the names of the chromosomes to keep.  Those come from the VCF's tabix index when there is one (its sequence
    # names: a few KB, before the VCF itself is touched), else they are guessed from the text once it is gunzipped
    # (vcf.contig_names_guess); either way they are checked against the parsed table below -- names that do not cover the table discard
    # the prefetch and the BAM is read as before.
    pre = None
    pre_ok = early is not None

    def start_prefetch(guess):
        nonlocal pre
        pre = {"guess": set(guess), "ready": threading.Event()}
        try:
            pre["key"] = (int(args.mapq.split(",")[0]), float(args.isize.split(",")[0]), int(args.paired_end.split(",")[0]))
        except ValueError:
            pre = None
            return

        def _prefetch():
            try:
                t0 = time.perf_counter()
                early["thread"].join()
                if "err" in early:
                    raise early["err"]
                pre["mapper"] = early["mapper"]
                pre["ready"].set()
                its = {}
                mq0, isz0, pe0 = pre["key"]
                if os.environ.get("PHZ_TIMING"):
                    sys.stderr.write("[phz timing]   device context %.3f s on its own thread; the BAM prefetch waited %.3f s for it and starts %.3f s into the run\n" % (
                        early["seconds"], time.perf_counter() - t0, time.perf_counter() - marks[0][1]))
                pre["shards"] = bamio.shards_from_bam_device(pre["mapper"].ctx, bam_list[0], its, mq0, args.remove_dups == 1, pe0 == 1, isz0, chroms=guess,
                                                             device="cuda:%d" % local)
                pre["interners"] = its
            except BaseException as e:               # reported by the ordinary path, which runs instead
                pre["err"] = e
            finally:
                pre["ready"].set()

        pre["thread"] = threading.Thread(target=_prefetch, daemon=True, name="phz-bam-prefetch")
        state["prefetch"] = pre["thread"]
        pre["thread"].start()

    if pre_ok:
        names = vcf.contig_names_from_tbi(args.vcf + ".tbi")
        if names:
            start_prefetch([args.chr_prefix + c for c in names])
    _tv0 = time.perf_counter()
    data = vcf.read_bytes(args.vcf, as_array=True)          # bgzipped: a uint8 array over the native reader's buffer (no 75 MB Python bytes object)
    _tv1 = time.perf_counter()
    if pre_ok and pre is None:
        guess = [args.chr_prefix + c for c in vcf.contig_names_guess(data)]
        if guess:
            start_prefetch(guess)
    sample_col = None
    # (the header sits at the top: only the first 20,000 lines are looked at, and only they are split -- split(b"\n", 20000) on the whole text copied the
    #  75 MB behind them once more, 0.05 s)
    head_end = 0
    full = data
    data = bytes(memoryview(full)[:8 << 20]) if not isinstance(full, (bytes, bytearray)) else full          # the header scan looks at bytes (a header beyond 8 MB: sample not found)
    for _ in range(20000):
        nxt = data.find(b"\n", head_end)
        if nxt < 0:
            head_end = len(data); break
        head_end = nxt + 1
        if data[head_end:head_end + 1] not in (b"#", b"\n", b"\r", b""):          # the first record: one more line is taken (it ends the loop below as it always did)
            nxt = data.find(b"\n", head_end)
            head_end = len(data) if nxt < 0 else nxt + 1
            break
    head_lines = data[:head_end].split(b"\n")
    data = full
    for raw in head_lines:
        if b"#CHR" in raw:
            cols = raw.decode().rstrip().split("\t")
            m = {cols[i]: i for i in range(9, len(cols))}
            if args.sample not in m:
                fatal_error("Sample '%s' not found in the input VCF file." % args.sample)
            sample_col = m[args.sample]
            break
        if raw and raw[0:1] != b"#":
            break
    if sample_col is None:
        fatal_error("Sample '%s' not found in the input VCF file." % args.sample)
    # cut -f 1-9,S | grep -v '0|0\|1|1' [| bedtools intersect -v -b blacklist]   (phaser.py:220-225) and the bedtools intersect of
    # --haplo_count_blacklist (:232-241) all run inside the native loader (interval lookups by binary search)
    load_kw = dict(chrom_of_interest=args.chr, pass_only=args.pass_only, include_indels=args.include_indels, chr_prefix=args.chr_prefix,
                   id_separator=args.id_separator, gw_phase_method=args.gw_phase_method, gw_af_field=args.gw_af_field,
                   contig_ban=(args.id_separator, ":"), threads=max(1, args.threads))
    if args.blacklist != "":
        say("    removing blacklisted variants and processing VCF...")
    if args.haplo_count_blacklist != "":
        say("#1b. Loading haplotypic count blacklist intervals...")
    _tv2 = time.perf_counter()
    vs = vcf.load_variants(data, sample_column=sample_col, grep_hom=True, drop_bed=load_bed(args.blacklist) if args.blacklist != "" else None,
                           mark_bed=load_bed(args.haplo_count_blacklist) if (args.haplo_count_blacklist != "" and args.chr_prefix == "") else None,
                           **load_kw)       # with --chr_prefix the reference's blacklist keys (VCF names) never match its lookups (prefixed names, :1070)
    if os.environ.get("PHZ_TIMING"):
        sys.stderr.write("[phz timing]   vcf: read + inflate %.3f s, contig guess + header %.3f s, parse + tables %.3f s (%d bytes of text)\n" % (
            _tv1 - _tv0, _tv2 - _tv1, time.perf_counter() - _tv2, len(data)))
    mark("vcf read + het-variant table")
    say("     creating variant mapping table...")
    say("          %d heterozygous sites being used for phasing (%d filtered, %d indels excluded, %d unphased)" %
        (vs.het_count, vs.filter_count, vs.indels_excluded, vs.unphased_count))
    say()
    if vs.het_count == 0:
        fatal_error("No heterozygous sites that passed all filters were included in the analysis, phASER cannot continue. "
                    "Check blacklist and pass_only arguments.")
    say("#2. Retrieving reads that overlap heterozygous sites...")
    # per-BAM lists (phaser.py:469-513)
    from collections import OrderedDict
    base = [os.path.basename(x).replace(".bam", "") for x in bam_list]
    counter = OrderedDict(); bam_names = []
    for x in base:
        if base.count(x) > 1:
            counter[x] = counter.get(x, 0) + 1
            bam_names.append(x + "." + str(counter[x]))
        else:
            bam_names.append(x)

    def per_bam(val, what):
        lst = val.split(",")
        if len(lst) == 1 and len(bam_list) > 1:
            lst = lst * len(bam_list)
        elif len(lst) != len(bam_list):
            fatal_error("Number of %s values and input BAMs does not match. Supply either one %s to be used for all BAMs or one %s per input BAM." % (what, what, what))
        return lst
    mapq_list = per_bam(args.mapq, "mapq")
    isize_list = list(map(float, per_bam(args.isize, "isize")))
    pe_list = per_bam(args.paired_end, "paired_end")
    excl = [x - 1 for x in map(int, args.haplo_count_bam_exclude.split(","))] if args.haplo_count_bam_exclude != "" else []
    cfg = Config(baseq=args.baseq, as_q_cutoff=args.as_q_cutoff, cc_threshold=args.cc_threshold, max_block_size=args.max_block_size,
                 id_separator=args.id_separator, unphased_vars=args.unphased_vars, gw_phase_method=args.gw_phase_method,
                 output_read_ids=args.output_read_ids, unique_ids=args.unique_ids, haplo_count_bam_exclude=excl, py_hash_order=args.py_hash_order,
                 include_indels=args.include_indels, host_threads=max(1, args.threads))
    if pre is not None:
        pre["ready"].wait()
    mapper0 = pre.get("mapper") if pre is not None else None
    if mapper0 is None and early is not None:          # no prefetch after all: the context made for it serves the ordinary path
        early["thread"].join()
        mapper0 = early.get("mapper")
    eng = Engine(vs, bam_names, cfg, device=local, mapper=mapper0)
    eng.spool_dir = os.path.dirname(os.path.abspath(args.o))       # ranks hand their row text to rank 0 through files next to the outputs
    device = "cuda:%d" % local
    interners: Dict[str, object] = {}
    any_sam = any(b.endswith(".sam") for b in bam_list)       # text inputs keep everything on the Python reader
    # multi-GPU: chromosomes are sharded across ranks by record count (LPT).  Before anything is decoded the count is not known; its
    # proxy is the compressed bytes each chromosome occupies in the BAMs (a few dozen BGZF member inflations per file); text inputs
    # fall back to the variant count
    if world > 1:
        weights = {c: 0.0 for c in vs.chroms}
        if not any_sam:
            for bam in bam_list:
                for name, nbytes in bamio.bam_ref_weights(bam, threads=max(0, args.threads if args.threads > 1 else 0)).items():
                    if name in weights:              # VCF names already carry --chr_prefix (they are the BAM's names)
                        weights[name] += float(nbytes)
        if sum(weights.values()) <= 0:
            weights = {c: float(len(v)) for c, v in vs.chroms.items()}
        owner = pdist.assign_chromosomes(weights, world)
        eng.set_owned([c for c in vs.chroms if owner[c] == rank])
    mine = set(eng.chrom_list)
    # While the BAMs are read: the per-variant tables of the row stage go to the GPU (0.07 s at genome scale, independent of the reads)
    warm = None
    if not any_sam and cfg.device_rows:
        import threading

        def _warm():
            try:
                import torch
                if not torch.cuda.is_available():
                    return
                from . import rowsdev
                from ._lib import Context
                # its OWN phz_ctx (stream, scratch, error string): the BAM prefetch thread may still be inside phz_bamdev_* on the
                # Engine's ctx, and a phz_ctx serves one thread at a time (include/phz.h)
                rowsdev.tables_for(eng, ctx=Context(local))
            except Exception:
                pass            # finish() does both itself when they are not there

        warm = threading.Thread(target=_warm, daemon=True)
        warm.start()
    bam_paths = []                 # which decoder read each BAM: "device" (phz_bamdev_*), "host" (phz_bam_*: the file was declined or PHZ_BAM_HOST=1), "sam" (text)
    for bi, (bam, mq, isz, pe) in enumerate(zip(bam_list, mapq_list, isize_list, pe_list)):
        say("     file: %s" % bam)
        say("          minimum mapq: %s" % mq)
        say("          mapping reads to variants...")
        if bam.endswith(".sam"):
            shards = samio.shards_from_sam(open(bam).read(), interners, isz)
            bam_paths.append("sam")
        elif any_sam:
            shards = bamio.shards_from_bam(bam, interners, int(mq), args.remove_dups == 1, int(pe) == 1, isz, chroms=mine)
            bam_paths.append("host")
        else:
            # BGZF inflate + record decode + filters + packing on the GPU (phz_bamdev_*); files it declines, and PHZ_BAM_HOST=1, go
            # through the host decoder (phz_bam_*, --threads host threads).  QNAME interning is host-side in both.
            shards = None
            declined = False
            if bi == 0 and pre is not None:
                pre["thread"].join()
                verdict = "discarded (%s)" % ("error in the prefetch" if "err" in pre else "chromosomes %s not in the guess" % sorted(mine - pre["guess"])
                                              if not mine <= pre["guess"] else "filters differ")
                if "err" not in pre and mine <= pre["guess"] and pre["key"] == (int(mq), isz, int(pe)):
                    if pre.get("shards") is None:
                        declined = True                  # the device path declined the file: straight to the host decoder
                        verdict = "device path declined the file"
                    else:
                        shards = {c: s for c, s in pre["shards"].items() if c in mine}
                        interners.update({c: it for c, it in pre["interners"].items() if c in mine})
                        verdict = "used"
                if os.environ.get("PHZ_TIMING"):
                    sys.stderr.write("[phz timing]   bam prefetch during the VCF parse: %s\n" % verdict)
                pre.pop("shards", None); pre.pop("interners", None)
            if shards is None and not declined and os.environ.get("PHZ_BAM_HOST") != "1":
                shards = bamio.shards_from_bam_device(eng.ctx, bam, interners, int(mq), args.remove_dups == 1, int(pe) == 1, isz, chroms=mine,
                                                      device=device)
            bam_paths.append("device" if shards is not None else "host")
            if shards is None:
                shards = bamio.shards_from_bam_native(bam, interners, int(mq), args.remove_dups == 1, int(pe) == 1, isz, chroms=mine,
                                                      threads=max(0, args.threads if args.threads > 1 else 0))
        mark("bam decode + filters + qname interning")
        items = []
        for chrom in vs.chroms:
            if chrom in shards and chrom in mine:
                names = None
                if args.output_read_ids == 1 or (args.py_hash_order == 1 and os.environ.get("PHZ_PYORDER_PYTHON") == "1"):
                    names = interners[chrom].names          # a list of str: the host row twin and the pure-Python raw-byte twin index it
                elif args.py_hash_order == 1:
                    names = interners[chrom].pool() if hasattr(interners[chrom], "pool") else interners[chrom].names      # (blob, offsets): what the native raw-byte tier reads
                items.append((chrom, shards[chrom].to(device), len(interners[chrom]), names))
        eng.add_shards(bi, items)                # all chromosomes of the BAM in one K_map submission
        for it in items:
            say("               completed chromosome %s..." % it[0])
        for chrom in interners:
            if chrom in eng.n_qid:
                eng.n_qid[chrom] = len(interners[chrom])
        mark("H2D + K_map")
        say("          processing mapped reads...")
        n_before = len(eng.log)
        eng.close_bam(bi)
        eng.resolve_cutoffs()          # (the CLI prints the BAM's cutoff line here; a caller that streams passes leaves it to finish())
        mark("AS cutoff")
        for line in eng.log[n_before:]:
            say(line)
    say("#3. Identifying connected variants...")
    say("     calculating sequencing noise level...")
    n_before = len(eng.log)
    if warm is not None:
        warm.join()
    if arena is not None:
        arena.join()
    files = eng.finish(chunks=True)
    mark("tally + pair tests + components + block phasing + rows")
    if os.environ.get("PHZ_TIMING"):
        sys.stderr.write("[phz timing]   finish: %s\n" % ", ".join("%s %.3f" % (k, v) for k, v in eng.stats.items()))
    if files is not None:
        for line in eng.log[n_before:]:
            say(line)
        say("#4. Identifying haplotype blocks...")
        say("#5. Phasing blocks...")
        say("#6. Outputting haplotypes...")
        writer = None
        five = [(args.o + "." + name + ".txt", body) for name, body in files.items()]
        if args.write_vcf == 1:                 # the five files go out on a helper thread while the phased VCF is put together
            import threading
            werr = []

            def _write():
                try:
                    pdist.write_files(five, threads=max(1, min(16, args.threads)))
                except BaseException as e:      # re-raised on the main thread
                    werr.append(e)

            writer = threading.Thread(target=_write)
            writer.start()
        else:
            pdist.write_files(five, threads=max(1, min(16, args.threads)))
            mark("write the five files")
        up = pc = 0
        if args.write_vcf == 1:
            from . import vcfout
            say("#7. Outputting phased VCF...")
            if args.gw_phase_vcf == 1:
                say("     GT field is being updated with phASER genome wide phase when applicable. This can be changed using the --gw_phase_vcf argument.")
            elif args.gw_phase_vcf == 2:
                say("     GT field is being updated with either phASER genome wide phase or phASER block phase with PS specified, depending on phase anchoring quality.")
            else:
                say("     GT field is not being updated with phASER genome wide phase. This can be changed using the --gw_phase_vcf argument.")
            vtxt, up, pc = vcfout.phased_vcf_text(data, sample_col, eng, args.id_separator, args.chr, args.gw_phase_vcf,
                                                  args.gw_phase_vcf_min_confidence, threads=max(1, args.threads), as_bytes=True)
            writer.join()
            if werr:
                raise werr[0]
            mark("phased VCF text (+ the five files on a helper thread)")
            say("     Compressing and tabix indexing output VCF...")
            if not vcfout.write_bgzf(args.o + ".vcf.gz", vtxt, max(0, args.threads if args.threads > 1 else 0), index="vcf"):
                say("     WARNING: the VCF is not position-sorted, no tabix index written")
            mark("phased VCF bgzf + tabix")
        say('')
        say("     COMPLETED using %d reads in %d seconds using %d GPU(s)" % (eng.total_lines, time.time() - start, world))
        say("     PHASED  %d of %d all variants (= %f) with at least one other variant" %
            (eng.phased, vs.het_count, float(eng.phased) / float(vs.het_count)))
        if args.write_vcf == 1:
            if vs.unphased_count > 0:
                say("     GENOME WIDE PHASED  %d of %d unphased variants (= %f)" % (up, vs.unphased_count, float(up) / float(vs.unphased_count)))
            say("     GENOME WIDE PHASE CORRECTED  %d of %d variants (= %f)" % (pc, vs.het_count, float(pc) / float(vs.het_count)))
        # which of this build's paths did the work (not a line of the reference): a run that fell back to a host twin says so
        say("     HOT PATH  bam: %s, rows: %s%s" % (",".join(bam_paths) if bam_paths else "-", getattr(eng, "rows_path", "-"),
                                                    " (%s)" % eng.rows_fallback if getattr(eng, "rows_fallback", None) else ""))
        say('')
        say("The End.")
        global LAST_STAGE_SECONDS
        LAST_STAGE_SECONDS = {name: t - t_prev for (_, t_prev), (name, t) in zip(marks[:-1], marks[1:])}
        LAST_STAGE_SECONDS["total"] = marks[-1][1] - marks[0][1]
        if os.environ.get("PHZ_TIMING"):
            for (_, t_prev), (name, t) in zip(marks[:-1], marks[1:]):
                sys.stderr.write("[phz timing] %-55s %8.3f s\n" % (name, t - t_prev))
            sys.stderr.write("[phz timing] %-55s %8.3f s\n" % ("total", marks[-1][1] - marks[0][1]))
    pdist.cleanup_spool()          # (barrier) every rank removes its spool file once rank 0 has written the outputs
    return 0


if __name__ == "__main__":
    sys.exit(main())
