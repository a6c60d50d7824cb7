"""Helper of tests/test_pyorder.py, run as a subprocess with PYTHONHASHSEED=0: host stages of one fixture with Config.py_hash_order=1 -> for every
output file 'raw' if the bytes equal the reference's file, else the first differing line."""
import gzip
import json
import os
import pickle
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "oracle"))
GOLD = os.path.join(REPO, "tests", "golden")


def main():
    import numpy as np
    from helpers import OUTPUTS, stub_gpu_stages, option_case_kwargs
    from phasing_oracle import bam_display_names
    from phaser_amd import vcf
    from phaser_amd.engine import Config, Engine
    case = sys.argv[1]
    gz = lambda p: gzip.open(p, "rt").read()
    load = {}; cfg = {}
    if case.startswith("opts_"):
        meta = json.load(open(os.path.join(GOLD, "pipe_opts", "cases.json")))
        load, cfg, _, _ = option_case_kwargs(case[5:], meta["cases"][case[5:]], meta["blacklist"])
        d = os.path.join(GOLD, "pipe_opts", case[5:]); vcf_text = open(os.path.join(GOLD, "pipe_opts", "in.vcf")).read(); bams = ["o1.bam", "o2.bam"]
    else:
        d = os.path.join(GOLD, case)
        if case == "c1":
            import hashlib
            from phaser_amd import synth
            g = json.load(open(os.path.join(d, "meta.json")))["gen"]
            v, gs, ge, w = synth.make_variants(g["region"][0], g["region"][1], g["region"][2], g["n_snps"], g["vseed"], n_genes=g["n_genes"])
            vcf_text = "\n".join(synth.vcf_lines([v])) + "\n"; bams = ["c1.bam"]
        else:
            vcf_text = open(os.path.join(d, "in.vcf")).read()
            bams = {"pipe_one": ["a.bam"], "pipe_two": ["t1.bam", "t2.bam"], "pipe_sparse": ["s1.bam", "s2.bam", "s3.bam"], "pipe_indel": ["i.bam"]}.get(case, ["n.bam"])
        if case.startswith("pipe_noisy"):
            cfg["max_block_size"] = json.load(open(os.path.join(d, "meta.json")))["max_block_size"]
        if case == "pipe_indel":
            load["include_indels"] = 1; cfg["include_indels"] = 1
    load = dict(load); cfg = dict(cfg)
    inc = load.pop("include_indels", 0); cfg.pop("include_indels", None)
    vs = vcf.load_variants(vcf_text, include_indels=inc, **load)
    saved = pickle.load(gzip.open(os.path.join(GOLD, "tally", case + ".pkl.gz"), "rb"))

    class _M:
        class ctx:
            lib = None
        device = None
    eng = Engine(vs, bam_display_names(bams), Config(include_indels=inc, py_hash_order=1, **cfg), mapper=_M())
    eng.n_qid.update(saved["n_qid"]); eng.qnames.update(saved["qnames"])
    stub_gpu_stages(eng, saved)

    def kept_lines():
        out = {}
        for c in eng.chrom_list:
            R = saved["tally"][c]
            per = [None] * len(eng.bam_names)
            for b, base, n in R["bam_offsets"]:
                sl = slice(base, base + n)
                keep = R["line_cls"][sl] != 255
                per[b] = (R["line_qid"][sl][keep], R["line_var"][sl][keep], R["line_cls"][sl][keep])
            out[c] = per
        return out
    eng.kept_lines = kept_lines
    out = eng.finish()
    res = {}
    for name in OUTPUTS:
        want = gz(os.path.join(d, "out.%s.txt.gz" % name))
        if out[name] == want:
            res[name] = "raw"
        else:
            a = out[name].split("\n"); b = want.split("\n")
            i = next((k for k in range(min(len(a), len(b))) if a[k] != b[k]), min(len(a), len(b)))
            res[name] = "line %d: got %r want %r" % (i, a[i][:200] if i < len(a) else None, b[i][:200] if i < len(b) else None)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
