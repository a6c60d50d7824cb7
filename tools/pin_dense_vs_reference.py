#!/usr/bin/env python3
"""Build-container pin (needs /root/reference): ONE dense-variant sample (het SNPs every few bases inside two genes, two BAMs with shared QNAMEs) through the
reference's own process_vcf and through oracle/phasing_oracle.py, canonical comparison of the five files.  The reference needs minutes on these shapes
(and does not finish on 1,000-base reads), which is why they are not part of tools/fuzz_oracle_phasing.py's default draw.
usage: PYTHONHASHSEED=0 tools/pin_dense_vs_reference.py n_snps pairs L max_block_size err seed"""
import os, sys, tempfile, time, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools")); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "oracle")); sys.path.insert(0, REPO)
import make_golden as mg
import phasing_oracle as po
from helpers import OUTPUTS, canonical
from phaser_amd import synth
phaser, rvm = mg.build_reference()
n_snps, pairs, L, mbs, err, seed = [int(x) if i != 4 else float(x) for i, x in enumerate(sys.argv[1:7])]
contigs=[("chr7",159345973)]
v,gs,ge,w=synth.make_variants("chr7",1,600_000,n_snps,seed,n_genes=2)
names=["d1.bam","d2.bam"]; sams={b:{} for b in names}
for bi,b in enumerate(names):
    rb=synth.make_reads(v,gs,ge,w,pairs,seed+bi+100,L=L,qname_prefix="q",err_rate=err)
    rf=rb.select(synth.samtools_keep(rb,255))
    sams[b]["chr7"]="\n".join(synth.sam_lines(rf,contigs))+"\n"
vcf_text="\n".join(synth.vcf_lines([v]))+"\n"
t0=time.time()
with tempfile.TemporaryDirectory() as tmp:
    want, calls = mg.run_pipeline(phaser, rvm, vcf_text, sams, tmp, max_block_size=mbs)
t1=time.time()
ph=po.Phaser(po.bam_display_names(names), max_block_size=mbs)
pool,_,_=po.load_vcf(vcf_text)
for b in names: ph.add_bam([calls[(b,c)] for c in pool])
got=ph.finish()
bad=[n for n in OUTPUTS if canonical(n,got[n])!=canonical(n,want[n])]
lines=sum(calls[(b,"chr7")].count("\n") for b in names)
print("snps %d pairs %d L %d mbs %d err %.3f: call lines %d, hap rows %d, reference %.1fs oracle %.1fs -> %s" % (n_snps,pairs,L,mbs,err,lines,want["haplotypes"].count("\n"),t1-t0,time.time()-t1,"OK" if not bad else "DIFF "+str(bad)), flush=True)
