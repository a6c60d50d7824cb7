/*
 * oracle/rvm_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of phASER's read->variant allele mapper.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or run it;
 * the product path (phaser_amd/) never does and fails loudly without its HIP library.
 *
 * Parity status: PINNED.  tests/test_oracle_mapper.py checks this file byte-for-byte
 * against call TSVs and micro known-answers that tools/make_golden.py produced by
 * running the reference's own Cython-compiled read_variant_map.py in the build container
 * (tests/golden/kat_micro.json, mapper_small/, c1/calls.tsv.gz, pipe_one/calls.a.chr22.tsv.gz).
 *
 * What it restates (reference = /root/reference/phaser/read_variant_map.py):
 *   do_read_variant_map :3-124   driver: SAM text on stdin x variant table -> one TSV line per hit
 *   split_read          :165-234 baseq masking, CIGAR walk, N-split segments, insertion map
 *   identify_allele     :236-258 slice + insertion splice + 'D' strip, "" / "N" suppressed
 * The streaming variant buffer of :38-49/:106-112 is a sorted merge join whose net effect is
 * stateless per (record, segment, variant) -- SURVEY.md 3.3 -- so this file enumerates the
 * candidate variants of each segment directly.  Input contract (same as the reference's use
 * from phaser.py:1346): one chromosome per run, reads coordinate-sorted, variant table sorted.
 * The text front end (main) also follows the buffer on a stream that is NOT coordinate-sorted, where
 * it is not stateless: the buffer is the index range [b_lo, L) of the table -- L = variants consumed
 * so far (:88-93 skips those behind the record, :106-112 appends up to the segment's end; neither
 * ever rewinds), b_lo = consumed variants pruned because they lay behind some earlier record (:37-50,
 * done for EVERY record, also one the isize filter drops) -- and a record that steps backwards only
 * sees what is still inside it (pinned by tests/golden/mapper_unsorted/).
 *
 * Quirks reproduced on purpose (each has a known-answer fixture):
 *   - insertion keys are read-relative (:220) but looked up segment-relative (:246-251)
 *   - zip(bases, quals) truncates to the shorter string (:179); slices clamp at the end (:200)
 *   - later insertion at the same key overwrites the earlier one (dict, :220)
 *   - 'D' characters (deletion placeholders AND IUPAC D) are stripped after splicing (:254)
 */
#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* No fixed capacities anywhere in this file: the reference keeps CIGAR operations in a list and insertions in a dict, so a record
 * with hundreds of operations or insertions (long, indel-rich reads) must come out the same.  (Until round 3 the array front end cut a
 * CIGAR at 64 operations and a segment kept 64 insertions -- silently; tests/test_gpu_mapper.py::test_many_op_records_vs_oracle found it
 * when the PRODUCT, which has no such cap, disagreed with this file and agreed with the reference.) */
typedef struct {
    int start;          /* genome_start: offset of the segment from POS (:207 first field) */
    char *pseudo;       /* pseudo_read */
    int plen;
    int ins_first, n_ins;       /* the segment's insertions: entries [ins_first, ins_first + n_ins) of the split's insertion arrays */
} segment_t;

typedef struct {
    segment_t *seg;
    int n_seg, cap_seg;
    char *masked;       /* baseq-masked bases, length nb */
    int nb;
    char *pool;         /* storage for pseudo reads */
    size_t pool_cap;
    int *ins_key; const char **ins_ptr; int *ins_len;       /* insertions of all segments of the record (the dict of :217-222, per segment) */
    int ins_n, ins_cap;
} split_t;

static void split_free(split_t *s) { free(s->seg); free(s->masked); free(s->pool); free(s->ins_key); free((void *)s->ins_ptr); free(s->ins_len); memset(s, 0, sizeof *s); }

/* growable (op, length) list of one CIGAR */
typedef struct { char *op; int *len; int n, cap; } ops_t;
static void ops_push(ops_t *o, char c, int len) {
    if (o->n == o->cap) {
        o->cap = o->cap ? 2 * o->cap : 64;
        o->op = (char *)realloc(o->op, (size_t)o->cap); o->len = (int *)realloc(o->len, (size_t)o->cap * sizeof(int));
    }
    o->op[o->n] = c; o->len[o->n++] = len;
}
static void ops_free(ops_t *o) { free(o->op); free(o->len); memset(o, 0, sizeof *o); }

static segment_t *new_segment(split_t *s, int start, char *pseudo) {
    if (s->n_seg == s->cap_seg) {
        s->cap_seg = s->cap_seg ? 2 * s->cap_seg : 8;
        s->seg = (segment_t *)realloc(s->seg, s->cap_seg * sizeof(segment_t));
    }
    segment_t *g = &s->seg[s->n_seg++];
    memset(g, 0, sizeof *g);
    g->start = start; g->pseudo = pseudo; g->ins_first = s->ins_n;
    return g;
}

/* slice [a, a+len) of a string of length n with Python clamping; returns count, sets *from */
static int clamp_slice(int a, int len, int n, int *from) {
    int lo = a < n ? a : n, hi = (a + len) < n ? (a + len) : n;
    if (lo < 0) lo = 0;
    *from = lo;
    return hi > lo ? hi - lo : 0;
}

/* read_variant_map.py:165-234.  ops: n_ops pairs (op char, length). */
static void split_read(split_t *s, const char *seq, int nseq, const char *qual, int nqual, int baseq,
                       const char *op, const int *oplen, int n_ops) {
    int nb = nseq < nqual ? nseq : nqual;
    s->n_seg = 0; s->ins_n = 0;
    s->masked = (char *)realloc(s->masked, nb + 1);
    s->nb = nb;
    for (int i = 0; i < nb; i++) s->masked[i] = ((int)(unsigned char)qual[i] - 33 >= baseq) ? seq[i] : 'N';
    /* worst-case pseudo length: all M bases + all D lengths */
    size_t need = 1;
    for (int i = 0; i < n_ops; i++) need += (size_t)(op[i] == 'D' ? oplen[i] : 0);
    need += (size_t)nb + (size_t)n_ops + 1;
    if (need > s->pool_cap) { s->pool_cap = need * 2; s->pool = (char *)realloc(s->pool, s->pool_cap); }
    char *w = s->pool;
    int read_pos = 0, genome_start = 0, genome_pos = 0;
    segment_t *cur = new_segment(s, 0, w);
    for (int i = 0; i < n_ops; i++) {
        char c = op[i]; int len = oplen[i];
        if (c == 'M' || c == 'X' || c == '=') {
            int from, cnt = clamp_slice(read_pos, len, nb, &from);
            memcpy(w, s->masked + from, cnt); w += cnt; cur->plen += cnt;
            read_pos += len; genome_pos += len;
        } else if (c == 'N') {
            genome_pos += len; genome_start = genome_pos;
            cur = new_segment(s, genome_start, w);
        } else if (c == 'D') {
            memset(w, 'D', len); w += len; cur->plen += len;
            genome_pos += len;
        } else if (c == 'I') {
            int from, cnt = clamp_slice(read_pos, len, nb, &from);
            int key = genome_pos - 1, slot = -1;
            for (int k = cur->ins_first; k < cur->ins_first + cur->n_ins; k++) if (s->ins_key[k] == key) slot = k;   /* dict overwrite */
            if (slot < 0) {
                if (s->ins_n == s->ins_cap) {
                    s->ins_cap = s->ins_cap ? 2 * s->ins_cap : 64;
                    s->ins_key = (int *)realloc(s->ins_key, (size_t)s->ins_cap * sizeof(int));
                    s->ins_ptr = (const char **)realloc((void *)s->ins_ptr, (size_t)s->ins_cap * sizeof(char *));
                    s->ins_len = (int *)realloc(s->ins_len, (size_t)s->ins_cap * sizeof(int));
                }
                slot = s->ins_n++; cur->n_ins++;
            }
            s->ins_key[slot] = key; s->ins_ptr[slot] = s->masked + from; s->ins_len[slot] = cnt;
            read_pos += len;
        } else if (c == 'S') {
            read_pos += len;
        } /* H, P and anything else: no effect (:227-229) */
    }
}

/* read_variant_map.py:236-258; returns allele length (0 = no call), writes into out (cap bytes) */
static int identify_allele(const split_t *s, const segment_t *g, int read_pos, int vpos, int ref_len, char *out, int cap) {
    int rs = vpos - (read_pos + g->start), re = rs + ref_len, n = 0;
    if (rs < 0 || re > g->plen) return 0;
    for (int p = rs; p < re; p++) {
        char c = g->pseudo[p];
        if (c != 'D' && n < cap) out[n++] = c;
        for (int k = g->ins_first; k < g->ins_first + g->n_ins; k++) if (s->ins_key[k] == p)
            for (int j = 0; j < s->ins_len[k]; j++) { char d = s->ins_ptr[k][j]; if (d != 'D' && n < cap) out[n++] = d; }
    }
    if (n == 1 && out[0] == 'N') return 0;
    return n;
}

static long lower_bound_i32(const int32_t *a, long n, long key) {
    long lo = 0, hi = n;
    while (lo < hi) { long m = (lo + hi) >> 1; if (a[m] < key) lo = m + 1; else hi = m; }
    return lo;
}

/* ----------------------------------------------------------------------------------------------
 * Array front end (used by the GPU parity tests and the bench cpu_baseline leg).
 * seq: base codes 0..3 = ACGT, 4 = N, row stride L;  qual: phred, row stride L;  cigar: len<<4|op.
 * Output: calls in mapper order; code 0..3 = single base ACGT, 4 = any other string (text in o_str,
 * 32 bytes per call, NUL padded).  Returns the number of calls (may exceed cap; only cap are written).
 */
static const char OPCH[] = "MIDNSHP=X???????";

long rvm_oracle_map_soa(long n, const int32_t *pos, const int64_t *cigar_off, const uint32_t *cigar,
                        const uint8_t *seq, const uint8_t *qual, int L, int baseq,
                        long nv, const int32_t *vpos, const uint8_t *vreflen,
                        long cap, int32_t *o_read, int32_t *o_var, uint8_t *o_code, char *o_str) {
    split_t s; memset(&s, 0, sizeof s);
    char *sq = (char *)malloc(L + 1), *ql = (char *)malloc(L + 1);
    ops_t o; memset(&o, 0, sizeof o);
    long nc = 0;
    int max_reflen = 1;
    for (long i = 0; i < nv; i++) if (vreflen[i] > max_reflen) max_reflen = vreflen[i];
    for (long r = 0; r < n; r++) {
        for (int j = 0; j < L; j++) { sq[j] = "ACGTN"[seq[r * L + j] > 4 ? 4 : seq[r * L + j]]; ql[j] = (char)(qual[r * L + j] + 33); }
        o.n = 0;
        for (int64_t k = cigar_off[r]; k < cigar_off[r + 1]; k++) { uint32_t c = cigar[k]; ops_push(&o, OPCH[c & 15], (int)(c >> 4)); }
        split_read(&s, sq, L, ql, L, baseq, o.op, o.len, o.n);
        for (int g = 0; g < s.n_seg; g++) {
            const segment_t *sg = &s.seg[g];
            long lo = (long)pos[r] + sg->start, hi = lo + sg->plen;
            for (long v = lower_bound_i32(vpos, nv, lo); v < nv && vpos[v] < hi; v++) {
                char buf[32];
                int len = identify_allele(&s, sg, pos[r], vpos[v], vreflen[v], buf, 31);
                if (!len) continue;
                if (nc < cap) {
                    o_read[nc] = (int32_t)r; o_var[nc] = (int32_t)v;
                    uint8_t code = 4;
                    if (len == 1) { const char *p = strchr("ACGT", buf[0]); if (p && buf[0]) code = (uint8_t)(p - "ACGT"); }
                    o_code[nc] = code;
                    if (o_str) { memset(o_str + nc * 32, 0, 32); memcpy(o_str + nc * 32, buf, len); }
                }
                nc++;
            }
        }
    }
    free(sq); free(ql); split_free(&s); ops_free(&o);
    return nc;
}

/* Known-answer front end: one record given as text, one variant; returns allele per segment joined by '|' */
int rvm_oracle_kat(int pos, const char *seq, const char *qual, const char *cigar, int baseq, int vpos, int ref_len,
                   char *out, int cap) {
    ops_t o; memset(&o, 0, sizeof o); long num = 0;
    if (strcmp(cigar, "*") != 0)
        for (const char *p = cigar; *p; p++) {
            if (isdigit((unsigned char)*p)) num = num * 10 + (*p - '0');
            else { ops_push(&o, *p, (int)num); num = 0; }
        }
    split_t s; memset(&s, 0, sizeof s);
    split_read(&s, seq, (int)strlen(seq), qual, (int)strlen(qual), baseq, o.op, o.len, o.n);
    ops_free(&o);
    int n = 0;
    for (int g = 0; g < s.n_seg; g++) {
        char buf[256];
        int len = identify_allele(&s, &s.seg[g], pos, vpos, ref_len, buf, 255);
        if (g && n < cap - 1) out[n++] = '|';
        for (int j = 0; j < len && n < cap - 1; j++) out[n++] = buf[j];
    }
    out[n] = 0;
    int ns = s.n_seg;
    split_free(&s);
    return ns;
}

/* ----------------------------------------------------------------------------------------------
 * Text front end = call_read_variant_map.py:14-26 + do_read_variant_map :3-124.
 * Usage: rvm_oracle --variant_table T --baseq B --o OUT [--isize_cutoff I] [--splice 1] < in.sam
 */
#ifdef RVM_ORACLE_MAIN
typedef struct { char *chr, *id, *rsid, *gt, *maf; int pos, ref_len; } var_t;

static char *dupfield(const char *s, size_t n) { char *d = (char *)malloc(n + 1); memcpy(d, s, n); d[n] = 0; return d; }

int main(int argc, char **argv) {
    const char *table = NULL, *outp = NULL; int baseq = 10; double isize = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--variant_table")) table = argv[i + 1];
        else if (!strcmp(argv[i], "--baseq")) baseq = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--o")) outp = argv[i + 1];
        else if (!strcmp(argv[i], "--isize_cutoff")) isize = atof(argv[i + 1]);
    }
    if (!table || !outp) { fprintf(stderr, "usage: rvm_oracle --variant_table T --baseq B --o OUT [--isize_cutoff I] < sam\n"); return 2; }
    FILE *ft = fopen(table, "r"); if (!ft) { perror(table); return 2; }
    var_t *vars = NULL; long nv = 0, capv = 0; int32_t *vpos = NULL;
    char *line = NULL; size_t lcap = 0; ssize_t ll;
    while ((ll = getline(&line, &lcap, ft)) > 0) {
        while (ll && (line[ll - 1] == '\n' || line[ll - 1] == '\r')) line[--ll] = 0;
        char *f[8]; int nf = 0; char *p = line;
        while (nf < 8) { f[nf++] = p; char *t = strchr(p, '\t'); if (!t) break; *t = 0; p = t + 1; }
        if (nf < 8) continue;
        if (nv == capv) { capv = capv ? capv * 2 : 1024; vars = (var_t *)realloc(vars, capv * sizeof(var_t)); vpos = (int32_t *)realloc(vpos, capv * sizeof(int32_t)); }
        var_t *v = &vars[nv];
        v->chr = dupfield(f[0], strlen(f[0])); v->pos = atoi(f[1]); v->id = dupfield(f[2], strlen(f[2]));
        v->rsid = dupfield(f[3], strlen(f[3])); v->ref_len = atoi(f[5]); v->gt = dupfield(f[6], strlen(f[6]));
        v->maf = dupfield(f[7], strlen(f[7]));
        vpos[nv++] = v->pos;
    }
    fclose(ft);
    FILE *fo = fopen(outp, "w"); if (!fo) { perror(outp); return 2; }
    split_t s; memset(&s, 0, sizeof s);
    ops_t cig; memset(&cig, 0, sizeof cig);
    char **col = NULL; int ccap = 0;
    char *abuf = (char *)malloc(1 << 16);
    long b_lo = 0, L = 0;       /* the reference's variant buffer = table entries [b_lo, L) */
    while ((ll = getline(&line, &lcap, stdin)) > 0) {
        while (ll && (line[ll - 1] == '\n' || line[ll - 1] == '\r' || line[ll - 1] == ' ' || line[ll - 1] == '\t')) line[--ll] = 0; /* rstrip */
        if (line[0] == '@') continue;
        int nc = 0; char *p = line;
        for (;;) {
            if (nc == ccap) { ccap = ccap ? 2 * ccap : 32; col = (char **)realloc(col, ccap * sizeof(char *)); }
            col[nc++] = p; char *t = strchr(p, '\t'); if (!t) break; *t = 0; p = t + 1;
        }
        if (nc < 11) continue;
        int read_pos = atoi(col[3]);
        long tl = labs(atol(col[8]));
        {   /* :37-50: consumed variants behind this record leave the buffer, whatever happens to the record afterwards */
            long lb = lower_bound_i32(vpos, nv, read_pos);
            long cut = lb < L ? lb : L;
            if (cut > b_lo) b_lo = cut;
            if (!(isize == 0 || (double)tl <= isize)) continue;
            if (L < lb) { L = lb; b_lo = lb; }      /* :88-93: variants behind the record are skipped for good */
        }
        const char *as_str = ""; char as_norm[32];
        for (int i = 11; i < nc; i++) if (!strncmp(col[i], "AS:", 3)) {      /* last AS: tag wins (:56-59) */
            const char *c2 = strchr(col[i] + 3, ':');
            if (c2) { snprintf(as_norm, sizeof as_norm, "%ld", atol(c2 + 1)); as_str = as_norm; }
        }
        long num = 0;
        cig.n = 0;
        for (const char *c = col[5]; *c; c++) {
            if (*c >= '0' && *c <= '9') num = num * 10 + (*c - '0');
            else { ops_push(&cig, *c, (int)num); num = 0; }
        }
        split_read(&s, col[9], (int)strlen(col[9]), col[10], (int)strlen(col[10]), baseq, cig.op, cig.len, cig.n);
        for (int g = 0; g < s.n_seg; g++) {
            const segment_t *sg = &s.seg[g];
            long lo = (long)read_pos + sg->start, hi = lo + sg->plen;
            while (L < nv && vpos[L] <= hi) L++;      /* :106-112: the buffer grows up to the segment's end (inclusive) */
            long v0 = lower_bound_i32(vpos, nv, lo);
            if (v0 < b_lo) v0 = b_lo;                 /* only on a stream out of coordinate order */
            for (long v = v0; v < nv && vpos[v] < hi; v++) {
                int len = identify_allele(&s, sg, read_pos, vars[v].pos, vars[v].ref_len, abuf, (1 << 16) - 1);
                if (!len) continue;
                abuf[len] = 0;
                fprintf(fo, "%s\t%s\t%s\t%s\t%s\t%s\t%s\n", col[0], vars[v].id, vars[v].rsid, abuf, as_str, vars[v].gt, vars[v].maf);
            }
        }
    }
    fclose(fo);
    return 0;
}
#endif
