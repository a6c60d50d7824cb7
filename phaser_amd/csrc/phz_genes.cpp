// Host side of phaser_gene_ae (phaser_gene_ae/phaser_gene_ae.py:57-80): multi-threaded parse of a
// haplotypic_counts.txt into the arrays the gene-level kernels read.  Per row the fields the script uses; per variant its
// position and id text; per (row, haplotype) the read-label sequence of aReads / bReads turned into
//   lab_pos[p]   position of the variant the p-th label belongs to
//   lab_prev[p]  index (within the same row and haplotype) of the previous occurrence of the same label, -1 if none
// which is all a distinct-count over any subset of the row's variants needs (phz_genes.hip).  Labels are compared as text
// by the reference (set of strings); here they are keyed by their text too, so "7" and "07" stay different reads.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "phz.h"
#include "phz_text.h"

struct phz_hc {
    std::vector<int32_t> contig, start, stop, a_count, b_count, total, phase, bam;
    std::vector<double> gw_stat, maf;
    std::vector<int64_t> var_off;                  // [n_rows+1]
    std::vector<int32_t> var_pos;                  // per variant
    std::vector<int64_t> var_id_off; std::vector<int32_t> var_id_len;     // id text inside the input buffer
    std::vector<int64_t> lab_off_a, lab_off_b;     // [n_rows+1] label ranges per row (empty for single-variant rows)
    std::vector<int32_t> lab_pos_a, lab_prev_a, lab_pos_b, lab_prev_b;
    std::vector<std::string> contig_names, bam_names;
    std::string names_blob; std::vector<int64_t> names_off;     // contig names then bam names, NUL separated
    int has_maf = 0;
    int status = 0;
    std::string error;
};

namespace {

struct Chunk {
    std::vector<int32_t> contig, start, stop, a_count, b_count, total, phase, bam;
    std::vector<double> gw_stat, maf;
    std::vector<int32_t> nvar;
    std::vector<int32_t> var_pos; std::vector<int64_t> var_id_off; std::vector<int32_t> var_id_len;
    std::vector<int64_t> nlab_a, nlab_b;
    std::vector<int32_t> lab_pos_a, lab_prev_a, lab_pos_b, lab_prev_b;
    std::vector<std::string> contig_names, bam_names;
    std::unordered_map<std::string, int> cmap, bmap;
    int status = 0; std::string error;
};

int intern(std::unordered_map<std::string, int> &m, std::vector<std::string> &names, std::string_view s) {
    auto it = m.find(std::string(s));
    if (it != m.end()) return it->second;
    const int id = (int)names.size();
    names.emplace_back(s); m.emplace(std::string(s), id);
    return id;
}

bool to_ll(std::string_view s, long long *out) {
    if (s.empty()) return false;
    char *e = nullptr;
    std::string tmp(s);
    *out = strtoll(tmp.c_str(), &e, 10);
    if (*e == '.') { double d = strtod(tmp.c_str(), &e); *out = (long long)d; }      // pandas would hand int(1.0) the same way
    return *e == 0;
}

// one haplotype's label text "l,l,l;l,l;..." -> per-label variant position + previous occurrence of the same label text
bool parse_labels(std::string_view f, const int32_t *vpos, int nvar, std::vector<int32_t> &lab_pos, std::vector<int32_t> &lab_prev,
                  int64_t *count, std::unordered_map<std::string_view, int32_t> &last) {
    last.clear();
    int v = 0; int64_t n = 0;
    size_t i = 0;
    const size_t base = lab_pos.size();
    while (true) {
        size_t j = i;
        while (j < f.size() && f[j] != ',' && f[j] != ';') j++;
        if (j > i) {                                   // blank labels are removed by the reference (:206-207)
            if (v >= nvar) return false;
            std::string_view lab = f.substr(i, j - i);
            auto it = last.find(lab);
            const int32_t here = (int32_t)(lab_pos.size() - base);
            if (it == last.end()) { lab_prev.push_back(-1); last.emplace(lab, here); }
            else { lab_prev.push_back(it->second); it->second = here; }
            lab_pos.push_back(vpos[v]);
            n++;
        }
        if (j >= f.size()) break;
        if (f[j] == ';') v++;
        i = j + 1;
    }
    *count = n;
    return true;
}

struct Cols { int contig = -1, start = -1, stop = -1, variants = -1, a = -1, b = -1, total = -1, phase = -1, gw = -1, maf = -1, bam = -1, ar = -1, br = -1, n = 0; };

void parse_lines(const char *text, const std::vector<int64_t> &ls, size_t lo, size_t hi, const Cols &K, std::string_view sep, Chunk &c) {
    std::vector<std::string_view> f;
    std::unordered_map<std::string_view, int32_t> last;
    std::vector<int32_t> vpos;
    for (size_t li = lo; li < hi; li++) {
        const char *p = text + ls[li]; const char *e = text + ls[li + 1] - 1;       // line without its '\n'
        if (e > p && e[-1] == '\r') e--;
        if (e <= p) continue;
        f.clear();
        const char *q = p;
        while (true) {
            const char *t = (const char *)memchr(q, '\t', (size_t)(e - q));
            if (!t) { f.emplace_back(q, (size_t)(e - q)); break; }
            f.emplace_back(q, (size_t)(t - q)); q = t + 1;
        }
        while ((int)f.size() < K.n) f.emplace_back();
        long long st, sp, ac, bc, tc;
        if (!to_ll(f[K.start], &st) || !to_ll(f[K.stop], &sp) || !to_ll(f[K.a], &ac) || !to_ll(f[K.b], &bc) || !to_ll(f[K.total], &tc)) {
            c.status = PHZ_E_ARG; c.error = "haplotypic_counts: non-numeric start/stop/count field"; return;
        }
        c.contig.push_back(intern(c.cmap, c.contig_names, f[K.contig]));
        c.bam.push_back(intern(c.bmap, c.bam_names, f[K.bam]));
        c.start.push_back((int32_t)st); c.stop.push_back((int32_t)sp); c.a_count.push_back((int32_t)ac); c.b_count.push_back((int32_t)bc);
        c.total.push_back((int32_t)tc);
        const std::string_view ph = f[K.phase];
        c.phase.push_back(ph == "0/1" ? 0 : (ph == "0|1" ? 1 : (ph == "1|0" ? 2 : 3)));
        c.gw_stat.push_back(strtod(std::string(f[K.gw]).c_str(), nullptr));
        c.maf.push_back(K.maf >= 0 ? strtod(std::string(f[K.maf]).c_str(), nullptr) : 0.0);
        // variants: id text + position = second separator-delimited field (:183-187)
        const std::string_view vs = f[K.variants];
        vpos.clear();
        size_t i = 0; int nv = 0;
        while (true) {
            size_t j = vs.find(',', i);
            if (j == std::string_view::npos) j = vs.size();
            const std::string_view id = vs.substr(i, j - i);
            if (nv == 0) {
                size_t cnt = 0, pos = 0;
                while (!sep.empty() && (pos = id.find(sep, pos)) != std::string_view::npos) { cnt++; pos += sep.size(); }
                if (cnt < 3) { c.status = PHZ_E_ARG; c.error = "ERROR - ID separator not found in variant ID, please ensure that --id_separator is set correctly."; return; }
            }
            const size_t s1 = id.find(sep);
            long long pv = 0;
            bool ok = s1 != std::string_view::npos;
            if (ok) {
                const size_t s2 = id.find(sep, s1 + sep.size());
                ok = to_ll(id.substr(s1 + sep.size(), (s2 == std::string_view::npos ? id.size() : s2) - s1 - sep.size()), &pv);
            }
            if (!ok) { c.status = PHZ_E_ARG; c.error = "haplotypic_counts: variant id without a numeric position"; return; }
            vpos.push_back((int32_t)pv);
            c.var_pos.push_back((int32_t)pv); c.var_id_off.push_back((int64_t)(id.data() - text)); c.var_id_len.push_back((int32_t)id.size());
            nv++;
            if (j >= vs.size()) break;
            i = j + 1;
        }
        c.nvar.push_back(nv);
        int64_t na = 0, nb = 0;
        if (nv > 1) {
            if (!parse_labels(f[K.ar], vpos.data(), nv, c.lab_pos_a, c.lab_prev_a, &na, last) ||
                !parse_labels(f[K.br], vpos.data(), nv, c.lab_pos_b, c.lab_prev_b, &nb, last)) {
                c.status = PHZ_E_ARG; c.error = "haplotypic_counts: more read-label groups than variants in a row"; return;
            }
        }
        c.nlab_a.push_back(na); c.nlab_b.push_back(nb);
    }
}

template <class T> void append(std::vector<T> &dst, const std::vector<T> &src) { dst.insert(dst.end(), src.begin(), src.end()); }

}  // namespace

extern "C" int phz_hc_parse(const char *text, int64_t len, const char *id_separator, int threads, phz_hc **out) {
    if (!text || len < 0 || !id_separator || !out) return PHZ_E_ARG;
    phz_hc *h = new phz_hc();
    *out = h;
    std::vector<int64_t> ls(1, 0);
    for (const char *p = text, *e = text + len; p < e;) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
        if (!nl) { ls.push_back(len + 1); break; }          // last line without newline: pretend one follows
        ls.push_back((int64_t)(nl - text) + 1); p = nl + 1;
    }
    if (ls.size() < 2) { h->status = PHZ_E_ARG; h->error = "empty haplotypic_counts file"; return h->status; }
    // header (:78-80)
    Cols K;
    {
        std::string_view hd(text, (size_t)(ls[1] - 1 - ls[0]));
        if (!hd.empty() && hd.back() == '\r') hd.remove_suffix(1);
        size_t i = 0; int k = 0;
        while (true) {
            size_t j = hd.find('\t', i);
            if (j == std::string_view::npos) j = hd.size();
            const std::string_view name = hd.substr(i, j - i);
            if (name == "contig") K.contig = k; else if (name == "start") K.start = k; else if (name == "stop") K.stop = k;
            else if (name == "variants") K.variants = k; else if (name == "aCount") K.a = k; else if (name == "bCount") K.b = k;
            else if (name == "totalCount") K.total = k; else if (name == "blockGWPhase") K.phase = k; else if (name == "gwStat") K.gw = k;
            else if (name == "max_haplo_maf") K.maf = k; else if (name == "bam") K.bam = k; else if (name == "aReads") K.ar = k;
            else if (name == "bReads") K.br = k;
            k++;
            if (j >= hd.size()) break;
            i = j + 1;
        }
        K.n = k;
    }
    if (K.bam < 0) { h->status = PHZ_E_UNSUPPORTED; h->error = "ERROR - this version of phaser_gene_ae is only compatible with results from phASER v1.0.0+"; return h->status; }
    if (K.contig < 0 || K.start < 0 || K.stop < 0 || K.variants < 0 || K.a < 0 || K.b < 0 || K.total < 0 || K.phase < 0 || K.gw < 0 || K.ar < 0 || K.br < 0) {
        h->status = PHZ_E_ARG; h->error = "haplotypic_counts: missing column"; return h->status;
    }
    h->has_maf = K.maf >= 0;
    const size_t nlines = ls.size() - 1;
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, threads), (nlines + 4095) / 4096));
    const size_t nchunks = (size_t)nt * 4;
    std::vector<Chunk> ch(nchunks);
    std::atomic<size_t> next(0);
    const std::string_view sep(id_separator);
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= nchunks) break;
            const size_t lo = 1 + (nlines - 1) * i / nchunks, hi = 1 + (nlines - 1) * (i + 1) / nchunks;
            parse_lines(text, ls, lo, hi, K, sep, ch[i]);
        }
    };
    if (nt == 1) work();
    else { std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(work); for (auto &t : th) t.join(); }
    std::unordered_map<std::string, int> cmap, bmap;
    h->var_off.push_back(0); h->lab_off_a.push_back(0); h->lab_off_b.push_back(0);
    for (auto &c : ch) {
        if (c.status) { h->status = c.status; h->error = c.error; return h->status; }
        std::vector<int> cre(c.contig_names.size()), bre(c.bam_names.size());
        for (size_t i = 0; i < c.contig_names.size(); i++) cre[i] = intern(cmap, h->contig_names, c.contig_names[i]);
        for (size_t i = 0; i < c.bam_names.size(); i++) bre[i] = intern(bmap, h->bam_names, c.bam_names[i]);
        for (auto x : c.contig) h->contig.push_back(cre[x]);
        for (auto x : c.bam) h->bam.push_back(bre[x]);
        append(h->start, c.start); append(h->stop, c.stop); append(h->a_count, c.a_count); append(h->b_count, c.b_count); append(h->total, c.total);
        append(h->phase, c.phase); append(h->gw_stat, c.gw_stat); append(h->maf, c.maf);
        for (auto n : c.nvar) h->var_off.push_back(h->var_off.back() + n);
        append(h->var_pos, c.var_pos); append(h->var_id_off, c.var_id_off); append(h->var_id_len, c.var_id_len);
        for (auto n : c.nlab_a) h->lab_off_a.push_back(h->lab_off_a.back() + n);
        for (auto n : c.nlab_b) h->lab_off_b.push_back(h->lab_off_b.back() + n);
        append(h->lab_pos_a, c.lab_pos_a); append(h->lab_prev_a, c.lab_prev_a); append(h->lab_pos_b, c.lab_pos_b); append(h->lab_prev_b, c.lab_prev_b);
        c = Chunk();
    }
    for (auto &s : h->contig_names) { h->names_off.push_back((int64_t)h->names_blob.size()); h->names_blob += s; h->names_blob += '\0'; }
    for (auto &s : h->bam_names) { h->names_off.push_back((int64_t)h->names_blob.size()); h->names_blob += s; h->names_blob += '\0'; }
    h->names_off.push_back((int64_t)h->names_blob.size());
    return PHZ_OK;
}

extern "C" int phz_hc_view(const phz_hc *h, phz_hc_arrays *v) {
    if (!h || !v) return PHZ_E_ARG;
    memset(v, 0, sizeof(*v));
    v->n_rows = (int64_t)h->start.size(); v->n_vars = (int64_t)h->var_pos.size();
    v->n_lab_a = (int64_t)h->lab_pos_a.size(); v->n_lab_b = (int64_t)h->lab_pos_b.size();
    v->n_contigs = (int32_t)h->contig_names.size(); v->n_bams = (int32_t)h->bam_names.size(); v->has_maf = h->has_maf;
    v->contig = h->contig.data(); v->start = h->start.data(); v->stop = h->stop.data(); v->a_count = h->a_count.data(); v->b_count = h->b_count.data();
    v->total = h->total.data(); v->phase = h->phase.data(); v->bam = h->bam.data(); v->gw_stat = h->gw_stat.data(); v->maf = h->maf.data();
    v->var_off = h->var_off.data(); v->var_pos = h->var_pos.data(); v->var_id_off = h->var_id_off.data(); v->var_id_len = h->var_id_len.data();
    v->lab_off_a = h->lab_off_a.data(); v->lab_off_b = h->lab_off_b.data();
    v->lab_pos_a = h->lab_pos_a.data(); v->lab_prev_a = h->lab_prev_a.data(); v->lab_pos_b = h->lab_pos_b.data(); v->lab_prev_b = h->lab_prev_b.data();
    v->names = h->names_blob.data(); v->names_off = h->names_off.data();
    return PHZ_OK;
}

extern "C" const char *phz_hc_error(const phz_hc *h) { return h ? h->error.c_str() : ""; }

extern "C" void phz_hc_free(phz_hc *h) { delete h; }

// Output rows of phaser_gene_ae (:147-163) for every BAM (in the order given) and feature: phased counts when they are at
// least as large as the best unphased block, else that block; log2(a/b) the way the script computes it
// (zero_divide / zero_log with math.log(x, 2) = log(x) / log(2)), numbers printed as Python's str() prints them.
extern "C" int phz_gene_rows(const phz_gene_rows_in *in, char **out, int64_t *out_len) {
    if (!in || !out || !out_len) return PHZ_E_ARG;
    const phz_gene_rows_in &I = *in;
    using phztext::put_int; using phztext::put_pyfloat;
    const std::vector<uint32_t> chr_off = phztext::pool_offsets(I.feat_chr, I.feat_chr_len), name_off = phztext::pool_offsets(I.feat_name, I.feat_name_len),
                                bam_off = phztext::pool_offsets(I.bam_names, I.bam_names_len);
    if ((int64_t)chr_off.size() != I.n_features + 1 || (int64_t)name_off.size() != I.n_features + 1) return PHZ_E_ARG;
    auto at = [](const char *b, const std::vector<uint32_t> &off, int64_t i) { return std::string_view(b + off[(size_t)i], off[(size_t)i + 1] - off[(size_t)i] - 1); };
    const int threads = std::max(1, I.threads);
    const int64_t nf = I.n_features;
    const int64_t total = (int64_t)I.n_bam_order * nf;
    const int64_t step = 4096;
    const int64_t nchunks = (total + step - 1) / step;
    std::vector<std::string> parts((size_t)nchunks);
    std::atomic<int64_t> next(0);
    auto work = [&]() {
        for (;;) {
            const int64_t c = next.fetch_add(1);
            if (c >= nchunks) break;
            std::string &o = parts[(size_t)c];
            for (int64_t t = c * step; t < std::min(total, (c + 1) * step); t++) {
                const int b = I.bam_order[t / nf];
                const int64_t fi = t % nf, kx = (int64_t)b * nf + fi;
                const long long a = I.A[kx], bb = I.B[kx], ua = I.UA[kx], ub = I.UB[kx];
                const bool phased = a + bb >= ua + ub;
                const long long x = phased ? a : ua, y = phased ? bb : ub, tc = x + y;
                if (tc < I.min_cov) continue;
                const int64_t *ids = phased ? I.pv_sorted + I.pv_lo[kx] : I.u_var + I.best_lo[kx];
                const int64_t nids = phased ? I.pv_hi[kx] - I.pv_lo[kx] : I.best_hi[kx] - I.best_lo[kx];
                o.append(at(I.feat_chr, chr_off, fi)); o += '\t'; put_int(o, I.feat_start[fi]); o += '\t'; put_int(o, I.feat_stop[fi]); o += '\t';
                o.append(at(I.feat_name, name_off, fi)); o += '\t'; put_int(o, x); o += '\t'; put_int(o, y); o += '\t'; put_int(o, tc); o += '\t';
                const double ratio = y == 0 ? INFINITY : (double)x / (double)y;
                const double l2 = ratio == 0 ? -INFINITY : log(ratio) / log(2.0);
                put_pyfloat(o, l2); o += '\t'; put_int(o, nids); o += '\t';
                for (int64_t k = 0; k < nids; k++) { if (k) o += ','; o.append(I.text + I.var_id_off[ids[k]], (size_t)I.var_id_len[ids[k]]); }
                o += phased ? "\t1\t" : "\t0\t";
                o.append(at(I.bam_names, bam_off, b)); o += '\n';
            }
        }
    };
    const int nt = (int)std::min<int64_t>(threads, std::max<int64_t>(1, nchunks));
    if (nt <= 1) work();
    else { std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(work); for (auto &t : th) t.join(); }
    size_t sz = 0;
    for (auto &p2 : parts) sz += p2.size();
    char *buf = (char *)malloc(sz + 1);
    if (!buf) return PHZ_E_NOMEM;
    size_t w = 0;
    for (auto &p2 : parts) { memcpy(buf + w, p2.data(), p2.size()); w += p2.size(); }
    buf[w] = 0;
    *out = buf; *out_len = (int64_t)w;
    return PHZ_OK;
}
