// Native het-variant loader (host side of SURVEY.md 8(a) M0): VCF text -> per-chromosome variant tables, multi-threaded.
// Restates the filter of phaser/phaser.py:396-433 (GT present, no '.', more than one distinct allele character, FILTER holds
// PASS unless --pass_only 0), the table of generate_mapping_table :1355-1413 (unique id, SNP-only unless --include_indels,
// maf from the INFO AF field only with --gw_phase_method 1) and the per-variant fields generate_variant_dict :1418-1462
// derives from a table row (individual's alleles in allele-index order, alleles in GT order when phased, maf number).
// Strings are handed back as separator-joined pools (one '\n' after every item), numbers as arrays.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "phz.h"
#include "phz_text.h"

namespace {

struct Cols {
    std::vector<int32_t> pos;
    std::vector<uint8_t> ref_len, a0, a1, is_ref, black;      // is_ref: 2 per variant; black: --haplo_count_blacklist hit
    std::vector<int8_t> phase_idx;                    // 2 per variant
    std::vector<double> maf;
    std::string uid, rsid_field, rsid, ref, all_alleles, alleles, phase, gt, maf_text, maf_str, allele2;
    int64_t n = 0;
    void append(const Cols &o) {
        pos.insert(pos.end(), o.pos.begin(), o.pos.end()); ref_len.insert(ref_len.end(), o.ref_len.begin(), o.ref_len.end());
        a0.insert(a0.end(), o.a0.begin(), o.a0.end()); a1.insert(a1.end(), o.a1.begin(), o.a1.end());
        black.insert(black.end(), o.black.begin(), o.black.end());
        is_ref.insert(is_ref.end(), o.is_ref.begin(), o.is_ref.end()); phase_idx.insert(phase_idx.end(), o.phase_idx.begin(), o.phase_idx.end());
        maf.insert(maf.end(), o.maf.begin(), o.maf.end());
        uid += o.uid; rsid_field += o.rsid_field; rsid += o.rsid; ref += o.ref; all_alleles += o.all_alleles; alleles += o.alleles;
        phase += o.phase; gt += o.gt; maf_text += o.maf_text; maf_str += o.maf_str; allele2 += o.allele2;
        n += o.n;
    }
};

struct Chunk {
    std::vector<std::string> names;                 // chromosomes in first-appearance order within the chunk
    std::vector<Cols> cols;
    std::unordered_map<std::string, int> idx;
    int64_t filter_count = 0, unphased = 0, excluded = 0;
    int status = 0; std::string error;
};

using phztext::put_pyfloat;
using phztext::split;

// BED intervals of one file: per chromosome sorted by start and merged, queried by binary search
struct BedIndex {
    std::unordered_map<std::string, std::vector<std::pair<int64_t, int64_t>>> by_chrom;
    bool empty() const { return by_chrom.empty(); }
    void build(int64_t n, const char *const *chrom, const int64_t *start, const int64_t *end) {
        for (int64_t i = 0; i < n; i++) if (end[i] > start[i]) by_chrom[chrom[i]].emplace_back(start[i], end[i]);
        for (auto &kv : by_chrom) {
            auto &v = kv.second;
            std::sort(v.begin(), v.end());
            size_t w = 0;
            for (size_t i = 0; i < v.size(); i++) {
                if (w && v[i].first <= v[w - 1].second) v[w - 1].second = std::max(v[w - 1].second, v[i].second);
                else v[w++] = v[i];
            }
            v.resize(w);
        }
    }
    // does [s, e) share at least one base with an interval of `chrom`?
    bool hit(std::string_view chrom, int64_t s, int64_t e) const {
        auto it = by_chrom.find(std::string(chrom));
        if (it == by_chrom.end()) return false;
        const auto &v = it->second;
        // first interval whose end is beyond s (merged intervals: ends are ascending too)
        size_t lo = 0, hi = v.size();
        while (lo < hi) { const size_t m = (lo + hi) >> 1; if (v[m].second <= s) lo = m + 1; else hi = m; }
        return lo < v.size() && v[lo].first < e;
    }
};

// Python float(): accepts surrounding whitespace, inf/nan spellings; here: strtod over the whole token
bool py_float(std::string_view s, double *out) {
    std::string t(s);
    size_t a = 0, b = t.size();
    while (a < b && isspace((unsigned char)t[a])) a++;
    while (b > a && isspace((unsigned char)t[b - 1])) b--;
    if (a == b) return false;
    t = t.substr(a, b - a);
    char *e = nullptr;
    *out = strtod(t.c_str(), &e);
    return *e == 0;
}

inline uint8_t base_code(std::string_view s) {
    if (s.size() != 1) return 255;
    switch (s[0]) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 255; }
}

// the lines that START in [b0, b1) (both line starts, or the end of the text): every worker finds its own line ends -- a line index over the whole text
// (75 MB of memchr and a 12 MB vector on one thread) was a tenth of the call
void parse_lines(const char *text, int64_t len, int64_t b0, int64_t b1, const phz_vcf_opts &O, const BedIndex &drop,
                 const BedIndex &mark, Chunk &c) {
    std::vector<std::string_view> f, fmt, sf, alts, every, info, ind, ph, afs_s;
    std::vector<char> g;
    std::vector<double> afs;
    const std::string_view coi(O.chrom_of_interest ? O.chrom_of_interest : ""), prefix(O.chr_prefix ? O.chr_prefix : ""),
        sep(O.id_separator ? O.id_separator : "_"), af_field(O.gw_af_field ? O.gw_af_field : "AF");
    for (int64_t at = b0; at < b1;) {
        const char *p = text + at;
        const char *e = (const char *)memchr(p, '\n', (size_t)(len - at));
        if (!e) e = text + len;
        at = (int64_t)(e - text) + 1;
        if (e <= p || *p == '#') continue;
        split(std::string_view(p, (size_t)(e - p)), '\t', f);
        if (O.grep_hom) {      // cut -f 1-9,S | grep -v '0|0\|1|1' (phaser.py:220-225): the pattern anywhere in those columns drops the line
            bool hom = false;
            for (size_t k = 0; k < f.size(); k++)
                if (k < 9 || (int)k == O.sample_column)
                    if (f[k].find("0|0") != std::string_view::npos || f[k].find("1|1") != std::string_view::npos) hom = true;
            if (hom) continue;
        }
        const std::string_view chrom0 = f[0];
        // --chr: the reference reads `tabix -h vcf chr:` (:206-208), so records of other contigs never reach any check
        if (!coi.empty() && coi != chrom0) continue;
        if (!drop.empty() || !mark.empty()) {
            if (f.size() < 4) { c.status = PHZ_E_ARG; c.error = "VCF line with too few columns"; return; }
        }
        long long pos_bed = 0;
        if (!drop.empty() || !mark.empty()) {
            std::string t(f[1]); char *e2 = nullptr; pos_bed = strtoll(t.c_str(), &e2, 10);
            if (t.empty() || *e2) { c.status = PHZ_E_ARG; c.error = "VCF POS is not an integer"; return; }
            const int64_t rl = (int64_t)std::max<size_t>(1, f[3].size());
            if (!drop.empty() && drop.hit(chrom0, pos_bed - 1, pos_bed - 1 + rl)) continue;      // bedtools intersect -v (:220)
        }
        for (int b = 0; b < O.n_contig_ban; b++)
            if (chrom0.find(O.contig_ban[b]) != std::string_view::npos) {
                c.status = PHZ_E_ARG;
                c.error = std::string("     FATAL ERROR: Character '") + O.contig_ban[b] + "' must not be present in contig name. Please change id separtor "
                          "using --id_separator to a character not found in the contig names and try again.";
                return;
            }
        std::string chrom(prefix); chrom += chrom0;
        auto it = c.idx.find(chrom);
        int ci;
        if (it == c.idx.end()) { ci = (int)c.names.size(); c.idx.emplace(chrom, ci); c.names.push_back(chrom); c.cols.emplace_back(); }
        else ci = it->second;
        if ((int)f.size() <= 8 || (int)f.size() <= O.sample_column) { c.status = PHZ_E_ARG; c.error = "VCF line with too few columns"; return; }
        split(f[8], ':', fmt);
        int gti = -1;
        for (size_t k = 0; k < fmt.size(); k++) if (fmt[k] == "GT") { gti = (int)k; break; }
        if (gti < 0) continue;
        split(f[(size_t)O.sample_column], ':', sf);
        if (gti >= (int)sf.size()) { c.status = PHZ_E_ARG; c.error = "VCF sample column has fewer fields than FORMAT"; return; }
        const std::string_view geno = sf[(size_t)gti];
        g.assign(geno.begin(), geno.end());
        if (std::find(g.begin(), g.end(), '.') != g.end()) continue;
        bool phased = false, is_unphased = false;
        { auto q = std::find(g.begin(), g.end(), '|'); if (q != g.end()) { phased = true; g.erase(q); } }       // list.remove: first one only
        { auto q = std::find(g.begin(), g.end(), '/'); if (q != g.end()) { is_unphased = true; g.erase(q); } }
        {
            bool distinct = false;
            for (size_t k = 1; k < g.size(); k++) if (g[k] != g[0]) distinct = true;
            if (!distinct) continue;
        }
        if (O.pass_only != 0) {
            split(f[6], ';', info);
            bool pass = false;
            for (auto &x : info) if (x == "PASS") pass = true;
            if (!pass) { c.filter_count++; continue; }
        }
        c.unphased += is_unphased ? 1 : 0;
        split(f[4], ',', alts);
        every.clear(); every.push_back(f[3]);
        for (auto &x : alts) every.push_back(x);
        size_t maxlen = 0;
        for (auto &x : every) maxlen = std::max(maxlen, x.size());
        if (!(maxlen == 1 || O.include_indels == 1)) { c.excluded++; continue; }
        Cols &K = c.cols[(size_t)ci];
        // maf (:1381-1396)
        bool have_maf = false; double maf = 0;
        if (O.gw_phase_method == 1) {
            split(f[7], ';', info);
            std::string_view val; bool found = false;
            for (auto &item : info) {
                const size_t eq = item.find('=');
                if (eq == std::string_view::npos) continue;
                if (item.substr(0, eq) == af_field) {              // later duplicates overwrite earlier ones (dict)
                    const size_t eq2 = item.find('=', eq + 1);
                    val = item.substr(eq + 1, (eq2 == std::string_view::npos ? item.size() : eq2) - eq - 1);
                    found = true;
                }
            }
            if (found) {
                split(val, ',', afs_s);
                afs.clear();
                for (auto &x : afs_s) { double d; if (!py_float(x, &d)) { c.status = PHZ_E_ARG; c.error = "could not convert AF value to float"; return; } afs.push_back(d); }
                if (afs.size() == alts.size()) {
                    bool any = false; double best = 0;
                    for (char ch : g) {
                        if (ch < '0' || ch > '9') { c.status = PHZ_E_ARG; c.error = "genotype character is not an allele index"; return; }
                        const int ai = ch - '0';
                        if (ai == 0) continue;
                        if (ai - 1 >= (int)afs.size()) { c.status = PHZ_E_ARG; c.error = "genotype allele index beyond ALT"; return; }
                        const double a = afs[(size_t)ai - 1], m = a < 1 - a ? a : 1 - a;      // min(af, 1 - af), first argument wins ties
                        if (!any || m < best) { best = m; any = true; }
                    }
                    if (any) { have_maf = true; maf = best; }
                }
            }
        }
        // the individual's alleles in allele-index order (:1433-1435), alleles in GT order when phased (:1437-1443)
        ind.clear();
        for (size_t i = 0; i < every.size() && i < 10; i++)
            if (std::find(g.begin(), g.end(), (char)('0' + i)) != g.end()) ind.push_back(every[i]);
        ph.clear();
        if (phased) {
            for (char ch : g) {
                if (ch < '0' || ch > '9' || (size_t)(ch - '0') >= every.size()) { c.status = PHZ_E_ARG; c.error = "genotype allele index beyond ALT"; return; }
                ph.push_back(every[(size_t)(ch - '0')]);
            }
        } else { ph.push_back("-"); ph.push_back("-"); }
        long long posv = 0;
        { std::string t(f[1]); char *e2 = nullptr; posv = strtoll(t.c_str(), &e2, 10); if (t.empty() || *e2) { c.status = PHZ_E_ARG; c.error = "VCF POS is not an integer"; return; } }
        K.pos.push_back((int32_t)posv);
        if (f[3].size() > 255) {       // the general mapper's window is len(REF) (read_variant_map.py:237-244): never silently shortened
            c.status = PHZ_E_UNSUPPORTED; c.error = "REF allele longer than 255 bases at " + chrom + ":" + std::string(f[1]);
            return;
        }
        K.ref_len.push_back((uint8_t)f[3].size());
        K.black.push_back((!mark.empty() && mark.hit(chrom0, pos_bed - 1, pos_bed - 1 + (int64_t)std::max<size_t>(1, f[3].size()))) ? 1 : 0);
        std::string uid(chrom); uid += sep; uid += f[1];
        for (auto &x : every) { uid += sep; uid += x; }
        K.uid += uid; K.uid += '\n';
        K.rsid_field += f[2]; K.rsid_field += '\n';
        if (f[2] == "." || f[2].empty()) K.rsid += uid; else K.rsid += f[2];
        K.rsid += '\n';
        K.ref += f[3]; K.ref += '\n';
        for (size_t i = 0; i < every.size(); i++) { if (i) K.all_alleles += ','; K.all_alleles += every[i]; }
        K.all_alleles += '\n';
        for (size_t i = 0; i < ind.size(); i++) { if (i) K.alleles += ','; K.alleles += ind[i]; }
        K.alleles += '\n';
        for (size_t i = 0; i < ph.size(); i++) { if (i) K.phase += ','; K.phase += ph[i]; }
        K.phase += '\n';
        K.gt += geno; K.gt += '\n';
        if (have_maf) { put_pyfloat(K.maf_text, maf); put_pyfloat(K.maf_str, maf); K.maf.push_back(maf); }
        else { K.maf_text += "None"; K.maf_str += '0'; K.maf.push_back(0.0); }
        K.maf_text += '\n'; K.maf_str += '\n';
        const std::string_view i0 = ind.size() > 0 ? ind[0] : std::string_view(), i1 = ind.size() > 1 ? ind[1] : std::string_view();
        K.allele2 += i0; K.allele2 += '\n'; K.allele2 += i1; K.allele2 += '\n';
        K.a0.push_back(ind.size() > 0 ? base_code(i0) : 255); K.a1.push_back(ind.size() > 1 ? base_code(i1) : 255);
        K.is_ref.push_back(i0 == f[3] ? 1 : 0); K.is_ref.push_back(i1 == f[3] ? 1 : 0);
        auto pidx = [&](std::string_view a) -> int8_t { for (size_t k = 0; k < ph.size(); k++) if (ph[k] == a) return (int8_t)k; return -1; };
        K.phase_idx.push_back(pidx(i0)); K.phase_idx.push_back(pidx(i1));
        K.n++;
    }
}

}  // namespace

struct phz_vcf {
    std::vector<std::string> names;
    std::vector<Cols> cols;
    int64_t filter_count = 0, unphased = 0, excluded = 0, het = 0;
    int status = 0; std::string error;
};

extern "C" int phz_vcf_parse(const char *text, int64_t len, const phz_vcf_opts *opts, phz_vcf **out) {
    if (!text || len < 0 || !opts || !out) return PHZ_E_ARG;
    phz_vcf *h = new phz_vcf();
    *out = h;
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, opts->threads), ((size_t)len + (400u << 10) - 1) / (400u << 10)));      // ~8,000 lines per thread at least
    const size_t nchunks = len ? (size_t)nt * 4 : 0;
    std::vector<Chunk> ch(nchunks);
    // chunk i = the lines starting in [cut[i], cut[i + 1]): cut[i] = start of the first line at or after byte len * i / nchunks
    std::vector<int64_t> cut(nchunks + 1, len);
    if (nchunks) cut[0] = 0;
    for (size_t i = 1; i < nchunks; i++) {
        const int64_t raw = (int64_t)((__int128)len * (int64_t)i / (int64_t)nchunks);
        const char *nl = raw > 0 ? (const char *)memchr(text + raw - 1, '\n', (size_t)(len - raw + 1)) : nullptr;
        cut[i] = raw > 0 ? (nl ? (int64_t)(nl - text) + 1 : len) : 0;
    }
    BedIndex drop, mark;
    if (opts->n_drop > 0) drop.build(opts->n_drop, opts->drop_chrom, opts->drop_start, opts->drop_end);
    if (opts->n_mark > 0) mark.build(opts->n_mark, opts->mark_chrom, opts->mark_start, opts->mark_end);
    std::atomic<size_t> next(0);
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= nchunks) break;
            parse_lines(text, len, cut[i], cut[i + 1], *opts, drop, mark, ch[i]);
        }
    };
    if (nt == 1) work();
    else { std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(work); for (auto &t : th) t.join(); }
    // merge: the chunks' pieces of a chromosome, in chunk order (= file order), are appended by ONE thread per chromosome (the pieces of 1.5 M variants are
    // ~150 MB of column text: one thread doing all of them was a quarter of the call)
    std::unordered_map<std::string, int> idx;
    std::vector<std::vector<std::pair<size_t, size_t>>> pieces;
    for (size_t ci = 0; ci < ch.size(); ci++) {
        Chunk &c = ch[ci];
        if (c.status) { h->status = c.status; h->error = c.error; return h->status; }
        h->filter_count += c.filter_count; h->unphased += c.unphased; h->excluded += c.excluded;
        for (size_t k = 0; k < c.names.size(); k++) {
            auto it = idx.find(c.names[k]);
            int gi;
            if (it == idx.end()) { gi = (int)h->names.size(); idx.emplace(c.names[k], gi); h->names.push_back(c.names[k]); h->cols.emplace_back(); pieces.emplace_back(); }
            else gi = it->second;
            pieces[(size_t)gi].emplace_back(ci, k);
        }
    }
    std::vector<int> unsorted(h->names.size(), 0);
    std::atomic<size_t> next_chrom(0);
    auto merge = [&]() {
        for (;;) {
            const size_t g = next_chrom.fetch_add(1);
            if (g >= pieces.size()) break;
            Cols &K = h->cols[g];
            size_t n = 0, txt = 0;
            for (auto &pc : pieces[g]) { const Cols &o = ch[pc.first].cols[pc.second]; n += (size_t)o.n; txt += o.uid.size(); }
            K.pos.reserve(n); K.ref_len.reserve(n); K.a0.reserve(n); K.a1.reserve(n); K.black.reserve(n); K.is_ref.reserve(2 * n); K.phase_idx.reserve(2 * n); K.maf.reserve(n);
            K.uid.reserve(txt);
            for (auto &pc : pieces[g]) { Cols &o = ch[pc.first].cols[pc.second]; K.append(o); o = Cols(); }
            for (size_t i = 1; i < K.pos.size(); i++)
                if (K.pos[i] < K.pos[i - 1]) { unsorted[g] = 1; break; }
        }
    };
    {
        const int mt = (int)std::max<size_t>(1, std::min<size_t>((size_t)nt, pieces.size()));
        if (mt == 1) merge();
        else { std::vector<std::thread> th; for (int t = 0; t < mt; t++) th.emplace_back(merge); for (auto &t : th) t.join(); }
    }
    for (size_t k = 0; k < h->names.size(); k++) {
        h->het += h->cols[k].n;
        if (unsorted[k]) {
            h->status = PHZ_E_ARG;
            h->error = "     FATAL ERROR: VCF records of " + h->names[k] + " are not sorted by position.";
            return h->status;
        }
    }
    return PHZ_OK;
}

extern "C" int phz_vcf_summary(const phz_vcf *h, int32_t *n_chroms, int64_t *het, int64_t *filter_count, int64_t *indels_excluded, int64_t *unphased) {
    if (!h) return PHZ_E_ARG;
    if (n_chroms) *n_chroms = (int32_t)h->names.size();
    if (het) *het = h->het;
    if (filter_count) *filter_count = h->filter_count;
    if (indels_excluded) *indels_excluded = h->excluded;
    if (unphased) *unphased = h->unphased;
    return PHZ_OK;
}

extern "C" int phz_vcf_chrom(const phz_vcf *h, int32_t i, phz_vcf_table *t) {
    if (!h || !t || i < 0 || (size_t)i >= h->names.size()) return PHZ_E_ARG;
    const Cols &K = h->cols[(size_t)i];
    memset(t, 0, sizeof(*t));
    t->name = h->names[(size_t)i].c_str(); t->n = K.n;
    t->pos = K.pos.data(); t->ref_len = K.ref_len.data(); t->a0 = K.a0.data(); t->a1 = K.a1.data(); t->is_ref = K.is_ref.data();
    t->phase_idx = K.phase_idx.data(); t->maf = K.maf.data(); t->blacklisted = K.black.data();
    const std::string *pools[11] = {&K.uid, &K.rsid_field, &K.rsid, &K.ref, &K.all_alleles, &K.alleles, &K.phase, &K.gt, &K.maf_text, &K.maf_str, &K.allele2};
    for (int k = 0; k < 11; k++) { t->pool[k] = pools[k]->data(); t->pool_len[k] = (int64_t)pools[k]->size(); }
    return PHZ_OK;
}

extern "C" const char *phz_vcf_error(const phz_vcf *h) { return h ? h->error.c_str() : ""; }

extern "C" void phz_vcf_free(phz_vcf *h) { delete h; }
