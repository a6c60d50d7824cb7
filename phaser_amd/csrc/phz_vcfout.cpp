// Native write_vcf (phaser/phaser.py:1661-1855): the sample's VCF with the phASER FORMAT tags
// PG / PB / PI / PW / PC / PM (/ PS) filled from the haplotype blocks.  Input is the original VCF text and the sample's
// column (the reference feeds `gunzip -c | cut -f 1-9,S`), plus per chromosome the block arrays phz_rows_format returned
// and the variant table's string pools.  Header lines are handled in order, data lines in parallel.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "phz.h"
#include "phz_text.h"

namespace {

using phztext::put_int;
using phztext::put_pyfloat;
using phztext::split;

struct Hit { int32_t chrom, block, i; };       // chromosome, block within it, position of the variant inside the block

struct ChromIdx {
    std::vector<uint32_t> uid_off, rsid_off, alle_off, maf_off;
    std::vector<int32_t> blk_of;                // block of every slot e of the block arrays
    std::vector<int64_t> blk_start;             // prefix of blk_size
    std::vector<std::string> names, stat_txt;   // per block: PB text, PC text
};

const char *TAGS[6] = {"PG", "PB", "PI", "PW", "PC", "PM"};

struct Work {
    const phz_vcfout_chrom *chroms;
    int n_chroms;
    std::vector<ChromIdx> idx;
    // uid -> (chromosome, slot e of its block arrays): flat open-addressing table, filled by one thread per chromosome (atomic claims).
    // entry = tag (24 bits of the hash) << 40 | chromosome << 32 | e; the uid text decides equality
    std::unique_ptr<std::atomic<uint64_t>[]> table; uint64_t tmask = 0;
    int sample_column, gw_phase_vcf;
    double min_confidence;
    std::string_view sep, coi;
};

inline uint64_t uid_hash(std::string_view s) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (unsigned char ch : s) { h ^= ch; h *= 0x100000001b3ull; }
    h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
    return h;
}

std::string_view pool_at(const char *b, const std::vector<uint32_t> &off, int64_t i) {
    return std::string_view(b + off[(size_t)i], off[(size_t)i + 1] - off[(size_t)i] - 1);
}

void join(std::string &o, const std::vector<std::string_view> &v, char sep) {
    for (size_t i = 0; i < v.size(); i++) { if (i) o += sep; o.append(v[i]); }
}
void join(std::string &o, const std::vector<std::string> &v, char sep) {
    for (size_t i = 0; i < v.size(); i++) { if (i) o += sep; o += v[i]; }
}

// one data line (:1741-1850); returns false when the line is dropped (--chr filter)
bool data_line(const Work &W, std::string_view line, std::string &o, int64_t &unphased_phased, int64_t &corrections, int *status) {
    // scratch containers are per thread and keep their capacity from line to line (a genome is millions of lines)
    static thread_local std::vector<std::string_view> c, fmt, x, all_alleles, ind, alts, xs, tsplit;
    static thread_local std::vector<std::string> fmt2, sf, xv;
    split(line, '\t', c);
    if ((int)c.size() <= 8 || (int)c.size() <= W.sample_column) { *status = PHZ_E_ARG; return false; }
    if (!W.coi.empty() && c[0] != W.coi) return false;
    std::string sample(c[(size_t)W.sample_column]);
    std::string format(c[8]);
    if (format.find("GT") != std::string::npos) {
        split(c[8], ':', fmt);
        int gt_index = -1;
        for (size_t k = 0; k < fmt.size(); k++) if (fmt[k] == "GT") { gt_index = (int)k; break; }
        if (gt_index < 0) { *status = PHZ_E_ARG; return false; }       // "GT" only as part of another key: the reference raises here
        split(sample, ':', x);
        if (gt_index >= (int)x.size()) { *status = PHZ_E_ARG; return false; }
        std::string genotype(x[(size_t)gt_index]);
        { size_t q = genotype.find('|'); if (q != std::string::npos) genotype.erase(q, 1); }
        { size_t q = genotype.find('/'); if (q != std::string::npos) genotype.erase(q, 1); }
        all_alleles.clear(); all_alleles.push_back(c[3]);
        { split(c[4], ',', alts); for (auto &a : alts) all_alleles.push_back(a); }
        const size_t n_fields = fmt.size();
        if (x.size() < n_fields) sample.append(n_fields - x.size(), ':');
        fmt2.assign(fmt.begin(), fmt.end());
        for (const char *tag : TAGS) if (std::find(fmt2.begin(), fmt2.end(), tag) == fmt2.end()) fmt2.emplace_back(tag);
        auto fi = [&](const char *tag) { return (size_t)(std::find(fmt2.begin(), fmt2.end(), tag) - fmt2.begin()); };
        // id rebuilt WITHOUT --chr_prefix and with str(int(POS)), as the reference does (:1763)
        std::string uid(c[0]);
        { long long pv = strtoll(std::string(c[1]).c_str(), nullptr, 10); uid += W.sep; put_int(uid, pv); }
        for (auto &a : all_alleles) { uid += W.sep; uid.append(a); }
        bool found = false; Hit H{0, 0, 0};
        if (W.table) {
            const uint64_t hh = uid_hash(uid);
            for (uint64_t sl = hh & W.tmask;; sl = (sl + 1) & W.tmask) {
                const uint64_t ent = W.table[sl].load(std::memory_order_relaxed);
                if (ent == ~0ull) break;
                if ((ent >> 40) != (hh >> 40)) continue;
                const int ci = (int)((ent >> 32) & 0xFF); const int64_t e = (int64_t)(ent & 0xFFFFFFFFull);
                const phz_vcfout_chrom &Cc = W.chroms[ci];
                if (pool_at(Cc.uid, W.idx[(size_t)ci].uid_off, Cc.blk_var[e]) == std::string_view(uid)) {
                    const int32_t b = W.idx[(size_t)ci].blk_of[(size_t)e];
                    H = Hit{ci, b, (int32_t)(e - W.idx[(size_t)ci].blk_start[(size_t)b])}; found = true; break;
                }
            }
        }
        sf.clear();
        auto split_sample = [&]() { split(sample, ':', tsplit); sf.assign(tsplit.begin(), tsplit.end()); if (sf.size() < fmt2.size()) sf.resize(fmt2.size()); };
        if (found) {
            const phz_vcfout_chrom &C = W.chroms[H.chrom];
            const ChromIdx &X = W.idx[(size_t)H.chrom];
            const int64_t e = X.blk_start[(size_t)H.block] + H.i;
            const int g = C.blk_var[e];
            const int ha = C.blk_hap[e];
            split(pool_at(C.alleles, X.alle_off, g), ',', ind);
            std::string alleles_out[2], gw_out[2];
            const int order[2] = {ha, 1 - ha};
            for (int k = 0; k < 2; k++) {
                const int a = order[k];
                if (a >= (int)ind.size()) { *status = PHZ_E_ARG; return false; }
                int vidx = -1;
                for (size_t q = 0; q < all_alleles.size(); q++) if (all_alleles[q] == ind[(size_t)a]) { vidx = (int)q; break; }
                if (vidx < 0) { *status = PHZ_E_ARG; return false; }
                // gw[i] = corrected phase of the allele on haplotype A / B, re-ordered to allele-index order
                const int c0 = C.blk_cor[2 * e], c1 = C.blk_cor[2 * e + 1];
                const int gw = a == 0 ? (ha == 0 ? c0 : c1) : (ha == 0 ? c1 : c0);
                if (gw == 0 || gw == 1) { gw_out[gw].clear(); put_int(gw_out[gw], vidx); }
                put_int(alleles_out[k], vidx);
            }
            const double stat = C.blk_stat[H.block];
            {
                split(sample, ':', xs);
                xv.assign(xs.begin(), xs.end());
                const std::string new_phase = gw_out[0] + "|" + gw_out[1];
                bool changed = false;
                if (stat >= W.min_confidence) {
                    const std::string &cur = xv[(size_t)gt_index];
                    if (cur.find('|') != std::string::npos && cur != new_phase) corrections++;
                    if (cur.find('/') != std::string::npos && cur != "./." && cur != new_phase) unphased_phased++;
                    if (W.gw_phase_vcf == 1 || W.gw_phase_vcf == 2) { xv[(size_t)gt_index] = new_phase; changed = true; }
                }
                if (W.gw_phase_vcf == 2 && stat < W.min_confidence) { xv[(size_t)gt_index] = alleles_out[0] + "|" + alleles_out[1]; changed = true; }
                if (changed) { sample.clear(); join(sample, xv, ':'); }
            }
            split_sample();
            sf[fi("PG")] = alleles_out[0] + "|" + alleles_out[1];
            sf[fi("PB")] = X.names[(size_t)H.block];
            { std::string t; put_int(t, C.first_block_index + H.block + 1); sf[fi("PI")] = t; }
            sf[fi("PM")] = std::string(pool_at(C.maf_str, X.maf_off, C.blk_maxmaf[H.block]));
            sf[fi("PW")] = gw_out[0] + "|" + gw_out[1];
            sf[fi("PC")] = X.stat_txt[(size_t)H.block];
            if (W.gw_phase_vcf == 2 && stat < W.min_confidence) {
                if (std::find(fmt2.begin(), fmt2.end(), "PS") == fmt2.end()) { fmt2.emplace_back("PS"); sf.emplace_back(); }
                std::string t; put_int(t, C.first_block_index + H.block + 1); sf[fi("PS")] = t;
            }
        } else {
            split(sample, ':', xs);
            const std::string pw(xs[(size_t)gt_index]);
            split_sample();
            std::string sorted_gt(genotype);
            std::sort(sorted_gt.begin(), sorted_gt.end());
            std::string pg;
            for (size_t k = 0; k < sorted_gt.size(); k++) { if (k) pg += '/'; pg += sorted_gt[k]; }
            sf[fi("PG")] = pg; sf[fi("PB")] = "."; sf[fi("PI")] = "."; sf[fi("PM")] = "."; sf[fi("PW")] = pw; sf[fi("PC")] = ".";
        }
        format.clear(); join(format, fmt2, ':');
        sample.clear(); join(sample, sf, ':');
    }
    for (int k = 0; k < 8; k++) { o.append(c[(size_t)k]); o += '\t'; }
    o += format; o += '\t'; o += sample; o += '\n';
    return true;
}

}  // namespace

extern "C" int phz_vcf_phase_text(const char *text, int64_t len, int32_t sample_column, const char *id_separator, const char *chrom_of_interest,
                                  int32_t gw_phase_vcf, double min_confidence, const phz_vcfout_chrom *chroms, int32_t n_chroms, int32_t threads,
                                  char **out, int64_t *out_len, int64_t *unphased_phased, int64_t *corrections) {
    if (!text || len < 0 || !out || !out_len || (!chroms && n_chroms)) return PHZ_E_ARG;
    *out = nullptr; *out_len = 0;
    const bool timing = getenv("PHZ_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[phz timing]     vcf out: %-36s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    Work W;
    W.chroms = chroms; W.n_chroms = n_chroms; W.sample_column = sample_column; W.gw_phase_vcf = gw_phase_vcf; W.min_confidence = min_confidence;
    W.sep = id_separator ? id_separator : "_"; W.coi = chrom_of_interest ? chrom_of_interest : "";
    W.idx.resize((size_t)n_chroms);
    if (n_chroms > 256) return PHZ_E_ARG;
    size_t total_vars = 0;
    for (int ci = 0; ci < n_chroms; ci++) total_vars += (size_t)chroms[ci].n_blk_vars;
    if (total_vars >= (1ull << 32)) return PHZ_E_ARG;
    if (total_vars) {
        uint64_t cap = 1024;
        while (cap < 2 * total_vars) cap <<= 1;
        W.table.reset(new std::atomic<uint64_t>[cap]);
        W.tmask = cap - 1;
        const int nt0 = std::max(1, threads);
        std::vector<std::thread> th;
        for (int t = 0; t < nt0; t++) th.emplace_back([&, t] { for (uint64_t i = cap * (uint64_t)t / (uint64_t)nt0; i < cap * (uint64_t)(t + 1) / (uint64_t)nt0; i++) W.table[i].store(~0ull, std::memory_order_relaxed); });
        for (auto &x : th) x.join();
    }
    {   // per chromosome, in parallel: pool offsets, block starts, PB / PC texts, table entries of its block variants
        std::atomic<int> next(0);
        auto work = [&] {
            for (;;) {
                const int ci = next.fetch_add(1);
                if (ci >= n_chroms) break;
                const phz_vcfout_chrom &C = chroms[ci];
                ChromIdx &X = W.idx[(size_t)ci];
                X.uid_off = phztext::pool_offsets(C.uid, C.uid_len); X.rsid_off = phztext::pool_offsets(C.rsid, C.rsid_len);
                X.alle_off = phztext::pool_offsets(C.alleles, C.alleles_len); X.maf_off = phztext::pool_offsets(C.maf_str, C.maf_str_len);
                X.blk_start.assign((size_t)C.n_blocks + 1, 0);
                for (int64_t b = 0; b < C.n_blocks; b++) X.blk_start[(size_t)b + 1] = X.blk_start[(size_t)b] + C.blk_size[b];
                X.names.resize((size_t)C.n_blocks); X.stat_txt.resize((size_t)C.n_blocks);
                X.blk_of.resize((size_t)C.n_blk_vars);
                for (int64_t b = 0; b < C.n_blocks; b++) {
                    std::string &nm = X.names[(size_t)b];
                    for (int64_t e = X.blk_start[(size_t)b]; e < X.blk_start[(size_t)b + 1]; e++) {
                        if (e > X.blk_start[(size_t)b]) nm += ',';
                        const std::string_view r = pool_at(C.rsid, X.rsid_off, C.blk_var[e]);
                        for (char ch : r) nm += ch == ':' ? '_' : ch;
                        X.blk_of[(size_t)e] = (int32_t)b;
                        const std::string_view u = pool_at(C.uid, X.uid_off, C.blk_var[e]);
                        const uint64_t hh = uid_hash(u);
                        const uint64_t ent = (hh >> 40 << 40) | ((uint64_t)ci << 32) | (uint64_t)e;
                        // a uid that is already there (the same variant listed twice) keeps the LATER entry, as the dict assignment did
                        for (uint64_t sl = hh & W.tmask;; sl = (sl + 1) & W.tmask) {
                            uint64_t cur = W.table[sl].load(std::memory_order_relaxed);
                            if (cur == ~0ull) { if (W.table[sl].compare_exchange_strong(cur, ent, std::memory_order_relaxed)) break; }
                            if ((cur >> 40) == (hh >> 40) && ((cur >> 32) & 0xFF) == (uint64_t)ci &&
                                pool_at(C.uid, X.uid_off, C.blk_var[(int64_t)(cur & 0xFFFFFFFFull)]) == u) {
                                while (cur < ent && !W.table[sl].compare_exchange_weak(cur, ent, std::memory_order_relaxed)) {}
                                break;
                            }
                        }
                    }
                    if (C.blk_stat_int[b]) X.stat_txt[(size_t)b] = "1"; else put_pyfloat(X.stat_txt[(size_t)b], C.blk_stat[b]);
                }
            }
        };
        const int ntc = std::max(1, std::min(threads, n_chroms));
        if (ntc == 1) work();
        else { std::vector<std::thread> th; for (int t = 0; t < ntc; t++) th.emplace_back(work); for (auto &x : th) x.join(); }
    }
    lap("block names + variant lookup");
    // lines
    std::vector<int64_t> ls(1, 0);
    for (const char *p = text, *e = text + len; p < e;) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
        if (!nl) { ls.push_back(len + 1); break; }
        ls.push_back((int64_t)(nl - text) + 1); p = nl + 1;
    }
    const size_t nlines = ls.size() - 1;
    auto line_at = [&](size_t i) { return std::string_view(text + ls[i], (size_t)(ls[i + 1] - 1 - ls[i])); };
    // header lines, sequentially (:1688-1738); they sit at the top of the file
    std::string head, format_text;
    size_t first_data = nlines;
    for (size_t i = 0; i < nlines; i++) {
        const std::string_view line = line_at(i);
        if (line.empty()) continue;
        if (line[0] != '#') { first_data = i; break; }
        if (line.find("##FORMAT") != std::string_view::npos) { format_text.append(line); format_text += '\n'; head.append(line); head += '\n'; }
        else if (line.substr(0, 6) == "#CHROM") {
            static const char *DESC[6][2] = {{"PG", "phASER Local Genotype"}, {"PB", "phASER Local Block"},
                                             {"PI", "phASER Local Block Index (unique for each block)"}, {"PM", "phASER Local Block Maximum Variant MAF"},
                                             {"PW", "phASER Genome Wide Genotype"}, {"PC", "phASER Genome Wide Confidence"}};
            for (auto &d : DESC)
                if (format_text.find(std::string("##FORMAT=<ID=") + d[0] + ",") == std::string::npos)
                    head += std::string("##FORMAT=<ID=") + d[0] + ",Number=1,Type=String,Description=\"" + d[1] + "\">\n";
            if (gw_phase_vcf == 2 && format_text.find("##FORMAT=<ID=PS,") == std::string::npos)
                head += "##FORMAT=<ID=PS,Number=1,Type=String,Description=\"Phase Set\">\n";
            std::vector<std::string_view> c; split(line, '\t', c);
            if ((int)c.size() <= sample_column) return PHZ_E_ARG;
            for (int k = 0; k < 9; k++) { head.append(c[(size_t)k]); head += '\t'; }
            head.append(c[(size_t)sample_column]); head += '\n';
        } else if (line.substr(0, 2) == "##") { head.append(line); head += '\n'; }
        else {                 // a '#' line that is neither: the cut would have kept its first columns; pass it through cut
            std::vector<std::string_view> c; split(line, '\t', c);
            if ((int)c.size() > sample_column && c.size() > 9) { for (int k = 0; k < 9; k++) { head.append(c[(size_t)k]); head += '\t'; } head.append(c[(size_t)sample_column]); head += '\n'; }
            else { head.append(line); head += '\n'; }
        }
    }
    lap("line index + header");
    const size_t ndata = nlines - first_data;
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, threads), (ndata + 8191) / 8192));
    const size_t nchunks = ndata ? (size_t)nt * 4 : 0;
    struct Part { std::string o; int64_t up = 0, pc = 0; int status = 0; };
    std::vector<Part> parts(nchunks);
    std::atomic<size_t> next(0);
    auto work = [&]() {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= nchunks) break;
            Part &P = parts[k];
            const size_t lo = first_data + ndata * k / nchunks, hi = first_data + ndata * (k + 1) / nchunks;
            for (size_t i = lo; i < hi && !P.status; i++) {
                const std::string_view line = line_at(i);
                if (line.empty()) continue;
                if (line[0] == '#') { P.o.append(line); P.o += '\n'; continue; }
                data_line(W, line, P.o, P.up, P.pc, &P.status);
            }
        }
    };
    if (nt == 1) work();
    else { std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(work); for (auto &t : th) t.join(); }
    lap("data lines (threads)");
    size_t total = head.size();
    int64_t up = 0, pc = 0;
    for (auto &P : parts) { if (P.status) return P.status; total += P.o.size(); up += P.up; pc += P.pc; }
    char *buf = (char *)malloc(total + 1);
    if (!buf) return PHZ_E_NOMEM;
    size_t w = 0;
    memcpy(buf, head.data(), head.size()); w += head.size();
    {
        std::vector<size_t> at(parts.size() + 1, w);
        for (size_t k = 0; k < parts.size(); k++) at[k + 1] = at[k] + parts[k].o.size();
        std::atomic<size_t> nx(0);
        auto cp = [&] { for (;;) { const size_t k = nx.fetch_add(1); if (k >= parts.size()) break; memcpy(buf + at[k], parts[k].o.data(), parts[k].o.size()); std::string().swap(parts[k].o); } };
        if (nt == 1) cp();
        else { std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(cp); for (auto &x : th) x.join(); }
        w = at[parts.size()];
    }
    buf[w] = 0;
    lap("concatenation");
    *out = buf; *out_len = (int64_t)w;
    if (unphased_phased) *unphased_phased = up;
    if (corrections) *corrections = pc;
    return PHZ_OK;
}
