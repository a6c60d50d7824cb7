// Host stage C2 of the phasing path in native, multi-threaded code: block phasing and the text rows of the five
// output files for one chromosome.  Restates, on integer-indexed arrays,
//   phase_v3 / resolve_phase / sub_block_phase / split_by_weak      phaser/phaser.py:2107-2324
//   the block output loop (haplotypes, haplotypic_counts, allele_config)  phaser/phaser.py:865-1172
//   singleton rows :1180-1239, variant_connections rows :691-695, allelic_counts rows :737-749
// Inputs are what K_tally / phz_components and the ordering stage produced (plain arrays); outputs are text buffers
// in the reference's row order plus the per-block arrays write_vcf needs.  Numbers are printed the way Python prints
// them (str(int), repr(float), str(numpy.float64)) because that is what ends up in the files (SURVEY.md 8.1 rule 7).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <sys/resource.h>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_set>
#include <vector>

#include "phz.h"
#include "phz_text.h"

namespace {
std::atomic<long long> g_phase_ns{0};     // PHZ_TIMING: thread-time spent inside phase_component (block phasing proper)
bool g_phase_timing = false;


using phztext::Pool;
using phztext::put_int;
using phztext::put_pyfloat;

// ------------------------------------------------------------------------------------------------ block phasing
// Allele graph of one component: node 2*i + a = allele a of the i-th (position-sorted) variant.
struct AlleleGraph {
    int n = 0;
    std::vector<std::vector<int>> adj;            // allele-level neighbours
    std::vector<std::pair<int, int>> vedges;      // variant-level edges (ties included), i < j not required
    std::vector<char> mark;                       // scratch, all zero between calls
};

std::string flip(const std::string &c) {
    std::string o(c);
    for (auto &ch : o) ch = ch == '-' ? '-' : (ch == '0' ? '1' : '0');
    return o;
}

// resolve_phase (:2172-2207): flood fill from (variant lo, allele 0); accepted iff the component has exactly hi-lo
// nodes (not necessarily one per variant: the string then skips variants without a node -- quirk kept)
bool resolve(AlleleGraph &g, int lo, int hi, bool clean, std::string &out) {
    const int n = hi - lo;
    std::vector<int> stack, seen;
    const int seed = 2 * lo;
    g.mark[seed] = 1; seen.push_back(seed); stack.push_back(seed);
    while (!stack.empty()) {
        const int x = stack.back(); stack.pop_back();
        for (int y : g.adj[x]) {
            if (clean && (y < 2 * lo || y >= 2 * hi)) continue;
            if (!g.mark[y]) { g.mark[y] = 1; seen.push_back(y); stack.push_back(y); }
        }
    }
    const bool ok = (int)seen.size() == n;
    if (ok) {
        out.clear();
        for (int i = lo; i < hi; i++) {
            if (i < g.n && g.mark[2 * i]) out += '0';
            else if (i < g.n && g.mark[2 * i + 1]) out += '1';
        }
    }
    for (int x : seen) g.mark[x] = 0;
    return ok;
}

// supporting allele edges of configuration cfg over variants idx0, idx0+1, ... (zip truncates to the shorter; :2236-2249)
int score_cfg(AlleleGraph &g, int idx0, int idx_n, const std::string &cfg) {
    const int L = std::min(idx_n, (int)cfg.size());
    for (int k = 0; k < L; k++)
        if (cfg[k] != '-') g.mark[2 * (idx0 + k) + (cfg[k] == '1')] = 1;
    int s = 0;
    for (int k = 0; k < L; k++)
        if (cfg[k] != '-') {
            const int node = 2 * (idx0 + k) + (cfg[k] == '1');
            for (int y : g.adj[node]) s += g.mark[y];
        }
    for (int k = 0; k < L; k++)
        if (cfg[k] != '-') g.mark[2 * (idx0 + k) + (cfg[k] == '1')] = 0;
    return s;
}

// sub_block_phase without a given configuration (:2209-2258): optional clean flood fill, else brute force over the
// configurations whose first allele is 0; a tie for the best score gives all '-'
int best_free(AlleleGraph &g, int base, int n, bool attempt, std::string &c0, std::string &c1) {
    if (attempt && resolve(g, base, base + n, true, c0)) { c1 = flip(c0); return 0; }
    if (n > 30) return PHZ_E_UNSUPPORTED;       // 2^(n-1) configurations: the reference does not finish either
    uint64_t m[64];
    for (int k = 0; k < 2 * n; k++) {
        uint64_t x = 0;
        for (int y : g.adj[2 * base + k]) {
            const int l = y - 2 * base;
            if (l >= 0 && l < 2 * n) x |= 1ull << l;
        }
        m[k] = x;
    }
    long best_s = -1; uint64_t best_c = 0; long ties = 0;
    const uint64_t ncode = 1ull << (n - 1);
    for (uint64_t code = 0; code < ncode; code++) {
        uint64_t chosen = 1;                                     // variant 0 -> allele 0
        for (int k = 1; k < n; k++) chosen |= 1ull << (2 * k + (int)((code >> (n - 1 - k)) & 1));
        long s = 0;
        for (uint64_t c = chosen; c; c &= c - 1) s += __builtin_popcountll(m[__builtin_ctzll(c)] & chosen);
        if (s > best_s) { best_s = s; best_c = code; ties = 1; }
        else if (s == best_s) ties++;
    }
    if (ties == 1) {
        c0.assign(1, '0');
        for (int k = 1; k < n; k++) c0 += ((best_c >> (n - 1 - k)) & 1) ? '1' : '0';
        c1 = flip(c0);
    } else {
        c0.assign((size_t)n, '-'); c1 = c0;
    }
    return 0;
}

// sub_block_phase with two phased neighbours (:2225-2258): the 4 joint configurations, complements skipped
void best_given(AlleleGraph &g, int idx0, int idx_n, const std::string cur[2], const std::string nxt[2], std::string out[2]) {
    struct E { std::string c, inv; int s; };
    std::vector<E> sc;
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) {
            std::string c = cur[a] + nxt[b];
            std::string inv = flip(c);
            bool dup = false;
            for (auto &e : sc) if (e.c == c || e.c == inv) dup = true;
            if (dup) continue;
            const int s = score_cfg(g, idx0, idx_n, c);
            sc.push_back({c, inv, s});
        }
    int top = -1, nwin = 0, win = 0;
    for (auto &e : sc) top = std::max(top, e.s);
    for (size_t i = 0; i < sc.size(); i++) if (sc[i].s == top) { nwin++; win = (int)i; }
    if (nwin == 1) { out[0] = sc[win].c; out[1] = sc[win].inv; }
    else { out[0].assign((size_t)idx_n, '-'); out[1] = out[0]; }
}

// split_by_weak (:2271-2324) -> fragment start offsets (last entry = n)
int weak_split(const AlleleGraph &g, int max_size, std::vector<int> &bounds) {
    const int n = g.n;
    std::vector<long> weak((size_t)n + 2, 0);           // weak[p], 2 <= p <= n-2: variant edges (u, w) with u < p <= w
    {
        std::vector<long> diff((size_t)n + 2, 0);
        for (auto &e : g.vedges) {
            const int i = std::min(e.first, e.second), j = std::max(e.first, e.second);
            if (i == j) continue;
            diff[i + 1] += 1; diff[j + 1] -= 1;           // p in (i, j]
        }
        long run = 0;
        for (int p = 0; p <= n; p++) { run += diff[p]; weak[p] = run; }
    }
    long maxw = 0;
    for (int p = 2; p < n - 1; p++) maxw = std::max(maxw, weak[p]);
    std::vector<char> inpts((size_t)n + 2, 0);
    int biggest = n;
    long level = 1;
    bounds.assign({0, n});
    while (biggest > max_size || level == 1) {
        for (int p = 2; p < n - 1; p++)
            if (weak[p] == level && !inpts[p + 1] && !inpts[p - 1]) inpts[p] = 1;
        bounds.assign(1, 0);
        for (int p = 2; p < n - 1; p++) if (inpts[p]) bounds.push_back(p);
        bounds.push_back(n);
        biggest = 0;
        for (size_t i = 1; i < bounds.size(); i++) biggest = std::max(biggest, bounds[i] - bounds[i - 1]);
        level++;
        if (level > maxw + 1 && biggest > max_size) return PHZ_E_UNSUPPORTED;   // the reference loops forever here
    }
    return 0;
}

// phase_v3 (:2107-2170) -> final sub-blocks as (first local variant, configuration of haplotype A)
int phase_component(AlleleGraph &g, int max_block_size, std::vector<std::pair<int, std::string>> &res) {
    const int n = g.n;
    std::vector<std::string> fin;
    std::string c0, c1;
    if (resolve(g, 0, n, false, c0)) {
        fin.push_back(c0);
    } else {
        const int xmax = max_block_size == 0 ? n : max_block_size;
        std::vector<int> bounds;
        int st = weak_split(g, xmax, bounds);
        if (st) return st;
        const int nf = (int)bounds.size() - 1;
        std::vector<std::string> p0((size_t)nf), p1((size_t)nf);
        for (int f = 0; f < nf; f++) {
            st = best_free(g, bounds[f], bounds[f + 1] - bounds[f], nf > 1, p0[f], p1[f]);
            if (st) return st;
        }
        std::string cur[2] = {p0[0], p1[0]};
        int start = 0;
        for (int f = 1; f < nf; f++) {
            const std::string nxt[2] = {p0[f], p1[f]};
            const long total = (long)cur[0].size() + (long)cur[1].size() + (long)nxt[0].size() + (long)nxt[1].size();
            const int used = (int)((total + 1) / 2);
            const int hi = std::min(n, start + used);
            std::string out[2];
            best_given(g, start, std::max(0, hi - start), cur, nxt, out);
            if (out[0].find('-') != std::string::npos) {
                fin.push_back(cur[0]); start = used; cur[0] = nxt[0]; cur[1] = nxt[1];      // start is ASSIGNED (:2152)
            } else {
                cur[0] = out[0]; cur[1] = out[1];
            }
        }
        fin.push_back(cur[0]);
    }
    int vi = 0;
    for (auto &s : fin) {
        const int first = vi;
        vi += (int)s.size();
        if (!s.empty() && s[0] != '-') res.emplace_back(first, s);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ read label ranks
// distinct values of q[0..n) numbered by first appearance; label[i] = number of q[i]; ids = distinct values in that order
struct Ranker {
    std::vector<int32_t> key, val;
    // scratch of emit_block, kept across the blocks of a chunk: a block used to construct (and free) fifteen small containers
    struct Scratch {
        std::vector<int8_t> phs[2], cor[2];
        std::vector<int32_t> pool, allq, label, ids[2];
        std::vector<int> used_vars, black;
        std::vector<size_t> vlen;
        std::string labels[2], stat_txt;
    } s;
    void run(const std::vector<int32_t> &q, std::vector<int32_t> *label, std::vector<int32_t> *ids, int64_t *ndistinct) {
        size_t cap = 16;
        while (cap < q.size() * 2) cap <<= 1;
        key.assign(cap, -1); val.resize(cap);
        const size_t mask = cap - 1;
        int32_t next = 0;
        if (label) label->resize(q.size());
        if (ids) ids->clear();
        for (size_t i = 0; i < q.size(); i++) {
            const int32_t x = q[i];
            size_t h = ((uint32_t)x * 2654435761u) & mask;
            while (key[h] != -1 && key[h] != x) h = (h + 1) & mask;
            if (key[h] == -1) { key[h] = x; val[h] = next++; if (ids) ids->push_back(x); }
            if (label) (*label)[i] = val[h];
        }
        *ndistinct = next;
    }
};

struct Ctx {
    const phz_rows_in *in;
    Pool uid, rsid, alle, maftxt, qname;
    std::vector<uint8_t> phased;       // [nv]
    std::string_view name_of(int g) const { return in->unique_ids ? uid.at(g) : rsid.at(g); }
    // read list of (variant g, allele k, BAM b): QNAME ids of its kept lines in line order (phaser.py:1318-1322)
    const int32_t *rl(int g, int k, int b, int64_t *n) const {
        const size_t e = ((size_t)2 * g + k) * (size_t)in->nb + (size_t)b;
        *n = (int64_t)in->rl_start[e + 1] - (int64_t)in->rl_start[e];
        return in->rl_qid + in->rl_start[e];
    }
    // ... over all BAMs (the BAM lists are adjacent, in BAM order = line order across BAMs)
    const int32_t *rl_all(int g, int k, int64_t *n) const {
        const size_t e = ((size_t)2 * g + k) * (size_t)in->nb;
        *n = (int64_t)in->rl_start[e + in->nb] - (int64_t)in->rl_start[e];
        return in->rl_qid + in->rl_start[e];
    }
    int64_t lines_of(int g) const {
        const size_t e = (size_t)2 * g * (size_t)in->nb;
        return (int64_t)in->rl_start[e + 2 * (size_t)in->nb] - (int64_t)in->rl_start[e];
    }
};

struct BlockChunk {
    std::string hap, ase, cfg;
    std::vector<int32_t> bvar, bsize, bmaxmaf;
    std::vector<uint8_t> bhap, bstat_int;
    std::vector<int8_t> bcor;
    std::vector<double> bstat;
    int64_t phased = 0;
    int status = 0;
    int64_t est_hap = 0, est_ase = 0, est_cfg = 0;      // byte estimates of the three texts: one reservation instead of a growth chain
};

// one final (phased) block: rows of haplotypes.txt, haplotypic_counts.txt, allele_config.txt (:865-1172)
void emit_block(const Ctx &C, const std::vector<int> &vars, const std::string &ha, double sup_edges, double tot_edges, Ranker &rk,
                BlockChunk &o) {
    const phz_rows_in &I = *C.in;
    const int n = (int)vars.size();
    std::string hb = flip(ha);
    const std::string *hx[2] = {&ha, &hb};
    int minpos = I.pos[vars[0]], maxpos = I.pos[vars[0]];
    for (int g : vars) { minpos = std::min(minpos, I.pos[g]); maxpos = std::max(maxpos, I.pos[g]); }
    // phase indices of each haplotype's alleles in the VCF phase (-1 = not phased there)
    std::vector<int8_t> (&phs)[2] = rk.s.phs;
    int64_t counts[2];
    std::vector<int32_t> &pool = rk.s.pool;
    for (int h = 0; h < 2; h++) {
        phs[h].resize((size_t)n);
        pool.clear();
        for (int i = 0; i < n; i++) {
            const int g = vars[i], k = (*hx[h])[i] - '0';
            phs[h][i] = I.phase_idx[2 * g + k];
            int64_t m; const int32_t *q = C.rl_all(g, k, &m);
            pool.insert(pool.end(), q, q + m);
        }
        rk.run(pool, nullptr, nullptr, &counts[h]);
    }
    int nknown = 0, ksum = 0, first_known = -1;
    bool all_equal = true, any_nan = false;
    for (int i = 0; i < n; i++) {
        if (phs[0][i] >= 0) {
            if (first_known < 0) first_known = phs[0][i];
            else if (phs[0][i] != first_known) all_equal = false;
            nknown++; ksum += phs[0][i];
        } else any_nan = true;
    }
    const int conc = all_equal ? 1 : 0;                       // usable values all the same (or none)
    // genome-wide phase (:945-1025)
    std::vector<int8_t> (&cor)[2] = rk.s.cor;
    cor[0] = phs[0]; cor[1] = phs[1];
    double stat = 0.5; bool stat_int = false;
    auto by_mean = [&]() {
        double m = (double)ksum / (double)nknown;
        if (m < 0.5) { cor[0].assign((size_t)n, 0); cor[1].assign((size_t)n, 1); }
        else if (m > 0.5) { cor[0].assign((size_t)n, 1); cor[1].assign((size_t)n, 0); }
        const double other = 1 - m;
        stat = m >= other ? m : other;
    };
    if (nknown > 0) {
        if (!any_nan && all_equal) { stat = 1; stat_int = true; }
        else if (I.gw_phase_method == 0) by_mean();
        else if (I.gw_phase_method == 1) {
            double w[2] = {0, 0};
            for (int i = 0; i < n; i++) {
                if (phs[0][i] == 0) w[0] += I.maf[vars[i]];
                else if (phs[0][i] == 1) w[1] += I.maf[vars[i]];
            }
            const double sw = w[0] + w[1];
            if (sw > 0) {
                stat = (w[0] >= w[1] ? w[0] : w[1]) / sw;
                if (w[0] > w[1]) { cor[0].assign((size_t)n, 0); cor[1].assign((size_t)n, 1); }
                else if (w[1] > w[0]) { cor[0].assign((size_t)n, 1); cor[1].assign((size_t)n, 0); }
            } else by_mean();
        }
    }
    std::string &stat_txt = rk.s.stat_txt;
    stat_txt.clear();
    if (stat_int) stat_txt = "1"; else put_pyfloat(stat_txt, stat);
    int maxmaf_g = vars[0];
    for (int g : vars) if (I.maf[g] > I.maf[maxmaf_g]) maxmaf_g = g;       // first maximal element, like max()
    auto phase_chars = [&](const std::vector<int8_t> &v, std::string &s) { for (int8_t x : v) s += x < 0 ? '-' : (char)('0' + x); };

    // ---- haplotypes.txt
    std::string &H = o.hap;
    H.append(I.chrom); H += '\t'; put_int(H, minpos); H += '\t'; put_int(H, maxpos); H += '\t'; put_int(H, maxpos - minpos); H += '\t';
    put_int(H, n); H += '\t';
    for (int i = 0; i < n; i++) { if (i) H += ','; H.append(C.name_of(vars[i])); }
    H += '\t';
    for (int h = 0; h < 2; h++) {
        if (h) H += '|';
        for (int i = 0; i < n; i++) { if (i) H += ','; H.append(C.alle.at(2 * vars[i] + ((*hx[h])[i] - '0'))); }
    }
    H += '\t'; put_int(H, counts[0]); H += '\t'; put_int(H, counts[1]); H += '\t'; put_int(H, counts[0] + counts[1]); H += '\t';
    put_pyfloat(H, sup_edges); H += '\t'; put_pyfloat(H, tot_edges); H += '\t';
    phase_chars(phs[0], H); H += '|'; phase_chars(phs[1], H); H += '\t'; put_int(H, conc); H += '\t';
    phase_chars(cor[0], H); H += '|'; phase_chars(cor[1], H); H += '\t'; H += stat_txt; H += '\n';

    // ---- haplotypic_counts.txt: one row per BAM (:1048-1125)
    std::vector<int> &used_vars = rk.s.used_vars, &black = rk.s.black;
    std::vector<int32_t> &allq = rk.s.allq, &label = rk.s.label, (&ids)[2] = rk.s.ids;
    std::vector<size_t> &vlen = rk.s.vlen;
    std::string (&labels)[2] = rk.s.labels;
    for (int b = 0; b < I.nb; b++) {
        if (I.bam_excluded && I.bam_excluded[b]) continue;
        used_vars.clear(); black.clear();
        int64_t ns[2];
        for (int h = 0; h < 2; h++) {
            allq.clear(); vlen.clear();
            for (int i = 0; i < n; i++) {
                const int g = vars[i];
                if (!(I.blacklisted && I.blacklisted[g])) {
                    const int k = (*hx[h])[i] - '0';
                    if (h == 0) used_vars.push_back(g);
                    int64_t m; const int32_t *q = C.rl(g, k, b, &m);
                    allq.insert(allq.end(), q, q + m);
                    vlen.push_back((size_t)m);
                } else if (h == 0) black.push_back(g);
            }
            rk.run(allq, &label, &ids[h], &ns[h]);
            labels[h].clear();
            size_t p0 = 0;
            for (size_t v = 0; v < vlen.size(); v++) {
                if (v) labels[h] += ';';
                for (size_t t = 0; t < vlen[v]; t++) { if (t) labels[h] += ','; put_int(labels[h], label[p0 + t]); }
                p0 += vlen[v];
            }
        }
        const int64_t cov = ns[0] + ns[1];
        if (cov <= 0) continue;
        std::string &A = o.ase;
        A.append(I.chrom); A += '\t'; put_int(A, minpos); A += '\t'; put_int(A, maxpos); A += '\t';
        for (size_t i = 0; i < used_vars.size(); i++) { if (i) A += ','; A.append(C.uid.at(used_vars[i])); }
        A += '\t'; put_int(A, (long long)used_vars.size()); A += '\t';
        for (size_t i = 0; i < black.size(); i++) { if (i) A += ','; A.append(C.uid.at(black[i])); }
        A += '\t'; put_int(A, (long long)black.size()); A += '\t';
        for (int h = 0; h < 2; h++) {
            bool firstv = true;
            for (int i = 0; i < n; i++) {
                const int g = vars[i];
                if (I.blacklisted && I.blacklisted[g]) continue;
                if (!firstv) A += ',';
                firstv = false;
                A.append(C.alle.at(2 * g + ((*hx[h])[i] - '0')));
            }
            A += '\t';
        }
        put_int(A, ns[0]); A += '\t'; put_int(A, ns[1]); A += '\t'; put_int(A, cov); A += '\t';
        A += cor[0][0] == 0 ? "0|1" : (cor[0][0] == 1 ? "1|0" : "0/1");
        A += '\t'; A += stat_txt; A += '\t';
        if (I.output_read_ids == 1) {
            for (int h = 0; h < 2; h++) {
                for (size_t i = 0; i < ids[h].size(); i++) { if (i) A += ','; A.append(C.qname.at(ids[h][i])); }
                A += '\t';
            }
        }
        A.append(C.maftxt.at(maxmaf_g)); A += '\t'; A.append(I.bam_names[b]); A += '\t'; A += labels[0]; A += '\t'; A += labels[1]; A += '\n';
    }

    // ---- allele_config.txt (:1160-1172)
    std::string &F = o.cfg;
    for (int i = 0; i < n; i++) {
        const int ga = vars[i];
        const bool ra = I.is_ref[2 * ga + (ha[i] - '0')];
        const std::string_view ua = C.uid.at(ga), sa = C.rsid.at(ga);
        for (int j = 0; j < n; j++) {
            if (i == j) continue;
            const int gb = vars[j];
            const bool rb = I.is_ref[2 * gb + (hb[j] - '0')];
            F.append(ua); F += '\t'; F.append(sa); F += '\t'; F.append(C.uid.at(gb)); F += '\t'; F.append(C.rsid.at(gb));
            F += (ra == rb) ? "\ttrans\n" : "\tcis\n";
        }
    }
    // ---- per-block arrays for write_vcf
    o.bsize.push_back(n); o.bstat.push_back(stat); o.bstat_int.push_back(stat_int ? 1 : 0); o.bmaxmaf.push_back(maxmaf_g);
    for (int i = 0; i < n; i++) {
        o.bvar.push_back(vars[i]); o.bhap.push_back((uint8_t)(ha[i] - '0'));
        o.bcor.push_back(cor[0][i]); o.bcor.push_back(cor[1][i]);
    }
    o.phased += n;
}

// components ranked [lo, hi) in first-key order: phase them and write their rows
void run_block_chunk(Ctx &C, int64_t lo, int64_t hi, BlockChunk &o) {
    const phz_rows_in &I = *C.in;
    const int64_t res_max = 256ll << 20;      // an estimate, not a bound: the strings still grow when it falls short
    o.hap.reserve((size_t)std::min(o.est_hap, res_max)); o.ase.reserve((size_t)std::min(o.est_ase, res_max)); o.cfg.reserve((size_t)std::min(o.est_cfg, res_max));
    Ranker rk;
    AlleleGraph g;
    std::vector<int> mem, vars, sub_of, alle_of, byid, idx;
    std::vector<long> sup, tot;
    std::string ha2 = "00";
    std::vector<std::pair<int, std::string>> subs;
    struct LE { int i, j, k; };
    std::vector<LE> edges;
    for (int64_t r = lo; r < hi && !o.status; r++) {
        const int64_t ci = I.comp_order[r];
        mem.assign(I.mem_s + I.comp_starts[ci], I.mem_s + I.comp_ends[ci]);
        std::sort(mem.begin(), mem.end(), [&](int a, int b) { return I.pos[a] != I.pos[b] ? I.pos[a] < I.pos[b] : a < b; });   // sort_var_ids :1884
        const int n = (int)mem.size();
        // The commonest component by far: two variants joined by one edge with a verdict.  resolve_phase floods (variant 0, allele 0)
        // to (variant 1, allele 0) for a cis edge and to (variant 1, allele 1) for a trans edge, i.e. one block "00" / "01" with
        // one supporting edge out of one (phaser.py:2172-2207, :876-895): nothing of the graph machinery below is needed.
        if (n == 2 && I.e_ends[ci] - I.e_starts[ci] == 1) {
            const int k = (int)I.cfgv[I.e_keep[I.eo[I.e_starts[ci]]]];
            if (k == 0 || k == 1) {
                vars.assign(mem.begin(), mem.end());
                C.phased[vars[0]] = 1; C.phased[vars[1]] = 1;
                ha2[1] = k == 0 ? '0' : '1';
                emit_block(C, vars, ha2, 1.0, 1.0, rk, o);
                continue;
            }
        }
        byid.assign(mem.begin(), mem.end());                        // for the global -> local lookup
        idx.resize((size_t)n);
        for (int i = 0; i < n; i++) idx[i] = i;
        std::sort(idx.begin(), idx.end(), [&](int a, int b) { return mem[a] < mem[b]; });
        for (int i = 0; i < n; i++) byid[i] = mem[idx[i]];
        auto loc = [&](int gid) { return idx[std::lower_bound(byid.begin(), byid.end(), gid) - byid.begin()]; };
        // the per-node neighbour vectors outlive the component (cleared, not destroyed), like every other temporary of this loop:
        // a genome is ~1M components of two or three variants
        g.n = n;
        if (g.adj.size() < (size_t)2 * n) g.adj.resize((size_t)2 * n);
        for (int k = 0; k < 2 * n; k++) g.adj[(size_t)k].clear();
        g.vedges.clear(); g.mark.assign((size_t)2 * n + 2, 0);
        edges.clear();
        for (int64_t t = I.e_starts[ci]; t < I.e_ends[ci]; t++) {
            const int64_t e = I.e_keep[I.eo[t]];
            const int i = loc(I.ea[e]), j = loc(I.eb[e]), k = (int)I.cfgv[e];
            edges.push_back({i, j, k});
            g.vedges.emplace_back(i, j);
            if (k == 0) {
                g.adj[2 * i].push_back(2 * j); g.adj[2 * j].push_back(2 * i);
                g.adj[2 * i + 1].push_back(2 * j + 1); g.adj[2 * j + 1].push_back(2 * i + 1);
            } else if (k == 1) {
                g.adj[2 * i].push_back(2 * j + 1); g.adj[2 * j + 1].push_back(2 * i);
                g.adj[2 * i + 1].push_back(2 * j); g.adj[2 * j].push_back(2 * i + 1);
            }
        }
        subs.clear();
        if (g_phase_timing) {
            const auto t0 = std::chrono::steady_clock::now();
            o.status = phase_component(g, I.max_block_size, subs);
            g_phase_ns.fetch_add((long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
        } else o.status = phase_component(g, I.max_block_size, subs);
        if (o.status) return;
        sub_of.assign((size_t)n, -1); alle_of.assign((size_t)n, 0);
        for (size_t s = 0; s < subs.size(); s++)
            for (size_t t = 0; t < subs[s].second.size(); t++) {
                const int li = subs[s].first + (int)t;
                if (li < n) { sub_of[li] = (int)s; alle_of[li] = subs[s].second[t] - '0'; }
            }
        // allele edges supporting / total inside each final block (:876-895; ordered pairs halved = edges)
        sup.assign(subs.size(), 0); tot.assign(subs.size(), 0);
        for (auto &e : edges) {
            if (e.k < 0 || sub_of[e.i] < 0 || sub_of[e.i] != sub_of[e.j]) continue;
            tot[sub_of[e.i]]++;
            const int linked = e.k == 0 ? alle_of[e.i] : 1 - alle_of[e.i];
            if (alle_of[e.j] == linked) sup[sub_of[e.i]]++;
        }
        for (size_t s = 0; s < subs.size(); s++) {
            vars.clear();
            std::string ha;
            for (size_t t = 0; t < subs[s].second.size(); t++) {
                const int li = subs[s].first + (int)t;
                if (li < n) { vars.push_back(mem[li]); ha += subs[s].second[t]; }
            }
            if (vars.empty()) continue;
            for (int v : vars) C.phased[v] = 1;
            emit_block(C, vars, ha, (double)sup[s], (double)tot[s], rk, o);
        }
    }
}

struct TextChunk { std::string a, b; int64_t rows = 0; };

// variant_connections rows (:691-695) for eorder[lo, hi)
void run_conn(const Ctx &C, int64_t lo, int64_t hi, TextChunk &o) {
    const phz_rows_in &I = *C.in;
    o.a.reserve((size_t)(hi - lo) * 72);
    for (int64_t t = lo; t < hi; t++) {
        const int64_t k = I.eorder[t];
        const int a = I.va[k], b = I.vb[k];
        std::string &S = o.a;
        S.append(C.uid.at(a)); S += '\t'; S.append(C.uid.at(b)); S += '\t'; put_int(S, I.sup[k]); S += '\t'; put_int(S, I.tot[k]); S += '\t';
        if (I.sup[k] == 0) S += '0';
        else if (I.tot[k] - I.sup[k] > 0) put_pyfloat(S, I.pv[k]);
        else S += '1';
        S += '\t';
        if (I.phase_idx[2 * a] >= 0 && I.phase_idx[2 * b] >= 0 && I.cis[k] != I.trans[k]) {
            const int pa = I.cis[k] > I.trans[k] ? I.phase_idx[2 * a] : I.phase_idx[2 * a + 1];
            S += pa == I.phase_idx[2 * b] ? '1' : '0';
        } else S += '.';
        S += '\n';
    }
}

// allelic_counts rows (:737-749) for the first-appearance keys [lo, hi)
void run_allelic(const Ctx &C, int64_t lo, int64_t hi, TextChunk &o) {
    const phz_rows_in &I = *C.in;
    o.a.reserve((size_t)(hi - lo) * 56);
    for (int64_t t = lo; t < hi; t++) {
        const int g = (int)I.key_g[t];
        const long long r0 = I.var_distinct[3 * g], r1 = I.var_distinct[3 * g + 1];
        if (r0 + r1 <= 0) continue;
        std::string &S = o.a;
        S.append(I.chrom); S += '\t'; put_int(S, I.pos[g]); S += '\t'; S.append(C.uid.at(g)); S += '\t'; S.append(C.alle.at(2 * g)); S += '\t';
        S.append(C.alle.at(2 * g + 1)); S += '\t'; put_int(S, r0); S += '\t'; put_int(S, r1); S += '\t'; put_int(S, r0 + r1); S += '\n';
        o.rows++;
    }
}

// singleton rows (:1180-1239) for the first-appearance keys [lo, hi): a = haplotypic_counts rows, b = haplotypes rows
void run_singles(const Ctx &C, int64_t lo, int64_t hi, TextChunk &o) {
    const phz_rows_in &I = *C.in;
    if (o.a.capacity() == 0) { o.a.reserve((size_t)(hi - lo) * 48); o.b.reserve((size_t)(hi - lo) * 40); }     // most keys sit in blocks
    std::vector<int32_t> q[2];
    for (int64_t t = lo; t < hi; t++) {
        const int g = (int)I.key_g[t];
        if ((long long)I.var_count[3 * g] + I.var_count[3 * g + 1] == 0 || C.phased[g]) continue;
        const bool is_phased = I.phase_idx[2 * g] >= 0;
        if (!(I.blacklisted && I.blacklisted[g])) {
            for (int b = 0; b < I.nb; b++) {
                if (I.bam_excluded && I.bam_excluded[b]) continue;
                for (int k = 0; k < 2; k++) {
                    int64_t m; const int32_t *src = C.rl(g, k, b, &m);
                    if (I.output_read_ids == 1) {          // the QNAMEs are listed: first-appearance order (what the block rows and the device stage write)
                        std::unordered_set<int32_t> seen;
                        q[k].clear();
                        for (int64_t t = 0; t < m; t++) if (seen.insert(src[t]).second) q[k].push_back(src[t]);
                    } else {
                        q[k].assign(src, src + m);
                        std::sort(q[k].begin(), q[k].end());
                        q[k].erase(std::unique(q[k].begin(), q[k].end()), q[k].end());
                    }
                }
                const long long n0 = (long long)q[0].size(), n1 = (long long)q[1].size();
                if (n0 + n1 <= 0) continue;
                std::string &A = o.a;
                A.append(I.chrom); A += '\t'; put_int(A, I.pos[g]); A += '\t'; put_int(A, I.pos[g]); A += '\t'; A.append(C.uid.at(g));
                A += "\t1\t\t0\t"; A.append(C.alle.at(2 * g)); A += '\t'; A.append(C.alle.at(2 * g + 1)); A += '\t';
                put_int(A, n0); A += '\t'; put_int(A, n1); A += '\t'; put_int(A, n0 + n1); A += '\t';
                if (is_phased) { put_int(A, I.phase_idx[2 * g]); A += '|'; put_int(A, I.phase_idx[2 * g + 1]); } else A += "0/1";
                A += "\t1\t";
                if (I.output_read_ids == 1) {
                    for (int k = 0; k < 2; k++) {
                        for (size_t i = 0; i < q[k].size(); i++) { if (i) A += ','; A.append(C.qname.at(q[k][i])); }
                        A += '\t';
                    }
                }
                A.append(C.maftxt.at(g)); A += '\t'; A.append(I.bam_names[b]); A += "\t\t\n";
            }
        }
        const long long d0 = I.var_distinct[3 * g], d1 = I.var_distinct[3 * g + 1];
        std::string &H = o.b;
        H.append(I.chrom); H += '\t'; put_int(H, (long long)I.pos[g] - 1); H += '\t'; put_int(H, I.pos[g]); H += "\t1\t1\t";
        H.append(C.name_of(g)); H += '\t'; H.append(C.alle.at(2 * g)); H += '|'; H.append(C.alle.at(2 * g + 1)); H += '\t';
        put_int(H, d0); H += '\t'; put_int(H, d1); H += '\t'; put_int(H, d0 + d1); H += "\t0\t0\t";
        std::string ps;
        if (is_phased) { put_int(ps, I.phase_idx[2 * g]); ps += '|'; put_int(ps, I.phase_idx[2 * g + 1]); } else ps = "-|-";
        H += ps; H += "\tnan\t"; H += ps; H += "\tnan\n";
        o.rows++;
    }
}

template <class F>
void parallel_chunks(int threads, int64_t nchunks, F fn) {
    if (nchunks <= 0) return;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads, nchunks));
    if (nt == 1) { for (int64_t i = 0; i < nchunks; i++) fn(i); return; }
    std::atomic<int64_t> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++)
        th.emplace_back([&]() { for (;;) { const int64_t i = next.fetch_add(1); if (i >= nchunks) break; fn(i); } });
    for (auto &t : th) t.join();
}

template <class T>
T *take_vec(const std::vector<T> &v) {
    T *p = (T *)malloc(std::max<size_t>(1, v.size() * sizeof(T)));
    if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

// chunk boundaries over keys [0, n) that never straddle a change of key_bam
std::vector<int64_t> key_chunks(const phz_rows_in &I, int64_t step) {
    std::vector<int64_t> b(1, 0);
    int64_t start = 0;
    for (int64_t t = 1; t <= I.n_keys; t++) {
        if (t == I.n_keys || I.key_bam[t] != I.key_bam[t - 1] || t - start >= step) { b.push_back(t); start = t; }
    }
    if (I.n_keys == 0) b.assign(1, 0);
    return b;
}

// ---- ordering stage (what the host did with numpy before): from the tested pairs, the component labels and the tally's
// first-appearance numbers to the orders the writers follow (SURVEY.md 8.1 rules 2, 4, 5)
struct Prep {
    std::vector<int32_t> ea, eb, va, vb, mem_s;
    std::vector<int64_t> eorder, comp_starts, comp_ends, comp_order, e_keep, eo, e_starts, e_ends, key_bam, key_g;
};

void derive(const phz_rows_in &in, Prep &P, phz_rows_in &I) {
    I = in;
    const int nv = in.nv;
    const int64_t ne = in.n_edges;
    const int32_t v0 = (int32_t)in.v0;
    // pairs in local indices, oriented by first appearance (phaser.py:667-678 enumerates from the earlier key)
    P.ea.resize((size_t)ne); P.eb.resize((size_t)ne); P.va.resize((size_t)ne); P.vb.resize((size_t)ne);
    for (int64_t e = 0; e < ne; e++) {
        const int32_t a = in.ea[e] - v0, b = in.eb[e] - v0;
        P.ea[(size_t)e] = a; P.eb[(size_t)e] = b;
        const bool swap = in.rank[b] < in.rank[a];
        P.va[(size_t)e] = swap ? b : a; P.vb[(size_t)e] = swap ? a : b;
    }
    // rows of variant_connections: hash order in the reference, (rank a, rank b) here
    {
        struct K { uint64_t ra, rb; int64_t e; };
        std::vector<K> k((size_t)ne);
        for (int64_t e = 0; e < ne; e++) k[(size_t)e] = {in.rank[P.va[(size_t)e]], in.rank[P.vb[(size_t)e]], e};
        std::sort(k.begin(), k.end(), [](const K &x, const K &y) { return x.ra != y.ra ? x.ra < y.ra : (x.rb != y.rb ? x.rb < y.rb : x.e < y.e); });
        P.eorder.resize((size_t)ne);
        for (int64_t e = 0; e < ne; e++) P.eorder[(size_t)e] = k[(size_t)e].e;
    }
    // components of the surviving graph: members grouped by label (= smallest member), components in label order
    std::vector<int32_t> deg((size_t)nv + 1, 0);
    for (int64_t e = 0; e < ne; e++) if (in.keep[e]) { deg[(size_t)P.ea[(size_t)e]]++; deg[(size_t)P.eb[(size_t)e]]++; }
    std::vector<int32_t> comp_of((size_t)nv + 1, -1);          // label -> component index
    std::vector<int64_t> csize;
    for (int g = 0; g < nv; g++) {
        if (!deg[(size_t)g]) continue;
        const int l = in.label[g] - v0;
        if (comp_of[(size_t)l] < 0) { comp_of[(size_t)l] = (int32_t)csize.size(); csize.push_back(0); }     // labels are met in ascending order (label <= member)
        csize[(size_t)comp_of[(size_t)l]]++;
    }
    const int64_t nc = (int64_t)csize.size();
    P.comp_starts.assign((size_t)nc, 0); P.comp_ends.assign((size_t)nc, 0);
    {
        int64_t acc = 0;
        for (int64_t c = 0; c < nc; c++) { P.comp_starts[(size_t)c] = acc; acc += csize[(size_t)c]; P.comp_ends[(size_t)c] = P.comp_starts[(size_t)c]; }
        P.mem_s.resize((size_t)acc);
        for (int g = 0; g < nv; g++) {
            if (!deg[(size_t)g]) continue;
            const int c = comp_of[(size_t)(in.label[g] - v0)];
            P.mem_s[(size_t)P.comp_ends[(size_t)c]++] = g;
        }
    }
    // block order = order in which the components' first keys enter the connectivity map (rule 4)
    {
        std::vector<std::pair<uint64_t, int64_t>> k((size_t)nc);
        for (int64_t c = 0; c < nc; c++) {
            uint64_t m = ~0ull;
            for (int64_t t = P.comp_starts[(size_t)c]; t < P.comp_ends[(size_t)c]; t++) m = std::min(m, in.rank[P.mem_s[(size_t)t]]);
            k[(size_t)c] = {m, c};
        }
        std::sort(k.begin(), k.end());
        P.comp_order.resize((size_t)nc);
        for (int64_t c = 0; c < nc; c++) P.comp_order[(size_t)c] = k[(size_t)c].second;
    }
    // surviving pairs grouped by component, pair order kept inside
    P.e_starts.assign((size_t)nc, 0); P.e_ends.assign((size_t)nc, 0);
    {
        std::vector<int64_t> cnt((size_t)nc, 0);
        int64_t nk = 0;
        for (int64_t e = 0; e < ne; e++) if (in.keep[e]) { cnt[(size_t)comp_of[(size_t)(in.label[P.ea[(size_t)e]] - v0)]]++; nk++; }
        int64_t acc = 0;
        for (int64_t c = 0; c < nc; c++) { P.e_starts[(size_t)c] = acc; P.e_ends[(size_t)c] = acc; acc += cnt[(size_t)c]; }
        P.e_keep.resize((size_t)nk); P.eo.resize((size_t)nk);
        for (int64_t e = 0; e < ne; e++)
            if (in.keep[e]) { const int c = comp_of[(size_t)(in.label[P.ea[(size_t)e]] - v0)]; P.e_keep[(size_t)P.e_ends[(size_t)c]++] = e; }
        for (int64_t t = 0; t < nk; t++) P.eo[(size_t)t] = t;
    }
    // first-appearance keys of the covered variants: (BAM of the first kept line, line) (rule 2)
    {
        struct K { int64_t bam, line; int32_t g; };
        std::vector<K> k;
        for (int g = 0; g < nv; g++) {
            const int64_t f = in.var_first[g];
            if (f < 0) continue;
            int64_t b = 0;
            for (int t = 0; t < in.nb; t++) if (f >= in.bam_line_lo[t] && f < in.bam_line_hi[t]) b = t;
            k.push_back({b, f, g});
        }
        std::sort(k.begin(), k.end(), [](const K &x, const K &y) { return x.bam != y.bam ? x.bam < y.bam : (x.line != y.line ? x.line < y.line : x.g < y.g); });
        P.key_bam.resize(k.size()); P.key_g.resize(k.size());
        for (size_t t = 0; t < k.size(); t++) { P.key_bam[t] = k[t].bam; P.key_g[t] = k[t].g; }
    }
    I.ea = P.ea.data(); I.eb = P.eb.data(); I.va = P.va.data(); I.vb = P.vb.data(); I.eorder = P.eorder.data();
    I.ncomp = nc; I.mem_s = P.mem_s.data(); I.comp_starts = P.comp_starts.data(); I.comp_ends = P.comp_ends.data();
    I.comp_order = P.comp_order.data(); I.e_keep = P.e_keep.data(); I.eo = P.eo.data(); I.e_starts = P.e_starts.data(); I.e_ends = P.e_ends.data();
    I.n_keys = (int64_t)P.key_g.size(); I.key_bam = P.key_bam.data(); I.key_g = P.key_g.data();
}

// everything one chromosome needs between the phases of phz_rows_format_multi
struct ChromState {
    Ctx C;
    Prep P;
    phz_rows_in I2;                           // the chromosome's input with the derived arrays plugged in (raw mode)
    std::vector<int64_t> cb, kb;              // block-chunk / key-chunk boundaries
    std::vector<int64_t> w;                   // weight of every component (block chunks are balanced by it)
    std::vector<int64_t> eh, ea, ec;          // ... and byte estimates of its haplotypes / haplotypic_counts / allele_config rows
    std::vector<BlockChunk> bc;
    std::vector<TextChunk> cc, ac, sc;
};

// the chunk buffers of one phz_rows_format_multi call; every phz_rows_out of the call holds a reference
struct RowsOwner {
    std::vector<ChromState> st;
    std::atomic<int> refs{0};
};

}  // namespace

// Several chromosomes in ONE thread pool (a genome's chromosomes differ 5x in size: per-chromosome calls would leave threads idle
// at every chromosome's tail).  in[i] / out[i] describe chromosome i; `threads` workers in total.
extern "C" int phz_rows_format_multi(const phz_rows_in *in, int n_chroms, phz_rows_out *out, int threads) {
    if ((!in || !out) && n_chroms) return PHZ_E_ARG;
    if (n_chroms <= 0) return PHZ_OK;
    threads = std::max(1, threads);
    const bool timing = getenv("PHZ_TIMING") != nullptr;
    g_phase_timing = timing;
    auto t_prev = std::chrono::steady_clock::now();
    // wall time and, next to it, the CPU time the whole process spent in the lap: under a container CPU quota the second one is what
    // a pass really costs (wall time then depends on how much of the period's quota is left)
    struct Cpu { double user, sys; };
    auto cpu_now = []() {
        rusage ru; getrusage(RUSAGE_SELF, &ru);
        return Cpu{(double)ru.ru_utime.tv_sec + 1e-6 * (double)ru.ru_utime.tv_usec, (double)ru.ru_stime.tv_sec + 1e-6 * (double)ru.ru_stime.tv_usec};
    };
    Cpu cpu_prev = timing ? cpu_now() : Cpu{0, 0};
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        const Cpu cpu = cpu_now();
        fprintf(stderr, "[phz timing]   rows: %-42s %7.1f ms wall, cpu %7.1f ms user + %7.1f ms system\n", what,
                std::chrono::duration<double, std::milli>(now - t_prev).count(), (cpu.user - cpu_prev.user) * 1e3, (cpu.sys - cpu_prev.sys) * 1e3);
        t_prev = now; cpu_prev = cpu;
    };
    for (int c = 0; c < n_chroms; c++) memset(&out[c], 0, sizeof(out[c]));
    RowsOwner *keep = new RowsOwner();
    keep->st.resize((size_t)n_chroms);
    std::vector<ChromState> &st = keep->st;
    // ---- phase A: per-chromosome setup + component weights
    std::vector<int64_t> totals((size_t)n_chroms, 0);
    std::vector<const phz_rows_in *> use((size_t)n_chroms);
    parallel_chunks(threads, n_chroms, [&](int64_t c) {
        ChromState &S = st[(size_t)c];
        if (in[c].raw) { derive(in[c], S.P, S.I2); use[(size_t)c] = &S.I2; } else use[(size_t)c] = &in[c];
        const phz_rows_in &I = *use[(size_t)c];
        S.C.in = use[(size_t)c];
        S.C.uid = {I.uid_off, I.uid}; S.C.rsid = {I.rsid_off, I.rsid}; S.C.alle = {I.allele_off, I.allele}; S.C.maftxt = {I.maf_off, I.maf_txt};
        S.C.qname = {I.qname_off, I.qname};
        S.C.phased.assign((size_t)I.nv + 1, 0);
        S.w.resize((size_t)I.ncomp); S.eh.resize((size_t)I.ncomp); S.ea.resize((size_t)I.ncomp); S.ec.resize((size_t)I.ncomp);
        int64_t total = 0;
        const int64_t idlen = I.nv > 0 ? (int64_t)(I.uid_off[I.nv] / (uint32_t)I.nv) + 2 : 24;      // mean id text + separator
        for (int64_t r = 0; r < I.ncomp; r++) {
            const int64_t ci = I.comp_order[r];
            const int64_t sz = I.comp_ends[ci] - I.comp_starts[ci];
            int64_t x = 4, lines = 0;
            for (int64_t t = I.comp_starts[ci]; t < I.comp_ends[ci]; t++) { const int64_t l = S.C.lines_of((int)I.mem_s[t]); lines += l; x += 2 + l + sz; }
            S.w[(size_t)r] = x; total += x;
            S.eh[(size_t)r] = 160 + sz * (2 * idlen + 12);
            S.ea[(size_t)r] = (int64_t)I.nb * (200 + sz * (idlen + 8) + lines * 9);
            S.ec[(size_t)r] = sz * (sz - 1) * (4 * idlen + 8);
        }
        totals[(size_t)c] = total;
    });
    lap("ordering stage (per chromosome)");
    int64_t grand = 0;
    for (int64_t t : totals) grand += t;
    const int64_t target = std::max<int64_t>(4096, grand / ((int64_t)threads * 8) + 1);
    struct Task { int c; int kind; int64_t i; };      // kind 0: block chunk i, 1: connection chunk i
    std::vector<Task> tasks;
    const int64_t estep = 16384;
    for (int c = 0; c < n_chroms; c++) {
        const phz_rows_in &I = *use[(size_t)c];
        ChromState &S = st[(size_t)c];
        S.cb.assign(1, 0);
        int64_t acc = 0;
        for (int64_t r = 0; r < I.ncomp; r++) {
            acc += S.w[(size_t)r];
            if (acc >= target) { S.cb.push_back(r + 1); acc = 0; }
        }
        if (S.cb.back() != I.ncomp) S.cb.push_back(I.ncomp);
        S.bc.resize(S.cb.size() - 1);
        for (size_t i = 0; i + 1 < S.cb.size(); i++)
            for (int64_t r = S.cb[i]; r < S.cb[i + 1]; r++) {
                S.bc[i].est_hap += S.eh[(size_t)r]; S.bc[i].est_ase += S.ea[(size_t)r]; S.bc[i].est_cfg += S.ec[(size_t)r];
            }
        S.cc.resize((size_t)((I.n_edges + estep - 1) / estep));
        for (size_t i = 0; i < S.bc.size(); i++) tasks.push_back({c, 0, (int64_t)i});
        for (size_t i = 0; i < S.cc.size(); i++) tasks.push_back({c, 1, (int64_t)i});
    }
    // ---- phase B1: blocks (phasing + rows) and connection rows
    parallel_chunks(threads, (int64_t)tasks.size(), [&](int64_t t) {
        const Task &k = tasks[(size_t)t];
        ChromState &S = st[(size_t)k.c];
        if (k.kind == 0) run_block_chunk(S.C, S.cb[(size_t)k.i], S.cb[(size_t)k.i + 1], S.bc[(size_t)k.i]);
        else run_conn(S.C, k.i * estep, std::min<int64_t>(use[(size_t)k.c]->n_edges, (k.i + 1) * estep), S.cc[(size_t)k.i]);
    });
    lap("block phasing + block / connection rows");
    if (timing) fprintf(stderr, "  phz_rows:   of which phase_component (phase_v3 + brute force) %.4f thread-seconds over %d threads\n", g_phase_ns.exchange(0) / 1e9, threads);
    for (auto &S : st) for (auto &c : S.bc) if (c.status) { const int code = c.status; delete keep; return code; }
    // ---- phase B2: allelic counts + singleton rows (need to know which variants ended up in a block)
    tasks.clear();
    for (int c = 0; c < n_chroms; c++) {
        ChromState &S = st[(size_t)c];
        S.kb = key_chunks(*use[(size_t)c], 8192);
        S.ac.resize(S.kb.size() - 1); S.sc.resize(S.kb.size() - 1);
        for (size_t i = 0; i + 1 < S.kb.size(); i++) tasks.push_back({c, 2, (int64_t)i});
    }
    parallel_chunks(threads, (int64_t)tasks.size(), [&](int64_t t) {
        const Task &k = tasks[(size_t)t];
        ChromState &S = st[(size_t)k.c];
        run_allelic(S.C, S.kb[(size_t)k.i], S.kb[(size_t)k.i + 1], S.ac[(size_t)k.i]);
        if (use[(size_t)k.c]->unphased_vars == 1) run_singles(S.C, S.kb[(size_t)k.i], S.kb[(size_t)k.i + 1], S.sc[(size_t)k.i]);
    });
    lap("allelic counts + singleton rows");
    // ---- phase C: hand the chunk buffers over as they are (no concatenation); they stay alive through the shared owner
    bool nomem = false;
    auto parts = [&](const std::vector<const std::string *> &src, const std::vector<int32_t> *bam, phz_text_parts *P) {
        std::vector<const char *> ptr; std::vector<int64_t> len; std::vector<int32_t> bm;
        for (size_t i = 0; i < src.size(); i++) {
            if (src[i]->empty()) continue;
            ptr.push_back(src[i]->data()); len.push_back((int64_t)src[i]->size());
            if (bam) bm.push_back((*bam)[i]);
        }
        P->n = (int64_t)ptr.size();
        P->ptr = take_vec(ptr); P->len = take_vec(len); P->bam = bam ? take_vec(bm) : nullptr;
        if (!P->ptr || !P->len || (bam && !P->bam)) nomem = true;
    };
    for (int c = 0; c < n_chroms; c++) {
        const phz_rows_in &I = *use[(size_t)c];
        ChromState &S = st[(size_t)c];
        phz_rows_out &O = out[c];
        std::vector<const std::string *> p_hap, p_ase, p_cfg, p_conn, p_all, p_sa, p_sh;
        std::vector<int32_t> kbam;
        std::vector<int32_t> bvar, bsize, bmaxmaf; std::vector<uint8_t> bhap, bstat_int; std::vector<int8_t> bcor; std::vector<double> bstat;
        int64_t phased = 0;
        for (auto &x : S.bc) {
            p_hap.push_back(&x.hap); p_ase.push_back(&x.ase); p_cfg.push_back(&x.cfg); phased += x.phased;
            if (I.want_vcf) {
                bvar.insert(bvar.end(), x.bvar.begin(), x.bvar.end()); bhap.insert(bhap.end(), x.bhap.begin(), x.bhap.end());
                bcor.insert(bcor.end(), x.bcor.begin(), x.bcor.end()); bmaxmaf.insert(bmaxmaf.end(), x.bmaxmaf.begin(), x.bmaxmaf.end());
                bstat.insert(bstat.end(), x.bstat.begin(), x.bstat.end()); bstat_int.insert(bstat_int.end(), x.bstat_int.begin(), x.bstat_int.end());
            }
            bsize.insert(bsize.end(), x.bsize.begin(), x.bsize.end());
        }
        for (auto &x : S.cc) p_conn.push_back(&x.a);
        int64_t arows = 0;
        for (size_t i = 0; i + 1 < S.kb.size(); i++) {
            const int64_t b = S.kb[i] < I.n_keys ? I.key_bam[S.kb[i]] : 0;
            p_all.push_back(&S.ac[i].a); p_sa.push_back(&S.sc[i].a); p_sh.push_back(&S.sc[i].b); arows += S.ac[i].rows;
            kbam.push_back((int32_t)b);
        }
        parts(p_conn, nullptr, &O.conn); parts(p_hap, nullptr, &O.hap); parts(p_ase, nullptr, &O.ase); parts(p_cfg, nullptr, &O.cfg);
        parts(p_all, &kbam, &O.allelic); parts(p_sa, &kbam, &O.single_ase); parts(p_sh, &kbam, &O.single_hap);
        O.allelic_rows = arows;
        O.n_blocks = (int64_t)bsize.size(); O.phased = phased;
        O.blk_size = take_vec(bsize);
        O.n_blk_vars = (int64_t)bvar.size();
        O.blk_var = take_vec(bvar); O.blk_hap = take_vec(bhap); O.blk_cor = take_vec(bcor); O.blk_stat = take_vec(bstat);
        O.blk_stat_int = take_vec(bstat_int); O.blk_maxmaf = take_vec(bmaxmaf);
        O.owner = keep;
    }
    keep->refs = n_chroms;
    lap("hand-over of the text buffers");
    if (nomem) { for (int c = 0; c < n_chroms; c++) phz_rows_free(&out[c]); return PHZ_E_NOMEM; }
    return PHZ_OK;
}

extern "C" int phz_rows_format(const phz_rows_in *in, phz_rows_out *out) {
    if (!in || !out) return PHZ_E_ARG;
    return phz_rows_format_multi(in, 1, out, in->threads);
}

// phase_v3 on one connected component (variants position-sorted, local indices): the same routine phz_rows_format runs
// per component, exposed so that it can be checked against the reference restatement on arbitrary graphs
extern "C" int phz_phase_block(int32_t n, int64_t n_edges, const int32_t *edge_i, const int32_t *edge_j, const int8_t *edge_cfg,
                               int32_t max_block_size, int32_t *sub_first, int32_t *sub_len, char *config, int32_t *n_subs) {
    if (n <= 0 || n_edges < 0 || !sub_first || !sub_len || !config || !n_subs) return PHZ_E_ARG;
    AlleleGraph g;
    g.n = n; g.adj.assign((size_t)2 * n, {}); g.mark.assign((size_t)2 * n + 2, 0);
    for (int64_t e = 0; e < n_edges; e++) {
        const int i = edge_i[e], j = edge_j[e], k = edge_cfg[e];
        if (i < 0 || j < 0 || i >= n || j >= n || i == j) return PHZ_E_ARG;
        g.vedges.emplace_back(i, j);
        if (k == 0) {
            g.adj[2 * i].push_back(2 * j); g.adj[2 * j].push_back(2 * i);
            g.adj[2 * i + 1].push_back(2 * j + 1); g.adj[2 * j + 1].push_back(2 * i + 1);
        } else if (k == 1) {
            g.adj[2 * i].push_back(2 * j + 1); g.adj[2 * j + 1].push_back(2 * i);
            g.adj[2 * i + 1].push_back(2 * j); g.adj[2 * j].push_back(2 * i + 1);
        }
    }
    std::vector<std::pair<int, std::string>> subs;
    const int st = phase_component(g, max_block_size, subs);
    if (st) return st;
    int w = 0, k = 0;
    for (auto &s : subs) {
        int len = 0;
        for (size_t t = 0; t < s.second.size(); t++)
            if (s.first + (int)t < n) { config[w++] = s.second[t]; len++; }
        sub_first[k] = s.first; sub_len[k] = len; k++;
    }
    *n_subs = k;
    return PHZ_OK;
}

// The host half of the pair test's text: phaser.py:693 writes str(p) of the binomial p-value.  The caller evaluates scipy once per distinct
// (supporting, total) pair -- the used slots of phz_rowsdev_pair_keys -- and this lays the values and their repr() text out by slot, the form
// phz_rowsdev_run takes: slot_pv[s] = value (1.0 for an empty slot), the text of slot s = txt[txt_off[s], txt_off[s + 1] - 1) followed by one
// separator byte.  Returns the bytes written, or -1 when txt_cap is too small (n_slots + 32 bytes per used slot always fits).
extern "C" int64_t phz_pair_slot_text(const uint32_t *used, const double *pv, int64_t n_used, int64_t n_slots, double *slot_pv, uint32_t *txt_off,
                                      char *txt, int64_t txt_cap) {
    if (n_used < 0 || n_slots < 0 || (n_used && (!used || !pv)) || !slot_pv || !txt_off || !txt) return -1;
    for (int64_t sidx = 0; sidx < n_slots; sidx++) slot_pv[sidx] = 1.0;
    int64_t at = 0, next = 0;               // bytes written, next slot to lay out
    std::string r;
    for (int64_t i = 0; i < n_used; i++) {
        const int64_t sidx = used[i];
        if (sidx < next || sidx >= n_slots) return -1;             // ascending, inside the table
        if (at + (sidx - next) + 40 > txt_cap) return -1;
        for (; next < sidx; next++) { txt_off[next] = (uint32_t)at; txt[at++] = '\n'; }
        slot_pv[sidx] = pv[i];
        r.clear();
        phztext::put_pyfloat(r, pv[i]);
        txt_off[next++] = (uint32_t)at;
        memcpy(txt + at, r.data(), r.size()); at += (int64_t)r.size();
        txt[at++] = '\n';
    }
    if (at + (n_slots - next) > txt_cap) return -1;
    for (; next < n_slots; next++) { txt_off[next] = (uint32_t)at; txt[at++] = '\n'; }
    txt_off[n_slots] = (uint32_t)at;
    return at;
}

// host helper in front of it: the occupied slots of the pair-key table (keys as phz_rowsdev_pair_keys left them: total << 32 | supporting, all ones = empty), in slot
// order, as the arrays the caller hands to scipy.stats.binom.cdf (k as float64, n as int64: the argument types of the reference's call after numpy's conversion).
// Returns the number of occupied slots (<= n_slots = the capacity of every output array).
extern "C" int64_t phz_pair_slots_used(const uint64_t *keys, int64_t n_slots, uint32_t *used, double *sup, int64_t *tot) {
    if (n_slots < 0 || (n_slots && (!keys || !used || !sup || !tot))) return -1;
    int64_t n = 0;
    for (int64_t i = 0; i < n_slots; i++) {
        const uint64_t k = keys[i];
        if (k == ~0ull) continue;
        used[n] = (uint32_t)i; sup[n] = (double)(uint32_t)(k & 0xFFFFFFFFull); tot[n] = (int64_t)(k >> 32);
        n++;
    }
    return n;
}

extern "C" void phz_rows_free(phz_rows_out *o) {
    if (!o) return;
    for (phz_text_parts *P : {&o->conn, &o->hap, &o->ase, &o->cfg, &o->allelic, &o->single_ase, &o->single_hap}) {
        free((void *)P->ptr); free((void *)P->len); free((void *)P->bam);
    }
    free(o->blk_size); free(o->blk_var); free(o->blk_hap);
    free(o->blk_cor); free(o->blk_stat); free(o->blk_stat_int); free(o->blk_maxmaf);
    if (o->owner) {
        RowsOwner *k = (RowsOwner *)o->owner;
        if (k->refs.fetch_sub(1) == 1) delete k;
    }
    memset(o, 0, sizeof(*o));
}
