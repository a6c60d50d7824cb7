// Native SAM-text front end of the mapper seam (Seam 1): what phaser/read_variant_map.py:25-64 does per input line -- header
// contigs, field split, |TLEN| filter, last AS: tag -- plus the packing of soa.pack_sam (normalised CIGAR ops, 2-bit bases, quality
// bytes with the non-ACGT flag), multi-threaded, and the mapper's output lines (read_variant_map.py:117) formatted from the K_map
// call list.  `python -m phaser_amd.call_read_variant_map` streams stdin through these instead of per-line Python.
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

namespace {

using phztext::put_int;

struct Rec {
    const char *line;             // start of the record's line
    uint32_t f[12];               // offsets of the first 11 fields + end of field 10, relative to `line`
    uint32_t len;                 // line length after rstrip
    int32_t pos;
    int32_t as; uint8_t has_as;
    uint32_t n_ops, nb;
};

struct SamShard {
    std::string name;
    std::vector<Rec> recs;
    std::vector<int32_t> pos, aln;
    std::vector<uint32_t> cigar_off, cigar, seq_off, qname_off;
    std::vector<uint8_t> has_as, seq2, qual;
    std::vector<char> qnames;
};

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

// Python int(): optional sign, digits (the reference would raise on anything else; so do we, as a status)
bool parse_int(const char *p, const char *e, long long *out) {
    while (p < e && is_space(*p)) p++;
    while (e > p && is_space(e[-1])) e--;
    if (p == e) return false;
    bool neg = false;
    if (*p == '+' || *p == '-') { neg = *p == '-'; p++; }
    if (p == e) return false;
    long long v = 0;
    for (; p < e; p++) {
        if (*p == '_') continue;                       // int("1_000") == 1000
        if (*p < '0' || *p > '9') return false;
        v = v * 10 + (*p - '0');
        if (v > (1ll << 40)) return false;
    }
    *out = neg ? -v : v;
    return true;
}

constexpr int OPC_M = 0, OPC_I = 1, OPC_D = 2, OPC_N = 3, OPC_S = 4, OPC_EQ = 7, OPC_X = 8, OPC_G = 9;

inline int op_code(char c) {
    switch (c) { case 'M': return 0; case 'I': return 1; case 'D': return 2; case 'N': return 3; case 'S': return 4; case 'H': return 5;
                 case 'P': return 6; case '=': return 7; case 'X': return 8; default: return -1; }
}

// soa.pack_sam's op normalisation on CIGAR text: ops written to out (may be null to count)
int norm_ops_text(const char *c, const char *e, long nb, uint32_t *out) {
    int n = 0;
    long read_pos = 0;
    unsigned long long num = 0;
    for (; c < e; c++) {
        if (*c >= '0' && *c <= '9') { num = num * 10 + (unsigned)(*c - '0'); if (num > (1ull << 28)) num = (1ull << 28); continue; }
        const int op = op_code(*c);
        const uint32_t len = (uint32_t)num;
        num = 0;
        if (op == OPC_M || op == OPC_EQ || op == OPC_X) {
            const long lo = read_pos < nb ? read_pos : nb, hi = read_pos + (long)len < nb ? read_pos + (long)len : nb;
            const uint32_t avail = (uint32_t)(hi > lo ? hi - lo : 0);
            if (avail == len) { if (out) out[n] = (len << 4) | (uint32_t)op; n++; }
            else {
                if (avail) { if (out) out[n] = (avail << 4) | (uint32_t)op; n++; }
                if (out) out[n] = ((len - avail) << 4) | (uint32_t)OPC_G; n++;
            }
            read_pos += len;
        } else if (op == OPC_I) {
            const long lo = read_pos < nb ? read_pos : nb, hi = read_pos + (long)len < nb ? read_pos + (long)len : nb;
            const uint32_t avail = (uint32_t)(hi > lo ? hi - lo : 0);
            if (out) out[n] = (avail << 4) | 1u; n++;
            read_pos += len;
        } else if (op == OPC_S) {
            if (out) out[n] = (len << 4) | 4u; n++;
            read_pos += len;
        } else if (op == OPC_D || op == OPC_N) {
            if (out) out[n] = (len << 4) | (uint32_t)op; n++;
        }
        // H, P and unknown characters have no effect in the reference (:227-229)
    }
    return n;
}

int n_threads(int want) {
    if (want > 0) return want;
    unsigned h = std::thread::hardware_concurrency();
    return h ? (int)(h > 32 ? 32 : h) : 4;
}

template <class F>
void par_for(int nt, size_t n, F fn) {
    if (n == 0) return;
    nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)nt, (n + 4095) / 4096));
    if (nt == 1) { fn((size_t)0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([&, t] { fn(n * (size_t)t / (size_t)nt, n * (size_t)(t + 1) / (size_t)nt); });
    for (auto &x : th) x.join();
}

}  // namespace

struct phz_sam {
    std::vector<std::string> contigs;           // @SQ names in header order
    std::vector<SamShard> shards;               // chromosomes in first-appearance order
    int64_t n_records = 0;                      // alignment lines read (before the TLEN filter), like the mapper's read_counter
    int stream_order = 1;                       // 1: every chromosome is ONE run of the stream, in coordinate order, counting ALL alignment lines
    std::string err;
    const char *text = nullptr;
};

extern "C" {

int phz_sam_parse(const char *text, int64_t len, double isize_cutoff, int threads, phz_sam **out) {
    if (!text || len < 0 || !out) return PHZ_E_ARG;
    phz_sam *h = new phz_sam();
    *out = h;
    h->text = text;
    const int nt = n_threads(threads);
    // ---- line table
    std::vector<int64_t> ls(1, 0);
    for (const char *p = text, *e = text + len; p < e;) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
        if (!nl) { ls.push_back(len + 1); break; }
        ls.push_back((int64_t)(nl - text) + 1); p = nl + 1;
    }
    const size_t nlines = ls.size() - 1;
    // ---- per line: classify, split, filter (parallel over line ranges, merged in order)
    // runs: the chromosome runs of ALL records of the part (also the ones the isize filter drops: the reference prunes its variant buffer by
    // every record, read_variant_map.py:37-50), with their first / last POS -- the stream is taken here only when every chromosome is one run
    // in coordinate order; anything else is the Python path's business (forward-only buffer semantics / refusal)
    struct Run { std::string_view chrom; int32_t first, last; };
    struct Part { std::vector<Rec> recs; std::vector<std::string_view> chroms; std::vector<std::string> contigs; std::vector<Run> runs; bool disorder = false;
                  int64_t n = 0; int status = 0; std::string err; };
    const size_t nparts = nlines ? (size_t)std::max(1, std::min<int>(nt * 4, (int)((nlines + 2047) / 2048))) : 0;
    std::vector<Part> parts(nparts);
    std::atomic<size_t> next(0);
    auto work = [&] {
        for (;;) {
            const size_t pi = next.fetch_add(1);
            if (pi >= nparts) break;
            Part &P = parts[pi];
            for (size_t li = nlines * pi / nparts; li < nlines * (pi + 1) / nparts && !P.status; li++) {
                const char *p = text + ls[li]; const char *e = text + ls[li + 1] - 1;
                if (e > text + len) e = text + len;
                while (e > p && is_space(e[-1])) e--;                       // line.rstrip()
                if (e <= p) {                                               // "" -> columns[0] == "": treated as a record by the reference and
                    P.status = PHZ_E_ARG; P.err = "empty line in SAM input"; break;      // crashing there (IndexError); a status here
                }
                if (*p == '@') {
                    if (e - p >= 3 && p[1] == 'S' && p[2] == 'Q') {
                        // contigs.append(columns[1].split(":")[1])
                        const char *t1 = (const char *)memchr(p, '\t', (size_t)(e - p));
                        if (!t1) { P.status = PHZ_E_ARG; P.err = "@SQ line without fields"; break; }
                        const char *f1 = t1 + 1;
                        const char *t2 = (const char *)memchr(f1, '\t', (size_t)(e - f1));
                        const char *f1e = t2 ? t2 : e;
                        const char *c1 = (const char *)memchr(f1, ':', (size_t)(f1e - f1));
                        if (!c1) { P.status = PHZ_E_ARG; P.err = "@SQ field without ':'"; break; }
                        const char *c2 = (const char *)memchr(c1 + 1, ':', (size_t)(f1e - c1 - 1));
                        P.contigs.emplace_back(c1 + 1, (size_t)((c2 ? c2 : f1e) - c1 - 1));
                    }
                    continue;
                }
                P.n++;
                Rec r;
                r.line = p; r.len = (uint32_t)(e - p);
                int nf = 0;
                const char *q = p;
                r.f[0] = 0;
                while (nf < 11) {
                    const char *t = (const char *)memchr(q, '\t', (size_t)(e - q));
                    if (!t) break;
                    nf++;
                    r.f[nf] = (uint32_t)(t + 1 - p);
                    q = t + 1;
                }
                if (nf < 10) { P.status = PHZ_E_ARG; P.err = "SAM record with fewer than 11 fields"; break; }
                if (nf == 10) r.f[11] = r.len + 1;                          // no optional fields: field 10 ends with the line
                long long v;
                if (!parse_int(p + r.f[3], p + r.f[4] - 1, &v)) { P.status = PHZ_E_ARG; P.err = "SAM POS is not an integer"; break; }
                r.pos = (int32_t)v;
                if (!parse_int(p + r.f[8], p + r.f[9] - 1, &v)) { P.status = PHZ_E_ARG; P.err = "SAM TLEN is not an integer"; break; }
                const double tl = v < 0 ? -(double)v : (double)v;
                {
                    const std::string_view cn(p + r.f[2], (size_t)(r.f[3] - 1 - r.f[2]));
                    if (P.runs.empty() || P.runs.back().chrom != cn) P.runs.push_back(Run{cn, r.pos, r.pos});
                    else { if (r.pos < P.runs.back().last) P.disorder = true; P.runs.back().last = r.pos; }
                }
                if (!(isize_cutoff == 0 || tl <= isize_cutoff)) continue;
                // AS = last optional field that starts with "AS:" -> int(field.split(":")[2])
                r.has_as = 0; r.as = 0;
                if (nf == 11) {
                    const char *o = p + r.f[11];
                    while (o < e) {
                        const char *t = (const char *)memchr(o, '\t', (size_t)(e - o));
                        const char *oe = t ? t : e;
                        if (oe - o >= 3 && o[0] == 'A' && o[1] == 'S' && o[2] == ':') {
                            const char *c2 = (const char *)memchr(o + 3, ':', (size_t)(oe - o - 3));
                            if (!c2) { P.status = PHZ_E_ARG; P.err = "AS tag without a value"; break; }
                            const char *c3 = (const char *)memchr(c2 + 1, ':', (size_t)(oe - c2 - 1));
                            if (!parse_int(c2 + 1, c3 ? c3 : oe, &v)) { P.status = PHZ_E_ARG; P.err = "AS value is not an integer"; break; }
                            r.as = (int32_t)v; r.has_as = 1;
                        }
                        if (!t) break;
                        o = t + 1;
                    }
                    if (P.status) break;
                }
                const uint32_t l_seq = r.f[10] - 1 - r.f[9], l_qual = r.f[11] - 1 - r.f[10];
                r.nb = l_seq < l_qual ? l_seq : l_qual;
                r.n_ops = (uint32_t)norm_ops_text(p + r.f[5], p + r.f[6] - 1, (long)r.nb, nullptr);
                P.recs.push_back(r);
                P.chroms.emplace_back(p + r.f[2], (size_t)(r.f[3] - 1 - r.f[2]));
            }
        }
    };
    if (nt == 1 || nparts <= 1) work();
    else { std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(work); for (auto &x : th) x.join(); }
    {
        std::unordered_map<std::string, int> seen;
        std::string cur; int32_t last = 0; bool any = false;
        for (auto &P : parts) {
            if (P.status) { h->err = P.err; return P.status; }
            if (P.disorder) h->stream_order = 0;
            for (auto &r : P.runs) {
                if (any && cur == r.chrom) { if (r.first < last) h->stream_order = 0; }
                else {
                    cur.assign(r.chrom);
                    if (!seen.emplace(cur, 1).second) h->stream_order = 0;       // a chromosome that comes back
                }
                last = r.last; any = true;
            }
        }
    }
    std::unordered_map<std::string, int> idx;
    for (auto &P : parts) {
        if (P.status) { h->err = P.err; return P.status; }
        h->n_records += P.n;
        for (auto &c : P.contigs) h->contigs.push_back(c);
        for (size_t i = 0; i < P.recs.size(); i++) {
            std::string name(P.chroms[i]);
            auto it = idx.find(name);
            int si;
            if (it == idx.end()) { si = (int)h->shards.size(); idx.emplace(name, si); h->shards.emplace_back(); h->shards.back().name = name; }
            else si = it->second;
            h->shards[(size_t)si].recs.push_back(P.recs[i]);
        }
        std::vector<Rec>().swap(P.recs);
    }
    // ---- pack every chromosome (soa.pack_sam): offsets by prefix sum, records in parallel
    for (auto &S : h->shards) {
        const size_t m = S.recs.size();
        for (size_t i = 1; i < m; i++)
            if (S.recs[i].pos < S.recs[i - 1].pos) { h->err = "records are not coordinate-sorted"; return PHZ_E_UNSUPPORTED; }
        S.pos.resize(m); S.aln.resize(m); S.has_as.resize(m);
        S.cigar_off.resize(m + 1); S.seq_off.resize(m + 1); S.qname_off.resize(m + 1);
        uint64_t co = 0, so = 0, qo = 0;
        for (size_t i = 0; i < m; i++) {
            const Rec &r = S.recs[i];
            S.cigar_off[i] = (uint32_t)co; S.seq_off[i] = (uint32_t)so; S.qname_off[i] = (uint32_t)qo;
            co += r.n_ops; so += (r.nb + 3) / 4; qo += r.f[1] - 1;
        }
        if (co >= (1ull << 31) || so >= (1ull << 31) || qo >= (1ull << 32)) { h->err = "shard exceeds 32-bit offsets"; return PHZ_E_UNSUPPORTED; }
        S.cigar_off[m] = (uint32_t)co; S.seq_off[m] = (uint32_t)so; S.qname_off[m] = (uint32_t)qo;
        S.cigar.resize(co); S.seq2.assign(so, 0); S.qual.assign(so * 4, 0); S.qnames.resize(qo);
        par_for(nt, m, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) {
                const Rec &r = S.recs[i];
                const char *p = r.line;
                S.pos[i] = r.pos; S.aln[i] = r.has_as ? r.as : 0; S.has_as[i] = r.has_as;
                memcpy(S.qnames.data() + S.qname_off[i], p, r.f[1] - 1);
                norm_ops_text(p + r.f[5], p + r.f[6] - 1, (long)r.nb, S.cigar.data() + S.cigar_off[i]);
                const char *sq = p + r.f[9], *ql = p + r.f[10];
                uint8_t *o2 = S.seq2.data() + S.seq_off[i];
                uint8_t *oq = S.qual.data() + (size_t)S.seq_off[i] * 4;
                for (uint32_t j = 0; j < r.nb; j++) {
                    int q = (int)(unsigned char)ql[j] - 33;
                    q = q < 0 ? 0 : (q > 127 ? 127 : q);
                    uint8_t code;
                    switch (sq[j]) {
                        case 'A': code = 0; break; case 'C': code = 1; break; case 'G': code = 2; break; case 'T': code = 3; break;
                        case 'N': case 'D': code = 0; q |= 0x80; break;         // behaves like N (an IUPAC 'D' is stripped downstream: same outcome)
                        default: code = 1; q |= 0x80; break;                    // any other character: a call that matches no allele
                    }
                    o2[j >> 2] |= (uint8_t)(code << (2 * (j & 3)));
                    oq[j] = (uint8_t)q;
                }
            }
        });
    }
    return PHZ_OK;
}

const char *phz_sam_error(const phz_sam *h) { return h ? h->err.c_str() : ""; }
void phz_sam_free(phz_sam *h) { delete h; }
int64_t phz_sam_n_records(const phz_sam *h) { return h->n_records; }
int phz_sam_stream_order(const phz_sam *h) { return h ? h->stream_order : 0; }
int phz_sam_n_contigs(const phz_sam *h) { return (int)h->contigs.size(); }
const char *phz_sam_contig(const phz_sam *h, int i) { return h->contigs[(size_t)i].c_str(); }
int phz_sam_n_shards(const phz_sam *h) { return (int)h->shards.size(); }

int phz_sam_shard(phz_sam *h, int i, phz_host_shard *out) {
    if (!h || !out || i < 0 || (size_t)i >= h->shards.size()) return PHZ_E_ARG;
    SamShard &s = h->shards[(size_t)i];
    out->ref_name = s.name.c_str();
    out->n_reads = (int64_t)s.pos.size(); out->n_ops = (int64_t)s.cigar.size(); out->n_seq_bytes = (int64_t)s.seq2.size();
    out->pos = s.pos.data(); out->cigar_off = s.cigar_off.data(); out->cigar = s.cigar.data(); out->seq_off = s.seq_off.data();
    out->seq2 = s.seq2.data(); out->qual = s.qual.data(); out->aln_score = s.aln.data(); out->has_as = s.has_as.data();
    out->qname_off = s.qname_off.data(); out->qnames = s.qnames.data();
    return PHZ_OK;
}

// The mapper's output lines for one chromosome (read_variant_map.py:117): qname, id, rsid, allele, AS, genotype, maf.
// id / rsid / gt / maf: sep pools over the chromosome's table rows (item v = bytes [off[v], off[v+1]-1)).  Single-base calls print
// from `code`; composite ones (code 4) copy the read characters aux0 / aux1 point at, masked by baseq, 'D' stripped.
int phz_sam_calls_tsv(const phz_sam *h, int shard, int64_t n_calls, const int32_t *read_idx, const int32_t *var_idx, const uint8_t *code,
                      const uint32_t *aux0, const uint32_t *aux1, int baseq, const uint32_t *id_off, const char *id,
                      const uint32_t *rsid_off, const char *rsid, const uint32_t *gt_off, const char *gt, const uint32_t *maf_off,
                      const char *maf, int threads, char **out, int64_t *out_len) {
    if (!h || shard < 0 || (size_t)shard >= h->shards.size() || n_calls < 0 || !out || !out_len) return PHZ_E_ARG;
    const SamShard &S = h->shards[(size_t)shard];
    const int nt = n_threads(threads);
    const size_t nchunks = (size_t)std::max<int64_t>(1, std::min<int64_t>((int64_t)nt * 4, (n_calls + 8191) / 8192));
    std::vector<std::string> parts(nchunks);
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    auto item = [](const uint32_t *off, const char *b, int64_t v) { return std::string_view(b + off[v], off[v + 1] - off[v] - 1); };
    auto work = [&] {
        for (;;) {
            const size_t ci = next.fetch_add(1);
            if (ci >= nchunks) break;
            std::string &T = parts[ci];
            for (int64_t k = n_calls * (int64_t)ci / (int64_t)nchunks; k < n_calls * (int64_t)(ci + 1) / (int64_t)nchunks; k++) {
                const int64_t r = read_idx[k], v = var_idx[k];
                if (r < 0 || (size_t)r >= S.recs.size()) { bad = 1; continue; }
                const Rec &R = S.recs[(size_t)r];
                const char *p = R.line;
                T.append(p, R.f[1] - 1); T += '\t';
                T.append(item(id_off, id, v)); T += '\t'; T.append(item(rsid_off, rsid, v)); T += '\t';
                if (code[k] < 4) T += "ACGT"[code[k]];
                else {
                    const char *sq = p + R.f[9], *ql = p + R.f[10];
                    auto ch = [&](uint32_t x) { if (x >= R.nb) return; const char c = ((int)(unsigned char)ql[x] - 33) >= baseq ? sq[x] : 'N'; if (c != 'D') T += c; };
                    if (aux0[k] != 0xFFFFFFFFu) ch(aux0[k]);
                    const uint32_t ilen = aux1[k] & 0xFFF, ioff = aux1[k] >> 12;
                    for (uint32_t x = ioff; x < ioff + ilen; x++) ch(x);
                }
                T += '\t';
                if (R.has_as) put_int(T, R.as);
                T += '\t'; T.append(item(gt_off, gt, v)); T += '\t'; T.append(item(maf_off, maf, v)); T += '\n';
            }
        }
    };
    if (nt == 1 || nchunks == 1) work();
    else { std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(work); for (auto &x : th) x.join(); }
    if (bad) return PHZ_E_ARG;
    size_t total = 0;
    for (auto &s : parts) total += s.size();
    char *buf = (char *)malloc(total + 1);
    if (!buf) return PHZ_E_NOMEM;
    size_t off = 0;
    for (auto &s : parts) { memcpy(buf + off, s.data(), s.size()); off += s.size(); }
    buf[total] = 0;
    *out = buf; *out_len = (int64_t)total;
    return PHZ_OK;
}

}  // extern "C"
