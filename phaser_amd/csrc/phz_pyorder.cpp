// Raw-byte tier, native (SURVEY.md 8(a) "T1"; the Python twin is phaser_amd/pyorder.py): rows and read labels of three of phASER's five files in the
// order CPython 3.10 gives the reference's `set` objects of strings (phaser/phaser.py:660-678, :930, :1059, :1086, :1106-1115, :1181-1239) when it
// hashes strings deterministically (PYTHONHASHSEED=0: how the golden files were written).  Nothing of CPython is linked or called: the two pieces of
// the interpreter the order depends on are restated here --
//   * the hash of a str: SipHash-2-4 over the string's bytes with the all-zero key a disabled hash randomisation leaves (Python/pyhash.c, the default
//     algorithm up to 3.10), -1 mapped to -2;
//   * the set: open addressing over a power-of-two table of (key, hash) entries, LINEAR_PROBES = 9 neighbours tried before the perturbed jump
//     i = i * 5 + 1 + (perturb >>= 5), growth to 4 x used (2 x beyond 50,000) once fill * 5 >= mask * 3, re-insertion in table order on growth,
//     iteration in table order; `a - b` iterates a and adds what b lacks to a fresh set, or (len(a) / 4 > len(b)) copies a and discards b's keys
//     (Objects/setobject.c: set_add_entry, set_table_resize, set_insert_clean, set_merge, set_difference).
// -- and the replay itself is the same sequence of insertions pyorder.replay makes with real sets.  tests/test_pyorder.py checks hash and set against
// the running interpreter and the replay against the Python twin and the reference's bytes on every fixture.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <string_view>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "phz.h"

namespace {

// ------------------------------------------------------------------------------------------------ CPython's str hash (seed 0)
inline uint64_t rotl(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }
#define PHZ_SIPROUND do { v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32); v2 += v3; v3 = rotl(v3, 16); v3 ^= v2; \
                          v0 += v3; v3 = rotl(v3, 21); v3 ^= v0; v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32); } while (0)
int64_t py_str_hash(const char *s, size_t len) {
    if (len == 0) return 0;
    const uint64_t k0 = 0, k1 = 0;
    uint64_t b = (uint64_t)len << 56;
    uint64_t v0 = k0 ^ 0x736f6d6570736575ULL, v1 = k1 ^ 0x646f72616e646f6dULL, v2 = k0 ^ 0x6c7967656e657261ULL, v3 = k1 ^ 0x7465646279746573ULL;
    const uint8_t *in = (const uint8_t *)s;
    size_t left = len;
    while (left >= 8) {
        uint64_t mi; memcpy(&mi, in, 8);          // little endian
        in += 8; left -= 8;
        v3 ^= mi; PHZ_SIPROUND; PHZ_SIPROUND; v0 ^= mi;
    }
    uint64_t t = 0;
    for (size_t i = 0; i < left; i++) t |= (uint64_t)in[i] << (8 * i);
    b |= t;
    v3 ^= b; PHZ_SIPROUND; PHZ_SIPROUND; v0 ^= b;
    v2 ^= 0xff;
    PHZ_SIPROUND; PHZ_SIPROUND; PHZ_SIPROUND; PHZ_SIPROUND;
    const int64_t h = (int64_t)((v0 ^ v1) ^ (v2 ^ v3));
    return h == -1 ? -2 : h;
}

// ------------------------------------------------------------------------------------------------ CPython 3.10's set, keys = ids of strings
// An entry holds the id of a string (>= 0), its hash, and the state: EMPTY (never used), ACTIVE, DUMMY (discarded).  Two ids are the same key when
// equal, or when their hashes and strings are equal (`eq`, given by the owner of the strings).
struct PySet {
    static constexpr int LINEAR_PROBES = 9, PERTURB_SHIFT = 5, MINSIZE = 8;
    struct Entry { int64_t hash; int32_t key; uint8_t state; };
    enum { EMPTY = 0, ACTIVE = 1, DUMMY = 2 };
    std::vector<Entry> table;
    size_t mask = MINSIZE - 1, fill = 0, used = 0;
    PySet() { table.assign(MINSIZE, Entry{0, -1, EMPTY}); }

    template <class Eq> bool contains(int32_t key, int64_t hash, Eq eq) const {
        size_t perturb = (size_t)hash, i = (size_t)hash & mask;
        for (;;) {
            const Entry *e = &table[i];
            int probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
            do {
                if (e->state == EMPTY) return false;
                if (e->state == ACTIVE && e->hash == hash && (e->key == key || eq(e->key, key))) return true;
                e++;
            } while (probes--);
            perturb >>= PERTURB_SHIFT;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    static void insert_clean(std::vector<Entry> &t, size_t m, int32_t key, int64_t hash) {
        size_t perturb = (size_t)hash, i = (size_t)hash & m;
        for (;;) {
            Entry *e = &t[i];
            if (e->state == EMPTY) { *e = Entry{hash, key, ACTIVE}; return; }
            if (i + LINEAR_PROBES <= m)
                for (int j = 0; j < LINEAR_PROBES; j++) { e++; if (e->state == EMPTY) { *e = Entry{hash, key, ACTIVE}; return; } }
            perturb >>= PERTURB_SHIFT;
            i = (i * 5 + 1 + perturb) & m;
        }
    }
    void resize(size_t minused) {
        size_t newsize = MINSIZE;
        while (newsize <= minused) newsize <<= 1;
        std::vector<Entry> nt(newsize, Entry{0, -1, EMPTY});
        for (const Entry &e : table) if (e.state == ACTIVE) insert_clean(nt, newsize - 1, e.key, e.hash);
        table.swap(nt); mask = newsize - 1; fill = used;
    }
    template <class Eq> void add(int32_t key, int64_t hash, Eq eq) {
        size_t perturb = (size_t)hash, i = (size_t)hash & mask;
        Entry *freeslot = nullptr;
        for (;;) {
            Entry *e = &table[i];
            int probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
            do {
                if (e->state == EMPTY) {
                    if (freeslot) { *freeslot = Entry{hash, key, ACTIVE}; used++; return; }
                    *e = Entry{hash, key, ACTIVE}; fill++; used++;
                    if (fill * 5 < mask * 3) return;
                    resize(used > 50000 ? used * 2 : used * 4);
                    return;
                }
                if (e->state == ACTIVE) { if (e->hash == hash && (e->key == key || eq(e->key, key))) return; }
                else freeslot = e;          // (CPython keeps the LAST dummy seen on the probe path)
                e++;
            } while (probes--);
            perturb >>= PERTURB_SHIFT;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    template <class Eq> void discard(int32_t key, int64_t hash, Eq eq) {
        size_t perturb = (size_t)hash, i = (size_t)hash & mask;
        for (;;) {
            Entry *e = &table[i];
            int probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
            do {
                if (e->state == EMPTY) return;
                if (e->state == ACTIVE && e->hash == hash && (e->key == key || eq(e->key, key))) { e->state = DUMMY; e->key = -1; e->hash = -1; used--; return; }
                e++;
            } while (probes--);
            perturb >>= PERTURB_SHIFT;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    // set(other) -- set_merge into an empty set
    void copy_from(const PySet &o) {
        if (o.used == 0) return;
        if ((fill + o.used) * 5 >= mask * 3) resize((used + o.used) * 2);
        if (fill == 0 && mask == o.mask && o.fill == o.used) { table = o.table; fill = o.fill; used = o.used; return; }
        fill = o.used; used = o.used;
        for (const Entry &e : o.table) if (e.state == ACTIVE) insert_clean(table, mask, e.key, e.hash);
    }
    template <class F> void for_each(F f) const { for (const Entry &e : table) if (e.state == ACTIVE) f(e.key); }
};

// a - b (set_difference)
template <class Eq> PySet py_set_difference(const PySet &a, const PySet &b, Eq eq) {
    PySet r;
    if ((a.used >> 2) > b.used) {          // copy a, discard what b holds (iterating b)
        r.copy_from(a);
        for (const PySet::Entry &e : b.table) if (e.state == PySet::ACTIVE) r.discard(e.key, e.hash, eq);
        // set_difference_update_internal: "If more than 1/4th are dummies, then resize them away"
        if ((r.fill - r.used) > r.mask / 4) r.resize(r.used > 50000 ? r.used * 2 : r.used * 4);
        return r;
    }
    for (const PySet::Entry &e : a.table) if (e.state == PySet::ACTIVE && !b.contains(e.key, e.hash, eq)) r.add(e.key, e.hash, eq);
    return r;
}

// ------------------------------------------------------------------------------------------------ string pools
struct Pool {
    const char *b = nullptr; const uint32_t *off = nullptr; int64_t n = 0; int sep = 0;        // sep 1: every item is followed by one separator byte
    std::string_view at(int64_t i) const { return std::string_view(b + off[i], (size_t)(off[i + 1] - off[i]) - (size_t)sep); }
};

std::vector<std::string_view> split(std::string_view s, char c) {
    std::vector<std::string_view> out;
    size_t p = 0;
    for (;;) {
        const size_t q = s.find(c, p);
        if (q == std::string_view::npos) { out.push_back(s.substr(p)); break; }
        out.push_back(s.substr(p, q - p)); p = q + 1;
    }
    return out;
}
void put_int(std::string &o, long long v) { char buf[24]; const int n = snprintf(buf, sizeof buf, "%lld", v); o.append(buf, (size_t)n); }

struct PairHash { size_t operator()(const std::pair<int64_t, int64_t> &p) const { return std::hash<int64_t>()(p.first * 0x9E3779B97F4A7C15LL ^ p.second); } };

}  // namespace

struct phz_pyorder { std::string conn, hap, ase, error; };

extern "C" int64_t phz_py_str_hash(const char *s, int64_t len) { return py_str_hash(s, (size_t)(len < 0 ? 0 : len)); }

// Test hook: the iteration order of set(items) followed by optional discards, and of (set(items) - set(minus)): what the emulation above gives for the
// strings of a pool.  order_out receives item indices (first occurrence of equal strings); returns their number.  mode 0: set(items[0..n));
// mode 1: set(items[0..n_a)) - set(items[n_a..n))
extern "C" int64_t phz_py_set_order(const char *blob, const uint32_t *off, int64_t n, int64_t n_a, int32_t mode, int32_t *order_out) {
    Pool P; P.b = blob; P.off = off; P.n = n; P.sep = 0;
    std::vector<int64_t> h((size_t)n);
    for (int64_t i = 0; i < n; i++) { const std::string_view s = P.at(i); h[(size_t)i] = py_str_hash(s.data(), s.size()); }
    auto eq = [&](int32_t x, int32_t y) { return P.at(x) == P.at(y); };
    PySet a, b;
    for (int64_t i = 0; i < (mode ? n_a : n); i++) a.add((int32_t)i, h[(size_t)i], eq);
    int64_t k = 0;
    if (!mode) { a.for_each([&](int32_t key) { order_out[k++] = key; }); return k; }
    for (int64_t i = n_a; i < n; i++) b.add((int32_t)i, h[(size_t)i], eq);
    PySet d = py_set_difference(a, b, eq);
    d.for_each([&](int32_t key) { order_out[k++] = key; });
    return k;
}

extern "C" void phz_pyorder_free(phz_pyorder *h) { delete h; }
extern "C" const char *phz_pyorder_error(const phz_pyorder *h) { return h ? h->error.c_str() : ""; }
extern "C" int phz_pyorder_text(const phz_pyorder *h, int which, const char **text, int64_t *len) {
    if (!h || !text || !len || which < 0 || which > 2) return PHZ_E_ARG;
    const std::string &s = which == 0 ? h->conn : (which == 1 ? h->hap : h->ase);
    *text = s.data(); *len = (int64_t)s.size();
    return PHZ_OK;
}

extern "C" int phz_pyorder_replay(const phz_pyorder_in *I, const char *conn, int64_t conn_len, const char *hap, int64_t hap_len, const char *ase, int64_t ase_len,
                                  phz_pyorder **out) {
    if (!I || !out || !conn || !hap || !ase || I->n_chroms < 0 || I->n_bams < 1) return PHZ_E_ARG;
    phz_pyorder *H = new phz_pyorder();
    *out = H;
    auto fail = [&](const char *m) { H->error = m; return PHZ_E_ARG; };
    const int NC = I->n_chroms, NB = I->n_bams;
    // ---- joint variant space (chromosomes in VCF order), pools, hashes of the uid strings
    std::vector<int64_t> vbase((size_t)NC + 1, 0);
    std::vector<Pool> uidp((size_t)NC), namep((size_t)NC), allep((size_t)NC), rsidp((size_t)NC);
    for (int c = 0; c < NC; c++) {
        vbase[(size_t)c + 1] = vbase[(size_t)c] + I->nv[c];
        uidp[(size_t)c] = Pool{I->uid[c], I->uid_off[c], I->nv[c], 1};
        allep[(size_t)c] = Pool{I->allele2[c], I->allele2_off[c], 2 * I->nv[c], 1};
        rsidp[(size_t)c] = Pool{I->rsid[c], I->rsid_off[c], I->nv[c], 1};
        namep[(size_t)c] = Pool{I->qname[c], I->qname_off[c], I->nq[c], 0};
    }
    const int64_t NV = vbase[(size_t)NC];
    std::vector<int32_t> chrom_of((size_t)NV);
    std::vector<int64_t> uhash((size_t)NV);
    for (int c = 0; c < NC; c++)
        for (int64_t v = 0; v < I->nv[c]; v++) {
            const int64_t g = vbase[(size_t)c] + v;
            chrom_of[(size_t)g] = c;
            const std::string_view s = uidp[(size_t)c].at(v);
            uhash[(size_t)g] = py_str_hash(s.data(), s.size());
        }
    auto uid_of = [&](int64_t g) { const int c = chrom_of[(size_t)g]; return uidp[(size_t)c].at(g - vbase[(size_t)c]); };
    auto ueq = [&](int32_t x, int32_t y) { return uid_of(x) == uid_of(y); };
    if (NV >= (1ll << 31)) return fail("too many variants");
    std::unordered_map<std::string_view, int32_t> gid_of;
    gid_of.reserve((size_t)NV * 2);
    for (int64_t g = 0; g < NV; g++) gid_of.emplace(uid_of(g), (int32_t)g);          // (the first of two variants with one uid string wins, as a dict key would)

    // ---- dict_variant_reads order (rule 2), read_vars (rule 3 + the overwrite of :576-581), haplo_reads[(uid, allele)][bam]
    std::vector<int32_t> dvr;                                   // variants in first-appearance order
    std::vector<uint8_t> in_dvr((size_t)NV, 0);
    struct RV { std::vector<int32_t> order; std::unordered_map<int32_t, int32_t> pos; std::vector<std::vector<int32_t>> lists; };      // QNAME id -> list of variants
    std::vector<RV> read_vars((size_t)NC);
    std::vector<int> rv_chrom_order;                            // chromosomes in the order they enter read_vars
    std::vector<uint8_t> rv_has((size_t)NC, 0);
    // haplo_reads as CSR over (variant, allele, bam): counts first
    const size_t NL = (size_t)NV * 2 * (size_t)NB;
    std::vector<uint32_t> hr_start(NL + 1, 0);
    auto excluded = [&](int b) { return I->bam_excluded && I->bam_excluded[b]; };
    for (int b = 0; b < NB; b++)
        for (int c = 0; c < NC; c++) {
            const size_t s = (size_t)c * NB + b;
            if (!I->line_qid[s] || I->n_lines[s] == 0 || excluded(b)) continue;
            for (int64_t i = 0; i < I->n_lines[s]; i++) {
                const int k = I->line_cls[s][i];
                if (k < 2) hr_start[((size_t)(vbase[(size_t)c] + I->line_var[s][i]) * 2 + (size_t)k) * NB + (size_t)b + 1]++;
            }
        }
    for (size_t i = 0; i < NL; i++) hr_start[i + 1] += hr_start[i];
    std::vector<int32_t> hr_q(hr_start[NL]);
    {
        std::vector<uint32_t> cur(hr_start.begin(), hr_start.end() - 1);
        for (int b = 0; b < NB; b++) {
            std::vector<int> per_file;
            std::vector<RV> rvs;
            for (int c = 0; c < NC; c++) {
                const size_t s = (size_t)c * NB + b;
                if (!I->line_qid[s] || I->n_lines[s] == 0) continue;          // a call file without a kept line returns chromosome "" (phaser.py:1299)
                RV rv;
                for (int64_t i = 0; i < I->n_lines[s]; i++) {
                    const int32_t q = I->line_qid[s][i], g = (int32_t)(vbase[(size_t)c] + I->line_var[s][i]);
                    const int k = I->line_cls[s][i];
                    if (q < 0 || q >= I->nq[c] || I->line_var[s][i] < 0 || I->line_var[s][i] >= I->nv[c]) return fail("call line outside the id spaces");
                    if (!in_dvr[(size_t)g]) { in_dvr[(size_t)g] = 1; dvr.push_back(g); }
                    if (k < 2) {
                        auto it = rv.pos.find(q);
                        int32_t at;
                        if (it == rv.pos.end()) { at = (int32_t)rv.order.size(); rv.pos.emplace(q, at); rv.order.push_back(q); rv.lists.emplace_back(); }
                        else at = it->second;
                        rv.lists[(size_t)at].push_back(g);
                        if (!excluded(b)) hr_q[cur[((size_t)g * 2 + (size_t)k) * NB + (size_t)b]++] = q;
                    }
                }
                per_file.push_back(c); rvs.push_back(std::move(rv));
            }
            for (int c : per_file) if (!rv_has[(size_t)c]) { rv_has[(size_t)c] = 1; rv_chrom_order.push_back(c); }
            for (size_t t = 0; t < per_file.size(); t++) {
                RV &tgt = read_vars[(size_t)per_file[t]]; RV &rv = rvs[t];
                for (size_t j = 0; j < rv.order.size(); j++) {
                    const int32_t q = rv.order[j];
                    auto it = tgt.pos.find(q);
                    if (it == tgt.pos.end()) { tgt.pos.emplace(q, (int32_t)tgt.order.size()); tgt.order.push_back(q); tgt.lists.push_back(std::move(rv.lists[j])); }
                    else tgt.lists[(size_t)it->second] = std::move(rv.lists[j]);          // a later BAM replaces the list of a QNAME already seen (:576-581)
                }
            }
        }
    }
    // ---- connectivity map -> sets -> order of the tested pairs (:1265-1285, :660-678)
    std::vector<std::pair<int32_t, int32_t>> pair_order;
    {
        std::unordered_set<std::pair<int64_t, int64_t>, PairHash> tested;
        for (int c : rv_chrom_order) {
            RV &rv = read_vars[(size_t)c];
            std::vector<int32_t> ov_order; std::unordered_map<int32_t, int32_t> ov_pos; std::vector<std::vector<int32_t>> ov;
            for (size_t j = 0; j < rv.order.size(); j++) {
                const std::vector<int32_t> &lst = rv.lists[j];
                for (int32_t v : lst)
                    for (int32_t o : lst)
                        if (o != v && !(uhash[(size_t)o] == uhash[(size_t)v] && uid_of(o) == uid_of(v))) {
                            auto it = ov_pos.find(v);
                            int32_t at;
                            if (it == ov_pos.end()) { at = (int32_t)ov_order.size(); ov_pos.emplace(v, at); ov_order.push_back(v); ov.emplace_back(); }
                            else at = it->second;
                            ov[(size_t)at].push_back(o);
                        }
            }
            for (size_t t = 0; t < ov_order.size(); t++) {
                PySet s;
                for (int32_t o : ov[t]) s.add(o, uhash[(size_t)o], ueq);
                const int32_t a = ov_order[t];
                s.for_each([&](int32_t b2) {
                    const std::pair<int64_t, int64_t> k1(a, b2), k2(b2, a);
                    if (!tested.count(k1) && !tested.count(k2)) { pair_order.emplace_back(a, b2); tested.insert(k1); }
                });
            }
        }
    }
    // ---- variant_connections.txt
    {
        const std::string_view text(conn, (size_t)conn_len);
        std::vector<std::string_view> rows = split(text, '\n');
        if (rows.empty()) return fail("variant_connections: empty text");
        std::unordered_map<std::pair<int64_t, int64_t>, std::string_view, PairHash> by_pair;
        size_t nbody = 0;
        for (size_t r = 1; r < rows.size(); r++) {
            if (rows[r].empty()) continue;
            const size_t t1 = rows[r].find('\t'), t2 = t1 == std::string_view::npos ? t1 : rows[r].find('\t', t1 + 1);
            if (t2 == std::string_view::npos) return fail("variant_connections: malformed row");
            auto ia = gid_of.find(rows[r].substr(0, t1)), ib = gid_of.find(rows[r].substr(t1 + 1, t2 - t1 - 1));
            if (ia == gid_of.end() || ib == gid_of.end()) return fail("variant_connections: unknown variant id");
            by_pair[{ia->second, ib->second}] = rows[r];
            nbody++;
        }
        if (pair_order.size() != nbody) return fail("variant_connections: replayed pairs do not match the tested pairs");
        std::string &o = H->conn;
        o.reserve((size_t)conn_len + 16);
        o.append(rows[0]); o += '\n';
        for (auto &pr : pair_order) {
            auto it = by_pair.find({pr.first, pr.second});
            if (it != by_pair.end()) { o.append(it->second); o += '\n'; continue; }
            it = by_pair.find({pr.second, pr.first});
            if (it == by_pair.end()) return fail("variant_connections: a replayed pair has no row");
            const std::string_view row = it->second;
            const size_t t1 = row.find('\t'), t2 = row.find('\t', t1 + 1);
            o.append(uid_of(pr.first)); o += '\t'; o.append(uid_of(pr.second)); o.append(row.substr(t2)); o += '\n';
        }
    }
    // ---- blocks, singletons, blacklist
    std::vector<uint8_t> black((size_t)NV, 0);
    for (int c = 0; c < NC; c++)
        if (I->blacklisted[c]) for (int64_t v = 0; v < I->nv[c]; v++) black[(size_t)(vbase[(size_t)c] + v)] = I->blacklisted[c][v];
    std::vector<int32_t> singletons;
    {
        PySet a, b;
        for (int32_t g : dvr) a.add(g, uhash[(size_t)g], ueq);
        for (int64_t k = 0; k < I->n_blocks; k++)
            for (int64_t t = I->blk_off[k]; t < I->blk_off[k + 1]; t++) {
                const int32_t g = (int32_t)(vbase[(size_t)I->blk_chrom[k]] + I->blk_var[t]);
                b.add(g, uhash[(size_t)g], ueq);
            }
        PySet d = py_set_difference(a, b, ueq);
        d.for_each([&](int32_t g) { singletons.push_back(g); });
    }
    // ---- haplotypic_counts.txt
    {
        const std::string_view text(ase, (size_t)ase_len);
        std::vector<std::string_view> rows = split(text, '\n');
        if (rows.empty()) return fail("haplotypic_counts: empty text");
        std::vector<std::string_view> body;
        for (size_t r = 1; r < rows.size(); r++) if (!rows[r].empty()) body.push_back(rows[r]);
        std::string &o = H->ase;
        o.reserve((size_t)ase_len + 16);
        o.append(rows[0]); o += '\n';
        size_t k = 0;
        std::vector<std::string_view> bam_names((size_t)NB);
        for (int b = 0; b < NB; b++) bam_names[(size_t)b] = std::string_view(I->bam_names[b]);
        std::string used_s, bl_s, lab, ids;
        std::vector<int32_t> used;
        for (int64_t bk = 0; bk < I->n_blocks; bk++) {
            const int c = I->blk_chrom[bk];
            const std::string_view cname(I->chrom_names[c]);
            used.clear(); used_s.clear();
            for (int64_t t = I->blk_off[bk]; t < I->blk_off[bk + 1]; t++) {
                const int32_t g = (int32_t)(vbase[(size_t)c] + I->blk_var[t]);
                if (!black[(size_t)g]) { if (!used.empty()) used_s += ','; used_s.append(uid_of(g)); used.push_back(g); }
            }
            PySet bset;
            for (int h = 0; h < 2; h++)
                for (int64_t t = I->blk_off[bk]; t < I->blk_off[bk + 1]; t++) {
                    const int32_t g = (int32_t)(vbase[(size_t)c] + I->blk_var[t]);
                    if (black[(size_t)g]) bset.add(g, uhash[(size_t)g], ueq);
                }
            bl_s.clear();
            bset.for_each([&](int32_t g) { if (!bl_s.empty()) bl_s += ','; bl_s.append(uid_of(g)); });
            for (int b = 0; b < NB; b++) {
                if (excluded(b)) continue;
                if (k >= body.size()) break;
                std::vector<std::string_view> f = split(body[k], '\t');
                if (f.size() < 18) return fail("haplotypic_counts: malformed row");
                const size_t nf = f.size();
                if (!(f[0] == cname && f[3] == used_s && f[nf - 3] == bam_names[(size_t)b] && atoll(std::string(f[4]).c_str()) == (long long)used.size())) continue;
                k++;
                const Pool &NP = namep[(size_t)c];
                std::vector<int64_t> nhash;
                auto neq = [&](int32_t x, int32_t y) { return NP.at(x) == NP.at(y); };
                std::string labels[2], idl[2];
                for (int h = 0; h < 2; h++) {
                    std::vector<std::string_view> hx = f[7 + (size_t)h].empty() ? std::vector<std::string_view>() : split(f[7 + (size_t)h], ',');
                    PySet s;
                    std::vector<std::pair<uint32_t, uint32_t>> var_reads;          // [lo, hi) into hr_q per used variant (empty: none)
                    const size_t nz = std::min(used.size(), hx.size());
                    for (size_t t = 0; t < nz; t++) {
                        const int32_t g = used[t];
                        const int cc = chrom_of[(size_t)g]; const int64_t lv = g - vbase[(size_t)cc];
                        int ai = -1;
                        if (allep[(size_t)cc].at(2 * lv) == hx[t]) ai = 0; else if (allep[(size_t)cc].at(2 * lv + 1) == hx[t]) ai = 1;
                        if (ai < 0) return fail("haplotypic_counts: an allele of a block row is not one of the variant's alleles");
                        const size_t e = ((size_t)g * 2 + (size_t)ai) * NB + (size_t)b;
                        var_reads.emplace_back(hr_start[e], hr_start[e + 1]);
                        for (uint32_t p = hr_start[e]; p < hr_start[e + 1]; p++) {
                            const int32_t q = hr_q[p];
                            const std::string_view nm = NP.at(q);
                            s.add(q, py_str_hash(nm.data(), nm.size()), neq);
                        }
                    }
                    std::unordered_map<int32_t, int32_t> index;
                    index.reserve(s.used * 2);
                    int32_t pos = 0;
                    s.for_each([&](int32_t q) { if (pos) idl[h] += ','; idl[h].append(NP.at(q)); index.emplace(q, pos++); });
                    std::string &L = labels[h];
                    for (size_t t = 0; t < var_reads.size(); t++) {
                        if (t) L += ';';
                        for (uint32_t p = var_reads[t].first; p < var_reads[t].second; p++) {
                            if (p > var_reads[t].first) L += ',';
                            auto it = index.find(hr_q[p]);
                            put_int(L, it->second);
                        }
                    }
                }
                for (size_t j = 0; j < nf; j++) {
                    if (j) o += '\t';
                    if (j == 5) o += bl_s;
                    else if (j == nf - 2) o += labels[0];
                    else if (j == nf - 1) o += labels[1];
                    else if (I->output_read_ids == 1 && nf == 20 && j == 14) o += idl[0];
                    else if (I->output_read_ids == 1 && nf == 20 && j == 15) o += idl[1];
                    else o.append(f[j]);
                }
                o += '\n';
            }
        }
        // singleton rows: grouped by variant, emitted in the order of the set
        std::unordered_map<int32_t, std::vector<std::string>> by_var;
        size_t n_single_in = 0;
        for (size_t r = k; r < body.size(); r++) {
            std::vector<std::string_view> f = split(body[r], '\t');
            if (f.size() < 18) return fail("haplotypic_counts: malformed singleton row");
            auto iv = gid_of.find(f[3]);
            if (iv == gid_of.end()) return fail("haplotypic_counts: unknown variant id in a singleton row");
            const int32_t g = iv->second;
            std::string row;
            const size_t nf = f.size();
            if (I->output_read_ids == 1 && nf == 20) {          // singleton rows list set(haplo_reads[allele][bam]) (:1196-1204, :1219-1220)
                int b = -1;
                for (int t = 0; t < NB; t++) if (f[nf - 3] == std::string_view(I->bam_names[t])) { b = t; break; }
                if (b < 0) return fail("haplotypic_counts: unknown BAM name in a singleton row");
                const int cc = chrom_of[(size_t)g];
                const Pool &NP = namep[(size_t)cc];
                auto neq = [&](int32_t x, int32_t y) { return NP.at(x) == NP.at(y); };
                std::string col[2];
                for (int k2 = 0; k2 < 2; k2++) {
                    const size_t e = ((size_t)g * 2 + (size_t)k2) * NB + (size_t)b;
                    PySet s;
                    for (uint32_t p = hr_start[e]; p < hr_start[e + 1]; p++) { const std::string_view nm = NP.at(hr_q[p]); s.add(hr_q[p], py_str_hash(nm.data(), nm.size()), neq); }
                    s.for_each([&](int32_t q) { if (!col[k2].empty()) col[k2] += ','; col[k2].append(NP.at(q)); });
                }
                for (size_t j = 0; j < nf; j++) { if (j) row += '\t'; if (j == 14) row += col[0]; else if (j == 15) row += col[1]; else row.append(f[j]); }
            } else row.assign(body[r]);
            by_var[g].push_back(std::move(row));
            n_single_in++;
        }
        size_t n_single = 0;
        for (int32_t g : singletons) {
            auto it = by_var.find(g);
            if (it == by_var.end()) continue;
            for (const std::string &r : it->second) { o.append(r); o += '\n'; n_single++; }
        }
        if (n_single != n_single_in) return fail("haplotypic_counts: singleton rows do not match the replayed singletons");
    }
    // ---- haplotypes.txt: block rows stay, singleton rows in the order of the set
    {
        const std::string_view text(hap, (size_t)hap_len);
        std::vector<std::string_view> rows = split(text, '\n');
        if (rows.empty()) return fail("haplotypes: empty text");
        std::vector<std::string_view> body;
        for (size_t r = 1; r < rows.size(); r++) if (!rows[r].empty()) body.push_back(rows[r]);
        std::string &o = H->hap;
        o.reserve((size_t)hap_len + 16);
        o.append(rows[0]); o += '\n';
        const size_t nblk = (size_t)I->n_blocks;
        if (body.size() < nblk) return fail("haplotypes: fewer rows than blocks");
        for (size_t r = 0; r < nblk; r++) { o.append(body[r]); o += '\n'; }
        // key of a singleton row: (chromosome, end position, name)
        struct Key { std::string s; };
        std::unordered_map<std::string, std::vector<std::string_view>> by_key;
        std::unordered_map<std::string, size_t> taken;
        size_t n_single_in = 0;
        for (size_t r = nblk; r < body.size(); r++) {
            std::vector<std::string_view> f = split(body[r], '\t');
            if (f.size() < 6) return fail("haplotypes: malformed singleton row");
            std::string key; key.append(f[0]); key += '\t'; key.append(f[2]); key += '\t'; key.append(f[5]);
            by_key[key].push_back(body[r]);
            n_single_in++;
        }
        size_t n_single = 0;
        if (I->unphased_vars == 1)
            for (int32_t g : singletons) {
                const int c = chrom_of[(size_t)g]; const int64_t lv = g - vbase[(size_t)c];
                std::string key; key.append(I->chrom_names[c]); key += '\t'; put_int(key, (long long)I->pos[c][lv]); key += '\t';
                key.append(I->unique_ids ? uid_of(g) : rsidp[(size_t)c].at(lv));
                auto it = by_key.find(key);
                if (it == by_key.end()) continue;
                size_t &t = taken[key];
                if (t < it->second.size()) { o.append(it->second[t++]); o += '\n'; n_single++; }
            }
        if (n_single != n_single_in) return fail("haplotypes: singleton rows do not match the replayed singletons");
    }
    return PHZ_OK;
}
