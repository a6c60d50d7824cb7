// Native BGZF / BAM decode + structure-of-arrays packing + QNAME interning (host side of the path's input,
// SURVEY.md 8(f) next-1).  Replaces, for one BAM, what phASER obtains from
//   samtools view -h BAM 'chr': | samtools view -Sh [-F 0x400] [-f 2] -q MAPQ -      (phaser/phaser.py:1346, :505-513)
// plus the per-record parsing the mapper does on the SAM text (read_variant_map.py:27-64).
// Multi-threaded raw-deflate inflate of the BGZF members (zlib), one sequential hop over the record chain,
// multi-threaded packing into exactly the arrays soa.pack_sam() builds (tests compare them bit for bit).
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <chrono>
#include <memory>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <map>
#include <mutex>
#include <vector>

#include "phz.h"
#include "phz_internal.h"

namespace {
// PHZ_TIMING=1: wall-clock laps of the host stages on stderr
struct Laps {
    bool on; const char *tag; std::chrono::steady_clock::time_point t;
    explicit Laps(const char *tag_) : on(getenv("PHZ_TIMING") != nullptr), tag(tag_), t(std::chrono::steady_clock::now()) {}
    void lap(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[phz timing]     %s: %-36s %7.1f ms\n", tag, what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};
}  // namespace

namespace {

// inflate target: plain malloc (a std::vector would zero-fill tens of GB on one thread before the parallel inflate writes them)
struct RawBuf {
    uint8_t *p = nullptr; size_t n = 0, mapped = 0;
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    RawBuf(RawBuf &&o) noexcept : p(o.p), n(o.n), mapped(o.mapped) { o.p = nullptr; o.n = 0; o.mapped = 0; }
    RawBuf &operator=(RawBuf &&o) noexcept { if (this != &o) { drop(); p = o.p; n = o.n; mapped = o.mapped; o.p = nullptr; o.n = 0; o.mapped = 0; } return *this; }
    ~RawBuf() { drop(); }
    void drop() { if (mapped) munmap(p, mapped); else free(p); p = nullptr; n = 0; mapped = 0; }
    // big buffers come from an anonymous mapping with transparent huge pages requested: the threads that fill them take
    // 512x fewer page faults than with 4 KB pages
    bool resize(size_t m) {
        drop();
        if (m >= (64u << 20)) {
            const size_t len = (m + 1 + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
            void *q = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (q != MAP_FAILED) { madvise(q, len, MADV_HUGEPAGE); p = (uint8_t *)q; n = m; mapped = len; return true; }
        }
        p = (uint8_t *)malloc(m ? m + 1 : 1); n = p ? m : 0;
        return p != nullptr;
    }
    uint8_t *data() { return p; }
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
    // hands the memory to a caller that will free() it: only valid for malloc'd buffers, so copy out of a mapping
    uint8_t *release() {
        uint8_t *q;
        if (mapped) { q = (uint8_t *)malloc(n + 1); if (q) memcpy(q, p, n + 1); drop(); }
        else { q = p; p = nullptr; n = 0; }
        return q;
    }
};

// std::vector whose resize() leaves new elements uninitialised: the per-record arrays of a shard are written in full by the
// parallel pack, a value-initialising resize would first sweep gigabytes of zeros on one thread
template <class T>
struct NoInit : std::allocator<T> {
    template <class U> struct rebind { using other = NoInit<U>; };
    NoInit() = default;
    template <class U> NoInit(const NoInit<U> &) {}
    template <class U, class... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U; else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
template <class T> using RawVec = std::vector<T, NoInit<T>>;

struct Shard {
    std::string name;
    RawVec<int32_t> pos, aln;
    RawVec<uint32_t> cigar_off, cigar, seq_off, qname_off;
    RawVec<uint8_t> has_as;
    RawBuf seq2, qual;                         // filled record by record in the parallel pack (no serial zero fill)
    RawVec<char> qnames;
};

struct Range { size_t rec_begin, rec_end; };

inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline int32_t rdi32(const uint8_t *p) { int32_t v; memcpy(&v, p, 4); return v; }
inline uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }

// value of the last AS tag; returns false when absent
bool aux_as(const uint8_t *p, const uint8_t *e, int32_t *out) {
    bool found = false;
    while (p + 3 <= e) {
        const uint8_t t0 = p[0], t1 = p[1], ty = p[2];
        p += 3;
        int sz = 0;
        switch (ty) {
            case 'c': case 'C': case 'A': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'Z': case 'H': { const uint8_t *z = (const uint8_t *)memchr(p, 0, (size_t)(e - p)); if (!z) return found;
                                   p = z + 1; continue; }
            case 'B': { if (p + 5 > e) return found; const uint8_t st = p[0]; const uint32_t cnt = rd32(p + 1);
                        int es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
                        if ((size_t)cnt * (size_t)es > (size_t)(e - p) - 5) return found;
                        p += 5 + (size_t)cnt * es; continue; }
            default: return found;
        }
        if (p + sz > e) return found;
        if (t0 == 'A' && t1 == 'S' && ty != 'A' && ty != 'f') {
            int32_t v = 0;
            switch (ty) {
                case 'c': v = (int8_t)p[0]; break;
                case 'C': v = p[0]; break;
                case 's': { int16_t x; memcpy(&x, p, 2); v = x; break; }
                case 'S': v = rd16(p); break;
                case 'i': v = rdi32(p); break;
                case 'I': v = (int32_t)rd32(p); break;
            }
            *out = v; found = true;
        }
        p += sz;
    }
    return found;
}

// normalised op list of one record (see soa.pack_sam): returns the number of ops written
inline int norm_ops(const uint8_t *cig, int n_cig, int nb, uint32_t *out) {
    int n = 0;
    long read_pos = 0;
    for (int i = 0; i < n_cig; i++) {
        const uint32_t c = rd32(cig + 4 * i);
        const uint32_t op = c & 15, len = c >> 4;
        if (op == 0 || op == 7 || op == 8) {
            long lo = read_pos < nb ? read_pos : nb, hi = read_pos + (long)len < nb ? read_pos + (long)len : nb;
            const uint32_t avail = (uint32_t)(hi > lo ? hi - lo : 0);
            if (avail == len) { if (out) out[n] = c; n++; }
            else {
                if (avail) { if (out) out[n] = (avail << 4) | op; n++; }
                if (out) out[n] = ((len - avail) << 4) | 9u; n++;
            }
            read_pos += len;
        } else if (op == 1) {
            long lo = read_pos < nb ? read_pos : nb, hi = read_pos + (long)len < nb ? read_pos + (long)len : nb;
            const uint32_t avail = (uint32_t)(hi > lo ? hi - lo : 0);
            if (out) out[n] = (avail << 4) | 1u; n++;
            read_pos += len;
        } else if (op == 4) {
            if (out) out[n] = c; n++;
            read_pos += len;
        } else if (op == 2 || op == 3) {
            if (out) out[n] = c; n++;
        }
    }
    return n;
}

struct Bam {
    RawBuf data;                               // inflated BAM stream
    std::vector<std::pair<std::string, int32_t>> refs;
    size_t first_record = 0;
    std::vector<Shard> shards;                 // result of the last decode, indexed by position in `order`
    std::string err;
};

// one raw-deflate stream per thread, reset between members (a BGZF file is ~16 members per MB: the per-member
// inflateInit2 / inflateEnd pair is an allocation of the 7 KB state each time)
struct ZState {
    z_stream zs; bool live = false;
    ~ZState() { if (live) inflateEnd(&zs); }
};
bool inflate_block(const uint8_t *src, size_t csize, uint8_t *dst, size_t isize) {
    static thread_local ZState Z;
    if (!Z.live) {
        memset(&Z.zs, 0, sizeof Z.zs);
        if (inflateInit2(&Z.zs, -15) != Z_OK) return false;
        Z.live = true;
    } else if (inflateReset(&Z.zs) != Z_OK) return false;
    Z.zs.next_in = (Bytef *)src; Z.zs.avail_in = (uInt)csize;
    Z.zs.next_out = dst; Z.zs.avail_out = (uInt)isize;
    const int rc = inflate(&Z.zs, Z_FINISH);
    return rc == Z_STREAM_END && Z.zs.avail_out == 0;
}

// the CRC32 of a member's payload against the value in its trailer
static bool crc_ok(const uint8_t *p, size_t n, uint32_t want) {
    return (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)p, (uInt)n) == want;
}

int n_threads(int want) {
    if (want > 0) return want;
    unsigned h = std::thread::hardware_concurrency();
    return h ? (int)(h > 32 ? 32 : h) : 4;
}

// Whole-file BGZF inflate: member table from the BC extra fields, members inflated in parallel into one buffer.
// PHZ_E_UNSUPPORTED = a gzip member without the BGZF extra field (plain gzip: not splittable, caller falls back).
template <class Vec>
int inflate_bgzf_file(const char *path, int threads, Vec &outbuf) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return PHZ_E_ARG;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 28) { close(fd); return PHZ_E_ARG; }
    const size_t fsz = (size_t)st.st_size;
    const uint8_t *f = (const uint8_t *)mmap(nullptr, fsz, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (f == MAP_FAILED) return PHZ_E_NOMEM;
    struct Blk { size_t off, csize, isize, dst; uint32_t crc; };
    std::vector<Blk> blks;
    size_t off = 0, total = 0;
    int status = PHZ_OK;
    while (off + 18 <= fsz) {
        if (f[off] != 0x1f || f[off + 1] != 0x8b) { status = PHZ_E_ARG; break; }
        if (!(f[off + 3] & 4)) { status = PHZ_E_UNSUPPORTED; break; }
        const uint16_t xlen = rd16(f + off + 10);
        size_t x = off + 12, xe = x + xlen;
        uint32_t bsize = 0;
        while (x + 4 <= xe) {
            const uint16_t slen = rd16(f + x + 2);
            if (f[x] == 'B' && f[x + 1] == 'C' && slen == 2) bsize = (uint32_t)rd16(f + x + 4) + 1;
            x += 4 + slen;
        }
        if (!bsize) { status = PHZ_E_UNSUPPORTED; break; }
        if (off + bsize > fsz || bsize < (uint32_t)xlen + 20) { status = PHZ_E_ARG; break; }
        const uint32_t isize = rd32(f + off + bsize - 4);
        blks.push_back({off + 12 + xlen, (size_t)bsize - xlen - 20, isize, total, rd32(f + off + bsize - 8)});
        total += isize;
        off += bsize;
    }
    if (status == PHZ_OK && blks.empty()) status = PHZ_E_ARG;
    if (status != PHZ_OK) { munmap((void *)f, fsz); return status; }
    if (!outbuf.resize(total)) { munmap((void *)f, fsz); return PHZ_E_NOMEM; }
    const int nt = n_threads(threads);
    std::atomic<size_t> next(0);
    std::atomic<bool> bad(false);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++)
        th.emplace_back([&] {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= blks.size()) break;
                if (blks[i].isize && !inflate_block(f + blks[i].off, blks[i].csize, (uint8_t *)outbuf.data() + blks[i].dst, blks[i].isize)) bad = true;
                else if (!crc_ok((const uint8_t *)outbuf.data() + blks[i].dst, blks[i].isize, blks[i].crc)) bad = true;      // (htslib checks the same after inflating a block)
            }
        });
    for (auto &t : th) t.join();
    munmap((void *)f, fsz);
    return bad ? PHZ_E_ARG : PHZ_OK;
}

// BAM header of an inflated stream prefix: 0 = parsed, 1 = need more bytes, 2 = malformed.  Every field is checked against the
// size before it is used (untrusted input).
int parse_bam_header(const uint8_t *d, size_t dn, std::vector<std::pair<std::string, int32_t>> &refs, size_t *first_record) {
    if (dn < 12) return 1;
    if (memcmp(d, "BAM\1", 4) != 0) return 2;
    const int32_t l_text = rdi32(d + 4);
    if (l_text < 0) return 2;
    if ((size_t)l_text > dn - 12) return 1;
    size_t p = 8 + (size_t)l_text;
    const int32_t n_ref = rdi32(d + p); p += 4;
    if (n_ref < 0) return 2;
    for (int32_t i = 0; i < n_ref; i++) {
        if (p + 4 > dn) return 1;
        const int32_t l = rdi32(d + p); p += 4;
        if (l < 1) return 2;
        if ((size_t)l > dn - p || dn - p - (size_t)l < 4) return 1;
        refs.emplace_back(std::string((const char *)d + p, (size_t)(l - 1)), 0); p += (size_t)l;
        refs.back().second = rdi32(d + p); p += 4;
    }
    *first_record = p;
    return 0;
}

// can offset p of the inflated bytes d[0, n) start a record?  -> offset of the next record, 0 when it cannot
size_t plausible_at(const uint8_t *d, size_t n, size_t p, int n_ref) {
    if (p + 36 > n) return 0;
    const int32_t bs = rdi32(d + p);
    if (bs < 32 || bs > (1 << 24) || p + 4 + (size_t)bs > n) return 0;
    const uint8_t *r = d + p + 4;
    const int32_t ref = rdi32(r), pos0 = rdi32(r + 4), l_seq = rdi32(r + 16), nref = rdi32(r + 20);
    const uint32_t l_rn = r[8], n_cig = rd16(r + 12), flag = rd16(r + 14);
    // a QNAME has at least one character (SAM: [!-?A-~]{1,254}; an empty name would make every zero byte a candidate), FLAG has 12
    // defined bits, mate position >= -1
    if (ref < -1 || ref >= n_ref || nref < -1 || nref >= n_ref || pos0 < -1 || l_seq < 0 || l_rn < 2 || flag >= 4096 || rdi32(r + 24) < -1) return 0;
    const size_t need = 32 + (size_t)l_rn + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    if (need > (size_t)bs) return 0;
    if (r[32 + l_rn - 1] != 0) return 0;                 // read name is NUL-terminated
    for (uint32_t k = 0; k + 1 < l_rn; k++) if (r[32 + k] < 33 || r[32 + k] > 126) return 0;
    return p + 4 + (size_t)bs;
}

// member table of a BGZF file (nothing inflated)
struct BgzfMap {
    struct Blk { size_t off, csize, isize, dst; uint32_t crc; };          // crc: CRC32 of the member's payload, from its trailer
    const uint8_t *f = nullptr; size_t fsz = 0, total = 0;
    int fd = -1;
    std::vector<Blk> blks;
    ~BgzfMap() { if (f) munmap((void *)f, fsz); if (fd >= 0) close(fd); }
    // the whole file mapped (on first use): the host decoder inflates out of the mapping
    const uint8_t *map() {
        if (!f && fd >= 0) {
            void *m = mmap(nullptr, fsz, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) f = (const uint8_t *)m;
        }
        return f;
    }
    // n bytes at `off`: out of the mapping when there is one, else pread
    // (rfd: a descriptor of the reading thread's own -- sixteen threads on ONE descriptor spend their time on its reference count)
    bool fetch(size_t off, size_t n, uint8_t *dst, int rfd = -1) const {
        if (off > fsz || n > fsz - off) return false;
        if (f) { memcpy(dst, f + off, n); return true; }
        if (rfd < 0) rfd = fd;
        size_t got = 0;
        while (got < n) {
            const ssize_t r = pread(rfd, dst + got, n - got, (off_t)(off + got));
            if (r <= 0) return false;
            got += (size_t)r;
        }
        return true;
    }
    // the compressed bytes of member k (its deflate stream): a pointer into the mapping, or into `tmp`
    const uint8_t *member(size_t k, std::vector<uint8_t> &tmp) const {
        if (f) return f + blks[k].off;
        tmp.resize(blks[k].csize + 1);
        return fetch(blks[k].off, blks[k].csize, tmp.data()) ? tmp.data() : nullptr;
    }
    // sparse: only the member headers (and a few members) are wanted -- they are read with pread (one 64-byte read per member: the
    // trailer of one member and the header of the next are neighbours), nothing is mapped.  Reading them through a mapping left one
    // page-table entry per member behind: 55 ms of munmap for a 3.8 GB BAM, on top of 250,000 page faults.
    int open(const char *path, bool sparse = false) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return PHZ_E_ARG;
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size < 28) return PHZ_E_ARG;
        fsz = (size_t)st.st_size;
        if (getenv("PHZ_BGZF_NO_MAP")) sparse = true;           // tests: the pread walk under the host decoder too (which maps the file afterwards)
        if (!sparse && !map()) return PHZ_E_NOMEM;
        constexpr size_t PEEK = 64;
        // one member header out of h = the `avail` bytes at `off` (avail = min(PEEK, fsz - off) at least): 0 = not a BGZF member header,
        // else the member's total size; *xl = XLEN
        auto header_in = [&](size_t off, const uint8_t *h, size_t avail, uint16_t *xl, int *why, int rfd) -> uint32_t {
            *why = PHZ_E_ARG;
            if (off + 18 > fsz || avail < 18 || h[0] != 0x1f || h[1] != 0x8b) return 0;
            if (!(h[3] & 4)) { *why = PHZ_E_UNSUPPORTED; return 0; }
            const uint16_t xlen = rd16(h + 10);
            std::vector<uint8_t> wide;
            if (12 + (size_t)xlen > avail) {                   // an extra field longer than the peek (BGZF writers use 6 bytes)
                if (off + 12 + xlen > fsz) { *why = PHZ_E_UNSUPPORTED; return 0; }
                wide.resize(12 + (size_t)xlen);
                if (!fetch(off, wide.size(), wide.data(), rfd)) return 0;
                h = wide.data();
            }
            size_t x = 12;
            const size_t xe = 12 + (size_t)xlen;
            uint32_t bsize = 0;
            while (x + 4 <= xe) {
                const uint16_t slen = rd16(h + x + 2);
                if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2 && x + 6 <= xe) bsize = (uint32_t)rd16(h + x + 4) + 1;
                x += 4 + (size_t)slen;
            }
            if (!bsize) { *why = PHZ_E_UNSUPPORTED; return 0; }
            if (off + bsize > fsz || bsize < (uint32_t)xlen + 20) return 0;
            *xl = xlen;
            return bsize;
        };
        auto header = [&](size_t off, uint16_t *xl, int *why, int rfd) -> uint32_t {
            uint8_t h[PEEK];
            const size_t avail = off < fsz ? std::min(PEEK, fsz - off) : 0;
            if (avail < 18 || !fetch(off, avail, h, rfd)) { *why = PHZ_E_ARG; return 0; }
            return header_in(off, h, avail, xl, why, rfd);
        };
        auto own_fd = [&]() -> int { return f ? -1 : ::open(path, O_RDONLY); };       // -1: the shared one (or the mapping)
        // the member chain from `off` up to (not beyond) `stop`: -> where it arrived, members appended without their dst.  One read per
        // member: the last four bytes of member i (ISIZE) and the header of member i+1.
        auto walk = [&](size_t off, size_t stop, std::vector<Blk> &out, int *status, int rfd) -> size_t {
            uint8_t h[8 + PEEK];
            size_t avail = 0;
            if (off < stop && off + 18 <= fsz) { avail = std::min(PEEK, fsz - off); if (!fetch(off, avail, h + 8, rfd)) { *status = PHZ_E_ARG; return off; } }
            while (off < stop && off + 18 <= fsz) {
                uint16_t xlen; int why;
                const uint32_t bsize = header_in(off, h + 8, avail, &xlen, &why, rfd);
                if (!bsize) { *status = why; return off; }
                const size_t nxt = off + bsize;
                avail = std::min(PEEK, fsz - nxt);
                if (!fetch(nxt - 8, 8 + avail, h, rfd)) { *status = PHZ_E_ARG; return off; }      // the trailer (CRC32, ISIZE) of this member and the header of the next
                const uint32_t crc = rd32(h), isz = rd32(h + 4);
                if (isz > 65536u) { *status = PHZ_E_ARG; return off; }        // BGZF: a member inflates to at most 64 KiB; the trailer is not trusted beyond that
                out.push_back({off + 12 + xlen, (size_t)bsize - xlen - 20, isz, 0, crc});
                off = nxt;
            }
            return off;
        };
        // Big files: the chain is walked in K segments at once.  A segment's first member is GUESSED (the first offset where four
        // member headers follow each other) and VERIFIED: segment k must arrive exactly where segment k+1 starts; any mismatch falls
        // back to the one sequential walk.  (250,000 members per genome, one small read each.)
        bool done = false;
        size_t par_min = 256u << 20;
        { const char *e = getenv("PHZ_BGZF_PAR_MIN"); if (e) par_min = (size_t)atoll(e); }       // tests force the segmented walk on small files
        // (the pread walk has no address-space lock to fight over: as many segments as the host has threads, 16..48)
        const int hw = (int)std::thread::hardware_concurrency();
        const int K = fsz >= par_min ? (f ? 16 : std::max(16, std::min(48, hw))) : 1;
        if (K > 1) {
            std::vector<size_t> start((size_t)K + 1, fsz);
            start[0] = 0;
            std::vector<std::vector<Blk>> part((size_t)K);
            std::vector<size_t> arrive((size_t)K, 0);
            std::vector<int> stt((size_t)K, PHZ_OK);
            std::vector<std::thread> th;
            for (int k = 1; k < K; k++)
                th.emplace_back([&, k] {
                    const size_t g = fsz / (size_t)K * (size_t)k, lim = std::min(fsz, g + (1u << 20));
                    std::vector<uint8_t> win;
                    size_t w0 = g;                               // the window holds [w0, w0 + win.size())
                    const int rfd = own_fd();
                    for (size_t p = g; p < lim; p++) {
                        if (p >= w0 + win.size()) { w0 = p; win.resize(std::min<size_t>(128u << 10, lim - p)); if (!fetch(w0, win.size(), win.data(), rfd)) break; }
                        if (win[p - w0] != 0x1f) continue;
                        size_t q = p; int ok = 0;
                        while (ok < 4) { uint16_t xl; int why; const uint32_t bs = header(q, &xl, &why, rfd); if (!bs) break; ok++; q += bs; if (q >= fsz) { ok = 4; break; } }
                        if (ok >= 4) { start[(size_t)k] = p; break; }
                    }
                    if (rfd >= 0) close(rfd);
                });
            for (auto &x : th) x.join();
            th.clear();
            for (int k = 0; k < K; k++)
                th.emplace_back([&, k] {
                    if (start[(size_t)k] >= fsz) { arrive[(size_t)k] = fsz; return; }
                    const int rfd = own_fd();
                    arrive[(size_t)k] = walk(start[(size_t)k], start[(size_t)k + 1], part[(size_t)k], &stt[(size_t)k], rfd);
                    if (rfd >= 0) close(rfd);
                });
            for (auto &x : th) x.join();
            bool ok = true;
            for (int k = 0; k < K; k++) {
                if (stt[(size_t)k] != PHZ_OK) ok = false;
                if (start[(size_t)k] < fsz && start[(size_t)k + 1] < fsz && arrive[(size_t)k] != start[(size_t)k + 1]) ok = false;
                if (k + 1 < K && start[(size_t)k + 1] < start[(size_t)k]) ok = false;
            }
            if (ok) {
                size_t n = 0;
                for (auto &v : part) n += v.size();
                blks.reserve(n);
                for (auto &v : part) blks.insert(blks.end(), v.begin(), v.end());
                done = true;
            }
        }
        if (!done) {
            blks.clear();
            int status = PHZ_OK;
            const size_t end = walk(0, fsz, blks, &status, -1);
            if (status != PHZ_OK && end + 18 <= fsz) return status;
        }
        total = 0;
        for (auto &b : blks) { b.dst = total; total += b.isize; }
        return blks.empty() ? PHZ_E_ARG : PHZ_OK;
    }
};

}  // namespace

struct phz_bam { Bam b; };

// QNAME table split into hash partitions so that one call can be processed by many threads; ids are still handed out in
// first-appearance order over the whole input (what a sequential dictionary would do), see phz_intern.
// open-addressing table hash -> value; equality of the key text is decided by the caller (hash match + string compare)
struct FlatMap {
    std::vector<uint64_t> hs; std::vector<int32_t> val;
    size_t mask = 0, count = 0;
    void reserve(size_t n) {          // room for n entries at load <= 0.5
        size_t cap = 16;
        while (cap < 2 * n) cap <<= 1;
        if (cap <= hs.size()) return;
        std::vector<uint64_t> oh; std::vector<int32_t> ov;
        oh.swap(hs); ov.swap(val);
        hs.assign(cap, 0); val.assign(cap, -1); mask = cap - 1;
        for (size_t i = 0; i < oh.size(); i++)
            if (ov[i] >= 0) { size_t k = (size_t)oh[i] & mask; while (val[k] >= 0) k = (k + 1) & mask; hs[k] = oh[i]; val[k] = ov[i]; }
    }
    template <class Eq> int32_t find(uint64_t h, Eq eq) const {
        if (hs.empty()) return -1;
        for (size_t k = (size_t)h & mask;; k = (k + 1) & mask) {
            if (val[k] < 0) return -1;
            if (hs[k] == h && eq(val[k])) return val[k];
        }
    }
    void insert(uint64_t h, int32_t v) {          // the key is known to be absent; reserve() was called
        size_t k = (size_t)h & mask;
        while (val[k] >= 0) k = (k + 1) & mask;
        hs[k] = h; val[k] = v; count++;
    }
};

inline uint64_t hash_name(const char *p, size_t n) {       // FNV-1a with a final mix
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= (uint8_t)p[i]; h *= 1099511628211ull; }
    h ^= h >> 32; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29;
    return h;
}

// QNAME table split into hash partitions so that one call can be processed by many threads; ids are still handed out in
// first-appearance order over the whole input (what a sequential dictionary would do), see phz_intern.
struct phz_interner {
    static constexpr int P = 64;
    struct Part {
        FlatMap ids;                               // name hash -> id (text compared through `names`)
        std::vector<std::unique_ptr<char[]>> arena; size_t arena_used = 0, arena_cap = 0;
        const char *keep(std::string_view s) {
            if (arena.empty() || arena_used + s.size() > arena_cap) {       // (an empty name must not find an empty arena)
                arena_cap = std::max<size_t>(1 << 20, s.size());
                arena.emplace_back(new char[arena_cap]); arena_used = 0;
            }
            char *d = arena.back().get() + arena_used;
            memcpy(d, s.data(), s.size()); arena_used += s.size();
            return d;
        }
    };
    Part part[P];
    std::vector<std::string_view> names;          // id -> name
};

// Plan of a chromosome-restricted read: BAM header, and the byte ranges [u0, u1) of the inflated stream (cut at record boundaries)
// that hold the records of the wanted references.  Shared by the host open (phz_bam_open_refs) and the device open (phz_bamdev_open).
struct BamPiece { uint64_t u0, u1; };
static int bam_plan(BgzfMap &M, const char *const *ref_names, int n_names, int64_t *ref_bytes, int max_refs, bool want_pieces,
                    std::vector<std::pair<std::string, int32_t>> &refs, std::vector<uint8_t> &head, size_t *first_record_out,
                    std::vector<BamPiece> &pieces, Laps &laps) {
    // header: inflate members until it parses
    size_t hb = 0, first_record = 0;
    std::vector<uint8_t> comp_tmp;
    for (;;) {
        if (hb >= M.blks.size()) return PHZ_E_ARG;
        const size_t old = head.size();
        head.resize(old + M.blks[hb].isize);
        if (M.blks[hb].isize) {
            const uint8_t *src = M.member(hb, comp_tmp);
            if (!src || !inflate_block(src, M.blks[hb].csize, head.data() + old, M.blks[hb].isize)) return PHZ_E_ARG;
        }
        hb++;
        refs.clear();
        const int rc = parse_bam_header(head.data(), head.size(), refs, &first_record);
        if (rc == 0) break;
        if (rc < 0 && head.size() > (1u << 30)) return PHZ_E_ARG;
        if (rc == 2) return PHZ_E_ARG;       // malformed, not merely short
    }
    const int n_ref = (int)refs.size();
    const size_t nb = M.blks.size();
    // first member that holds the first record
    size_t b0 = 0;
    while (b0 + 1 < nb && M.blks[b0 + 1].dst <= first_record) b0++;
    // probe(b): global uncompressed offset and refID of the first record that STARTS in member b or later members (up to 8 ahead).
    //   1  found;   0  the members from b to the end of the file hold no record start (past the last record);
    //  -1  no record boundary could be PROVEN in members b .. b+7 (records longer than the window, damaged members): the caller must
    //      not guess -- a wrong "past this reference" would silently drop records -- and gives the file to the full, order-agnostic open.
    // The window starts at three members and grows (x4, up to 256 members = 16 MB) while candidate chains run out of data, so records of
    // tens of kilobytes (long reads, large aux fields) are still chained 12 deep.
    auto probe = [&](size_t b, uint64_t *uoff, int32_t *ref) -> int {
        std::vector<uint8_t> tmp, ctmp;
        size_t b1 = b;
        for (; b1 < nb && b1 < b + 8; b1++) {
            for (size_t win = 3;; win *= 4) {
                tmp.clear();                                   // window = members b1 .. b1+win-1
                for (size_t k = b1; k < nb && k < b1 + win; k++) {
                    const size_t old = tmp.size();
                    tmp.resize(old + M.blks[k].isize);
                    if (M.blks[k].isize) {
                        const uint8_t *src = M.member(k, ctmp);
                        if (!src || !inflate_block(src, M.blks[k].csize, tmp.data() + old, M.blks[k].isize)) return -1;
                    }
                }
                if (b1 == b0) {                                // the true first record: no guess needed
                    const size_t p = first_record - M.blks[b0].dst;
                    if (p + 8 > tmp.size()) return 0;          // a BAM without records
                    *uoff = first_record; *ref = rdi32(tmp.data() + p + 4);
                    return 1;
                }
                const bool at_eof = b1 + win >= nb;
                const size_t lim = M.blks[b1].isize;
                bool short_of_data = false;
                for (size_t p = 0; p < lim; p++) {
                    size_t q = p; int ok = 0;
                    while (ok < 12) {
                        const size_t nx = plausible_at(tmp.data(), tmp.size(), q, n_ref);
                        if (!nx) {
                            // out of window with a header that still looks like a record: a larger window may complete the chain
                            if (ok > 0 || q == p) {
                                if (q + 36 <= tmp.size()) {
                                    const int32_t bs = rdi32(tmp.data() + q);
                                    if (bs >= 32 && bs <= (1 << 24) && q + 4 + (size_t)bs > tmp.size()) short_of_data = true;
                                } else if (ok > 0) short_of_data = true;
                            }
                            break;
                        }
                        ok++; q = nx;
                        if (at_eof && q + 36 > tmp.size()) { ok = 12; break; }      // the chain ran into the end of the file
                    }
                    if (ok >= 12) { *uoff = M.blks[b1].dst + p; *ref = rdi32(tmp.data() + p + 4); return 1; }
                }
                if (!short_of_data || at_eof || win >= 256) break;
            }
        }
        return b1 >= nb ? 0 : -1;
    };
    const int32_t INF = 0x7fffffff;
    bool uncertain = false;
    // first member b in [b0, nb] whose first starting record has refID >= r (unmapped = -1 sorts last; no boundary = past the end).
    // The refIDs met on the way must ascend with the member index: anything else is not a coordinate-sorted file.
    std::vector<std::pair<size_t, int32_t>> samples;
    auto search = [&](int32_t r) -> size_t {
        size_t lo = b0, hi = nb;
        while (lo < hi) {
            const size_t m = (lo + hi) >> 1;
            uint64_t u; int32_t ref;
            int32_t key = INF;
            const int pr = probe(m, &u, &ref);
            if (pr < 0) { uncertain = true; return nb; }
            if (pr > 0) key = ref < 0 ? INF : ref;
            samples.emplace_back(m, key);
            if (key >= r) hi = m; else lo = m + 1;
        }
        return lo;
    };
    // a header that declares another order settles it without a search
    {
        const int32_t l_text = rdi32(head.data() + 4);
        const std::string_view text((const char *)head.data() + 8, (size_t)l_text);
        const size_t hd = text.rfind("@HD", 0) == 0 ? 0 : std::string_view::npos;
        if (hd == 0) {
            const std::string_view line = text.substr(0, text.find('\n'));
            if (line.find("SO:unsorted") != std::string_view::npos || line.find("SO:queryname") != std::string_view::npos) return PHZ_E_UNSUPPORTED;
        }
    }
    std::vector<size_t> begin_blk((size_t)n_ref + 1, nb);
    std::vector<char> want((size_t)n_ref, 0);
    for (int i = 0; i < n_ref; i++) {
        if (!ref_names) { want[(size_t)i] = 1; continue; }
        for (int k = 0; k < n_names; k++) if (refs[(size_t)i].first == ref_names[k]) want[(size_t)i] = 1;
    }
    if (ref_bytes) {
        for (int i = 0; i <= n_ref; i++) begin_blk[(size_t)i] = i == 0 ? b0 : search(i);
        for (int i = 0; i < n_ref && i < max_refs; i++) {
            const size_t a = begin_blk[(size_t)i], b = begin_blk[(size_t)i + 1];
            const size_t fa = a < nb ? M.blks[a].off : M.fsz, fb = b < nb ? M.blks[b].off : M.fsz;
            ref_bytes[i] = (int64_t)(fb > fa ? fb - fa : 0);
        }
    }
    laps.lap("reference boundary search");
    *first_record_out = first_record;
    if (uncertain) return PHZ_E_UNSUPPORTED;
    if (!want_pieces) return PHZ_OK;
    // runs of consecutive wanted references -> byte ranges [u_begin, u_end) of the uncompressed stream, cut at record boundaries
    uint64_t total_u = M.total;
    for (int i = 0; i < n_ref;) {
        if (!want[(size_t)i]) { i++; continue; }
        int j = i;
        while (j + 1 < n_ref && want[(size_t)j + 1]) j++;
        const size_t bs = ref_bytes ? begin_blk[(size_t)i] : (i == 0 ? b0 : search(i));
        const size_t be = ref_bytes ? begin_blk[(size_t)j + 1] : search(j + 1);
        uint64_t u0 = first_record, u1 = total_u; int32_t rr;
        if (uncertain) return PHZ_E_UNSUPPORTED;
        const size_t sb = bs > b0 ? bs - 1 : b0;                  // records of reference i may begin in the member before bs
        if (sb > b0) { const int pr = probe(sb, &u0, &rr); if (pr < 0) return PHZ_E_UNSUPPORTED; if (pr == 0) u0 = M.blks[sb].dst; }
        if (be < nb) { const int pr = probe(be, &u1, &rr); if (pr < 0) return PHZ_E_UNSUPPORTED; if (pr == 0) u1 = total_u; }
        if (!pieces.empty() && u0 < pieces.back().u1) u0 = pieces.back().u1;       // the member before this run may belong to the previous piece
        if (u1 > u0) pieces.push_back({u0, u1});
        i = j + 1;
    }
    std::sort(samples.begin(), samples.end());
    for (size_t t = 1; t < samples.size(); t++)
        if (samples[t].second < samples[t - 1].second) return PHZ_E_UNSUPPORTED;      // refIDs do not ascend along the file: not coordinate-sorted
    return PHZ_OK;
}

int phz_bam_plan_file(const char *path, const char *const *ref_names, int n_names, PhzBamPlan *out) {
    Laps laps("bam plan");
    BgzfMap *M = new BgzfMap();
    if (int st = M->open(path, true)) { delete M; return st == PHZ_E_UNSUPPORTED ? PHZ_E_ARG : st; }
    laps.lap("member table");
    std::vector<BamPiece> pieces;
    if (int st = bam_plan(*M, ref_names, n_names, nullptr, 0, true, out->refs, out->head, &out->first_record, pieces, laps)) { delete M; return st; }
    for (auto &p : pieces) out->pieces.emplace_back(p.u0, p.u1);
    const size_t nb = M->blks.size();
    size_t b = 0;
    for (auto &p : pieces) {
        size_t lo = b, hi = nb;       // first member overlapping u0 (pieces ascend, members are taken once)
        while (lo < hi) { const size_t m = (lo + hi) >> 1; if (M->blks[m].dst + M->blks[m].isize <= p.u0) lo = m + 1; else hi = m; }
        for (b = lo; b < nb && M->blks[b].dst < p.u1; b++) {
            if (!out->members.empty() && out->members.back().dst == M->blks[b].dst) continue;
            out->members.push_back({(uint64_t)M->blks[b].off, (uint32_t)M->blks[b].csize, (uint32_t)M->blks[b].isize, (uint64_t)M->blks[b].dst, M->blks[b].crc});
        }
        if (b > lo) b--;              // the last member of this piece may also be the first of the next one
    }
    out->file = M->f; out->file_size = M->fsz; out->owner = M;
    laps.lap("pieces + member list");
    return PHZ_OK;
}

const uint8_t *phz_bam_plan_map(PhzBamPlan *p) {
    if (p && p->owner && !p->file) p->file = ((BgzfMap *)p->owner)->map();
    return p ? p->file : nullptr;
}

void phz_bam_plan_release(PhzBamPlan *p) {
    if (p && p->owner) { delete (BgzfMap *)p->owner; p->owner = nullptr; p->file = nullptr; }
}

extern "C" {

int phz_bam_open(const char *path, int threads, phz_bam **out) {
    *out = nullptr;
    phz_bam *h = new phz_bam();
    if (int st = inflate_bgzf_file(path, threads, h->b.data)) { delete h; return st == PHZ_E_UNSUPPORTED ? PHZ_E_ARG : st; }
    if (parse_bam_header(h->b.data.data(), h->b.data.size(), h->b.refs, &h->b.first_record) != 0) { delete h; return PHZ_E_ARG; }
    *out = h;
    return PHZ_OK;
}

// Chromosome-restricted open (what `samtools view BAM 'chr':` does with the .bai, phaser/phaser.py:1346; here without an index
// file): the BGZF member table is read from the member headers alone, the members in which the wanted references begin / end are
// found by binary search (a coordinate-sorted BAM keeps every reference's records contiguous; a probe inflates two members and
// looks for the first record boundary with the same plausibility chain the parallel record hop uses), and only the members that
// hold wanted records are inflated.  The result is a stream cut at record boundaries, i.e. a valid record chain: phz_bam_decode
// works on it unchanged.  ref_bytes (may be NULL) receives, per reference, the compressed bytes between the members where it and
// the next reference begin: a proxy of its record count, available before anything is decoded (LPT weights).
int phz_bam_open_refs(const char *path, int threads, const char *const *ref_names, int n_names, int64_t *ref_bytes, int max_refs,
                      phz_bam **out) {
    if (out) *out = nullptr;
    Laps laps(out ? "bam open" : "bam weights");
    BgzfMap M;
    if (int st = M.open(path, out == nullptr)) return st == PHZ_E_UNSUPPORTED ? PHZ_E_ARG : st;
    laps.lap("member table");
    phz_bam *h = out ? new phz_bam() : nullptr;
    std::vector<std::pair<std::string, int32_t>> refs_local;
    std::vector<std::pair<std::string, int32_t>> &refs = h ? h->b.refs : refs_local;
    std::vector<uint8_t> head;
    size_t first_record = 0;
    std::vector<BamPiece> pieces;
    if (int st = bam_plan(M, ref_names, n_names, ref_bytes, max_refs, h != nullptr, refs, head, &first_record, pieces, laps)) {
        delete h;
        // reference boundaries that cannot be proven (records longer than the probe window, a file that is not coordinate-sorted): the
        // whole file through the order-agnostic full open -- slower, never lossy; the caller's reference mask still selects the records
        if (st == PHZ_E_UNSUPPORTED && out) return phz_bam_open(path, threads, out);
        return st;
    }
    if (!h) return PHZ_OK;
    const size_t nb = M.blks.size();
    size_t b0 = 0;
    while (b0 + 1 < nb && M.blks[b0 + 1].dst <= first_record) b0++;
    size_t out_size = first_record;
    for (auto &p : pieces) out_size += (size_t)(p.u1 - p.u0);
    if (!h->b.data.resize(out_size)) { delete h; return PHZ_E_NOMEM; }
    memcpy(h->b.data.data(), head.data(), first_record);
    laps.lap("piece boundaries + buffer");
    // inflate plan: members fully inside a piece go straight to their place, edge members through a scratch buffer
    struct Job { size_t blk; size_t dst; size_t skip, take; };
    std::vector<Job> jobs;
    size_t base = first_record;
    for (auto &p : pieces) {
        size_t b = b0;
        {   // first member overlapping u0
            size_t lo = 0, hi = nb;
            while (lo < hi) { const size_t m = (lo + hi) >> 1; if (M.blks[m].dst + M.blks[m].isize <= p.u0) lo = m + 1; else hi = m; }
            b = lo;
        }
        for (; b < nb && M.blks[b].dst < p.u1; b++) {
            const uint64_t s0 = std::max<uint64_t>(M.blks[b].dst, p.u0), s1 = std::min<uint64_t>(M.blks[b].dst + M.blks[b].isize, p.u1);
            if (s1 > s0) jobs.push_back({b, base + (size_t)(s0 - p.u0), (size_t)(s0 - M.blks[b].dst), (size_t)(s1 - s0)});
        }
        base += (size_t)(p.u1 - p.u0);
    }
    const int nt = n_threads(threads);
    std::atomic<size_t> next(0);
    std::atomic<bool> bad(false);
    std::vector<std::thread> th;
    uint8_t *dstbuf = h->b.data.data();
    if (!M.map()) { delete h; return PHZ_E_NOMEM; }
    for (int t = 0; t < nt; t++)
        th.emplace_back([&] {
            std::vector<uint8_t> scratch;
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= jobs.size()) break;
                const Job &j = jobs[k];
                const auto &B = M.blks[j.blk];
                if (j.skip == 0 && j.take == B.isize) { if (!inflate_block(M.f + B.off, B.csize, dstbuf + j.dst, B.isize) || !crc_ok(dstbuf + j.dst, B.isize, B.crc)) bad = true; }
                else {
                    scratch.resize(B.isize);
                    if (!inflate_block(M.f + B.off, B.csize, scratch.data(), B.isize) || !crc_ok(scratch.data(), B.isize, B.crc)) { bad = true; continue; }
                    memcpy(dstbuf + j.dst, scratch.data() + j.skip, j.take);
                }
            }
        });
    for (auto &t : th) t.join();
    laps.lap("parallel inflate");
    if (bad) { delete h; return PHZ_E_ARG; }
    h->b.first_record = first_record;
    *out = h;
    return PHZ_OK;
}

int phz_bam_close(phz_bam *h) { delete h; return PHZ_OK; }
int phz_bam_n_ref(const phz_bam *h) { return (int)h->b.refs.size(); }
const char *phz_bam_ref_name(const phz_bam *h, int i) { return h->b.refs[(size_t)i].first.c_str(); }
int64_t phz_bam_ref_length(const phz_bam *h, int i) { return h->b.refs[(size_t)i].second; }

// Decode + filter + pack.  ref_mask[i] != 0 selects reference i (NULL = all).  Returns the number of references
// that received at least one record; shards are then read with phz_bam_shard().
int phz_bam_decode(phz_bam *h, const uint8_t *ref_mask, int min_mapq, int flag_required, int flag_forbidden, double isize_cutoff,
                   int threads, int *n_shards) {
    Bam &b = h->b;
    Laps laps("bam decode");
    const uint8_t *d = b.data.data();
    const size_t n = b.data.size();
    const int n_ref = (int)b.refs.size();
    // pass 1: hop over the record chain, apply the filters, size everything.  The chain is sequential by nature (every
    // record's length sits in its own header), so the stream is cut into segments whose first record boundary is GUESSED by
    // a plausibility scan, every segment is hopped by its own thread, and the guess is VERIFIED afterwards: segment k must
    // end exactly where segment k+1 began (segment 0 starts at the true first record, so this proves every boundary).
    // Any mismatch falls back to the plain sequential hop.
    struct Rec { size_t off; int32_t ref; uint32_t n_ops, nb; uint32_t l_qn; };      // l_qn: QNAME length without the NUL
    auto plausible = [&](size_t p) -> size_t {          // -> offset of the next record, 0 when p cannot start a record
        if (p + 36 > n) return 0;
        const int32_t bs = rdi32(d + p);
        if (bs < 32 || bs > (1 << 24) || p + 4 + (size_t)bs > n) return 0;
        const uint8_t *r = d + p + 4;
        const int32_t ref = rdi32(r), pos0 = rdi32(r + 4), l_seq = rdi32(r + 16), nref = rdi32(r + 20);
        const uint32_t l_rn = r[8], n_cig = rd16(r + 12), flag = rd16(r + 14);
        if (ref < -1 || ref >= n_ref || nref < -1 || nref >= n_ref || pos0 < -1 || l_seq < 0 || l_rn < 2 || flag >= 4096 || rdi32(r + 24) < -1) return 0;
        const size_t need = 32 + (size_t)l_rn + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
        if (need > (size_t)bs) return 0;
        if (r[32 + l_rn - 1] != 0) return 0;                 // read name is NUL-terminated
        for (uint32_t k = 0; k + 1 < l_rn; k++) if (r[32 + k] < 33 || r[32 + k] > 126) return 0;
        return p + 4 + (size_t)bs;
    };
    // hop() returns where the chain stopped; *corrupt is set when a record does not fit its own block_size (or the stream ends
    // inside a record): the fixed fields, QNAME, CIGAR, SEQ and QUAL of EVERY record are proven to lie inside the inflated
    // buffer here, so the packer below never reads past a record
    auto hop = [&](size_t p, size_t stop, auto &out, std::vector<int32_t> &last_pos, bool *unsorted, bool *corrupt) -> size_t {
        while (p < stop) {
            if (p + 4 > n) { *corrupt = true; return p; }
            const int32_t bs = rdi32(d + p);
            if (bs < 32 || (size_t)bs > n - p - 4) { *corrupt = true; return p; }
            const uint8_t *r = d + p + 4;
            const int32_t ref = rdi32(r);
            const uint32_t l_rn = r[8], mapq = r[9], n_cig = rd16(r + 12), flag = rd16(r + 14);
            const int32_t l_seq = rdi32(r + 16), tlen = rdi32(r + 28);
            if (l_seq < 0 || l_rn < 1 ||
                32 + (size_t)l_rn + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq > (size_t)bs) { *corrupt = true; return p; }
            bool keep = ref >= 0 && ref < n_ref && (!ref_mask || ref_mask[ref]) && (int)mapq >= min_mapq &&
                        ((int)flag & flag_required) == flag_required && ((int)flag & flag_forbidden) == 0;
            if (keep && isize_cutoff != 0) { const double tl = tlen < 0 ? -(double)tlen : (double)tlen; keep = tl <= isize_cutoff; }
            if (keep) {
                const int32_t pos0 = rdi32(r + 4);
                if (pos0 < last_pos[(size_t)ref]) *unsorted = true;           // BAM is not coordinate-sorted
                last_pos[(size_t)ref] = pos0;
                const uint8_t *cig = r + 32 + l_rn;
                const uint8_t *qual = cig + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2;
                uint32_t nb;
                if (l_seq <= 0) nb = 1;                               // SEQ '*' / QUAL '*': zip() keeps one character
                else nb = qual[0] == 0xFF ? 1u : (uint32_t)l_seq;     // QUAL '*'
                out.push_back({p + 4, ref, (uint32_t)norm_ops(cig, (int)n_cig, (int)nb, nullptr), nb, l_rn ? l_rn - 1 : 0u});
            }
            p += 4 + (size_t)bs;
        }
        return p;
    };
    RawVec<Rec> recs;
    bool unsorted = false;
    bool done = false;
    const int nt1 = n_threads(threads);
    size_t par_min = 64u << 20;
    { const char *e = getenv("PHZ_BAM_PAR_MIN"); if (e) par_min = (size_t)atoll(e); }
    if (nt1 > 1 && n - b.first_record > par_min) {
        const int K = nt1 * 4;
        std::vector<size_t> seg_start((size_t)K + 1, n);
        seg_start[0] = b.first_record;
        std::vector<std::vector<Rec>> part((size_t)K);
        std::vector<std::vector<int32_t>> lastp((size_t)K, std::vector<int32_t>((size_t)n_ref, -1)), firstp((size_t)K, std::vector<int32_t>((size_t)n_ref, -1));
        std::vector<size_t> seg_end((size_t)K, 0);
        std::vector<uint8_t> bad((size_t)K, 0);
        {   // guessed boundaries
            std::atomic<int> next(1);
            std::vector<std::thread> th;
            for (int t = 0; t < nt1; t++)
                th.emplace_back([&] {
                    for (;;) {
                        const int k = next.fetch_add(1);
                        if (k >= K) break;
                        size_t p = b.first_record + (n - b.first_record) / (size_t)K * (size_t)k;
                        const size_t limit = std::min(n, p + (size_t)(32 << 20));
                        size_t found = n;
                        for (; p < limit; p++) {
                            size_t q = p; int ok = 0;
                            while (ok < 12) { const size_t nx = plausible(q); if (!nx) break; ok++; q = nx; if (q + 4 > n) { ok = 12; break; } }
                            if (ok >= 12) { found = p; break; }
                        }
                        seg_start[(size_t)k] = found;
                    }
                });
            for (auto &x : th) x.join();
        }
        for (int k = 1; k <= K; k++) if (seg_start[(size_t)k] < seg_start[(size_t)k - 1]) seg_start[(size_t)k] = seg_start[(size_t)k - 1];
        {
            std::atomic<int> next(0);
            std::vector<std::thread> th;
            for (int t = 0; t < nt1; t++)
                th.emplace_back([&] {
                    for (;;) {
                        const int k = next.fetch_add(1);
                        if (k >= K) break;
                        part[(size_t)k].reserve((seg_start[(size_t)k + 1] - seg_start[(size_t)k]) / 180 + 64);
                        bool u = false, cr = false;
                        seg_end[(size_t)k] = hop(seg_start[(size_t)k], seg_start[(size_t)k + 1], part[(size_t)k], lastp[(size_t)k], &u, &cr);
                        bad[(size_t)k] = u ? 1 : 0;
                        if (cr) seg_end[(size_t)k] = (size_t)-1;          // never equals a segment start: forces the sequential hop
                        for (const Rec &x : part[(size_t)k])
                            if (firstp[(size_t)k][(size_t)x.ref] < 0) firstp[(size_t)k][(size_t)x.ref] = rdi32(d + x.off + 4) + 1;   // +1: 0 is a valid POS
                    }
                });
            for (auto &x : th) x.join();
        }
        bool ok = true;      // every guessed boundary must be where the previous segment's chain arrived
        for (int k = 0; k < K; k++)
            if (seg_start[(size_t)k + 1] < n && seg_end[(size_t)k] != seg_start[(size_t)k + 1]) ok = false;
        if (ok) {
            std::vector<size_t> at((size_t)K + 1, 0);
            for (int k = 0; k < K; k++) at[(size_t)k + 1] = at[(size_t)k] + part[(size_t)k].size();
            recs.resize(at[(size_t)K]);            // uninitialised; the segments copy themselves in (in parallel)
            std::vector<int32_t> last((size_t)n_ref, -1);
            for (int k = 0; k < K; k++) {
                if (bad[(size_t)k]) unsorted = true;
                for (int r2 = 0; r2 < n_ref; r2++) {
                    if (firstp[(size_t)k][(size_t)r2] >= 0 && firstp[(size_t)k][(size_t)r2] - 1 < last[(size_t)r2]) unsorted = true;
                    if (lastp[(size_t)k][(size_t)r2] >= 0 || firstp[(size_t)k][(size_t)r2] >= 0) last[(size_t)r2] = std::max(last[(size_t)r2], lastp[(size_t)k][(size_t)r2]);
                }
            }
            {
                std::atomic<int> next(0);
                std::vector<std::thread> th;
                for (int t = 0; t < nt1; t++)
                    th.emplace_back([&] {
                        for (;;) {
                            const int k = next.fetch_add(1);
                            if (k >= K) break;
                            if (!part[(size_t)k].empty()) memcpy(recs.data() + at[(size_t)k], part[(size_t)k].data(), part[(size_t)k].size() * sizeof(Rec));
                            std::vector<Rec>().swap(part[(size_t)k]);
                        }
                    });
                for (auto &x : th) x.join();
            }
            done = true;
        }
    }
    if (!done) {
        recs.clear(); unsorted = false;
        recs.reserve(n / 180 + 1024);
        std::vector<int32_t> last_pos((size_t)n_ref, -1);
        bool corrupt = false;
        hop(b.first_record, n, recs, last_pos, &unsorted, &corrupt);
        if (corrupt) { b.err = "truncated or corrupt BAM record"; return PHZ_E_ARG; }
    }
    laps.lap("record hop + filters");
    if (getenv("PHZ_TIMING")) fprintf(stderr, "[phz timing]   bam record hop: %s, %zu records kept\n", done ? "parallel segments, boundaries verified" : "sequential", recs.size());
    if (unsorted) return PHZ_E_UNSUPPORTED;
    // bucket by reference, in reference order (file order within a reference is preserved).  A coordinate-sorted BAM has its
    // references in ascending order already, so the permutation is the identity; anything else goes through a counting sort.
    std::vector<size_t> count((size_t)n_ref + 1, 0);
    const int ntb = n_threads(threads);
    {
        std::vector<std::vector<size_t>> hc((size_t)ntb, std::vector<size_t>((size_t)n_ref + 1, 0));
        std::vector<std::thread> th;
        for (int t = 0; t < ntb; t++)
            th.emplace_back([&, t] {
                const size_t lo = recs.size() * (size_t)t / (size_t)ntb, hi = recs.size() * (size_t)(t + 1) / (size_t)ntb;
                for (size_t i = lo; i < hi; i++) hc[(size_t)t][(size_t)recs[i].ref + 1]++;
            });
        for (auto &x : th) x.join();
        for (auto &h2 : hc) for (int i = 0; i <= n_ref; i++) count[(size_t)i] += h2[(size_t)i];
    }
    for (int i = 0; i < n_ref; i++) count[(size_t)i + 1] += count[(size_t)i];
    bool grouped = true;
    for (size_t i = 1; i < recs.size() && grouped; i += 4096) {        // sampled check first, exact check below only if it passes
        if (recs[i].ref < recs[i - 1].ref) grouped = false;
    }
    if (grouped) {
        std::atomic<bool> bad(false);
        std::vector<std::thread> th;
        for (int t = 0; t < ntb; t++)
            th.emplace_back([&, t] {
                const size_t lo = std::max<size_t>(1, recs.size() * (size_t)t / (size_t)ntb), hi = recs.size() * (size_t)(t + 1) / (size_t)ntb;
                for (size_t i = lo; i < hi; i++) if (recs[i].ref < recs[i - 1].ref) { bad = true; break; }
            });
        for (auto &x : th) x.join();
        grouped = !bad;
    }
    std::vector<uint32_t> order;
    if (!grouped) {
        order.resize(recs.size());
        std::vector<size_t> cur(count.begin(), count.end() - 1);
        for (size_t i = 0; i < recs.size(); i++) order[cur[(size_t)recs[i].ref]++] = (uint32_t)i;
    }
    auto rec_at = [&](size_t k) -> const Rec & { return grouped ? recs[k] : recs[order[k]]; };
    laps.lap("bucket by reference");
    b.shards.clear();
    for (int i = 0; i < n_ref; i++) {
        const size_t lo = count[(size_t)i], hi = count[(size_t)i + 1];
        if (hi == lo) continue;
        b.shards.emplace_back();
        Shard &s = b.shards.back();
        s.name = b.refs[(size_t)i].first;
        const size_t m = hi - lo;
        s.pos.resize(m); s.aln.resize(m); s.has_as.resize(m);
        s.cigar_off.resize(m + 1); s.seq_off.resize(m + 1); s.qname_off.resize(m + 1);
        // offsets = exclusive prefix sums of the per-record sizes: per-slice totals, then every slice fills its own part
        uint64_t co = 0, so = 0, qo = 0;
        {
            const int nsl = m < 65536 ? 1 : ntb;
            std::vector<uint64_t> tc((size_t)nsl + 1, 0), ts((size_t)nsl + 1, 0), tq((size_t)nsl + 1, 0);
            auto slice = [&](int t, size_t *a, size_t *z) { *a = m * (size_t)t / (size_t)nsl; *z = m * (size_t)(t + 1) / (size_t)nsl; };
            auto run = [&](auto fn) {
                if (nsl == 1) { fn(0); return; }
                std::vector<std::thread> th;
                for (int t = 0; t < nsl; t++) th.emplace_back([&, t] { fn(t); });
                for (auto &x : th) x.join();
            };
            run([&](int t) {
                size_t a, z; slice(t, &a, &z);
                uint64_t c2 = 0, s2 = 0, q2 = 0;
                for (size_t k = a; k < z; k++) { const Rec &x = rec_at(lo + k); c2 += x.n_ops; s2 += (x.nb + 3) / 4; q2 += x.l_qn; }
                tc[(size_t)t + 1] = c2; ts[(size_t)t + 1] = s2; tq[(size_t)t + 1] = q2;
            });
            for (int t = 0; t < nsl; t++) { tc[(size_t)t + 1] += tc[(size_t)t]; ts[(size_t)t + 1] += ts[(size_t)t]; tq[(size_t)t + 1] += tq[(size_t)t]; }
            co = tc[(size_t)nsl]; so = ts[(size_t)nsl]; qo = tq[(size_t)nsl];
            if (co >= (1ull << 31) || so >= (1ull << 31) || qo >= (1ull << 32)) { b.err = "shard exceeds 32-bit offsets"; return PHZ_E_UNSUPPORTED; }
            run([&](int t) {
                size_t a, z; slice(t, &a, &z);
                uint64_t c2 = tc[(size_t)t], s2 = ts[(size_t)t], q2 = tq[(size_t)t];
                for (size_t k = a; k < z; k++) {
                    const Rec &x = rec_at(lo + k);
                    s.cigar_off[k] = (uint32_t)c2; s.seq_off[k] = (uint32_t)s2; s.qname_off[k] = (uint32_t)q2;
                    c2 += x.n_ops; s2 += (x.nb + 3) / 4; q2 += x.l_qn;
                }
            });
        }
        s.cigar_off[m] = (uint32_t)co; s.seq_off[m] = (uint32_t)so; s.qname_off[m] = (uint32_t)qo;
        s.cigar.resize(co); s.qnames.resize(qo);
        if (!s.seq2.resize(so) || !s.qual.resize(so * 4)) return PHZ_E_NOMEM;
        // pass 2: pack in parallel
        const int nt = n_threads(threads);
        std::atomic<size_t> next(0);
        std::vector<std::thread> th;
        const size_t grain = 8192;
        for (int t = 0; t < nt; t++)
            th.emplace_back([&] {
                static const int8_t code_of[16] = {-2, 0, 1, -2, 2, -2, -2, -2, 3, -2, -2, -2, -2, -1, -2, -1};   // =ACMGRSVTWYHKDBN
                for (;;) {
                    const size_t k0 = next.fetch_add(grain);
                    if (k0 >= m) break;
                    const size_t k1 = k0 + grain < m ? k0 + grain : m;
                    for (size_t k = k0; k < k1; k++) {
                        const Rec &x = rec_at(lo + k);
                        const uint8_t *r = d + x.off;
                        const uint32_t l_rn = r[8], n_cig = rd16(r + 12);
                        const int32_t l_seq = rdi32(r + 16);
                        const int32_t bs = rdi32(r - 4);
                        s.pos[k] = rdi32(r + 4) + 1;
                        memcpy(s.qnames.data() + s.qname_off[k], r + 32, l_rn ? l_rn - 1 : 0);
                        const uint8_t *cig = r + 32 + l_rn;
                        norm_ops(cig, (int)n_cig, (int)x.nb, s.cigar.data() + s.cigar_off[k]);
                        const uint8_t *sq = cig + 4 * (size_t)n_cig;
                        const uint8_t *ql = sq + ((size_t)(l_seq > 0 ? l_seq : 0) + 1) / 2;
                        uint8_t *o2 = s.seq2.data() + s.seq_off[k];
                        uint8_t *oq = s.qual.data() + (size_t)s.seq_off[k] * 4;
                        memset(o2, 0, (size_t)(x.nb + 3) / 4);
                        memset(oq + (x.nb & ~3u), 0, (size_t)((x.nb + 3) / 4 * 4 - (x.nb & ~3u)));     // last group incl. padding
                        if (l_seq <= 0) {                  // SEQ '*' QUAL '*': one IUPAC-other character with phred 9
                            o2[0] = 1; oq[0] = (uint8_t)(9 | 0x80);
                        } else {
                            const bool noq = ql[0] == 0xFF;
                            for (uint32_t j = 0; j < x.nb; j++) {
                                const uint8_t nib = (j & 1) ? (sq[j >> 1] & 15) : (sq[j >> 1] >> 4);
                                const int c = code_of[nib];
                                uint8_t q = noq ? 9 : (ql[j] > 127 ? 127 : ql[j]);
                                uint8_t code2;
                                if (c >= 0) code2 = (uint8_t)c;
                                else { code2 = c == -1 ? 0 : 1; q |= 0x80; }
                                o2[j >> 2] |= (uint8_t)(code2 << (2 * (j & 3)));
                                oq[j] = q;
                            }
                        }
                        int32_t as = 0;
                        const uint8_t *aux = ql + (l_seq > 0 ? l_seq : 0);
                        const bool has = aux_as(aux, r + bs, &as);
                        s.aln[k] = has ? as : 0; s.has_as[k] = has ? 1 : 0;
                    }
                }
            });
        for (auto &t : th) t.join();
    }
    *n_shards = (int)b.shards.size();
    laps.lap("offsets + packing (all shards)");
    // the inflated stream is not needed once the shards are packed (decode is a one-shot call); unmapping ~10 GB takes most of a
    // second, so a helper thread does it while the caller goes on
    { RawBuf gone(std::move(b.data)); std::thread([g = std::move(gone)]() mutable { g.drop(); }).detach(); }
    laps.lap("release of the inflated stream");
    return PHZ_OK;
}

int phz_bam_shard(phz_bam *h, int i, phz_host_shard *out) {
    if (i < 0 || (size_t)i >= h->b.shards.size()) return PHZ_E_ARG;
    Shard &s = h->b.shards[(size_t)i];
    out->ref_name = s.name.c_str();
    out->n_reads = (int64_t)s.pos.size(); out->n_ops = (int64_t)s.cigar.size(); out->n_seq_bytes = (int64_t)s.seq2.size();
    out->pos = s.pos.data(); out->cigar_off = s.cigar_off.data(); out->cigar = s.cigar.data(); out->seq_off = s.seq_off.data();
    out->seq2 = s.seq2.data(); out->qual = s.qual.data(); out->aln_score = s.aln.data(); out->has_as = s.has_as.data();
    out->qname_off = s.qname_off.data(); out->qnames = s.qnames.data();
    return PHZ_OK;
}

int phz_interner_create(phz_interner **out) { *out = new phz_interner(); return PHZ_OK; }
int phz_interner_destroy(phz_interner *it) { delete it; return PHZ_OK; }
int64_t phz_interner_size(const phz_interner *it) { return (int64_t)it->names.size(); }

// QNAME -> id in first-appearance order (mates and the same template in later BAMs get the same id).  Parallel: names are
// bucketed by hash; every bucket resolves its names against its own table in input order and marks the first occurrence of
// each new name; a prefix sum over those marks numbers the new names in input order; a last sweep writes the ids.
int phz_intern(phz_interner *it, const char *blob, const uint32_t *off, int64_t n, int32_t *out_id) {
    constexpr int P = phz_interner::P;
    if (n <= 0) return PHZ_OK;
    const int nt = n_threads(0);
    auto par = [&](int64_t count, auto fn) {          // fn(lo, hi) over [0, count) in nt slices
        if (count < 65536 || nt == 1) { fn((int64_t)0, count); return; }
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back([&, t] { fn(count * t / nt, count * (t + 1) / nt); });
        for (auto &x : th) x.join();
    };
    auto name_of = [&](int64_t i) { return std::string_view(blob + off[i], off[i + 1] - off[i]); };
    // scratch arrays are default-initialised (no zero fill: every element is written before it is read)
    std::unique_ptr<uint64_t[]> hv(new uint64_t[(size_t)n]);
    par(n, [&](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; i++) hv[(size_t)i] = hash_name(blob + off[i], off[i + 1] - off[i]); });
    auto bucket_of = [&](int64_t i) { return (int)(hv[(size_t)i] >> 58); };        // top 6 bits; the table uses the low ones
    // bucket lists in input order: per-slice histograms -> offsets -> scatter (all parallel)
    const int ns = (n < 65536 || nt == 1) ? 1 : nt;
    std::vector<std::vector<int64_t>> hist((size_t)ns, std::vector<int64_t>(P, 0));
    par(n, [&](int64_t lo, int64_t hi) { auto &hc = hist[ns == 1 ? 0 : (size_t)((lo * nt + n - 1) / n)]; for (int64_t i = lo; i < hi; i++) hc[(size_t)bucket_of(i)]++; });
    std::vector<int64_t> start(P + 1, 0);
    for (int p = 0; p < P; p++) { int64_t c = 0; for (int t = 0; t < ns; t++) c += hist[(size_t)t][(size_t)p]; start[(size_t)p + 1] = start[(size_t)p] + c; }
    {   // turn the histograms into write cursors: slice t of bucket p starts after slices < t
        std::vector<int64_t> run(start.begin(), start.end() - 1);
        for (int t = 0; t < ns; t++) for (int p = 0; p < P; p++) { const int64_t c = hist[(size_t)t][(size_t)p]; hist[(size_t)t][(size_t)p] = run[(size_t)p]; run[(size_t)p] += c; }
    }
    std::unique_ptr<int32_t[]> order(new int32_t[(size_t)n]);
    par(n, [&](int64_t lo, int64_t hi) { auto &cur = hist[ns == 1 ? 0 : (size_t)((lo * nt + n - 1) / n)]; for (int64_t i = lo; i < hi; i++) order[(size_t)cur[(size_t)bucket_of(i)]++] = (int32_t)i; });
    // rep[i] >= 0: index of the first occurrence of this new name in the input; < 0: -(existing id) - 1
    std::unique_ptr<int32_t[]> rep(new int32_t[(size_t)n]);
    std::unique_ptr<uint8_t[]> first(new uint8_t[(size_t)n]);
    {
        std::atomic<int> next(0);
        auto work = [&] {
            for (;;) {
                const int p = next.fetch_add(1);
                if (p >= P) break;
                phz_interner::Part &T = it->part[p];
                FlatMap fresh;
                fresh.reserve((size_t)(start[(size_t)p + 1] - start[(size_t)p]));
                for (int64_t k = start[(size_t)p]; k < start[(size_t)p + 1]; k++) {
                    const int32_t i = order[(size_t)k];
                    const std::string_view nm = name_of(i);
                    const uint64_t h = hv[(size_t)i];
                    const int32_t known = T.ids.find(h, [&](int32_t id) { return it->names[(size_t)id] == nm; });
                    first[(size_t)i] = 0;
                    if (known >= 0) { rep[(size_t)i] = -known - 1; continue; }
                    const int32_t seen = fresh.find(h, [&](int32_t j) { return name_of(j) == nm; });
                    if (seen >= 0) { rep[(size_t)i] = seen; continue; }
                    fresh.insert(h, i);
                    rep[(size_t)i] = i; first[(size_t)i] = 1;
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < std::min(nt, P); t++) th.emplace_back(work);
        for (auto &x : th) x.join();
    }
    const int64_t base = (int64_t)it->names.size();
    std::unique_ptr<int32_t[]> rank(new int32_t[(size_t)n]);
    int64_t nnew = 0;
    {   // exclusive prefix sum of the first-occurrence marks: per-slice totals, then every slice numbers its own part
        std::vector<int64_t> tot((size_t)ns + 1, 0);
        par(n, [&](int64_t lo, int64_t hi) { int64_t c = 0; for (int64_t i = lo; i < hi; i++) c += first[(size_t)i]; tot[(ns == 1 ? 0 : (size_t)((lo * nt + n - 1) / n)) + 1] = c; });
        for (int t = 0; t < ns; t++) tot[(size_t)t + 1] += tot[(size_t)t];
        nnew = tot[(size_t)ns];
        par(n, [&](int64_t lo, int64_t hi) { int64_t c = tot[ns == 1 ? 0 : (size_t)((lo * nt + n - 1) / n)]; for (int64_t i = lo; i < hi; i++) { rank[(size_t)i] = (int32_t)c; c += first[(size_t)i]; } });
    }
    if (base + nnew > 0x7fffffff) return PHZ_E_ARG;
    par(n, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; i++) {
            const int32_t r = rep[(size_t)i];
            out_id[i] = r < 0 ? -r - 1 : (int32_t)(base + rank[(size_t)r]);
        }
    });
    // the new names enter their bucket's table (own copy of the text: the caller's blob goes away)
    it->names.resize((size_t)(base + nnew));
    {
        std::atomic<int> next(0);
        auto work = [&] {
            for (;;) {
                const int p = next.fetch_add(1);
                if (p >= P) break;
                phz_interner::Part &T = it->part[p];
                size_t add = 0;
                for (int64_t k = start[(size_t)p]; k < start[(size_t)p + 1]; k++) add += first[(size_t)order[(size_t)k]];
                T.ids.reserve(T.ids.count + add);
                for (int64_t k = start[(size_t)p]; k < start[(size_t)p + 1]; k++) {
                    const int32_t i = order[(size_t)k];
                    if (!first[(size_t)i]) continue;
                    const std::string_view nm = name_of(i);
                    const std::string_view kept(T.keep(nm), nm.size());
                    const int32_t id = (int32_t)(base + rank[(size_t)i]);
                    it->names[(size_t)id] = kept;
                    T.ids.insert(hv[(size_t)i], id);
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < std::min(nt, P); t++) th.emplace_back(work);
        for (auto &x : th) x.join();
    }
    return PHZ_OK;
}

// names of ids [0, size) as a blob + offsets (for --output_read_ids)
int phz_interner_names(const phz_interner *it, char *blob, int64_t blob_cap, uint32_t *off) {
    const size_t n = it->names.size();
    uint64_t o = 0;
    for (size_t i = 0; i < n; i++) {
        off[i] = (uint32_t)o;
        if ((int64_t)(o + it->names[i].size()) <= blob_cap) memcpy(blob + o, it->names[i].data(), it->names[i].size());
        o += it->names[i].size();
    }
    off[n] = (uint32_t)o;
    return (int64_t)o <= blob_cap ? PHZ_OK : PHZ_E_CAPACITY;
}

// ---- generic BGZF files (bgzip-compressed VCF in, phased VCF out) -------------------------------------------------------
static std::mutex g_mapped_mu;
static std::map<void *, size_t> g_mapped;          // buffers handed out by phz_bgzf_read that are mappings (phz_buf_free unmaps them)

// Reads a whole BGZF file with parallel member inflate.  PHZ_E_UNSUPPORTED for plain gzip (the caller streams it itself).
int phz_bgzf_read(const char *path, int threads, char **data, int64_t *len) {
    if (!path || !data || !len) return PHZ_E_ARG;
    *data = nullptr; *len = 0;
    RawBuf buf;
    if (int st = inflate_bgzf_file(path, threads, buf)) return st;
    *len = (int64_t)buf.size();
    buf.data()[buf.size()] = 0;
    // a big text stays where the workers inflated it (an anonymous mapping of huge pages): phz_buf_free unmaps it.  Copying it into a malloc'd block -- 75 MB
    // of a whole-genome VCF, fresh pages touched by one thread -- was a third of this call.
    if (buf.mapped) {
        std::lock_guard<std::mutex> lk(g_mapped_mu);
        g_mapped[(void *)buf.p] = buf.mapped;
        *data = (char *)buf.p;
        buf.p = nullptr; buf.n = 0; buf.mapped = 0;
        return PHZ_OK;
    }
    *data = (char *)buf.release();
    return PHZ_OK;
}

void phz_buf_free(char *p) {
    if (!p) return;
    size_t mapped = 0;
    {
        std::lock_guard<std::mutex> lk(g_mapped_mu);
        auto it = g_mapped.find((void *)p);
        if (it != g_mapped.end()) { mapped = it->second; g_mapped.erase(it); }
    }
    if (mapped) munmap(p, mapped); else free(p);
}

// Writes data as a BGZF file (60,000-byte members deflated in parallel, EOF marker at the end) -- what `bgzip` produces
// for the phased VCF (phaser.py:1851).  side (may be empty) runs on a thread of its own next to the deflate workers; members (may be
// NULL) receives the member table of the written file: compressed offset / uncompressed start per member incl. the EOF marker, plus
// one closing entry.
static int bgzf_write_impl(const char *path, const char *data, int64_t len, int threads, int level, const std::function<void()> &side,
                           std::vector<uint64_t> *m_coff, std::vector<uint64_t> *m_ustart) {
    if (!path || (!data && len) || len < 0) return PHZ_E_ARG;
    const size_t BLK = 60000;
    const size_t nblk = ((size_t)len + BLK - 1) / BLK;
    std::vector<std::string> out(nblk);
    std::atomic<size_t> next(0);
    std::atomic<bool> bad(false);
    int nt = threads > 0 ? threads : (int)std::min(64u, std::max(1u, std::thread::hardware_concurrency()));     // deflate scales with the cores
    if ((size_t)nt > nblk) nt = (int)std::max<size_t>(1, nblk);
    std::vector<std::thread> th;
    if (side) th.emplace_back(side);
    for (int t = 0; t < nt; t++)
        th.emplace_back([&] {
            std::vector<uint8_t> tmp(BLK + 1024);
            // one deflate state per thread, reset between members (the state is ~270 KB: a fresh one per member is an allocation and
            // its page faults 1,700 times per 100 MB)
            z_stream zs; memset(&zs, 0, sizeof zs);
            if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad = true; return; }
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= nblk) break;
                const size_t lo = i * BLK, n = std::min(BLK, (size_t)len - lo);
                if (deflateReset(&zs) != Z_OK) { bad = true; break; }
                zs.next_in = (Bytef *)(data + lo); zs.avail_in = (uInt)n;
                zs.next_out = tmp.data(); zs.avail_out = (uInt)tmp.size();
                const int rc = deflate(&zs, Z_FINISH);
                const size_t csize = tmp.size() - zs.avail_out;
                if (rc != Z_STREAM_END || csize + 26 > 65536) { bad = true; break; }
                std::string &o = out[i];
                const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
                o.assign((const char *)hdr, 16);
                const uint16_t bsize = (uint16_t)(csize + 25);
                o.push_back((char)(bsize & 0xff)); o.push_back((char)(bsize >> 8));
                o.append((const char *)tmp.data(), csize);
                const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)(data + lo), (uInt)n), isz = (uint32_t)n;
                for (int k = 0; k < 4; k++) o.push_back((char)((crc >> (8 * k)) & 0xff));
                for (int k = 0; k < 4; k++) o.push_back((char)((isz >> (8 * k)) & 0xff));
            }
            deflateEnd(&zs);
        });
    for (auto &t : th) t.join();
    if (bad) return PHZ_E_ARG;
    if (m_coff && m_ustart) {
        m_coff->clear(); m_ustart->clear();
        uint64_t c = 0;
        for (size_t i = 0; i < nblk; i++) { m_coff->push_back(c); m_ustart->push_back((uint64_t)i * BLK); c += out[i].size(); }
        m_coff->push_back(c); m_ustart->push_back((uint64_t)len);                  // the EOF marker: an empty member
        m_coff->push_back(c + 28); m_ustart->push_back((uint64_t)len);
    }
    FILE *fp = fopen(path, "wb");
    if (!fp) return PHZ_E_ARG;
    bool ok = true;
    for (auto &o : out) ok = ok && fwrite(o.data(), 1, o.size(), fp) == o.size();
    static const uint8_t eof_marker[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    ok = ok && fwrite(eof_marker, 1, 28, fp) == 28;
    ok = fclose(fp) == 0 && ok;
    return ok ? PHZ_OK : PHZ_E_ARG;
}

int phz_bgzf_write(const char *path, const char *data, int64_t len, int threads, int level) {
    return bgzf_write_impl(path, data, len, threads, level, std::function<void()>(), nullptr, nullptr);
}

// ---- BAM writer for synthetic read batches (test / benchmark tooling: there is no samtools in the image) ----------------------
// Encodes fixed-length reads held as arrays into BAM records (same bytes as phaser_amd/bamio.py:write_bam: bin 4680,
// mate fields = own reference / position, tags NH:i:1 and AS:i), in parallel, and writes them BGZF-compressed.
int phz_bam_write(const char *path, int n_ref, const char *const *ref_names, const int32_t *ref_lens, const phz_read_batch *batches,
                  int n_batches, int threads) {
    if (!path || n_ref < 0 || (!batches && n_batches)) return PHZ_E_ARG;
    std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
    for (int i = 0; i < n_ref; i++) { text += "@SQ\tSN:"; text += ref_names[i]; text += "\tLN:"; text += std::to_string(ref_lens[i]); text += "\n"; }
    std::string head("BAM\1", 4);
    auto put32 = [](std::string &o, int32_t v) { o.append((const char *)&v, 4); };
    put32(head, (int32_t)text.size()); head += text; put32(head, n_ref);
    for (int i = 0; i < n_ref; i++) {
        const std::string nm(ref_names[i]);
        put32(head, (int32_t)nm.size() + 1); head += nm; head.push_back('\0'); put32(head, ref_lens[i]);
    }
    // record sizes -> offsets
    std::vector<std::vector<uint64_t>> offs((size_t)n_batches);
    uint64_t total = head.size();
    static const uint8_t NT16[5] = {1, 2, 4, 8, 15};       // A C G T N
    for (int b = 0; b < n_batches; b++) {
        const phz_read_batch &B = batches[b];
        const size_t plen = strlen(B.qname_prefix);
        offs[(size_t)b].resize((size_t)B.n + 1);
        for (int64_t i = 0; i < B.n; i++) {
            offs[(size_t)b][(size_t)i] = total;
            char num[24]; const int nd = snprintf(num, sizeof num, "%d", B.qid[i]);
            const uint64_t nops = (uint64_t)(B.cigar_off[i + 1] - B.cigar_off[i]);
            total += 4 + 32 + plen + (size_t)nd + 1 + 4 * nops + ((uint64_t)B.L + 1) / 2 + (uint64_t)B.L + 14;
        }
        offs[(size_t)b][(size_t)B.n] = total;
    }
    std::vector<char> raw(total);
    memcpy(raw.data(), head.data(), head.size());
    const int nt = n_threads(threads);
    for (int b = 0; b < n_batches; b++) {
        const phz_read_batch &B = batches[b];
        const size_t plen = strlen(B.qname_prefix);
        std::atomic<int64_t> next(0);
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++)
            th.emplace_back([&] {
                for (;;) {
                    const int64_t lo = next.fetch_add(65536);
                    if (lo >= B.n) break;
                    const int64_t hi = std::min<int64_t>(B.n, lo + 65536);
                    for (int64_t i = lo; i < hi; i++) {
                        char *o = raw.data() + offs[(size_t)b][(size_t)i];
                        const uint64_t size = offs[(size_t)b][(size_t)i + 1] - offs[(size_t)b][(size_t)i];
                        char num[24]; const int nd = snprintf(num, sizeof num, "%d", B.qid[i]);
                        const int32_t nops = (int32_t)(B.cigar_off[i + 1] - B.cigar_off[i]);
                        auto w32 = [&](int32_t v) { memcpy(o, &v, 4); o += 4; };
                        auto w16 = [&](uint16_t v) { memcpy(o, &v, 2); o += 2; };
                        w32((int32_t)(size - 4)); w32(B.ref_id); w32(B.pos[i] - 1);
                        *o++ = (char)(plen + (size_t)nd + 1); *o++ = (char)B.mapq[i]; w16(4680); w16((uint16_t)nops); w16((uint16_t)B.flag[i]);
                        w32(B.L); w32(B.ref_id); w32(B.pos[i] - 1 > 0 ? B.pos[i] - 1 : 0); w32(B.tlen[i]);
                        memcpy(o, B.qname_prefix, plen); o += plen; memcpy(o, num, (size_t)nd); o += nd; *o++ = 0;
                        memcpy(o, B.cigar + B.cigar_off[i], 4 * (size_t)nops); o += 4 * (size_t)nops;
                        const uint8_t *sq = B.seq + (size_t)i * (size_t)B.L, *ql = B.qual + (size_t)i * (size_t)B.L;
                        for (int k = 0; k < B.L; k += 2) {
                            const uint8_t a = NT16[sq[k] > 4 ? 4 : sq[k]], c = k + 1 < B.L ? NT16[sq[k + 1] > 4 ? 4 : sq[k + 1]] : 0;
                            *o++ = (char)((a << 4) | c);
                        }
                        memcpy(o, ql, (size_t)B.L); o += B.L;
                        memcpy(o, "NHi", 3); o += 3; w32(1);
                        memcpy(o, "ASi", 3); o += 3; w32(B.aln_score[i]);
                    }
                }
            });
        for (auto &t : th) t.join();
    }
    return phz_bgzf_write(path, raw.data(), (int64_t)raw.size(), threads, 6);
}

// ---- tabix index (.tbi) of a BGZF-compressed, position-sorted VCF or BED file ----------------------------------------------
// What `tabix -p vcf` / `tabix -p bed` write next to the file (phaser.py:1851, phaser_expr_matrix.py:66): UCSC binning index
// (min_shift 14, depth 5) + 16 kb linear index over BGZF virtual offsets, itself BGZF-compressed.
namespace {
inline int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(4681 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(585 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(73 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(9 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(1 + (beg >> 26));
    return 0;
}
}  // namespace

// The index is gathered over UNCOMPRESSED text offsets (scan) and turned into BGZF virtual offsets when the member table is known
// (serialise): the scan of the phased VCF then runs next to its deflate workers instead of after a re-read of the written file.
namespace {
struct TbxIndex {
    struct Ref {
        std::string name;
        std::vector<std::pair<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>>> bins;   // in order of first use
        std::unordered_map<uint32_t, size_t> bin_idx;
        std::vector<uint64_t> ioff;
        uint64_t off_beg = 0, off_end = 0, n_rec = 0;
        int64_t last_beg = -1, first_beg = -1;
        uint32_t first_bin = 0xffffffffu, last_bin = 0xffffffffu;     // bins of the first / last record (what merge() needs at the seam)
    };
    std::vector<Ref> refs;
    std::unordered_map<std::string, size_t> ref_idx;

    // lines of d[lo, n) (lo at a line start); offsets are positions in d
    int scan(const char *d, size_t lo, size_t n, int preset) {
        size_t p = lo;
        int64_t last_ref = -1; uint32_t last_bin = 0xffffffffu;
        while (p < n) {
            const char *nl = (const char *)memchr(d + p, '\n', n - p);
            const size_t e = nl ? (size_t)(nl - d) : n;
            const size_t next = nl ? e + 1 : n;
            if (e > p && d[p] != '#') {
                // columns
                const char *c0 = d + p; const char *le = d + e;
                const char *t1 = (const char *)memchr(c0, '\t', (size_t)(le - c0));
                if (!t1) return PHZ_E_ARG;
                const char *t2 = (const char *)memchr(t1 + 1, '\t', (size_t)(le - t1 - 1));
                const std::string_view chrom_sv(c0, (size_t)(t1 - c0));
                int64_t beg, end;
                auto to_ll = [](const char *a, const char *z) {           // strtoll(..., 10) of [a, z): blanks, sign, digits
                    while (a < z && (*a == ' ' || *a == '\t')) a++;
                    bool neg = false;
                    if (a < z && (*a == '-' || *a == '+')) { neg = *a == '-'; a++; }
                    long long v = 0;
                    while (a < z && *a >= '0' && *a <= '9') { v = v * 10 + (*a - '0'); a++; }
                    return neg ? -v : v;
                };
                const long long v1 = to_ll(t1 + 1, t2 ? t2 : le);
                if (preset == 0) {          // VCF: POS is 1-based, the record covers len(REF) bases unless INFO carries END=
                    beg = v1 - 1;
                    const char *t3 = t2 ? (const char *)memchr(t2 + 1, '\t', (size_t)(le - t2 - 1)) : nullptr;        // end of ID
                    const char *t4 = t3 ? (const char *)memchr(t3 + 1, '\t', (size_t)(le - t3 - 1)) : nullptr;        // end of REF
                    end = beg + (t3 && t4 ? (int64_t)(t4 - t3 - 1) : 1);
                    const char *t = t4; int col = 4;
                    while (t && col < 7) { t = (const char *)memchr(t + 1, '\t', (size_t)(le - t - 1)); col++; }      // t = tab before INFO
                    if (t) {
                        const char *ie = (const char *)memchr(t + 1, '\t', (size_t)(le - t - 1));
                        std::string_view info(t + 1, (size_t)((ie ? ie : le) - t - 1));
                        size_t q = 0;
                        while (q < info.size()) {
                            size_t r = info.find(';', q); if (r == std::string_view::npos) r = info.size();
                            if (info.substr(q, 4) == "END=") { const long long ev = to_ll(info.data() + q + 4, info.data() + r); if (ev > beg) end = ev; }
                            q = r + 1;
                        }
                    }
                } else {                    // BED: 0-based start, end exclusive
                    if (!t2) return PHZ_E_ARG;
                    const char *t3 = (const char *)memchr(t2 + 1, '\t', (size_t)(le - t2 - 1));
                    beg = v1;
                    end = to_ll(t2 + 1, t3 ? t3 : le);
                }
                if (beg < 0) beg = 0;
                if (end <= beg) end = beg + 1;
                size_t ri;
                if (last_ref >= 0 && refs[(size_t)last_ref].name == chrom_sv) ri = (size_t)last_ref;        // the common case: same contig as the line before
                else {
                    const std::string chrom(chrom_sv);
                    auto it = ref_idx.find(chrom);
                    if (it == ref_idx.end()) { ri = refs.size(); ref_idx.emplace(chrom, ri); refs.emplace_back(); refs.back().name = chrom; }
                    else ri = it->second;
                }
                Ref &R = refs[ri];
                // tabix needs every contig in one run and ascending starts inside it (it refuses such files too)
                if (((int64_t)ri != last_ref && R.n_rec) || ((int64_t)ri == last_ref && beg < R.last_beg)) return PHZ_E_UNSUPPORTED;
                R.last_beg = beg;
                const uint64_t v0 = p, v1e = next;                           // uncompressed offsets here: virtual offsets at serialisation
                if ((int64_t)ri != last_ref) { last_bin = 0xffffffffu; last_ref = (int64_t)ri; if (!R.n_rec) R.off_beg = v0; }
                R.off_end = v1e; R.n_rec++;
                const uint32_t bin = (uint32_t)reg2bin(beg, end);
                if (R.n_rec == 1) { R.first_beg = beg; R.first_bin = bin; }
                R.last_bin = bin;
                if (bin != last_bin || R.bins.empty()) {
                    auto bi = R.bin_idx.find(bin);
                    size_t k;
                    if (bi == R.bin_idx.end()) { k = R.bins.size(); R.bin_idx.emplace(bin, k); R.bins.emplace_back(bin, std::vector<std::pair<uint64_t, uint64_t>>()); }
                    else k = bi->second;
                    R.bins[k].second.emplace_back(v0, v1e);
                    last_bin = bin;
                } else {
                    R.bins[R.bin_idx[bin]].second.back().second = v1e;
                }
                const size_t w0 = (size_t)(beg >> 14), w1 = (size_t)((end - 1) >> 14);
                if (R.ioff.size() <= w1) R.ioff.resize(w1 + 1, 0);
                for (size_t w = w0; w <= w1; w++) if (R.ioff[w] == 0) R.ioff[w] = v0;
            }
            p = next;
        }
        return PHZ_OK;
    }

    // Appends the index of the text that FOLLOWS this one (scanned on its own, as if it were a file) -- the result is what one scan over
    // both texts builds: a contig continues across the seam, its first record extends the open chunk when it falls into the bin of
    // the last record before the seam, bins keep the order of their first use, a linear-index window keeps its first offset.
    int merge(TbxIndex &b) {
        for (size_t k = 0; k < b.refs.size(); k++) {
            Ref &B = b.refs[k];
            if (!B.n_rec) continue;
            if (k == 0 && !refs.empty() && refs.back().name == B.name) {
                Ref &A = refs.back();
                if (B.first_beg < A.last_beg) return PHZ_E_UNSUPPORTED;
                for (auto &bn : B.bins) {
                    auto it = A.bin_idx.find(bn.first);
                    if (it == A.bin_idx.end()) { A.bin_idx.emplace(bn.first, A.bins.size()); A.bins.emplace_back(std::move(bn)); continue; }
                    auto &dst = A.bins[it->second].second;
                    size_t from = 0;
                    if (bn.first == B.first_bin && bn.first == A.last_bin) { dst.back().second = bn.second.front().second; from = 1; }
                    dst.insert(dst.end(), bn.second.begin() + (long)from, bn.second.end());
                }
                if (A.ioff.size() < B.ioff.size()) A.ioff.resize(B.ioff.size(), 0);
                for (size_t w = 0; w < B.ioff.size(); w++) if (A.ioff[w] == 0) A.ioff[w] = B.ioff[w];
                A.off_end = B.off_end; A.n_rec += B.n_rec; A.last_beg = B.last_beg; A.last_bin = B.last_bin;
            } else {
                if (ref_idx.count(B.name)) return PHZ_E_UNSUPPORTED;           // a contig in two runs
                ref_idx.emplace(B.name, refs.size());
                refs.emplace_back(std::move(B));
            }
        }
        return PHZ_OK;
    }

    // scan() over nt line-aligned pieces of d[0, n) at once, merged in order
    int scan_parallel(const char *d, size_t n, int preset, int nt) {
        if (nt > 1 && n / (size_t)nt < (1u << 12)) nt = (int)std::max<size_t>(1, n >> 12);
        if (nt <= 1) return scan(d, 0, n, preset);
        std::vector<size_t> cut((size_t)nt + 1, n);
        cut[0] = 0;
        for (int t = 1; t < nt; t++) {
            const size_t at = std::max(cut[(size_t)t - 1], n / (size_t)nt * (size_t)t);
            const char *nl = at < n ? (const char *)memchr(d + at, '\n', n - at) : nullptr;
            cut[(size_t)t] = nl ? (size_t)(nl - d) + 1 : n;
        }
        std::vector<TbxIndex> part((size_t)nt - 1);
        std::vector<int> status((size_t)nt, PHZ_OK);
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++) th.emplace_back([&, t] { status[(size_t)t] = part[(size_t)t - 1].scan(d, cut[(size_t)t], cut[(size_t)t + 1], preset); });
        status[0] = scan(d, 0, cut[1], preset);
        for (auto &x : th) x.join();
        for (int v : status) if (v != PHZ_OK) return v;
        for (auto &q : part) if (int v = merge(q)) return v;
        return PHZ_OK;
    }

    // coff / ustart: compressed offset and uncompressed start of every member, plus one closing entry
    void serialise(const std::vector<uint64_t> &coff, const std::vector<uint64_t> &ustart, int preset, std::string &o) {
        auto voff = [&](uint64_t upos) -> uint64_t {
            size_t b = (size_t)(std::upper_bound(ustart.begin() + 1, ustart.end(), upos) - (ustart.begin() + 1));
            if (b + 2 > ustart.size()) b = ustart.size() - 2;
            while (b + 2 < ustart.size() && ustart[b + 1] == ustart[b]) b++;           // skip empty members
            return (coff[b] << 16) | (upos - ustart[b]);
        };
        for (auto &R : refs) {
            for (size_t w = 1; w < R.ioff.size(); w++) if (R.ioff[w] == 0) R.ioff[w] = R.ioff[w - 1];
            for (auto &v : R.ioff) v = voff(v);
            for (auto &b : R.bins) for (auto &c : b.second) { c.first = voff(c.first); c.second = voff(c.second); }
            R.off_beg = voff(R.off_beg); R.off_end = voff(R.off_end);
        }
        o.assign("TBI\1", 4);
        auto p32 = [&](int32_t v) { o.append((const char *)&v, 4); };
        auto p64 = [&](uint64_t v) { o.append((const char *)&v, 8); };
        p32((int32_t)refs.size());
        if (preset == 0) { p32(2); p32(1); p32(2); p32(0); } else { p32(0x10000); p32(1); p32(2); p32(3); }
        p32('#'); p32(0);
        size_t l_nm = 0;
        for (auto &R : refs) l_nm += R.name.size() + 1;
        p32((int32_t)l_nm);
        for (auto &R : refs) { o += R.name; o.push_back('\0'); }
        for (auto &R : refs) {
            p32((int32_t)R.bins.size() + 1);
            for (auto &b : R.bins) {
                o.append((const char *)&b.first, 4); p32((int32_t)b.second.size());
                for (auto &c : b.second) { p64(c.first); p64(c.second); }
            }
            const uint32_t meta_bin = 37450; o.append((const char *)&meta_bin, 4); p32(2);       // htslib's pseudo-bin: file span + record counts
            p64(R.off_beg); p64(R.off_end); p64(R.n_rec); p64(0);
            p32((int32_t)R.ioff.size());
            for (uint64_t v : R.ioff) p64(v);
        }
        p64(0);                                                                                  // n_no_coor
    }
};
}  // namespace

int phz_tabix_build(const char *bgzf_path, int preset, int threads) {
    if (!bgzf_path || (preset != 0 && preset != 1)) return PHZ_E_ARG;
    // member table (compressed offset, uncompressed start) + inflated text
    int fd = open(bgzf_path, O_RDONLY);
    if (fd < 0) return PHZ_E_ARG;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return PHZ_E_ARG; }
    const size_t fsz = (size_t)st.st_size;
    const uint8_t *f = (const uint8_t *)mmap(nullptr, fsz, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (f == MAP_FAILED) return PHZ_E_NOMEM;
    std::vector<uint64_t> coff, ustart;
    {
        size_t off = 0; uint64_t u = 0;
        while (off + 18 <= fsz) {
            if (f[off] != 0x1f || f[off + 1] != 0x8b || !(f[off + 3] & 4)) { munmap((void *)f, fsz); return PHZ_E_ARG; }
            const uint16_t xlen = rd16(f + off + 10);
            size_t x = off + 12, xe = x + xlen; uint32_t bsize = 0;
            while (x + 4 <= xe) { const uint16_t slen = rd16(f + x + 2); if (f[x] == 'B' && f[x + 1] == 'C' && slen == 2) bsize = (uint32_t)rd16(f + x + 4) + 1; x += 4 + slen; }
            if (!bsize || off + bsize > fsz) { munmap((void *)f, fsz); return PHZ_E_ARG; }
            coff.push_back(off); ustart.push_back(u);
            u += rd32(f + off + bsize - 4);
            off += bsize;
        }
        coff.push_back(off); ustart.push_back(u);
    }
    munmap((void *)f, fsz);
    if (coff.size() < 2) return PHZ_E_ARG;
    RawBuf text;
    if (int s2 = inflate_bgzf_file(bgzf_path, threads, text)) return s2;
    TbxIndex ix;
    if (int s3 = ix.scan_parallel((const char *)text.data(), text.size(), preset, std::min(16, n_threads(threads)))) return s3;
    std::string o;
    ix.serialise(coff, ustart, preset, o);
    const std::string out_path = std::string(bgzf_path) + ".tbi";
    return phz_bgzf_write(out_path.c_str(), o.data(), (int64_t)o.size(), std::min(16, n_threads(threads)), 6);
}

// phz_bgzf_write + phz_tabix_build of the same text in one call: the index scan runs beside the deflate workers and nothing is read
// back.  The .gz is written in any case; PHZ_E_UNSUPPORTED = the text is not position-sorted, no .tbi written (as tabix refuses).
int phz_bgzf_write_indexed(const char *path, const char *data, int64_t len, int threads, int level, int preset) {
    if (preset != 0 && preset != 1) return PHZ_E_ARG;
    TbxIndex ix;
    int scan_status = PHZ_OK;
    std::vector<uint64_t> coff, ustart;
    const int scan_threads = std::min(16, n_threads(threads));
    const std::string out_path = std::string(path) + ".tbi";
    // An index left by an earlier run under the same name describes ANOTHER file from the moment the .gz is rewritten: it goes first,
    // so that every way out of this function leaves either the new index or none (tabix on a stale one returns wrong records silently).
    (void)unlink(out_path.c_str());
    const int st = bgzf_write_impl(path, data, len, threads, level, [&] { scan_status = ix.scan_parallel(data, (size_t)len, preset, scan_threads); }, &coff, &ustart);
    if (st != PHZ_OK) return st;
    if (scan_status != PHZ_OK) return scan_status;
    std::string o;
    ix.serialise(coff, ustart, preset, o);
    const int wst = phz_bgzf_write(out_path.c_str(), o.data(), (int64_t)o.size(), scan_threads, 6);
    if (wst != PHZ_OK) (void)unlink(out_path.c_str());
    return wst;
}

}  // extern "C"
