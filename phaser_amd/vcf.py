"""Het-variant loading for the hot path: VCF text -> per-chromosome variant tables.

The work is done by the native loader in libphz.so (phz_vcf_parse, phaser_amd/csrc/phz_vcf.cpp, host threads): the filter
of phaser/phaser.py:396-433 (GT present, no '.', more than one distinct allele, FILTER contains PASS unless
--pass_only 0), the table of generate_mapping_table :1355-1413 (unique id, SNP-only unless --include_indels) and the
per-variant fields the phasing core derives from a table row in generate_variant_dict :1418-1462.  This module wraps the
result: numeric columns are numpy arrays, string columns stay in the library's separator-joined pools (which is what the
native row writer reads) and turn into Python lists only when somebody asks for them.
"""
from __future__ import annotations

import ctypes as C
import gzip
from typing import Dict, List

import numpy as np

from . import _lib

_POOLS = ("uid", "rsid_field", "rsid", "ref", "all_alleles", "alleles", "phase", "gt", "maf_text", "maf_str", "allele2")
_SPLIT = ("all_alleles", "alleles", "phase")              # pools whose items are comma-joined lists


def sep_pool(strs):
    """n strings -> (uint32 offsets [n+1], bytes): string i = bytes[off[i] : off[i+1]-1] (one separator byte after each)."""
    if not strs:
        return np.zeros(1, dtype=np.uint32), b"\n"
    b = ("\n".join(strs) + "\n").encode()
    return pool_offsets(b, len(strs)), b


def pool_offsets(b: bytes, n: int) -> np.ndarray:
    ends = np.flatnonzero(np.frombuffer(b, dtype=np.uint8) == 10)
    if len(ends) != n or len(b) >= 2 ** 32:
        raise ValueError("string table holds a newline or exceeds 4 GiB")
    off = np.empty(n + 1, dtype=np.uint32)
    off[0] = 0; off[1:] = ends + 1
    return off


class ChromVariants:
    """One chromosome's het-variant table.  Arrays: pos int32 (VCF order, sorted), ref_len uint8, a0 / a1 uint8 base codes of
    the individual's two alleles (255 = not a single ACGT base), is_ref uint8 [2n], phase_idx int8 [2n], maf_val float64.
    String columns (lists, built on first use): uid, rsid_field (ID as written), rsid ('.' replaced by uid, :1451-1455), ref,
    all_alleles, alleles (individual's, allele-index order, :1430-1435), phase (GT order or ['-','-'], :1437-1443), gt,
    maf_text (str(maf) as written to the table), maf (float, or int 0 when not parseable, :1445-1449)."""

    def __init__(self, chrom: str, arrays: Dict[str, np.ndarray], raw: Dict[str, bytes]):
        self.chrom = chrom
        self.pos = arrays["pos"]; self.ref_len = arrays["ref_len"]; self.a0 = arrays["a0"]; self.a1 = arrays["a1"]
        self.is_ref = arrays["is_ref"]; self.phase_idx = arrays["phase_idx"]; self.maf_val = arrays["maf"]
        self.blacklisted = arrays.get("blacklisted")        # uint8 [n]: overlaps a --haplo_count_blacklist interval
        self._raw = raw
        self._pools = None

    def __len__(self):
        return len(self.pos)

    def __getattr__(self, name):
        if name in _POOLS and name != "allele2":
            text = self._raw[name].decode()
            items = text.split("\n")[:-1] if text else []
            if name in _SPLIT:
                items = [x.split(",") if x else [] for x in items]
            setattr(self, name, items)
            return items
        if name == "maf":
            vals = [0 if t == "0" else float(t) for t in self.__getattr__("maf_str")] if len(self) else []
            setattr(self, "maf", vals)
            return vals
        raise AttributeError(name)

    @property
    def is_general(self) -> bool:
        """True when the set is not pure SNPs (some REF longer than one base or some allele not a single ACGT base):
        such sets go through the general mapper (phz_map_reads_general)."""
        return bool((self.ref_len != 1).any() or (self.a0 == 255).any() or (self.a1 == 255).any())

    def pools(self):
        """Separator-joined string pools of the per-variant texts the native row writer prints (phz_rows_format)."""
        if self._pools is None:
            n = len(self)
            self._pools = {"uid": (pool_offsets(self._raw["uid"], n), self._raw["uid"]),
                           "rsid": (pool_offsets(self._raw["rsid"], n), self._raw["rsid"]),
                           "allele": (pool_offsets(self._raw["allele2"], 2 * n), self._raw["allele2"]),
                           "maf": (pool_offsets(self._raw["maf_str"], n), self._raw["maf_str"]),
                           "maf_val": self.maf_val}
        return self._pools

    def allele_pool(self):
        """-> (allele_off uint32 [2n+1], allele_bytes uint8) of the individual's two alleles per variant, no separators
        (the layout phz_variants_general wants)."""
        off, b = self.pools()["allele"]
        a = np.frombuffer(b, dtype=np.uint8)
        keep = a != 10
        out_off = (off.astype(np.int64) - np.arange(len(off), dtype=np.int64)).astype(np.uint32)      # one separator dropped per item
        return out_off, np.concatenate([a[keep], np.zeros(1, dtype=np.uint8)])

    def table_rows(self):
        return [[self.chrom, str(int(self.pos[i])), self.uid[i], self.rsid_field[i], ",".join(self.all_alleles[i]),
                 str(int(self.ref_len[i])), self.gt[i], self.maf_text[i]] for i in range(len(self))]


class VariantSet:
    def __init__(self, chroms, het_count, filter_count, indels_excluded, unphased_count):
        self.chroms: Dict[str, ChromVariants] = chroms      # insertion order = VCF order
        self.het_count = het_count; self.filter_count = filter_count
        self.indels_excluded = indels_excluded; self.unphased_count = unphased_count


class _NativeTable:
    """Owner of a phz_vcf handle: the string pools of the chromosomes stay in native memory and are copied out on first use only (the row stage
    reads four of the eleven; copying all of them for 1.5 M variants was 150 MB of memcpy per run)."""

    def __init__(self, lib, h):
        self.lib = lib; self.h = h

    def __del__(self):
        try:
            if self.h is not None:
                self.lib.phz_vcf_free(self.h); self.h = None
        except Exception:
            pass


class _LazyPools(dict):
    """raw[name] -> bytes of the pool, fetched from the native table when first asked for."""

    def __init__(self, owner, ptrs):
        super().__init__()
        self._owner = owner; self._ptrs = ptrs

    def __missing__(self, name):
        ptr, n = self._ptrs[name]
        b = C.string_at(ptr, n) if n else b""
        self[name] = b
        return b


def read_text(path: str) -> str:
    return read_bytes(path).decode()


class _NativeBuf:
    def __init__(self, lib, ptr):
        self.lib = lib; self.ptr = ptr

    def __del__(self):
        try:
            self.lib.phz_buf_free(self.ptr)
        except Exception:
            pass


def text_ptr(text):
    """(pointer, length, keep-alive) of a VCF text given as str / bytes / uint8 array (read_bytes(as_array=True): the inflated file where the native
    reader left it -- the 75 MB of a whole-genome VCF are never copied into a Python bytes object, 0.02-0.04 s of page faults in a fresh process)"""
    if isinstance(text, np.ndarray):
        a = text if (text.dtype == np.uint8 and text.flags.c_contiguous) else np.ascontiguousarray(text, dtype=np.uint8)
        return C.c_void_p(a.ctypes.data), int(a.size), a
    data = text.encode() if isinstance(text, str) else bytes(text)
    return C.cast(C.c_char_p(data), C.c_void_p), len(data), data


def read_bytes(path: str, threads: int = 0, as_array: bool = False):
    """The (decompressed) text of a VCF file: bytes, or with as_array a uint8 array over the native reader's buffer when the file is bgzipped."""
    if path.endswith(".gz") or path.endswith(".bgz"):
        # bgzip output is a chain of independent members: inflate them in parallel; plain gzip falls to the gzip module
        lib = _lib.load()
        p = C.c_void_p(); n = C.c_int64(0)
        st = lib.phz_bgzf_read(path.encode(), int(threads), C.byref(p), C.byref(n))
        if st == _lib.PHZ_OK:
            if as_array:
                return _lib.native_view(p.value, n.value, C.c_uint8, _NativeBuf(lib, p))
            try:
                return C.string_at(p, n.value)
            finally:
                lib.phz_buf_free(p)
        with gzip.open(path, "rb") as f:
            return f.read()
    with open(path, "rb") as f:
        return f.read()


def contig_names_from_tbi(path: str):
    """Sequence names of a tabix index (.tbi: BGZF-compressed; magic TBI\\1, n_ref, five format words, l_nm, the names NUL-terminated),
    or None when there is no such file / it is not a tabix index.  The names of the contigs that HAVE lines in the indexed file."""
    import os, struct, zlib
    try:
        if not os.path.isfile(path) or os.path.getsize(path) > (256 << 20):
            return None
        with gzip.open(path, "rb") as f:
            head = f.read(36)
            if len(head) < 36 or head[:4] != b"TBI\x01":
                return None
            n_ref, l_nm = struct.unpack_from("<i", head, 4)[0], struct.unpack_from("<i", head, 32)[0]
            if n_ref <= 0 or l_nm <= 0 or l_nm > (64 << 20):
                return None
            blob = f.read(l_nm)
        if len(blob) != l_nm:
            return None
        names = [x.decode() for x in blob.split(b"\x00") if x]
        return names if len(names) == n_ref else None
    except (OSError, EOFError, ValueError, struct.error, UnicodeDecodeError, zlib.error):
        return None


def contig_names_guess(data: bytes, max_probes: int = 50000):
    """Distinct CHROM values of a VCF text whose contigs come in runs (every tabix-able file), found by bisection between line probes:
    O(contigs x log lines) line lookups instead of a pass over the text.  A GUESS: a contig scattered inside another contig's run can
    be missed -- callers check the result against the parsed table (phaser.main does, before it trusts the BAM prefetch)."""
    if isinstance(data, np.ndarray):
        data = data.tobytes()               # (the rare path: a VCF without a tabix index next to it)
    n = len(data)
    p = 0
    while p < n and data[p:p + 1] == b"#":
        j = data.find(b"\n", p)
        if j < 0:
            return []
        p = j + 1
    if p >= n:
        return []

    def line_at(off):                    # start of the first line that begins at or after off
        if off <= p:
            return p
        j = data.find(b"\n", off - 1)
        return n if j < 0 else j + 1

    def chrom(s):
        e = data.find(b"\n", s)
        t = data.find(b"\t", s, n if e < 0 else e)
        return data[s:t] if t > 0 else None

    last = data.rfind(b"\n", 0, n - 1) + 1 if n > 1 else 0
    if last < p:
        last = p
    names = {}
    stack = [(p, last)]
    probes = 0
    while stack:
        a, b = stack.pop()
        ca, cb = chrom(a), chrom(b)
        for c in (ca, cb):
            if c:
                names[c] = 1
        if ca == cb or a >= b:
            continue
        m = line_at((a + b) // 2)
        if m >= b:
            m = line_at(a + 1)
            if m >= b:
                continue                 # neighbours
        probes += 1
        if probes > max_probes:
            return []
        stack.append((a, m)); stack.append((m, b))
    return [c.decode("latin1") for c in names]


def load_variants(vcf_text, sample_column: int = 9, chrom_of_interest: str = "", pass_only: int = 1,
                  include_indels: int = 0, chr_prefix: str = "", id_separator: str = "_", gw_phase_method: int = 0,
                  gw_af_field: str = "AF", contig_ban=("_", ":"), threads: int = 8, grep_hom: bool = False,
                  drop_bed=None, mark_bed=None) -> VariantSet:
    """vcf_text: str or bytes of the (decompressed) VCF.  Raises SystemExit with the reference's message on a banned
    contig character (phaser.py:386-392) and on unsorted records.  grep_hom=True applies the reference's
    `cut -f 1-9,S | grep -v '0|0\\|1|1'` pre-filter (phaser.py:220-225) inside the loader.  drop_bed / mark_bed: BED intervals as
    [(chrom, start, end)]: records overlapping drop_bed vanish (--blacklist, `bedtools intersect -v`), variants overlapping
    mark_bed get ChromVariants.blacklisted = 1 (--haplo_count_blacklist)."""
    lib = _lib.load()
    data_p, data_n, data = text_ptr(vcf_text)
    ban = [str(x).encode() for x in contig_ban]
    ban_arr = (C.c_char_p * max(1, len(ban)))(*ban) if ban else (C.c_char_p * 1)()
    keep = []

    def bed(iv):
        iv = list(iv or [])
        n = len(iv)
        names = (C.c_char_p * max(1, n))(*[str(x[0]).encode() for x in iv]) if n else (C.c_char_p * 1)()
        st_ = np.ascontiguousarray([int(x[1]) for x in iv], dtype=np.int64); en_ = np.ascontiguousarray([int(x[2]) for x in iv], dtype=np.int64)
        keep.extend([names, st_, en_])
        return n, names, C.c_void_p(st_.ctypes.data) if n else None, C.c_void_p(en_.ctypes.data) if n else None
    nd, dn, ds, de = bed(drop_bed); nm, mn, ms, me = bed(mark_bed)
    o = _lib.phz_vcf_opts(int(sample_column), chrom_of_interest.encode(), int(pass_only), int(include_indels), chr_prefix.encode(),
                          id_separator.encode(), int(gw_phase_method), gw_af_field.encode(), len(ban), ban_arr, max(1, int(threads)), 1 if grep_hom else 0,
                          nd, dn, ds, de, nm, mn, ms, me)
    h = C.c_void_p()
    st = lib.phz_vcf_parse(data_p, data_n, C.byref(o), C.byref(h))
    owner = _NativeTable(lib, h)          # frees the table when the last chromosome's lazy pools are gone
    if st != _lib.PHZ_OK:
        msg = (lib.phz_vcf_error(h) or b"").decode()
        if "FATAL ERROR" in msg:
            raise SystemExit(msg)
        raise _lib.PhzError(st, msg or "phz_vcf_parse failed")
    nch = C.c_int32(0); het = C.c_int64(0); fc = C.c_int64(0); ex = C.c_int64(0); un = C.c_int64(0)
    lib.phz_vcf_summary(h, C.byref(nch), C.byref(het), C.byref(fc), C.byref(ex), C.byref(un))
    chroms: Dict[str, ChromVariants] = {}
    for i in range(nch.value):
        t = _lib.phz_vcf_table()
        lib.phz_vcf_chrom(h, i, C.byref(t))
        n = int(t.n)

        def arr(ptr, count, dt):
            if count == 0:
                return np.zeros(0, dtype=dt)
            nb = count * np.dtype(dt).itemsize
            return np.frombuffer((C.c_char * nb).from_address(ptr), dtype=dt).copy()          # one copy, owned by numpy
        arrays = {"pos": arr(t.pos, n, np.int32), "ref_len": arr(t.ref_len, n, np.uint8), "a0": arr(t.a0, n, np.uint8),
                  "a1": arr(t.a1, n, np.uint8), "is_ref": arr(t.is_ref, 2 * n, np.uint8), "phase_idx": arr(t.phase_idx, 2 * n, np.int8),
                  "maf": arr(t.maf, n, np.float64), "blacklisted": arr(t.blacklisted, n, np.uint8)}
        raw = _LazyPools(owner, {name: (t.pool[k], int(t.pool_len[k])) for k, name in enumerate(_POOLS)})
        cv = ChromVariants(t.name.decode(), arrays, raw)
        chroms[cv.chrom] = cv
    return VariantSet(chroms, int(het.value), int(fc.value), int(ex.value), int(un.value))
