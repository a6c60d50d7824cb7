/*
 * phz.h -- C ABI of libphz.so: the MI355X (gfx950) implementation of phASER's read-backed
 * phasing hot path.  Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * Which reference interface each entry point replaces (paths relative to the phASER repo):
 *
 *   phz_map_reads        phaser/read_variant_map.py:3    do_read_variant_map(variant_table, baseq, o,
 *                        splice, isize_cutoff) -- reached today through the process boundary
 *                        phaser/phaser.py:1330-1353 call_mapping_script (bash | samtools | python).
 *                        One call == one (chromosome, BAM) shard, exactly the unit parallelize()
 *                        hands to a pool worker at phaser/phaser.py:533.
 *   phz_as_histogram     phaser/phaser.py:545-553        `cut -f 5` over all call files + numpy.percentile
 *   phz_tally            phaser/phaser.py:1287-1328      process_mapping_result (per-variant read lists),
 *                        :610-632 noise counters, :1265-1285 generate_connectivity_map,
 *                        :1594-1635 test_variant_connection's nine set intersections, :917-931 / :1086-1115 the
 *                        per-haplotype read lists of the block output loop (what SURVEY.md 8(b) calls phz_hap_counts)
 *   phz_components       phaser/phaser.py:1861-1882      build_haplotypes / :1985 build_haplotype_v3
 *
 * Deliberately NOT exported (SURVEY.md 8(b) suggested them):
 *   phz_load_variants    variant arrays travel with every call instead (6 B per SNP; nothing to keep resident between shards)
 *   phz_hap_counts       folded into phz_tally: the per-(variant, allele, BAM) read lists it leaves in HBM are exactly what the
 *                        haplotype-count loops of phaser/phaser.py:917-931 / :1086-1115 consume (phz_rowsdev_run, phz_rows_format)
 *   *_cpu twins          the CPU restatement of this path is test infrastructure (oracle/), never part of the product library:
 *                        every entry point here needs the GPU and fails loudly without one
 *
 * Conventions
 *   - every function returns 0 (PHZ_OK) or a negative phz_status; nothing throws across the ABI.
 *   - a phz_ctx owns one HIP stream, scratch buffers, the resident tally and an error string.  Calls on ONE ctx are serialised by a
 *     lock inside the ctx (thread-safe per phz_ctx): two threads that share a ctx take turns, whole call by whole
 *     call, and phz_last_error() then reports the LAST failing call of either.  For concurrency use one ctx per thread -- several
 *     may sit on one GPU, and device objects created through one (phz_rowsdev_create's tables) may be used through another ctx of
 *     the same device once the creating call has returned.
 *   - `space` says where the caller's pointers live: PHZ_HOST (the library stages through HBM itself)
 *     or PHZ_DEVICE (pointers are device pointers on the ctx's GPU; zero copies, used when the shard is
 *     already resident in HBM).
 *   - inputs of one call describe ONE chromosome: reads coordinate-sorted, variants position-sorted.
 *
 * Read shard layout (structure of arrays, see DESIGN.md "Data layout in HBM"):
 *   pos[n]            1-based leftmost aligned position (SAM POS)
 *   cigar_off[n+1]    op index range of read r is [cigar_off[r], cigar_off[r+1])
 *   cigar[n_ops]      BAM encoding len<<4|op, op in MIDNSHP=X (0..8); op 9 ('G', produced only by the
 *                     host packer for malformed records) advances the genome cursor without bases
 *   seq_off[n+1]      start of read r in units of 4 bases: byte offset into seq2, x4 = byte offset into qual
 *   seq2[...]         2 bits per base (A,C,G,T = 0..3), base j of a read in bits (2*(j&3)) of byte j>>2
 *   qual[...]         one byte per base: low 7 bits phred; bit 7 set = base is not ACGT and the 2-bit
 *                     code is a subtype (0: behaves like 'N', 1: other IUPAC symbol)
 *
 * Call list layout (mapper order: record, then segment, then variant position):
 *   read_idx, var_idx   indices into the shard / variant arrays
 *   code                0..3 = the allele is the single base A/C/G/T; 4 = any other string
 *                       (phz_map_reads_general additionally: 5 / 6 = equals the individual's allele 0 / 1 string)
 *   aux0                read offset of the base under the variant (0xFFFFFFFF: deletion placeholder)
 *   aux1                (read offset of spliced-in inserted bases << 12) | min(length,4095); 0 = none
 *                       aux0/aux1 let the host print the exact allele text; they never change a decision.
 */
#ifndef PHZ_H
#define PHZ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct phz_ctx phz_ctx;

typedef enum {
    PHZ_OK = 0,
    PHZ_E_ARG = -1,          /* bad argument / malformed shard */
    PHZ_E_HIP = -2,          /* HIP runtime error, see phz_last_error */
    PHZ_E_CAPACITY = -3,     /* output buffers too small; the needed count is returned */
    PHZ_E_UNSUPPORTED = -4,  /* e.g. variants with ref_len != 1 (indel mode) */
    PHZ_E_NOMEM = -5
} phz_status;

enum { PHZ_HOST = 0, PHZ_DEVICE = 1 };

typedef struct {
    int64_t n_reads, n_ops, n_seq_bytes;
    const int32_t *pos;
    const uint32_t *cigar_off;
    const uint32_t *cigar;
    const uint32_t *seq_off;
    const uint8_t *seq2;
    const uint8_t *qual;
    const uint8_t *bq;       /* optional (NULL = absent), device shards only: ONE byte per base holding both things a call needs -- bits 7:6 the 2-bit base, bits 5:0
                              * min(phred, 62), 63 = escape (a non-ACGT base or a phred above 62: seq2 / qual decide) -- at the index of the base's qual byte.
                              * When EVERY shard of a submission carries it, K_map's ONE instantiation reads it (one memory line per call instead of two; identical
                              * calls).  Measured on the whole-genome sample: no faster than the two planes (1.0928 against 1.0940 ms), so no producer writes it by default */
} phz_reads;

typedef struct {
    int64_t n;
    const int32_t *pos;      /* sorted ascending */
    const uint8_t *ref_len;  /* len(REF); only 1 (SNP) is accelerated in this version */
} phz_variants;

typedef struct {
    int64_t cap;             /* capacity of each array, in calls */
    int32_t *read_idx;
    int32_t *var_idx;
    uint8_t *code;
    uint32_t *aux0;          /* aux0 and aux1 may both be NULL: a caller that needs (record, variant, code) only -- the phasing stage -- */
    uint32_t *aux1;          /* saves their 8 of 17 output bytes per call (phz_map_reads, phz_map_reads_batch) */
} phz_calls;

/* One (chromosome, BAM) shard's call lines = K_map output + the per-record fields the phasing core reads
 * from the mapper's TSV (qname -> read_qid, AS column -> read_as). */
typedef struct {
    int64_t n_calls;
    const int32_t *read_idx;
    const int32_t *var_idx;      /* index into the shard's chromosome's variant table */
    const uint8_t *code;
    int64_t n_reads;
    const int32_t *read_qid;     /* template (QNAME) id per record; ids are per chromosome, shared by all BAMs */
    const int32_t *read_as;      /* AS:i per record, must fit int16 */
    const uint8_t *read_has_as;  /* NULL = every record carries AS */
    double as_cutoff;            /* numpy.percentile value (phaser.py:551); used when use_cutoff != 0 */
    int32_t use_cutoff;
    int32_t bam_index;
    int64_t var_base;            /* phz_tally over several chromosomes: index of the chromosome's first variant / first QNAME id */
    int64_t qid_base;            /* in the call's joint index spaces (0 for a single chromosome) */
    const int16_t *read_as16;    /* optional (NULL = absent): AS per record as ONE 2-byte plane with the has-AS flag folded in (SURVEY.md 8(a) M1 `as:int16`):
                                  * PHZ_AS16_NONE = the record carries no AS tag, +-PHZ_AS16_RANGE = an AS value outside [-32766, 32766] (refused like any value
                                  * outside int16).  When present the kernels gather this plane instead of read_as + read_has_as (2 instead of 5 bytes per record
                                  * touched, one memory line instead of two) */
    const double *as_cutoff_dev; /* optional (NULL = absent), DEVICE pointer to the four doubles phz_as_cutoff_enqueue writes: when present and use_cutoff != 0 the
                                  * kernels take the cutoff from [0] (and keep every line when [1] == 0: no record of the BAM carries an AS tag) instead of as_cutoff:
                                  * the percentile never visits the host between the histogram and the tally */
} phz_lines;
#define PHZ_AS16_NONE (-32768)
#define PHZ_AS16_RANGE 32767

#define PHZ_AS_BINS 65536        /* histogram bin = AS + 32768 */

/* What one phz_tally produced (sizes of the arrays phz_tally_fetch hands out) */
typedef struct {
    int64_t n_lines;         /* call lines of all shards */
    int64_t n_kept;          /* lines that passed the AS cutoff */
    int64_t n_edges;         /* distinct variant pairs */
    int64_t n_read_list;     /* kept ref/alt lines = entries of rl_qid */
    int64_t n_items;         /* distinct (QNAME, variant, class) */
    int64_t pair_events;     /* sum over QNAMEs of item pairs on different variants */
    int64_t noise_match;     /* sequencing-noise counters of these variants (phaser.py:610-632): ref+alt lines ... */
    int64_t noise_mismatch;  /* ... and other-allele lines, over variants whose other share is below 5 % */
} phz_tally_sizes;

/* Destination arrays of phz_tally_fetch; a NULL member is skipped.  Variant indices are positions in the call's joint variant
 * space (var_base + index), line numbers positions in the concatenation of the shards' lines in the order they were passed. */
typedef struct {
    int32_t *var_count;      /* [nv*3] kept call lines per (variant, class ref/alt/other), duplicates kept */
    int64_t *var_first;      /* [nv]   first kept line, -1 if none */
    int32_t *var_distinct;   /* [nv*3] distinct QNAMEs per (variant, class) */
    uint64_t *var_rank;      /* [nv] order in which variants enter the connectivity map (phaser.py:1271-1283): smallest
                              * (first ref/alt line of the QNAME << 32 | line) over surviving read_vars entries of QNAMEs
                              * with >= 2 distinct variants; UINT64_MAX when the variant never gets a key */
    uint8_t *line_cls;       /* [n_lines] 0 ref / 1 alt / 2 other / 255 dropped by the AS cutoff */
    int32_t *edge_a;         /* [n_edges] variant pair a < b, sorted by (a, b) */
    int32_t *edge_b;
    int32_t *edge_cells;     /* [n_edges*9] |S_a[i] & S_b[j]| at i*3+j, classes ref/alt/other */
    uint8_t *edge_linked;    /* [n_edges] 1 when some QNAME's surviving read_vars list holds both variants */
    int32_t *edge_cto;       /* [n_edges*3] the three sums of test_variant_connection (:1634-1636): same configuration (rr+aa),
                              * opposite (ar+ra), other (the five cells with an "other" allele) */
    uint32_t *rl_start;      /* [nv*2*n_bams + 1] read lists (phaser.py:1318-1322): the kept lines of (variant v, allele k, BAM b) */
    int32_t *rl_qid;         /* [n_read_list]     are rl_qid[rl_start[(2v+k)*n_bams+b] : rl_start[... + 1]] = their QNAME ids
                              *                   (chromosome-local, as passed in read_qid) in line order */
    int32_t *edge_stats;     /* [5*n_edges] five planes of n_edges: same-configuration count, opposite count, supporting = max of the
                              * two, total = all nine cells, chosen configuration 0 same / 1 opposite / -1 tie (:1637-1649) */
} phz_tally_out;

/* Variant table for the general (indel) mapper: per variant REF length and the individual's two allele strings. */
typedef struct {
    int64_t n;
    const int32_t *pos;            /* sorted ascending */
    const uint8_t *ref_len;        /* len(REF), 1..255 */
    const uint32_t *allele_off;    /* [2n+1]: allele 0 of variant v = bytes [off[2v], off[2v+1]), allele 1 = [off[2v+1], off[2v+2]) */
    const char *allele_bytes;
    int64_t n_allele_bytes;
} phz_variants_general;

/* timing slots for phz_get_timing */
enum { PHZ_T_MAP = 0, PHZ_T_ASHIST = 1, PHZ_T_TALLY = 2, PHZ_T_COMPONENTS = 3, PHZ_T_GENES = 4, PHZ_T_INFLATE = 5, PHZ_T_BAMPACK = 6, PHZ_T_ROWS = 7, PHZ_T_COUNT = 8 };

/* work counters accumulated over phz_tally calls since the last phz_reset_timing (the units of K_tally's byte model):
 * call lines seen, distinct (QNAME, variant, class) items, pair events = sum over QNAMEs of C(k, 2) item pairs on different
 * variants, distinct variant pairs (edges) */
enum { PHZ_C_LINES = 0, PHZ_C_ITEMS = 1, PHZ_C_PAIR_EVENTS = 2, PHZ_C_EDGES = 3, PHZ_C_FAR_LINES = 4 /* call lines outside their tile's variant window */,
       PHZ_C_DIRTY_LISTS = 5 /* read lists filled through a cursor and sorted (they hold a far line) */, PHZ_C_COUNT = 8 };

/* ---- Raw-byte tier, native: the rows of variant_connections / haplotypes / haplotypic_counts re-ordered and re-labelled the way
 * CPython 3.10 with PYTHONHASHSEED=0 orders the reference's sets of strings (phaser/phaser.py:660-678, :930, :1059, :1086, :1106-1115, :1181-1239).  The str
 * hash (SipHash-2-4, zero key) and the set (probe sequence, growth, difference) are restated in phz_pyorder.cpp; nothing of the interpreter is linked.
 * String pools: item i of a pool = blob[off[i], off[i + 1]) minus one trailing separator byte for the variant pools (uid, allele2 -- two per variant --,
 * rsid), no separator for the QNAME pools (id order, per chromosome).  Per-(chromosome, BAM) arrays are indexed [c * n_bams + b]; NULL = no call file. */
typedef struct {
    int32_t n_chroms, n_bams;
    const char *const *chrom_names; const char *const *bam_names;
    const int64_t *nv; const int32_t *const *pos;
    const char *const *uid; const uint32_t *const *uid_off;
    const char *const *allele2; const uint32_t *const *allele2_off;
    const char *const *rsid; const uint32_t *const *rsid_off;
    const int64_t *nq; const char *const *qname; const uint32_t *const *qname_off;
    const int32_t *const *line_qid; const int32_t *const *line_var; const uint8_t *const *line_cls; const int64_t *n_lines;      /* KEPT call lines, line order */
    const uint8_t *bam_excluded;               /* [n_bams] or NULL (--haplo_count_bam_exclude) */
    const uint8_t *const *blacklisted;         /* per chromosome [nv] or NULL: the variant is on the haplotype-count blacklist */
    int64_t n_blocks; const int32_t *blk_chrom; const int64_t *blk_off /* [n_blocks + 1] */; const int32_t *blk_var;              /* blocks in block order, chromosome-local variants */
    int32_t output_read_ids, unphased_vars, unique_ids;
} phz_pyorder_in;
typedef struct phz_pyorder phz_pyorder;
/* conn / hap / ase: the three files in the product's canonical order (what the fast path writes).  -> handle holding the three texts in the reference's raw order;
 * PHZ_E_ARG with a message (phz_pyorder_error) when the texts and the call lines do not belong together. */
int phz_pyorder_replay(const phz_pyorder_in *in, const char *conn, int64_t conn_len, const char *hap, int64_t hap_len, const char *ase, int64_t ase_len, phz_pyorder **out);
int phz_pyorder_text(const phz_pyorder *h, int which /* 0 variant_connections, 1 haplotypes, 2 haplotypic_counts */, const char **text, int64_t *len);
const char *phz_pyorder_error(const phz_pyorder *h);
void phz_pyorder_free(phz_pyorder *h);
/* the two restated pieces of the interpreter, exported for the tests: hash(str) under PYTHONHASHSEED=0, and the iteration order of set(items) (mode 0) or
 * of set(items[0, n_a)) - set(items[n_a, n)) (mode 1) as indices into the pool */
int64_t phz_py_str_hash(const char *s, int64_t len);
int64_t phz_py_set_order(const char *blob, const uint32_t *off, int64_t n, int64_t n_a, int32_t mode, int32_t *order_out);

int phz_version(void);
const char *phz_strerror(int status);
const char *phz_last_error(const phz_ctx *ctx);

int phz_device_count(int *n);
int phz_ctx_create(int device, phz_ctx **out);
int phz_ctx_destroy(phz_ctx *ctx);
int phz_ctx_sync(phz_ctx *ctx);
/* raw hipStream_t of the context (so a caller can order its own work against it) */
void *phz_ctx_stream(phz_ctx *ctx);

/* One chromosome's het-variant table made RESIDENT in the ctx (SURVEY.md 8(b) `phz_load_variants`; the POS / len(REF) columns of generate_mapping_table,
 * phaser/phaser.py:1355-1413, as the mapper reads them, read_variant_map.py:25-44).  slot in [0, 65536): one per chromosome; pos must be sorted (checked for host
 * arrays).  *resident receives DEVICE pointers owned by the ctx, valid until the slot is loaded again or the ctx is destroyed: pass it to
 * phz_map_reads(..., PHZ_DEVICE) / phz_map_reads_batch for every BAM's shard of that chromosome (the table is uploaded once, not per call). */
int phz_load_variants(phz_ctx *ctx, int slot, const int32_t *pos, const uint8_t *ref_len, int64_t n, int space, phz_variants *resident);

/* Read -> variant allele mapper.  On PHZ_E_CAPACITY *n_calls holds the required capacity. */
int phz_map_reads(phz_ctx *ctx, const phz_reads *reads, const phz_variants *vars, int baseq,
                  phz_calls *out, int64_t *n_calls, int space);

/* The (chromosome, BAM) shards of one fan-out -- parallelize(call_mapping_script, chromosomes) at phaser/phaser.py:533 --
 * submitted back to back on the ctx stream with ONE host wait.  Device pointers only (shards resident in HBM); reads[i],
 * vars[i], out[i], n_calls[i] describe shard i.  PHZ_E_CAPACITY when some out[i] is too small (its n_calls[i] holds the need). */
int phz_map_reads_batch(phz_ctx *ctx, int n_shards, const phz_reads *reads, const phz_variants *vars, int baseq,
                        const phz_calls *out, int64_t *n_calls);

/* AS histogram of one shard's call lines, ACCUMULATED into hist[PHZ_AS_BINS] (int64). */
int phz_as_histogram(phz_ctx *ctx, const phz_lines *shard, int64_t *hist, int space);
/* the same for several device-resident shards into one device-resident histogram, one host wait */
int phz_as_histogram_batch(phz_ctx *ctx, const phz_lines *shards, int n_shards, int64_t *hist);
/* the same for a caller that needs no all-reduce of the histogram: it stays on the device and the host gets its occupied bins, in bin
 * order (bin b holds AS == b - 32768; alignment scores live in a narrow band).  cap <= 4096; more occupied bins than cap: PHZ_E_CAPACITY with
 * *n_bins = their number (take phz_as_histogram_batch then). */
int phz_as_histogram_sparse(phz_ctx *ctx, const phz_lines *shards, int n_shards, int cap, int32_t *bins /* [cap] */, int64_t *counts /* [cap] */,
                            int32_t *n_bins);

/* DEVICE arrays: the AS column + has-AS flag (NULL = every record has one) of a shard -> the 2-byte plane of phz_lines.read_as16.  Enqueued on the ctx stream, no
 * host wait (its consumers run on the same stream). */
int phz_as_plane(phz_ctx *ctx, const int32_t *aln, const uint8_t *has_as, int64_t n, int16_t *out);
/* ... and the whole of phaser.py:545-553 for one BAM in one call and one host wait: histogram of its shards' AS column on the device, occupied bins back,
 * numpy.percentile(scores, q_percent) (default linear method, the same float64 operations) computed natively.  *found = 0: no record carries an AS tag.
 * PHZ_E_CAPACITY: more than 4,096 distinct scores (take phz_as_histogram_batch and the host formula then). */
int phz_as_cutoff(phz_ctx *ctx, const phz_lines *shards, int n_shards, double q_percent, double *cutoff, int32_t *found);
/* The same WITHOUT a host wait: histogram, order statistics and numpy.percentile's interpolation all on the device (the 64 Ki-bin histogram is scanned by one
 * workgroup; the same float64 operations as phz_as_cutoff, contraction off), enqueued on the ctx stream.  dev_out: DEVICE memory of four doubles --
 * [0] the cutoff, [1] 1.0 / 0.0 = some / no record carries an AS tag, [2] != 0: an AS value outside int16 was seen (the caller must refuse the input when
 * it reads the block back), [3] number of scores.  Point phz_lines.as_cutoff_dev of the BAM's shards at it for the phz_tally that follows on the same ctx. */
int phz_as_cutoff_enqueue(phz_ctx *ctx, const phz_lines *shards, int n_shards, double q_percent, double *dev_out);

/* Per-variant counters, distinct read sets, variant-pair co-occurrence cells and per-(variant, allele, BAM) read lists over
 * any number of (chromosome, BAM) shards in one submission.  Shards must be ordered by (chromosome, BAM); a chromosome's
 * shards share var_base / qid_base; nv / n_qid are the sizes of the joint index spaces.  a0/a1: the individual's two allele
 * base codes per variant (255 when an allele is not a single ACGT base).  The results stay resident in HBM until the next
 * phz_tally on this ctx: phz_tally_fetch copies what the caller wants, phz_components can use the edge list in place. */
int phz_tally(phz_ctx *ctx, const phz_lines *shards, int n_shards, int64_t nv, const uint8_t *a0, const uint8_t *a1,
              int64_t n_qid, int n_bams, phz_tally_sizes *sizes, int space);
int phz_tally_fetch(phz_ctx *ctx, const phz_tally_out *out, int space);
/* SURVEY.md 8(b) `phz_hap_counts`: DISTINCT reads per (variant, allele, BAM) read list of the resident tally = len(set(haplo_reads[allele][bam])), the
 * per-variant haplotype counts of phaser/phaser.py:1196-1204 (a block's union over its variants is the read-set stage of phz_rowsdev_run).
 * counts[(variant * 2 + allele) * n_bams + bam]; n_counts must equal variants x 2 x BAMs of the last phz_tally. */
int phz_hap_counts(phz_ctx *ctx, int32_t *counts, int64_t n_counts, int space);

/* Connected components of the variant graph restricted to edges with keep != 0: label[v] = smallest variant
 * index of v's component.  edge_a == edge_b == NULL: the edge list of the last phz_tally (n_edges must match). */
int phz_components(phz_ctx *ctx, int64_t nv, int64_t n_edges, const int32_t *edge_a, const int32_t *edge_b,
                   const uint8_t *keep, int32_t *label, int space);

/* ---- device side of the path's input (SURVEY.md 8(f) next-1 on the GPU): BGZF inflate, BAM record decode, filters and SoA packing in
 * HBM.  Replaces the same samtools pipeline as the host functions below (phaser/phaser.py:1346, :505-513). */
typedef struct {                 /* one BGZF member */
    uint64_t src;                /* byte offset of its raw deflate stream in the compressed buffer */
    uint32_t csize, isize;       /* compressed size of that stream; ISIZE (uncompressed size) from the member trailer */
    uint64_t dst;                /* byte offset of its output */
} phz_bgzf_member;

/* Inflates members whose compressed bytes are in DEVICE memory (16-byte aligned, readable for 16 bytes past the last member) into `out` (device).
 * *bad = 0, or a code > 0 when some member is not valid DEFLATE / does not produce ISIZE bytes: the output is unusable then. */
int phz_bgzf_inflate_device(phz_ctx *ctx, const uint8_t *comp, const phz_bgzf_member *members, int64_t n_members, uint8_t *out, int *bad);
/* CRC-32 of the inflated members in `out` (device) against `crc` (device; the CRC32 field of each member's trailer, in member order) -- the check htslib makes
 * after inflating a BGZF block, i.e. where the reference's `samtools view` (phaser/phaser.py:1346) stops on a damaged file.  *bad = 0, or 7 on a mismatch. */
int phz_bgzf_crc_device(phz_ctx *ctx, const uint8_t *out, const phz_bgzf_member *members, int64_t n_members, const uint32_t *crc, int *bad);

/* A coordinate-sorted BAM file decoded on the device: plan on the host (member table, header, chromosome ranges), then H2D of the
 * compressed members, K_inflate, record boundaries + filters + kept-record list, and k_pack into caller-allocated DEVICE arrays with
 * the layout of the phz_reads arrays plus the AS and QNAME columns.  PHZ_E_UNSUPPORTED: the file needs the host path (phz_bam_open*). */
typedef struct phz_bamdev phz_bamdev;
typedef struct { int32_t min_mapq, flag_required, flag_forbidden; double isize_cutoff; } phz_bam_filters;
typedef struct { int64_t n_reads, n_ops, n_seq_bytes, n_qname_bytes; } phz_bamdev_sizes;
typedef struct {                 /* device pointers; sizes from phz_bamdev_sizes_of: pos/aln_score/has_as [n_reads], *_off [n_reads+1], */
    int32_t *pos;                /* cigar [n_ops], seq2 [n_seq_bytes], qual [4*n_seq_bytes], qnames [n_qname_bytes] (no separators)    */
    uint32_t *cigar_off, *cigar, *seq_off;
    uint8_t *seq2, *qual;
    int32_t *aln_score;
    uint8_t *has_as;
    uint32_t *qname_off;
    char *qnames;
} phz_dev_shard;
int phz_bamdev_open(phz_ctx *ctx, const char *path, const char *const *ref_names, int n_names, const phz_bam_filters *filters, phz_bamdev **out);
int phz_bamdev_close(phz_bamdev *h);
int phz_bamdev_n_ref(const phz_bamdev *h);
const char *phz_bamdev_ref_name(const phz_bamdev *h, int i);
int64_t phz_bamdev_ref_length(const phz_bamdev *h, int i);
int phz_bamdev_sizes_of(const phz_bamdev *h, int ref, phz_bamdev_sizes *out);
int phz_bamdev_pack(phz_bamdev *h, const phz_dev_shard *dst, int n_dst);      /* dst[r] for reference r; entries of empty references are ignored */
/* QNAME ids of a device-resident shard, continuing the numbering of earlier BAMs: `store` / `store_off` [n_old + 1] hold the names of
 * ids [0, n_old) in id order (NULL / 0: none yet).  qid[n] = exactly what phz_intern assigns; first_idx[0, *n_new) = record of the
 * first occurrence of every new name, in id order.  All pointers device, n_new host. */
int phz_intern_device(phz_ctx *ctx, const char *qnames, const uint32_t *qname_off, int64_t n, const char *store, const uint32_t *store_off,
                      int64_t n_old, int32_t *qid, int32_t *first_idx, int64_t *n_new);
/* Appends the names of the new ids to the store, in two calls: dst == NULL fills dst_off[0 .. m] (byte offsets of the new names, the
 * first at base_bytes) and *total_bytes; dst != NULL copies the name bytes there. */
int phz_names_append_device(phz_ctx *ctx, const char *qnames, const uint32_t *qname_off, const int32_t *first_idx, int64_t m, uint32_t base_bytes,
                            uint32_t *dst_off, char *dst, int64_t *total_bytes);

/* ---- host side of the path's input: native BGZF/BAM decode, SoA packing, QNAME interning -------------------
 * Replaces `samtools view -h BAM 'chr': | samtools view -Sh [-F 0x400] [-f 2] -q MAPQ -` (phaser/phaser.py:1346)
 * and the mapper's per-record text parsing (phaser/read_variant_map.py:27-64).  Pure host code (zlib + threads). */
typedef struct phz_bam phz_bam;
typedef struct phz_interner phz_interner;

typedef struct {                 /* one reference sequence's filtered records, arrays owned by the phz_bam */
    const char *ref_name;
    int64_t n_reads, n_ops, n_seq_bytes;
    const int32_t *pos;
    const uint32_t *cigar_off;
    const uint32_t *cigar;
    const uint32_t *seq_off;
    const uint8_t *seq2;
    const uint8_t *qual;
    const int32_t *aln_score;    /* AS:i (0 when absent) */
    const uint8_t *has_as;
    const uint32_t *qname_off;   /* [n_reads+1] into qnames */
    const char *qnames;
} phz_host_shard;

int phz_bam_open(const char *path, int threads, phz_bam **out);      /* reads + inflates the file, parses the header */
/* Chromosome-restricted open (`samtools view BAM 'chr':`, phaser.py:1346, without an index file): only the BGZF members that can
 * hold records of the named references are inflated (found by binary search over the member table of a coordinate-sorted BAM);
 * ref_names == NULL opens everything.  ref_bytes (may be NULL, room for max_refs): compressed bytes per reference, a proxy of its
 * record count before anything is decoded.  out == NULL: only ref_bytes is produced (nothing but a few members is inflated). */
int phz_bam_open_refs(const char *path, int threads, const char *const *ref_names, int n_names, int64_t *ref_bytes, int max_refs,
                      phz_bam **out);
int phz_bam_close(phz_bam *bam);
int phz_bam_n_ref(const phz_bam *bam);
const char *phz_bam_ref_name(const phz_bam *bam, int i);
int64_t phz_bam_ref_length(const phz_bam *bam, int i);
/* one-shot (the inflated stream is released afterwards):
 * keep records with ref_mask[refID] != 0 (NULL = all), MAPQ >= min_mapq, (flag & required) == required,
 * (flag & forbidden) == 0 and |TLEN| <= isize_cutoff when isize_cutoff != 0 */
int phz_bam_decode(phz_bam *bam, const uint8_t *ref_mask, int min_mapq, int flag_required, int flag_forbidden,
                   double isize_cutoff, int threads, int *n_shards);
int phz_bam_shard(phz_bam *bam, int i, phz_host_shard *out);

/* ---- SAM text front end of the mapper seam (phaser/read_variant_map.py:25-64 per input line + the packing of one shard per
 * chromosome): @SQ contigs, field split, |TLEN| <= isize_cutoff (0 = no filter), AS = last AS: tag, records grouped per RNAME in
 * input order.  `text` must stay alive as long as the handle (records keep pointers into it). */
typedef struct phz_sam phz_sam;
int phz_sam_parse(const char *text, int64_t len, double isize_cutoff, int threads, phz_sam **out);
const char *phz_sam_error(const phz_sam *h);
void phz_sam_free(phz_sam *h);
int64_t phz_sam_n_records(const phz_sam *h);          /* alignment lines seen (before the TLEN filter) */
/* 1 when every chromosome is ONE run of the stream in coordinate order, counting ALL alignment lines (also the ones the TLEN filter
 * drops: the reference prunes its variant buffer by every record, read_variant_map.py:37-50).  Only then is the mapper's result the
 * stateless rule the kernels implement; the drop-in sends any other stream down its forward-only-buffer path (or refuses it). */
int phz_sam_stream_order(const phz_sam *h);
int phz_sam_n_contigs(const phz_sam *h);
const char *phz_sam_contig(const phz_sam *h, int i);
int phz_sam_n_shards(const phz_sam *h);
int phz_sam_shard(phz_sam *h, int i, phz_host_shard *out);
/* the mapper's output lines of one chromosome (read_variant_map.py:117) from a K_map call list; id / rsid / gt / maf are sep pools
 * over the chromosome's variant-table rows.  out is malloc'd (phz_buf_free). */
int phz_sam_calls_tsv(const phz_sam *h, int shard, int64_t n_calls, const int32_t *read_idx, const int32_t *var_idx, const uint8_t *code,
                      const uint32_t *aux0, const uint32_t *aux1, int baseq, const uint32_t *id_off, const char *id,
                      const uint32_t *rsid_off, const char *rsid, const uint32_t *gt_off, const char *gt, const uint32_t *maf_off,
                      const char *maf, int threads, char **out, int64_t *out_len);

/* whole BGZF file -> buffer owned by the library (release it with phz_buf_free and nothing else: a large text is an anonymous mapping, not a malloc'd block);
 * PHZ_E_UNSUPPORTED when the file is plain gzip */
int phz_bgzf_read(const char *path, int threads, char **data, int64_t *len);
void phz_buf_free(char *p);
/* data -> BGZF file: 60,000-byte members deflated in parallel + EOF marker (what bgzip writes, phaser.py:1851) */
int phz_bgzf_write(const char *path, const char *data, int64_t len, int threads, int level);

/* BAM writer for synthetic fixed-length read batches (test / benchmark tooling; one batch per reference, coordinate-sorted):
 * seq holds base codes 0..3 (4 = N), qual phred values, cigar BAM-coded words, QNAME = qname_prefix + str(qid). */
typedef struct {
    int64_t n;
    int32_t ref_id, L;
    const int32_t *pos, *flag, *mapq, *tlen, *aln_score, *qid;   /* pos 1-based */
    const int64_t *cigar_off;                                     /* [n+1] */
    const uint32_t *cigar;
    const uint8_t *seq, *qual;                                    /* [n*L] */
    const char *qname_prefix;
} phz_read_batch;
int phz_bam_write(const char *path, int n_ref, const char *const *ref_names, const int32_t *ref_lens, const phz_read_batch *batches,
                  int n_batches, int threads);

/* tabix index <bgzf_path>.tbi of a BGZF-compressed position-sorted file; preset 0 = VCF (tabix -p vcf), 1 = BED (tabix -p bed) */
int phz_tabix_build(const char *bgzf_path, int preset, int threads);
/* phz_bgzf_write + phz_tabix_build of the same text in one call (`bgzip` + `tabix -p vcf -f`, phaser.py:1851): the index is gathered
 * next to the deflate workers, nothing is read back.  PHZ_E_UNSUPPORTED: text not position-sorted -- the .gz is written, no .tbi. */
int phz_bgzf_write_indexed(const char *path, const char *data, int64_t len, int threads, int level, int preset);

int phz_interner_create(phz_interner **out);
int phz_interner_destroy(phz_interner *it);
int64_t phz_interner_size(const phz_interner *it);
int phz_intern(phz_interner *it, const char *blob, const uint32_t *off, int64_t n, int32_t *out_id);
int phz_interner_names(const phz_interner *it, char *blob, int64_t blob_cap, uint32_t *off);

/* Mapper for variant sets with indels (phASER --include_indels 1; read_variant_map.py:236-258 with ref_length > 1).
 * Same call list as phz_map_reads, but `code` is classified on the device against the allele strings:
 * 5 = text equals allele 0, 6 = equals allele 1, 0..3 = another single base, 4 = any other text.
 * Optional text pool (both NULL to skip): call_text_off[n_calls+1] / text_roff[] = read offsets of the characters of every
 * code-4 call, so the host can print the exact allele text.  On PHZ_E_CAPACITY *n_calls / *n_text hold the needs. */
int phz_map_reads_general(phz_ctx *ctx, const phz_reads *reads, const phz_variants_general *vars, int baseq,
                          phz_calls *out, int64_t *n_calls, uint32_t *call_text_off, uint32_t *text_roff,
                          int64_t text_cap, int64_t *n_text, int space);

/* ---- host stage C2: block phasing + the text rows of the five output files, one chromosome per call ----------------
 * Replaces phase_v3 (phaser/phaser.py:2107-2324), the block output loop (:865-1172), singleton rows (:1180-1239),
 * variant_connections rows (:691-695) and allelic_counts rows (:737-749).  Host arrays only; multi-threaded.
 * A "sep pool" is n strings joined by one separator byte: string i = bytes [off[i], off[i+1]-1). */
typedef struct {
    const char *chrom;
    int32_t nv;
    const int32_t *pos;
    const uint32_t *uid_off;    const char *uid;       /* unique ids */
    const uint32_t *rsid_off;   const char *rsid;      /* rsid, '.' already replaced by the unique id (:1451-1455) */
    const uint32_t *allele_off; const char *allele;    /* [2nv] the individual's alleles, index 2*v + k */
    const uint32_t *maf_off;    const char *maf_txt;   /* str(maf) per variant */
    const double *maf;
    const uint8_t *is_ref;        /* [2nv] allele k of v equals REF */
    const int8_t *phase_idx;      /* [2nv] position of allele k in the VCF phase, -1 when the genotype is unphased */
    const uint8_t *blacklisted;   /* [nv] --haplo_count_blacklist hit, may be NULL */
    const int32_t *var_count;     /* [3nv] from phz_tally */
    const int32_t *var_distinct;  /* [3nv] */
    /* read lists from phz_tally: the kept lines of (variant v, allele k, BAM b) carry the QNAME ids
     * rl_qid[rl_start[(2v+k)*nb+b] : rl_start[(2v+k)*nb+b+1]] in line order.  rl_start points at THIS chromosome's first entry of
     * the tally's array (its values index the tally's whole rl_qid, which is passed as is) */
    const uint32_t *rl_start;
    const int32_t *rl_qid;
    /* tested variant pairs (linked edges), oriented by first appearance: rows of variant_connections in eorder */
    int64_t n_edges;
    const int32_t *va, *vb, *ea, *eb;
    const int32_t *sup, *tot, *cis, *trans, *cfgv;
    const int64_t *eorder;
    const double *pv;
    /* connected components of the surviving graph */
    int64_t ncomp;
    const int32_t *mem_s;
    const int64_t *comp_starts, *comp_ends, *comp_order, *e_keep, *eo, *e_starts, *e_ends;
    /* first-appearance keys of covered variants, sorted by (BAM of first kept line, line) */
    int64_t n_keys;
    const int64_t *key_bam, *key_g;
    int32_t nb;
    const char *const *bam_names;
    const uint8_t *bam_excluded;  /* [nb], may be NULL */
    int32_t unique_ids, gw_phase_method, output_read_ids, unphased_vars, max_block_size, want_vcf, threads;
    const uint32_t *qname_off;  const char *qname;     /* QNAME per template id; only read when output_read_ids == 1 */
    /* raw != 0: the ordering stage runs inside the library.  Then ea / eb hold the tested pairs as indices of the tally's joint
     * variant space (v0 = this chromosome's first variant there), sorted by (ea, eb); sup / tot / cis / trans / cfgv / pv are
     * per tested pair; keep[e] != 0 when the pair survived the conflict test; rank / label / var_first are this chromosome's
     * slices of phz_tally's var_rank, phz_components' labels (joint space) and var_first; lines of BAM b occupy
     * [bam_line_lo[b], bam_line_hi[b]) of the tally's line space (-1, -1 when the BAM has no shard here).  va, vb, eorder, the
     * component arrays and the first-appearance keys are derived from those and need not be set. */
    int32_t raw;
    int64_t v0;
    const uint8_t *keep;
    const uint64_t *rank;
    const int32_t *label;
    const int64_t *var_first;
    const int64_t *bam_line_lo, *bam_line_hi;
} phz_rows_in;

/* Row text comes back as the buffers the worker threads filled, in output order (no concatenation pass: at whole-genome scale the
 * text is ~1 GB).  bam[i] (allelic counts / singleton rows only) = the first BAM the rows of part i are keyed to (rule 2). */
typedef struct {
    int64_t n;
    const char *const *ptr;
    const int64_t *len;
    const int32_t *bam;
} phz_text_parts;

typedef struct {
    phz_text_parts conn, hap, ase, cfg, allelic, single_ase, single_hap;     /* reference row order within the chromosome */
    int64_t allelic_rows, n_blocks, phased, n_blk_vars;
    int32_t *blk_size;        /* [n_blocks] variants per block */
    /* filled when want_vcf: per block variant lists and genome-wide phase for write_vcf (:1661-1855) */
    int32_t *blk_var;         /* [n_blk_vars] */
    uint8_t *blk_hap;         /* [n_blk_vars] allele index on haplotype A */
    int8_t *blk_cor;          /* [2*n_blk_vars] corrected phase of haplotype A / B alleles, -1 = none */
    double *blk_stat;         /* [n_blocks] gw confidence */
    uint8_t *blk_stat_int;    /* [n_blocks] 1 when the reference prints the int 1 */
    int32_t *blk_maxmaf;      /* [n_blocks] variant carrying max(maf) */
    void *owner;              /* keeps the text buffers alive until phz_rows_free */
} phz_rows_out;

int phz_rows_format(const phz_rows_in *in, phz_rows_out *out);
/* several chromosomes through one pool of `threads` workers: in[i] -> out[i] */
int phz_rows_format_multi(const phz_rows_in *in, int n_chroms, phz_rows_out *out, int threads);
void phz_rows_free(phz_rows_out *out);
/* phase_v3 (phaser.py:2107-2170) on one connected component: n position-sorted variants, edges (i, j, cfg 0 cis / 1 trans /
 * -1 tie) in local indices.  Outputs the final sub-blocks: first local variant, length, and haplotype-A allele characters
 * written back to back into config (capacity n); sub_first / sub_len need capacity n. */
int phz_phase_block(int32_t n, int64_t n_edges, const int32_t *edge_i, const int32_t *edge_j, const int8_t *edge_cfg,
                    int32_t max_block_size, int32_t *sub_first, int32_t *sub_len, char *config, int32_t *n_subs);

/* ---- device row stage: stages T7-O2 of the phasing path on the GPU, on the results phz_tally left in HBM --------------------
 * What it replaces in the reference (phaser/phaser.py): the bookkeeping of test_variant_connection :1594-1654 (the binomial p-value
 * itself stays the caller's scipy call, evaluated once per distinct argument pair), pruning :686-726, build_haplotypes :1861-1882,
 * phase_v3 :2107-2324, the output loops :691-695, :737-749, :865-1239.  The host twin is phz_rows_format_multi above; this stage declines
 * no option of the reference any more (--gw_phase_method 1 since round 5, --output_read_ids 1 since round 6: the QNAME lists of :1120-1123 /
 * :1196-1204 in first-appearance order, the canonical form of those Python sets).  No size of a component, a block or the read-count pairs
 * sends a pass to the host twin: components beyond the phasing kernel's limits are phased by the host routine inside the call (that component
 * only), the pair-key table grows on demand (PHZ_E_CAPACITY -> phz_rowsdev_set_pair_slots).
 *
 *   phz_rowsdev_create     upload the per-variant tables of this rank's chromosomes (joint variant index space of phz_tally)
 *   phz_rowsdev_pair_keys  stage 1: the distinct (total, supporting) read-count pairs of the pairs under test -> host
 *   phz_rowsdev_run        stage 2: verdicts, components, ordering, block phasing, read sets, the text of the five files in HBM
 *   phz_rowsdev_fetch_*    copy a finished text / the per-block arrays of write_vcf to host memory
 * String pools: item i of a pool = bytes [off[i], off[i+1] - 1) (one separator byte after every item). */
typedef struct phz_rowsdev phz_rowsdev;
enum { PHZ_PAIR_SLOTS = 16384 };      /* (a genome-wide sample has ~2,000 distinct count pairs; the table grows by itself, see phz_rowsdev_set_pair_slots) */
enum { PHZ_TXT_CONN = 0, PHZ_TXT_HAP = 1, PHZ_TXT_ASE = 2, PHZ_TXT_CFG = 3, PHZ_TXT_ALLELIC = 4, PHZ_TXT_SINGLE_ASE = 5, PHZ_TXT_SINGLE_HAP = 6,
       PHZ_TXT_COUNT = 7 };

typedef struct {
    int64_t nv;                       /* variants of all chromosomes of this rank, chromosomes in VCF order */
    int32_t n_chroms;
    const int64_t *chrom_v0;          /* [n_chroms + 1] first variant of every chromosome */
    const uint32_t *chrom_name_off; const char *chrom_names;
    const int32_t *pos;               /* [nv] */
    const uint32_t *uid_off; const char *uid;            /* nv items: unique id (chrom_pos_alleles, :1376) */
    const uint32_t *rsid_off; const char *rsid;          /* nv items: rsid, or the unique id when the VCF has '.' (:1451-1455) */
    const uint32_t *allele_off; const char *allele;      /* 2 nv items: the individual's two alleles in allele-index order */
    const uint32_t *maf_off; const char *maf_txt;        /* nv items: str(maf) */
    const double *maf;                /* [nv] */
    const uint8_t *is_ref;            /* [2 nv] allele k of variant v is the reference allele */
    const int8_t *phase_idx;          /* [2 nv] position of allele k in the VCF phase, -1 = unphased genotype */
    const uint8_t *blacklisted;       /* [nv] --haplo_count_blacklist, or NULL */
} phz_rowsdev_tables;

typedef struct {
    int32_t n_bams;
    const uint32_t *bam_name_off; const char *bam_names;
    const uint8_t *bam_excluded;      /* [n_bams] --haplo_count_bam_exclude, or NULL */
    int32_t n_shards;                 /* the (chromosome, BAM) shards of the phz_tally call: their line ranges and BAMs */
    const int64_t *shard_line_lo, *shard_line_hi;
    const int32_t *shard_bam;
    int32_t unique_ids, gw_phase_method, output_read_ids, unphased_vars, max_block_size, want_vcf;
    double cc_threshold;
    /* only read when output_read_ids == 1: the QNAMEs behind the template ids of the tally's read lists.  Template ids are per chromosome (phz_intern*),
     * so the names of all chromosomes sit in ONE string pool, chromosome after chromosome in the order of phz_rowsdev_tables, and
     * qname_base[c] = pool item of template id 0 of chromosome c ([n_chroms + 1]; the last entry = number of pool items). */
    const uint32_t *qname_off; const char *qname;
    const int64_t *qname_base;
    /* optional (NULL = off): a PAGE-LOCKED host region for the finished text.  When it holds all seven texts (each at a 4 KB boundary) the run copies every text there on a
     * second stream as soon as its writer kernel has finished -- the largest first -- and phz_rowsdev_result.host_off says where; otherwise host_off is -1 everywhere and
     * the caller takes the texts with phz_rowsdev_fetch_text as before. */
    void *host_text; int64_t host_text_cap;
} phz_rowsdev_opts;

typedef struct {
    int64_t bytes[PHZ_TXT_COUNT];
    /* byte offsets of the per-chromosome segments of every text (host arrays owned by the handle, valid until the next run):
     * CONN / HAP / ASE / CFG: [n_chroms + 1]; ALLELIC / SINGLE_*: [n_bams * n_chroms + 1], BAM of the first kept line major */
    const int64_t *seg_off[PHZ_TXT_COUNT];
    const int64_t *chrom_blocks, *chrom_blk_vars;        /* [n_chroms] blocks / block variants per chromosome */
    const int32_t *chrom_first_bam;                      /* [n_chroms] first BAM with a kept call line on the chromosome (-1: none): the reference lists the
                                                          * chromosomes of the block files in (that BAM, VCF order) order -- read_vars is keyed by the chromosome
                                                          * process_mapping_result returns, "" for a call file without kept lines (phaser.py:1299, :573-574) */
    int64_t n_blocks, n_blk_vars, phased, dropped, allelic_rows, n_components, n_linked, n_complex, n_exceptions, n_big_segments;
    double gpu_ms;                    /* HIP-event time of the sync-free sections of the run */
    int64_t host_off[PHZ_TXT_COUNT];  /* byte offset of every text in phz_rowsdev_opts.host_text, -1 = not copied (no region given, or too small) */
} phz_rowsdev_result;

int phz_rowsdev_create(phz_ctx *ctx, const phz_rowsdev_tables *tables, phz_rowsdev **out);
void phz_rowsdev_destroy(phz_rowsdev *h);
/* Size of the pair-key hash set: a power of two in [16, 2^28] (below PHZ_PAIR_SLOTS: tests only), PHZ_PAIR_SLOTS by default and kept by the handle.  phz_rowsdev_pair_keys returns
 * PHZ_E_CAPACITY when the distinct (supporting, total) pairs of a pass do not fit (very deep coverage): quadruple and call it again. */
int phz_rowsdev_set_pair_slots(phz_rowsdev *h, int64_t n_slots);
int64_t phz_rowsdev_pair_slots(const phz_rowsdev *h);
/* The (chromosome, BAM) shards of the phz_tally whose results the next phz_rowsdev_pair_keys reads -- line range and BAM per shard, in line order (= phz_rowsdev_opts.shard_*).
 * Optional: with them the first stage also enqueues the first-appearance keys of allelic_counts / the singleton rows (p-value-independent work that then runs while the
 * caller evaluates the p-values); phz_rowsdev_run checks them against its own options and redoes the keys when they differ.  phz_tally_pairs sets them itself. */
int phz_rowsdev_set_shards(phz_rowsdev *h, int32_t n_shards, const int64_t *line_lo, const int64_t *line_hi, const int32_t *bam);
int phz_rowsdev_pair_keys(phz_ctx *ctx, phz_rowsdev *h, uint64_t *keys_host /* [phz_rowsdev_pair_slots(h)] */);
/* phz_tally (same arguments) and phz_rowsdev_pair_keys in one call: no caller glue between the tally's last kernel and the first kernel of the row stage.
 * Returns phz_tally's status; *pair_status = phz_rowsdev_pair_keys' (PHZ_E_CAPACITY: phz_rowsdev_set_pair_slots, then phz_rowsdev_pair_keys alone). */
int phz_tally_pairs(phz_ctx *ctx, const phz_lines *shards, int n_shards, int64_t nv, const uint8_t *a0, const uint8_t *a1, int64_t n_qid, int n_bams,
                    phz_tally_sizes *sizes, int space, phz_rowsdev *h, uint64_t *keys_host, int32_t *pair_status);
/* host helper between the two stages: values and repr() text of the p-values laid out by slot (used[] ascending = the occupied slots of
 * phz_rowsdev_pair_keys, pv[i] = scipy.stats.binom.cdf for slot used[i]).  Returns the bytes of txt, -1 on bad arguments / txt_cap too small
 * (n_slots + 40 bytes per used slot always fits). */
/* host helper in front of it: the occupied slots of keys_host in slot order -> used[] (slot), sup[] (supporting count as float64), tot[] (total as int64): the argument
 * arrays of the binomial call; every output has capacity n_slots; returns their number (-1 on bad arguments). */
int64_t phz_pair_slots_used(const uint64_t *keys, int64_t n_slots, uint32_t *used, double *sup, int64_t *tot);
int64_t phz_pair_slot_text(const uint32_t *used, const double *pv, int64_t n_used, int64_t n_slots, double *slot_pv /* [n_slots] */,
                           uint32_t *txt_off /* [n_slots + 1] */, char *txt, int64_t txt_cap);
int phz_rowsdev_run(phz_ctx *ctx, phz_rowsdev *h, const phz_rowsdev_opts *opts, const double *slot_pv /* [phz_rowsdev_pair_slots(h)] */,
                    const uint32_t *slot_txt_off /* [phz_rowsdev_pair_slots(h) + 1] */, const char *slot_txt, phz_rowsdev_result *result);
int phz_rowsdev_fetch_text(phz_ctx *ctx, phz_rowsdev *h, int which, void *dst, int64_t bytes);
const void *phz_rowsdev_text_ptr(phz_rowsdev *h, int which);      /* device pointer of a finished text */
int phz_rowsdev_fetch_blocks(phz_ctx *ctx, phz_rowsdev *h, int32_t *blk_size, int32_t *blk_var, uint8_t *blk_hap, int8_t *blk_cor, double *blk_stat,
                             uint8_t *blk_stat_int, int32_t *blk_maxmaf);

/* phase_v3 (phaser/phaser.py:2107-2170; the worker `parallelize(phase_v3, ...)` fans out at :808) for a batch of connected components on the
 * GPU: flood fill, weak-point split, 2^(n-1) brute force shared by the lanes of a wave, stitching.  Component c = position-sorted variants
 * [comp_start[c], comp_start[c+1]) with the pairs [pair_start[c], pair_start[c+1]); pair_i / pair_j are LOCAL variant indices, pair_cfg 0 same
 * configuration / 1 opposite / -1 tie.  Host arrays in and out.  sub_of[v] = ordinal of v's final block inside its component (-1: none),
 * alle_of[v] = its allele on haplotype A, n_sub[c] = final blocks of component c.  Same results as phz_phase_block per component. */
int phz_phase_components(phz_ctx *ctx, int64_t n_comp, const uint32_t *comp_start, const uint32_t *pair_start, const int32_t *pair_i, const int32_t *pair_j,
                         const int8_t *pair_cfg, int32_t max_block_size, int32_t *sub_of, uint8_t *alle_of, uint32_t *n_sub);

/* Adopt tally results computed elsewhere (another process / device, or a fixture) as the resident results of this ctx: the arrays of
 * phz_tally_out + sizes, in `space`.  rl_list[i] = index (variant * 2 + allele) * n_bams + bam of the read list entry i belongs to. */
int phz_tally_import(phz_ctx *ctx, int64_t nv, int n_bams, const phz_tally_sizes *sizes, const phz_tally_out *arrays, const uint32_t *rl_list, int space);

/* ---- phaser_gene_ae (phaser_gene_ae/phaser_gene_ae.py): gene-level haplotypic counts from a haplotypic_counts.txt ---------
 * phz_hc_parse: multi-threaded parse of the file text (:78 pandas.read_csv + the per-row string splitting of :172-204).
 * The arrays stay owned by the handle; var_id_off / var_id_len point into the caller's text buffer. */
typedef struct phz_hc phz_hc;
typedef struct {
    int64_t n_rows, n_vars, n_lab_a, n_lab_b;
    int32_t n_contigs, n_bams, has_maf;
    const int32_t *contig, *start, *stop, *a_count, *b_count, *total, *bam;
    const int32_t *phase;          /* blockGWPhase: 0 "0/1", 1 "0|1", 2 "1|0", 3 anything else */
    const double *gw_stat, *maf;
    const int64_t *var_off;        /* [n_rows+1] */
    const int32_t *var_pos;        /* [n_vars] second separator-delimited field of the variant id (:186-187) */
    const int64_t *var_id_off;     /* [n_vars] id text = input[var_id_off : var_id_off + var_id_len] */
    const int32_t *var_id_len;
    const int64_t *lab_off_a, *lab_off_b;   /* [n_rows+1] label runs per row (empty for single-variant rows, :193-197) */
    const int32_t *lab_pos_a, *lab_prev_a, *lab_pos_b, *lab_prev_b;
    const char *names;             /* contig names then BAM names, NUL-terminated, at names_off[i] */
    const int64_t *names_off;      /* [n_contigs + n_bams + 1] */
} phz_hc_arrays;

int phz_hc_parse(const char *text, int64_t len, const char *id_separator, int threads, phz_hc **out);
int phz_hc_view(const phz_hc *h, phz_hc_arrays *view);
const char *phz_hc_error(const phz_hc *h);
void phz_hc_free(phz_hc *h);

/* K_genes: distinct reads per haplotype for every (row, feature) pair (variant_feature_reads :172-219).  A work item is a
 * slice of one row-haplotype label run: labels [item_lo, item_lo + item_n), run start item_run (lab_prev is run-relative).
 * pair_counts[2*pair + hap] receives the count; pair_begin / pair_end are the feature's BED start / stop. */
typedef struct {
    int64_t n_items;
    const int64_t *item_lo;
    const int32_t *item_n;
    const int64_t *item_run;
    const int32_t *item_pair;
    const uint8_t *item_hap;
    int64_t n_pairs;
    const int32_t *pair_begin, *pair_end;
    int64_t n_lab_a, n_lab_b;
    const int32_t *lab_pos_a, *lab_prev_a, *lab_pos_b, *lab_prev_b;
} phz_gene_work;

int phz_gene_counts(phz_ctx *ctx, const phz_gene_work *work, int32_t *pair_counts, int space);

/* Output rows of phaser_gene_ae (:147-163), threaded.  Keys are bam * n_features + feature.  Pools: items followed by '\n'.
 * out is malloc'd (phz_buf_free). */
typedef struct {
    int64_t n_features;
    const char *feat_chr;  int64_t feat_chr_len;
    const char *feat_name; int64_t feat_name_len;
    const int64_t *feat_start, *feat_stop;
    int32_t n_bam_order; const int32_t *bam_order;      /* BAM ids in output order */
    const char *bam_names; int64_t bam_names_len;
    const int64_t *A, *B, *UA, *UB;                      /* phased a / b, best unphased a / b per key */
    const int64_t *pv_lo, *pv_hi, *pv_sorted;            /* phased variants of a key = pv_sorted[pv_lo : pv_hi] */
    const int64_t *best_lo, *best_hi, *u_var;            /* variants of the best unphased block = u_var[best_lo : best_hi] */
    const char *text; const int64_t *var_id_off; const int32_t *var_id_len;   /* variant id text (phz_hc_arrays) */
    int64_t min_cov;
    int32_t threads;
} phz_gene_rows_in;
int phz_gene_rows(const phz_gene_rows_in *in, char **out, int64_t *out_len);

/* ---- native het-variant loader (phaser/phaser.py:396-433 filter, :1355-1413 table, :1418-1462 per-variant fields) ----------
 * phz_vcf_parse reads VCF text (header lines skipped) with host threads; phz_vcf_chrom hands out one chromosome's table.
 * String columns come as pools: every item is followed by one '\n'.  pool[]: 0 unique id, 1 ID column as written, 2 rsid
 * ('.' replaced by the unique id), 3 REF, 4 "REF,ALT..." , 5 the individual's alleles in allele-index order, 6 alleles in GT
 * order ("-,-" when unphased), 7 GT string, 8 str(maf) as written to the table ("None" without --gw_phase_method 1),
 * 9 str(maf number) ("0" when None), 10 the individual's first two alleles (2 items per variant).  The handle owns all memory. */
typedef struct phz_vcf phz_vcf;
typedef struct {
    int32_t sample_column;            /* 0-based column of the sample */
    const char *chrom_of_interest;    /* "" = all */
    int32_t pass_only, include_indels;
    const char *chr_prefix, *id_separator;
    int32_t gw_phase_method;
    const char *gw_af_field;
    int32_t n_contig_ban;
    const char *const *contig_ban;    /* strings that must not occur in a contig name (:386-392) */
    int32_t threads;
    int32_t grep_hom;                 /* 1: drop lines whose columns 1-9 + sample hold "0|0" or "1|1" (the grep of :220-225) */
    /* BED intervals (0-based, half-open), any order.  A VCF record is the interval [POS-1, POS-1+len(REF)) (bedtools).
     * drop_*: `bedtools intersect -v` of --blacklist (:220-221): overlapping records vanish before anything is counted.
     * mark_*: `bedtools intersect` of --haplo_count_blacklist (:232-241): overlapping variants get blacklisted[] = 1. */
    int64_t n_drop; const char *const *drop_chrom; const int64_t *drop_start, *drop_end;
    int64_t n_mark; const char *const *mark_chrom; const int64_t *mark_start, *mark_end;
} phz_vcf_opts;
typedef struct {
    const char *name;                 /* chr_prefix + CHROM */
    int64_t n;
    const int32_t *pos;
    const uint8_t *ref_len, *a0, *a1; /* len(REF) (a longer REF than 255 bases is refused); base code of the individual's allele 0 / 1 (255 = not one ACGT base) */
    const uint8_t *is_ref;            /* [2n] */
    const int8_t *phase_idx;          /* [2n] */
    const double *maf;
    const char *pool[11];
    int64_t pool_len[11];
    const uint8_t *blacklisted;       /* [n] 1 when the record overlaps a mark_* interval */
} phz_vcf_table;

int phz_vcf_parse(const char *text, int64_t len, const phz_vcf_opts *opts, phz_vcf **out);
int phz_vcf_summary(const phz_vcf *h, int32_t *n_chroms, int64_t *het, int64_t *filter_count, int64_t *indels_excluded, int64_t *unphased);
int phz_vcf_chrom(const phz_vcf *h, int32_t i, phz_vcf_table *table);
const char *phz_vcf_error(const phz_vcf *h);
void phz_vcf_free(phz_vcf *h);

/* ---- native write_vcf (phaser/phaser.py:1661-1855): the sample's VCF text with the phASER FORMAT tags --------------------------
 * One phz_vcfout_chrom per chromosome that has blocks: the variant table's pools (phz_vcf_chrom: 0 unique id, 2 rsid,
 * 5 individual's alleles, 9 str(maf)) and the block arrays of phz_rows_out; first_block_index = number of blocks of the
 * chromosomes before it (PI is 1-based over all chromosomes, :867).  out is malloc'd (phz_buf_free). */
typedef struct {
    const char *uid;     int64_t uid_len;
    const char *rsid;    int64_t rsid_len;
    const char *alleles; int64_t alleles_len;
    const char *maf_str; int64_t maf_str_len;
    int64_t n_blocks, n_blk_vars, first_block_index;
    const int32_t *blk_size, *blk_var, *blk_maxmaf;
    const uint8_t *blk_hap, *blk_stat_int;
    const int8_t *blk_cor;
    const double *blk_stat;
} phz_vcfout_chrom;

int phz_vcf_phase_text(const char *text, int64_t len, int32_t sample_column, const char *id_separator, const char *chrom_of_interest,
                       int32_t gw_phase_vcf, double min_confidence, const phz_vcfout_chrom *chroms, int32_t n_chroms, int32_t threads,
                       char **out, int64_t *out_len, int64_t *unphased_phased, int64_t *corrections);

/* Issue-rate microbenchmark (gfx950): wave64 instructions per second over the whole chip for kind 0 vector ALU (v_add_u32), 1 scalar ALU
 * (s_add_u32), 2 LDS (ds_read_b32), with waves_per_simd (1..8) waves resident on every SIMD; the ceilings K_map's instruction stream is
 * measured against (bench.py roofline.issue).  n_cu / clock_mhz: compute units and reported engine clock of the device. */
int phz_microbench(phz_ctx *ctx, int kind, int waves_per_simd, int iters, double *wave_insts_per_s, int *n_cu, int *clock_mhz);

/* Self-test of the library's device radix sort (stable LSD sort of (key, value) pairs over bit ranges; one launch per pass with decoupled look-back,
 * or the three-launch passes it replaced with three_launch = 1): host arrays in and out, keys of key_bytes = 4 or 8, ranges = nranges x [lo, hi) bit
 * ranges, least significant range first.  The row stage's ordering rules (phaser.py:1271-1283, :1310, :1870, :1884-1887) rest on it; the tests
 * compare it with a stable host sort.  No reference counterpart. */
int phz_selftest_sort(phz_ctx *ctx, int key_bytes, const void *keys, const uint32_t *vals, int64_t n, const int32_t *ranges, int nranges, int three_launch,
                      void *keys_out, uint32_t *vals_out);

/* Memory-side calibration (gfx950, round 5): one launch of an access pattern with a KNOWN byte count, for calibrating rocprofv3's FETCH_SIZE /
 * WRITE_SIZE on the patterns K_map uses (tools/prof_calib.sh; bench.py roofline.traffic applies the measured factors).  kind 0 / 1: coalesced
 * stream of 4 / 16 bytes per lane over the array; 2 / 3: ONE 1-byte load per `unit`-byte unit of the array (32..512), always inside the unit's
 * first 32 bytes, units visited in a pseudo-random permutation (2) or in address order (3); 4 / 5 / 6: coalesced stores of 4 / 8 / 16 bytes
 * per lane.  The array has 2^log2_bytes bytes (past the 256 MiB Infinity Cache from 29 up).  *known_bytes = bytes read / written by the lanes,
 * *requests = loads / stores, *seconds = best HIP-event time of `reps` launches.  No reference counterpart (measurement infrastructure). */
int phz_membench(phz_ctx *ctx, int kind, int log2_bytes, int unit, int reps, double *seconds, int64_t *known_bytes, int64_t *requests);

/* Kernel time measured with HIP events on the ctx stream: last launch, running total, launch count. */
int phz_get_timing(phz_ctx *ctx, int slot, float *last_ms, double *total_ms, int64_t *launches);
int phz_reset_timing(phz_ctx *ctx);
int phz_get_counter(phz_ctx *ctx, int slot, int64_t *value);

#ifdef __cplusplus
}
#endif
#endif /* PHZ_H */
