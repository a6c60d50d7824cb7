"""ctypes binding of libphz.so (C ABI declared in include/phz.h).

There is deliberately NO fallback: if the HIP library is missing or no GPU is visible the
product path raises.  The CPU restatement under oracle/ is test infrastructure only.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import os
import subprocess
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "libphz.so")
CSRC = os.path.join(HERE, "csrc")

PHZ_OK, PHZ_E_ARG, PHZ_E_HIP, PHZ_E_CAPACITY, PHZ_E_UNSUPPORTED, PHZ_E_NOMEM = 0, -1, -2, -3, -4, -5
PHZ_HOST, PHZ_DEVICE = 0, 1
PHZ_T_MAP, PHZ_T_ASHIST, PHZ_T_TALLY, PHZ_T_COMPONENTS, PHZ_T_GENES, PHZ_T_INFLATE, PHZ_T_BAMPACK, PHZ_T_ROWS = 0, 1, 2, 3, 4, 5, 6, 7
PHZ_C_LINES, PHZ_C_ITEMS, PHZ_C_PAIR_EVENTS, PHZ_C_EDGES, PHZ_C_FAR_LINES, PHZ_C_DIRTY_LISTS = 0, 1, 2, 3, 4, 5


class PhzError(RuntimeError):
    def __init__(self, status, msg=""):
        super().__init__("libphz status %d: %s" % (status, msg))
        self.status = status


class phz_reads(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_ops", C.c_int64), ("n_seq_bytes", C.c_int64),
                ("pos", C.c_void_p), ("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
                ("seq_off", C.c_void_p), ("seq2", C.c_void_p), ("qual", C.c_void_p), ("bq", C.c_void_p)]


class phz_variants(C.Structure):
    _fields_ = [("n", C.c_int64), ("pos", C.c_void_p), ("ref_len", C.c_void_p)]


class phz_variants_general(C.Structure):
    _fields_ = [("n", C.c_int64), ("pos", C.c_void_p), ("ref_len", C.c_void_p), ("allele_off", C.c_void_p),
                ("allele_bytes", C.c_void_p), ("n_allele_bytes", C.c_int64)]


class phz_calls(C.Structure):
    _fields_ = [("cap", C.c_int64), ("read_idx", C.c_void_p), ("var_idx", C.c_void_p), ("code", C.c_void_p),
                ("aux0", C.c_void_p), ("aux1", C.c_void_p)]


class phz_lines(C.Structure):
    _fields_ = [("n_calls", C.c_int64), ("read_idx", C.c_void_p), ("var_idx", C.c_void_p), ("code", C.c_void_p),
                ("n_reads", C.c_int64), ("read_qid", C.c_void_p), ("read_as", C.c_void_p), ("read_has_as", C.c_void_p),
                ("as_cutoff", C.c_double), ("use_cutoff", C.c_int32), ("bam_index", C.c_int32),
                ("var_base", C.c_int64), ("qid_base", C.c_int64), ("read_as16", C.c_void_p), ("as_cutoff_dev", C.c_void_p)]


class phz_tally_sizes(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("n_lines", "n_kept", "n_edges", "n_read_list", "n_items", "pair_events", "noise_match",
                                           "noise_mismatch")]


class phz_tally_out(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("var_count", "var_first", "var_distinct", "var_rank", "line_cls", "edge_a", "edge_b", "edge_cells",
                                          "edge_linked", "edge_cto", "rl_start", "rl_qid", "edge_stats")]


class phz_bam_filters(C.Structure):
    _fields_ = [("min_mapq", C.c_int32), ("flag_required", C.c_int32), ("flag_forbidden", C.c_int32), ("isize_cutoff", C.c_double)]


class phz_bamdev_sizes(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("n_reads", "n_ops", "n_seq_bytes", "n_qname_bytes")]


class phz_dev_shard(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("pos", "cigar_off", "cigar", "seq_off", "seq2", "qual", "aln_score", "has_as", "qname_off", "qnames")]


class phz_host_shard(C.Structure):
    _fields_ = [("ref_name", C.c_char_p), ("n_reads", C.c_int64), ("n_ops", C.c_int64), ("n_seq_bytes", C.c_int64),
                ("pos", C.c_void_p), ("cigar_off", C.c_void_p), ("cigar", C.c_void_p), ("seq_off", C.c_void_p),
                ("seq2", C.c_void_p), ("qual", C.c_void_p), ("aln_score", C.c_void_p), ("has_as", C.c_void_p),
                ("qname_off", C.c_void_p), ("qnames", C.c_void_p)]


class phz_rows_in(C.Structure):
    _fields_ = [("chrom", C.c_char_p), ("nv", C.c_int32), ("pos", C.c_void_p),
                ("uid_off", C.c_void_p), ("uid", C.c_void_p), ("rsid_off", C.c_void_p), ("rsid", C.c_void_p),
                ("allele_off", C.c_void_p), ("allele", C.c_void_p), ("maf_off", C.c_void_p), ("maf_txt", C.c_void_p),
                ("maf", C.c_void_p), ("is_ref", C.c_void_p), ("phase_idx", C.c_void_p), ("blacklisted", C.c_void_p),
                ("var_count", C.c_void_p), ("var_distinct", C.c_void_p),
                ("rl_start", C.c_void_p), ("rl_qid", C.c_void_p),
                ("n_edges", C.c_int64), ("va", C.c_void_p), ("vb", C.c_void_p), ("ea", C.c_void_p), ("eb", C.c_void_p),
                ("sup", C.c_void_p), ("tot", C.c_void_p), ("cis", C.c_void_p), ("trans", C.c_void_p), ("cfgv", C.c_void_p),
                ("eorder", C.c_void_p), ("pv", C.c_void_p),
                ("ncomp", C.c_int64), ("mem_s", C.c_void_p), ("comp_starts", C.c_void_p), ("comp_ends", C.c_void_p),
                ("comp_order", C.c_void_p), ("e_keep", C.c_void_p), ("eo", C.c_void_p), ("e_starts", C.c_void_p),
                ("e_ends", C.c_void_p),
                ("n_keys", C.c_int64), ("key_bam", C.c_void_p), ("key_g", C.c_void_p),
                ("nb", C.c_int32), ("bam_names", C.POINTER(C.c_char_p)), ("bam_excluded", C.c_void_p),
                ("unique_ids", C.c_int32), ("gw_phase_method", C.c_int32), ("output_read_ids", C.c_int32),
                ("unphased_vars", C.c_int32), ("max_block_size", C.c_int32), ("want_vcf", C.c_int32), ("threads", C.c_int32),
                ("qname_off", C.c_void_p), ("qname", C.c_void_p),
                ("raw", C.c_int32), ("v0", C.c_int64), ("keep", C.c_void_p), ("rank", C.c_void_p), ("label", C.c_void_p),
                ("var_first", C.c_void_p), ("bam_line_lo", C.c_void_p), ("bam_line_hi", C.c_void_p)]


class phz_text_parts(C.Structure):
    _fields_ = [("n", C.c_int64), ("ptr", C.POINTER(C.c_void_p)), ("len", C.POINTER(C.c_int64)), ("bam", C.POINTER(C.c_int32))]


class phz_rows_out(C.Structure):
    _fields_ = [(k, phz_text_parts) for k in ("conn", "hap", "ase", "cfg", "allelic", "single_ase", "single_hap")] + \
               [(k, C.c_int64) for k in ("allelic_rows", "n_blocks", "phased", "n_blk_vars")] + \
               [(k, C.c_void_p) for k in ("blk_size", "blk_var", "blk_hap", "blk_cor", "blk_stat", "blk_stat_int", "blk_maxmaf")] + \
               [("owner", C.c_void_p)]


PHZ_PAIR_SLOTS = 16384
PHZ_TXT_NAMES = ("conn", "hap", "ase", "cfg", "allelic", "single_ase", "single_hap")     # PHZ_TXT_* order
PHZ_TXT_COUNT = 7


class phz_rowsdev_tables(C.Structure):
    _fields_ = [("nv", C.c_int64), ("n_chroms", C.c_int32), ("chrom_v0", C.c_void_p), ("chrom_name_off", C.c_void_p), ("chrom_names", C.c_void_p),
                ("pos", C.c_void_p), ("uid_off", C.c_void_p), ("uid", C.c_void_p), ("rsid_off", C.c_void_p), ("rsid", C.c_void_p),
                ("allele_off", C.c_void_p), ("allele", C.c_void_p), ("maf_off", C.c_void_p), ("maf_txt", C.c_void_p), ("maf", C.c_void_p),
                ("is_ref", C.c_void_p), ("phase_idx", C.c_void_p), ("blacklisted", C.c_void_p)]


class phz_rowsdev_opts(C.Structure):
    _fields_ = [("n_bams", C.c_int32), ("bam_name_off", C.c_void_p), ("bam_names", C.c_void_p), ("bam_excluded", C.c_void_p),
                ("n_shards", C.c_int32), ("shard_line_lo", C.c_void_p), ("shard_line_hi", C.c_void_p), ("shard_bam", C.c_void_p),
                ("unique_ids", C.c_int32), ("gw_phase_method", C.c_int32), ("output_read_ids", C.c_int32), ("unphased_vars", C.c_int32),
                ("max_block_size", C.c_int32), ("want_vcf", C.c_int32), ("cc_threshold", C.c_double),
                ("qname_off", C.c_void_p), ("qname", C.c_void_p), ("qname_base", C.c_void_p), ("host_text", C.c_void_p), ("host_text_cap", C.c_int64)]


class phz_rowsdev_result(C.Structure):
    _fields_ = [("bytes", C.c_int64 * PHZ_TXT_COUNT), ("seg_off", C.POINTER(C.c_int64) * PHZ_TXT_COUNT), ("chrom_blocks", C.POINTER(C.c_int64)),
                ("chrom_blk_vars", C.POINTER(C.c_int64)), ("chrom_first_bam", C.POINTER(C.c_int32))] + \
               [(k, C.c_int64) for k in ("n_blocks", "n_blk_vars", "phased", "dropped", "allelic_rows", "n_components", "n_linked", "n_complex",
                                          "n_exceptions", "n_big_segments")] + [("gpu_ms", C.c_double), ("host_off", C.c_int64 * PHZ_TXT_COUNT)]


class phz_hc_arrays(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_vars", C.c_int64), ("n_lab_a", C.c_int64), ("n_lab_b", C.c_int64),
                ("n_contigs", C.c_int32), ("n_bams", C.c_int32), ("has_maf", C.c_int32)] + \
               [(k, C.c_void_p) for k in ("contig", "start", "stop", "a_count", "b_count", "total", "bam", "phase", "gw_stat", "maf",
                                          "var_off", "var_pos", "var_id_off", "var_id_len", "lab_off_a", "lab_off_b",
                                          "lab_pos_a", "lab_prev_a", "lab_pos_b", "lab_prev_b", "names", "names_off")]


class phz_gene_work(C.Structure):
    _fields_ = [("n_items", C.c_int64), ("item_lo", C.c_void_p), ("item_n", C.c_void_p), ("item_run", C.c_void_p),
                ("item_pair", C.c_void_p), ("item_hap", C.c_void_p), ("n_pairs", C.c_int64), ("pair_begin", C.c_void_p),
                ("pair_end", C.c_void_p), ("n_lab_a", C.c_int64), ("n_lab_b", C.c_int64), ("lab_pos_a", C.c_void_p),
                ("lab_prev_a", C.c_void_p), ("lab_pos_b", C.c_void_p), ("lab_prev_b", C.c_void_p)]


class phz_vcf_opts(C.Structure):
    _fields_ = [("sample_column", C.c_int32), ("chrom_of_interest", C.c_char_p), ("pass_only", C.c_int32), ("include_indels", C.c_int32),
                ("chr_prefix", C.c_char_p), ("id_separator", C.c_char_p), ("gw_phase_method", C.c_int32), ("gw_af_field", C.c_char_p),
                ("n_contig_ban", C.c_int32), ("contig_ban", C.POINTER(C.c_char_p)), ("threads", C.c_int32), ("grep_hom", C.c_int32),
                ("n_drop", C.c_int64), ("drop_chrom", C.POINTER(C.c_char_p)), ("drop_start", C.c_void_p), ("drop_end", C.c_void_p),
                ("n_mark", C.c_int64), ("mark_chrom", C.POINTER(C.c_char_p)), ("mark_start", C.c_void_p), ("mark_end", C.c_void_p)]


class phz_vcf_table(C.Structure):
    _fields_ = [("name", C.c_char_p), ("n", C.c_int64), ("pos", C.c_void_p), ("ref_len", C.c_void_p), ("a0", C.c_void_p), ("a1", C.c_void_p),
                ("is_ref", C.c_void_p), ("phase_idx", C.c_void_p), ("maf", C.c_void_p), ("pool", C.c_void_p * 11), ("pool_len", C.c_int64 * 11),
                ("blacklisted", C.c_void_p)]


class phz_pyorder_in(C.Structure):
    _fields_ = [("n_chroms", C.c_int32), ("n_bams", C.c_int32), ("chrom_names", C.POINTER(C.c_char_p)), ("bam_names", C.POINTER(C.c_char_p)),
                ("nv", C.c_void_p), ("pos", C.POINTER(C.c_void_p)),
                ("uid", C.POINTER(C.c_void_p)), ("uid_off", C.POINTER(C.c_void_p)), ("allele2", C.POINTER(C.c_void_p)), ("allele2_off", C.POINTER(C.c_void_p)),
                ("rsid", C.POINTER(C.c_void_p)), ("rsid_off", C.POINTER(C.c_void_p)),
                ("nq", C.c_void_p), ("qname", C.POINTER(C.c_void_p)), ("qname_off", C.POINTER(C.c_void_p)),
                ("line_qid", C.POINTER(C.c_void_p)), ("line_var", C.POINTER(C.c_void_p)), ("line_cls", C.POINTER(C.c_void_p)), ("n_lines", C.c_void_p),
                ("bam_excluded", C.c_void_p), ("blacklisted", C.POINTER(C.c_void_p)),
                ("n_blocks", C.c_int64), ("blk_chrom", C.c_void_p), ("blk_off", C.c_void_p), ("blk_var", C.c_void_p),
                ("output_read_ids", C.c_int32), ("unphased_vars", C.c_int32), ("unique_ids", C.c_int32)]


class phz_vcfout_chrom(C.Structure):
    _fields_ = [("uid", C.c_void_p), ("uid_len", C.c_int64), ("rsid", C.c_void_p), ("rsid_len", C.c_int64), ("alleles", C.c_void_p),
                ("alleles_len", C.c_int64), ("maf_str", C.c_void_p), ("maf_str_len", C.c_int64), ("n_blocks", C.c_int64),
                ("n_blk_vars", C.c_int64), ("first_block_index", C.c_int64), ("blk_size", C.c_void_p), ("blk_var", C.c_void_p),
                ("blk_maxmaf", C.c_void_p), ("blk_hap", C.c_void_p), ("blk_stat_int", C.c_void_p), ("blk_cor", C.c_void_p),
                ("blk_stat", C.c_void_p)]


class phz_read_batch(C.Structure):
    _fields_ = [("n", C.c_int64), ("ref_id", C.c_int32), ("L", C.c_int32), ("pos", C.c_void_p), ("flag", C.c_void_p), ("mapq", C.c_void_p),
                ("tlen", C.c_void_p), ("aln_score", C.c_void_p), ("qid", C.c_void_p), ("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
                ("seq", C.c_void_p), ("qual", C.c_void_p), ("qname_prefix", C.c_char_p)]


class phz_gene_rows_in(C.Structure):
    _fields_ = [("n_features", C.c_int64), ("feat_chr", C.c_void_p), ("feat_chr_len", C.c_int64), ("feat_name", C.c_void_p),
                ("feat_name_len", C.c_int64), ("feat_start", C.c_void_p), ("feat_stop", C.c_void_p), ("n_bam_order", C.c_int32),
                ("bam_order", C.c_void_p), ("bam_names", C.c_void_p), ("bam_names_len", C.c_int64), ("A", C.c_void_p), ("B", C.c_void_p),
                ("UA", C.c_void_p), ("UB", C.c_void_p), ("pv_lo", C.c_void_p), ("pv_hi", C.c_void_p), ("pv_sorted", C.c_void_p),
                ("best_lo", C.c_void_p), ("best_hi", C.c_void_p), ("u_var", C.c_void_p), ("text", C.c_void_p), ("var_id_off", C.c_void_p),
                ("var_id_len", C.c_void_p), ("min_cov", C.c_int64), ("threads", C.c_int32)]


PHZ_AS_BINS = 65536

# every symbol include/phz.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "phz_version": (C.c_int, []),
    "phz_strerror": (C.c_char_p, [C.c_int]),
    "phz_last_error": (C.c_char_p, [C.c_void_p]),
    "phz_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "phz_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "phz_ctx_destroy": (C.c_int, [C.c_void_p]),
    "phz_ctx_sync": (C.c_int, [C.c_void_p]),
    "phz_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "phz_map_reads": (C.c_int, [C.c_void_p, C.POINTER(phz_reads), C.POINTER(phz_variants), C.c_int,
                                C.POINTER(phz_calls), C.POINTER(C.c_int64), C.c_int]),
    "phz_map_reads_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(phz_reads), C.POINTER(phz_variants), C.c_int,
                                      C.POINTER(phz_calls), C.POINTER(C.c_int64)]),
    "phz_as_histogram": (C.c_int, [C.c_void_p, C.POINTER(phz_lines), C.c_void_p, C.c_int]),
    "phz_as_histogram_batch": (C.c_int, [C.c_void_p, C.POINTER(phz_lines), C.c_int, C.c_void_p]),
    "phz_as_histogram_sparse": (C.c_int, [C.c_void_p, C.POINTER(phz_lines), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "phz_pyorder_replay": (C.c_int, [C.POINTER(phz_pyorder_in), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "phz_pyorder_text": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "phz_pyorder_error": (C.c_char_p, [C.c_void_p]),
    "phz_pyorder_free": (None, [C.c_void_p]),
    "phz_py_str_hash": (C.c_int64, [C.c_char_p, C.c_int64]),
    "phz_py_set_order": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "phz_as_plane": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "phz_as_cutoff": (C.c_int, [C.c_void_p, C.POINTER(phz_lines), C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "phz_as_cutoff_enqueue": (C.c_int, [C.c_void_p, C.POINTER(phz_lines), C.c_int, C.c_double, C.c_void_p]),
    "phz_tally": (C.c_int, [C.c_void_p, C.POINTER(phz_lines), C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                            C.POINTER(phz_tally_sizes), C.c_int]),
    "phz_tally_pairs": (C.c_int, [C.c_void_p, C.POINTER(phz_lines), C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                  C.POINTER(phz_tally_sizes), C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "phz_tally_fetch": (C.c_int, [C.c_void_p, C.POINTER(phz_tally_out), C.c_int]),
    "phz_hap_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    "phz_load_variants": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(phz_variants)]),
    "phz_components": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "phz_bgzf_inflate_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int)]),
    "phz_bgzf_crc_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int)]),
    "phz_bamdev_open": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(phz_bam_filters), C.POINTER(C.c_void_p)]),
    "phz_bamdev_close": (C.c_int, [C.c_void_p]),
    "phz_bamdev_n_ref": (C.c_int, [C.c_void_p]),
    "phz_bamdev_ref_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "phz_bamdev_ref_length": (C.c_int64, [C.c_void_p, C.c_int]),
    "phz_bamdev_sizes_of": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(phz_bamdev_sizes)]),
    "phz_bamdev_pack": (C.c_int, [C.c_void_p, C.POINTER(phz_dev_shard), C.c_int]),
    "phz_intern_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_int64)]),
    "phz_names_append_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_int64)]),
    "phz_bam_open": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "phz_bam_open_refs": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "phz_bam_close": (C.c_int, [C.c_void_p]),
    "phz_bam_n_ref": (C.c_int, [C.c_void_p]),
    "phz_bam_ref_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "phz_bam_ref_length": (C.c_int64, [C.c_void_p, C.c_int]),
    "phz_bam_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int)]),
    "phz_bam_shard": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(phz_host_shard)]),
    "phz_sam_parse": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, C.c_int, C.POINTER(C.c_void_p)]),
    "phz_sam_error": (C.c_char_p, [C.c_void_p]),
    "phz_sam_free": (None, [C.c_void_p]),
    "phz_sam_n_records": (C.c_int64, [C.c_void_p]),
    "phz_sam_stream_order": (C.c_int, [C.c_void_p]),
    "phz_sam_n_contigs": (C.c_int, [C.c_void_p]),
    "phz_sam_contig": (C.c_char_p, [C.c_void_p, C.c_int]),
    "phz_sam_n_shards": (C.c_int, [C.c_void_p]),
    "phz_sam_shard": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(phz_host_shard)]),
    "phz_sam_calls_tsv": (C.c_int, [C.c_void_p, C.c_int, C.c_int64] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_int64)]),
    "phz_bgzf_read": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "phz_buf_free": (None, [C.c_void_p]),
    "phz_bgzf_write": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int64, C.c_int, C.c_int]),
    "phz_bam_write": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.POINTER(phz_read_batch), C.c_int, C.c_int]),
    "phz_tabix_build": (C.c_int, [C.c_char_p, C.c_int, C.c_int]),
    "phz_bgzf_write_indexed": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "phz_interner_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "phz_interner_destroy": (C.c_int, [C.c_void_p]),
    "phz_interner_size": (C.c_int64, [C.c_void_p]),
    "phz_intern": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "phz_interner_names": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "phz_map_reads_general": (C.c_int, [C.c_void_p, C.POINTER(phz_reads), C.POINTER(phz_variants_general), C.c_int,
                                        C.POINTER(phz_calls), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_int64,
                                        C.POINTER(C.c_int64), C.c_int]),
    "phz_rows_format": (C.c_int, [C.POINTER(phz_rows_in), C.POINTER(phz_rows_out)]),
    "phz_rows_format_multi": (C.c_int, [C.POINTER(phz_rows_in), C.c_int, C.POINTER(phz_rows_out), C.c_int]),
    "phz_rows_free": (None, [C.POINTER(phz_rows_out)]),
    "phz_rowsdev_create": (C.c_int, [C.c_void_p, C.POINTER(phz_rowsdev_tables), C.POINTER(C.c_void_p)]),
    "phz_rowsdev_destroy": (None, [C.c_void_p]),
    "phz_rowsdev_set_pair_slots": (C.c_int, [C.c_void_p, C.c_int64]),
    "phz_rowsdev_pair_slots": (C.c_int64, [C.c_void_p]),
    "phz_rowsdev_pair_keys": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "phz_pair_slot_text": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "phz_pair_slots_used": (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "phz_rowsdev_set_shards": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "phz_rowsdev_run": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(phz_rowsdev_opts), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(phz_rowsdev_result)]),
    "phz_rowsdev_fetch_text": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    "phz_rowsdev_text_ptr": (C.c_void_p, [C.c_void_p, C.c_int]),
    "phz_rowsdev_fetch_blocks": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_void_p] * 7),
    "phz_phase_components": (C.c_int, [C.c_void_p, C.c_int64] + [C.c_void_p] * 5 + [C.c_int32] + [C.c_void_p] * 3),
    "phz_tally_import": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.POINTER(phz_tally_sizes), C.POINTER(phz_tally_out), C.c_void_p, C.c_int]),
    "phz_phase_block": (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.POINTER(C.c_int32)]),
    "phz_hc_parse": (C.c_int, [C.c_void_p, C.c_int64, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "phz_hc_view": (C.c_int, [C.c_void_p, C.POINTER(phz_hc_arrays)]),
    "phz_hc_error": (C.c_char_p, [C.c_void_p]),
    "phz_hc_free": (None, [C.c_void_p]),
    "phz_gene_rows": (C.c_int, [C.POINTER(phz_gene_rows_in), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "phz_gene_counts": (C.c_int, [C.c_void_p, C.POINTER(phz_gene_work), C.c_void_p, C.c_int]),
    "phz_vcf_parse": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(phz_vcf_opts), C.POINTER(C.c_void_p)]),
    "phz_vcf_summary": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                  C.POINTER(C.c_int64)]),
    "phz_vcf_chrom": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(phz_vcf_table)]),
    "phz_vcf_error": (C.c_char_p, [C.c_void_p]),
    "phz_vcf_free": (None, [C.c_void_p]),
    "phz_vcf_phase_text": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_char_p, C.c_char_p, C.c_int32, C.c_double,
                                     C.POINTER(phz_vcfout_chrom), C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "phz_microbench": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "phz_selftest_sort": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "phz_membench": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "phz_get_timing": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double),
                                 C.POINTER(C.c_int64)]),
    "phz_reset_timing": (C.c_int, [C.c_void_p]),
    "phz_get_counter": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]),
}

_lib: Optional[C.CDLL] = None


def hip_sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 into phaser_amd/libphz.so (in-tree).  One object per source under
    csrc/build/ (compiled concurrently, rebuilt only when the source or a header changed), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = hip_sources()
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(REPO, "include", "phz.h")]
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    bdir = os.path.join(CSRC, "build")
    os.makedirs(bdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(REPO, "include"), "-I" + CSRC]
    jobs = []
    objs = []
    for src in srcs:
        obj = os.path.join(bdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append(["hipcc"] + flags + ["-c", src, "-o", obj])
    if not jobs and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(o) for o in objs):
        return LIB_PATH

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    run(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", LIB_PATH, "-lz", "-lpthread"])
    return LIB_PATH


def load() -> C.CDLL:
    """Load libphz.so and bind every declared symbol; raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  Importing torch FIRST makes the
    # dynamic loader resolve libphz.so's libamdhip64.so.N against that already-loaded copy, so the process
    # holds exactly one HIP runtime (two runtimes => the second one sees no device).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        # not a fallback: compile the same HIP sources in-tree when the toolchain is present, otherwise fail loudly
        import shutil
        if shutil.which("hipcc"):
            build()
        if not os.path.exists(LIB_PATH):
            raise PhzError(PHZ_E_HIP, "phaser_amd/libphz.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                                      "there is no CPU fallback for the product path")
    alt = os.environ.get("PHZ_LIB_PATH")          # debugging aid: a sanitizer build of the host-only translation units
    lib = C.CDLL(alt or LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        if alt and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)        # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class Context:
    """One phz_ctx (one HIP stream on one GPU).  Raises loudly when no GPU is usable."""

    def __init__(self, device: int = 0):
        self.lib = load()
        n = C.c_int(0)
        self.lib.phz_device_count(C.byref(n))
        if n.value <= 0:
            raise PhzError(PHZ_E_HIP, "no HIP device visible; the phaser_amd hot path needs an MI355X")
        h = C.c_void_p()
        st = self.lib.phz_ctx_create(device, C.byref(h))
        if st != PHZ_OK:
            raise PhzError(st, "phz_ctx_create failed")
        self.h = h
        self.device = device

    def check(self, st, allow=()):
        if st != PHZ_OK and st not in allow:
            raise PhzError(st, (self.lib.phz_last_error(self.h) or b"").decode() or
                           self.lib.phz_strerror(st).decode())
        return st

    def timing(self, slot=PHZ_T_MAP):
        last = C.c_float(); tot = C.c_double(); n = C.c_int64()
        self.check(self.lib.phz_get_timing(self.h, slot, C.byref(last), C.byref(tot), C.byref(n)))
        return last.value, tot.value, n.value

    def counter(self, slot):
        v = C.c_int64()
        self.check(self.lib.phz_get_counter(self.h, slot, C.byref(v)))
        return v.value

    def reset_timing(self):
        self.check(self.lib.phz_reset_timing(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.phz_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OwnedArray(np.ndarray):
    """numpy view of native memory that keeps the owning handle alive (zero-copy hand-over of library buffers)."""

    def __new__(cls, arr, owner):
        obj = arr.view(cls)
        obj._owner = owner
        return obj

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, "_owner", None)


_view_from_memory = C.pythonapi.PyMemoryView_FromMemory
_view_from_memory.restype = C.py_object
_view_from_memory.argtypes = [C.c_void_p, C.c_ssize_t, C.c_int]


_DTYPE_OF: dict = {}


def native_view(ptr, count: int, ctype, owner):
    """-> numpy array over `count` items of `ctype` at `ptr`, alive as long as the array (or anything built on it) is."""
    dt = _DTYPE_OF.get(ctype)
    if dt is None:
        dt = _DTYPE_OF[ctype] = np.dtype(ctype)      # numpy derives a dtype from a ctypes type in ~3 us: once per type, not per view
    if count == 0 or not ptr:
        return np.zeros(0, dtype=dt)
    # PyMemoryView_FromMemory + frombuffer: ~1 us; np.ctypeslib.as_array builds a ctypes array type per distinct length (~20 us)
    a = np.frombuffer(_view_from_memory(ptr if isinstance(ptr, int) else C.cast(ptr, C.c_void_p).value, count * dt.itemsize, 0x200), dtype=dt)
    return OwnedArray(a, owner)
