#!/bin/bash
# Build the host-only translation units (parsers, writers, BAM decoder, block phasing) with AddressSanitizer and run the CPU
# tests that exercise them against that build (PHZ_LIB_PATH).  Last run: round 5, after the pread member walk of the BAM plan (140 tests clean, CRC checks of the BGZF members included).
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
g++ -std=c++17 -O1 -g -fsanitize=address -fno-omit-frame-pointer -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$R/include -I$R/phaser_amd/csrc \
    $R/phaser_amd/csrc/phz_vcf.cpp $R/phaser_amd/csrc/phz_vcfout.cpp $R/phaser_amd/csrc/phz_genes.cpp $R/phaser_amd/csrc/phz_rows.cpp \
    $R/phaser_amd/csrc/phz_bam.cpp -o /tmp/libphz_asan.so -lz -lpthread
cd $R
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 PHZ_LIB_PATH=/tmp/libphz_asan.so LD_PRELOAD=$(g++ -print-file-name=libasan.so) \
    python -m pytest tests/test_native_robustness.py tests/test_host_stages.py tests/test_vcf_loader.py tests/test_gene_ae.py tests/test_bamio.py \
    tests/test_tabix.py tests/test_phase_block.py tests/test_expr_matrix.py -x -q -m "not gpu" -p no:cacheprovider
