#!/bin/bash
# builds /tmp/tail_harness (add -pg to CXXEXTRA for gprof); run: BM2_RESCUE_FLAT=1 BM2_CIGAR_FLAT=1 BM2_TAIL_PROF=1 /tmp/tail_harness /tmp/tail_in.bin <threads> <repeats>
R=$(cd $(dirname $0)/../.. && pwd)
/opt/rocm/lib/llvm/bin/clang++ -O3 -g $CXXEXTRA -std=c++17 -I$R/include $R/tools/tail_prof/harness.cpp $R/bwa-mem2_amd/csrc/sam_tail.cpp $R/bwa-mem2_amd/csrc/index_io.cpp \
    $R/bwa-mem2_amd/csrc/fastq_io.cpp -lpthread -o ${1:-/tmp/tail_harness}
