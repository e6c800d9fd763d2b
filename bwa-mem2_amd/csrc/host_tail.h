// host_tail.h -- pieces of the host tail shared between finish_regs.cpp and sam_tail.cpp
#pragma once
#include <stdint.h>
#include "../../include/bm2.h"

// mem_sort_dedup_patch (bwamem.cpp:292-353); query == NULL: no hit merging (mem_patch_reg returns 0, bwamem.cpp:181), the
// form mem_matesw calls it in (bwamem_pair.cpp:274)
int bm2h_sort_dedup_patch(const bm2_opt *opt, int64_t l_pac, const uint8_t *ref_string, const uint8_t *query, int n, bm2_alnreg_t *a);
