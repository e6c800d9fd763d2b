// integration/bm2_process_seqs.cpp -- the reference-side binding of libbm2, compiled INTO bwa-mem2 (oracle/Makefile, target `bm2`).
//
// bwa-mem2 has no plugin interface; its coarsest seam is mem_process_seqs (bwamem.cpp:1338-1390, declared in bwamem.h:331): one
// chunk of reads in, bseq1_t::sam strings out, called from step 1 of the kt_pipeline (fastmap.cpp:263-296) and written out by step 2
// (fastmap.cpp:303-322).  This file IS that function, with the reference's own signature, for a build in which bwamem.cpp is
// compiled with -Dmem_process_seqs=mem_process_seqs_reference: every caller in fastmap.cpp then lands here, and
// `bwa-mem2.bm2 mem ...` -- the reference's CLI, option parser, FASTQ reader (kseq), chunking, @HD/@SQ/@PG header and writer --
// aligns on the GPU.  Nothing of the reference is modified or copied; its headers are included from where they lie.
//
// What happens to a chunk here: bases -> 2-bit codes (bwamem.cpp:992-1000) on opt->n_threads host threads, then bm2_chunk_hits_sharded --
// the chunk cut at multiples of 512 reads over every visible GPU (BM2_DEVICES=n caps the number, BM2_DEVICE picks the first), each part
// through bm2_batch_upload / run (mem_kernel1_core + mem_kernel2_core up to :1152) / finish (mem_sort_dedup_patch) on its GPU's replica,
// the hits gathered in read order -- then ONE bm2_sam_pe_dev_multi / bm2_sam_se_dev_multi over the whole chunk (mem_pestat is chunk-wide; the rescue
// and CIGAR alignments run as device batches cut over the same GPUs, results back in task order), so the text does not depend on the number of GPUs.  The index is NOT loaded twice: the
// descriptor points at the arrays FMI_search::load_index and main_mem already hold (FMI_search keeps them private; a maintainer
// would add accessors -- this file opens the class with the preprocessor instead, to leave the reference's sources alone).
#include <atomic>
#include <mutex>
#include <thread>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define private public
#include "FMI_search.h"
#undef private
#include "bwamem.h"
#include "bntseq.h"
#include "bwa.h"
#include "../include/bm2.h"

extern char bwa_rg_id[256];                         // bwa.cpp:40
extern unsigned char nst_nt4_table[256];            // bntseq.cpp:38-55

namespace {
struct Gpu {
    std::vector<bm2_ctx *> ctx;                      // one per GPU, each with its index replica
    bm2_index_desc desc;
    std::vector<int64_t> ann_offset; std::vector<int32_t> ann_len, ann_is_alt; std::vector<const char *> ann_name, ann_anno;
    std::mutex mu;
};
Gpu g;

void die(const char *what) { fprintf(stderr, "[bm2] %s: %s\n", what, bm2_last_error()); exit(EXIT_FAILURE); }

void attach(const worker_t &w) {                    // once: the loaded index as a bm2_index_desc, replica to the GPU
    const FMI_search *f = w.fmi;
    const bntseq_t *bns = f->idx->bns;
    memset(&g.desc, 0, sizeof g.desc);
    g.desc.ref_len = f->reference_seq_len;
    for (int i = 0; i < 5; i++) g.desc.count[i] = f->count[i] - 1;      // load_index stored count + 1 (FMI_search.cpp:433-436); the library adds it itself
    g.desc.sentinel_index = f->sentinel_index;
    g.desc.cp_occ = f->cp_occ; g.desc.sa_ms_byte = f->sa_ms_byte; g.desc.sa_ls_word = f->sa_ls_word;
    g.desc.ref_string = w.ref_string; g.desc.l_pac = bns->l_pac; g.desc.n_seqs = bns->n_seqs;
    for (int i = 0; i < bns->n_seqs; i++) {
        g.ann_offset.push_back(bns->anns[i].offset); g.ann_len.push_back(bns->anns[i].len); g.ann_is_alt.push_back(bns->anns[i].is_alt);
        g.ann_name.push_back(bns->anns[i].name); g.ann_anno.push_back(bns->anns[i].anno);
    }
    g.desc.ann_offset = g.ann_offset.data(); g.desc.ann_len = g.ann_len.data(); g.desc.ann_is_alt = g.ann_is_alt.data();
    g.desc.ann_name = g.ann_name.data(); g.desc.ann_anno = g.ann_anno.data();
    setenv("BM2_MALLOC_TUNE", "1", 0);                          // this program is the library's host: it opts in to the allocator settings of the SAM tail
    const char *dev = getenv("BM2_DEVICE"), *cap = getenv("BM2_DEVICES");
    const int first = dev ? atoi(dev) : 0, visible = bm2_device_count();
    int n_gpu = visible - first;
    if (cap && atoi(cap) > 0 && atoi(cap) < n_gpu) n_gpu = atoi(cap);
    if (n_gpu < 1) n_gpu = 1;
    g.ctx.assign((size_t)n_gpu, nullptr);
    std::vector<std::thread> up;                                 // the replicas go up side by side
    for (int i = 0; i < n_gpu; i++) up.emplace_back([i, first]() { g.ctx[(size_t)i] = bm2_create(first + i, &g.desc); });
    for (auto &t : up) t.join();
    for (int i = 0; i < n_gpu; i++) if (!g.ctx[(size_t)i]) die("bm2_create");
    fprintf(stderr, "[bm2] index replica on %d GPU(s) (%ld bases); seed -> chain -> extend -> pair runs in libbm2\n", n_gpu, (long)bns->l_pac);
}
}  // namespace

void mem_process_seqs(mem_opt_t *opt, int64_t n_processed, int n, bseq1_t *seqs, const mem_pestat_t *pes0, worker_t &w) {
    // One chunk at a time: the contexts serve one host thread each.  Under fastmap.cpp's kt_pipeline this never waits (klib lets chunk i + 1 into
    // step 1 only when chunk i has left it); a caller that runs step 1 of two chunks side by side is serialised HERE, and is told so once.
    std::unique_lock<std::mutex> lock(g.mu, std::try_to_lock);
    if (!lock.owns_lock()) {
        static std::atomic<bool> told{false};
        if (!told.exchange(true)) fprintf(stderr, "[bm2] mem_process_seqs was entered by two threads at once: the chunks run one after the other (one device stage per process)\n");
        lock.lock();
    }
    if (g.ctx.empty()) attach(w);
    const double t0 = realtime();
    // ---- options: mem_opt_t -> the two structs of include/bm2.h (same names, same meaning)
    bm2_opt o; bm2_opt_init(&o);
    o.a = opt->a; o.b = opt->b; o.o_del = opt->o_del; o.e_del = opt->e_del; o.o_ins = opt->o_ins; o.e_ins = opt->e_ins;
    o.pen_clip5 = opt->pen_clip5; o.pen_clip3 = opt->pen_clip3; o.w = opt->w; o.zdrop = opt->zdrop;
    o.min_seed_len = opt->min_seed_len; o.split_width = opt->split_width; o.max_occ = opt->max_occ; o.max_chain_gap = opt->max_chain_gap;
    o.min_chain_weight = opt->min_chain_weight; o.max_chain_extend = opt->max_chain_extend; o.max_mem_intv = (int64_t)opt->max_mem_intv;
    o.split_factor = opt->split_factor; o.mask_level = opt->mask_level; o.drop_ratio = opt->drop_ratio; o.mask_level_redun = opt->mask_level_redun;
    memcpy(o.mat, opt->mat, 25);
    bm2_sam_opt so; bm2_sam_opt_init(&so);
    so.T = opt->T; so.flag = opt->flag; so.max_XA_hits = opt->max_XA_hits; so.max_XA_hits_alt = opt->max_XA_hits_alt;
    so.XA_drop_ratio = opt->XA_drop_ratio; so.mapQ_coef_len = opt->mapQ_coef_len; so.mapQ_coef_fac = opt->mapQ_coef_fac;
    so.pen_unpaired = opt->pen_unpaired; so.max_ins = opt->max_ins; so.max_matesw = opt->max_matesw; so.n_threads = opt->n_threads;
    so.rg_id = bwa_rg_id[0] ? bwa_rg_id : nullptr;
    // ---- reads: codes (the reference converts seqs[i].seq in place inside mem_kernel1_core; the strings stay untouched here)
    std::vector<int64_t> off((size_t)n + 1); std::vector<int32_t> len((size_t)n + 1);
    int64_t nb = 0;
    for (int i = 0; i < n; i++) { off[(size_t)i] = nb; len[(size_t)i] = seqs[i].l_seq; nb += seqs[i].l_seq; }
    std::vector<uint8_t> enc((size_t)nb + 1);
    std::vector<const char *> name((size_t)n + 1), comment((size_t)n + 1), qual((size_t)n + 1);
    {
        int nt = opt->n_threads > 0 ? opt->n_threads : 1;
        if (nt > n / 4096 + 1) nt = n / 4096 + 1;
        std::atomic<int> next(0);
        auto convert = [&]() {
            for (int lo; (lo = next.fetch_add(4096)) < n;)
                for (int i = lo; i < n && i < lo + 4096; i++) {
                    uint8_t *d = enc.data() + off[(size_t)i];
                    const char *s = seqs[i].seq;
                    for (int k = 0; k < seqs[i].l_seq; k++) d[k] = (unsigned char)s[k] < 4 ? (uint8_t)s[k] : nst_nt4_table[(unsigned char)s[k]];
                    name[(size_t)i] = seqs[i].name; comment[(size_t)i] = seqs[i].comment; qual[(size_t)i] = seqs[i].qual;
                    seqs[i].sam = nullptr;
                }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++) th.emplace_back(convert);
        convert();
        for (auto &t : th) t.join();
    }
    bm2_reads reads = { n, enc.data(), off.data(), len.data() };
    bm2_read_text txt = { name.data(), comment.data(), qual.data() };
    // ---- device: mem_kernel1_core + mem_kernel2_core over the chunk's parts, then the hits worker_sam receives, in read order
    std::vector<int64_t> aln_off((size_t)n + 1);
    int64_t n_aln = 0;
    std::vector<bm2_alnreg_t> aln((size_t)(4 * (int64_t)n + 1024));
    int rc = bm2_chunk_hits_sharded(g.ctx.data(), (int)g.ctx.size(), &reads, &o, aln.data(), (int64_t)aln.size(), aln_off.data(), &n_aln);
    if (rc == BM2_ECAP) { aln.resize((size_t)n_aln + 1); rc = bm2_chunk_hits_sharded(g.ctx.data(), (int)g.ctx.size(), &reads, &o, aln.data(), (int64_t)aln.size(), aln_off.data(), &n_aln); }
    if (rc) die("device stage");
    // ---- pairing / SAM text: mem_pestat + worker_sam
    int64_t cap = 3 * (nb + 200 * (int64_t)n) + (1 << 20), need = 0;
    char *text = nullptr;
    bm2_pestat pin[4];
    if (pes0) for (int d = 0; d < 4; d++) { pin[d].low = pes0[d].low; pin[d].high = pes0[d].high; pin[d].failed = pes0[d].failed; pin[d].pad = 0; pin[d].avg = pes0[d].avg; pin[d].std = pes0[d].std; }
    const bool pe = (opt->flag & MEM_F_PE) != 0;
    for (int attempt = 0;; ++attempt) {
        text = (char *)realloc(text, (size_t)cap + 1);
        if (!text) { fprintf(stderr, "[bm2] out of memory\n"); exit(EXIT_FAILURE); }
        // (the single-end tail reorders the hits in place: should the text not fit -- the capacity above is generous, so hardly ever -- the
        //  second attempt fetches the hits again instead of every chunk paying for a spare copy)
        if (attempt > 0 && !pe && bm2_chunk_hits_sharded(g.ctx.data(), (int)g.ctx.size(), &reads, &o, aln.data(), (int64_t)aln.size(), aln_off.data(), &n_aln)) die("device stage");
        rc = pe ? bm2_sam_pe_dev_multi(g.ctx.data(), (int)g.ctx.size(), &g.desc, &o, &so, &reads, &txt, aln.data(), aln_off.data(), n_processed, pes0 ? pin : nullptr, nullptr, text, cap, &need)
                : bm2_sam_se_dev_multi(g.ctx.data(), (int)g.ctx.size(), &g.desc, &o, &so, &reads, &txt, aln.data(), aln_off.data(), n_processed, text, cap, &need);
        if (rc != BM2_ECAP) break;
        cap = need + 16;
    }
    if (rc) die("SAM tail");
    text[need] = 0;
    // The chunk's records are ONE string, handed over as read 0's: step 2 prints the non-null strings in read order and frees them
    // (fastmap.cpp:307-316), so the output is the same bytes; a caller that wants one string per read has to cut it at the reads' QNAMEs.
    for (int i = 1; i < n; i++) seqs[i].sam = nullptr;
    if (n > 0) seqs[0].sam = text; else free(text);
    fprintf(stderr, "\t[0000][ M::%s] Processed %d reads in %.3f real sec (libbm2)\n", __func__, n, realtime() - t0);
}
