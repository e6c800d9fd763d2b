"""ctypes binding of libbm2.so (the C ABI declared in include/bm2.h).

Host-side plumbing only: numpy arrays in, numpy arrays out.  There is no CPU fallback --
if the library or a HIP device is missing every call raises.
(The package directory is `bwa-mem2_amd/`; put it on sys.path and `import bm2`.)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BM2_LIB") or os.path.join(_HERE, "libbm2.so")      # (BM2_LIB: another build of the same sources, e.g. an A/B variant)

BM2_OK, BM2_ENODEV, BM2_ENOMEM, BM2_EINVAL, BM2_ECAP, BM2_EUNSUP, BM2_EIO = 0, -1, -2, -3, -4, -5, -6

SMEM_DT = np.dtype([("rid", "<u4"), ("m", "<u4"), ("n", "<u4"), ("pad", "<u4"), ("k", "<i8"), ("l", "<i8"), ("s", "<i8")])
SEQPAIR_DT = np.dtype([(n, "<i4") for n in ("idr", "idq", "id", "len1", "len2", "h0", "seqid", "regid",
                                             "score", "tle", "gtle", "qle", "gscore", "max_off")])
DEVCHAIN_DT = np.dtype([("pos", "<i8"), ("seed_off", "<i8"), ("n", "<i4"), ("rid", "<i4"), ("w", "<i4"), ("kept", "<i4"),
                        ("first", "<i4"), ("is_alt", "<i4"), ("read", "<i4"), ("frac_rep", "<f4"), ("rmax0", "<i8"),
                        ("rmax1", "<i8"), ("reg0", "<i4"), ("pad", "<i4")])
DEVSEED_DT = np.dtype([("rbeg", "<i8"), ("qbeg", "<i4"), ("len", "<i4"), ("score", "<i4"), ("aln", "<i4")])
DEVREG_DT = np.dtype([("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("rid", "<i4"), ("score", "<i4"),
                      ("truesc", "<i4"), ("w", "<i4"), ("seedcov", "<i4"), ("seedlen0", "<i4"), ("frac_rep", "<f4"),
                      ("chain", "<i4")])
REG_DT = np.dtype([("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("rid", "<i4"), ("score", "<i4"),
                   ("truesc", "<i4"), ("w", "<i4"), ("seedcov", "<i4"), ("seedlen0", "<i4"), ("frac_rep", "<f4"),
                   ("pad", "<i4")])
ALNREG_DT = np.dtype([("rb", "<i8"), ("re", "<i8")] + [(n, "<i4") for n in ("qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub",
                      "sub_n", "w", "seedcov", "secondary", "secondary_all", "seedlen0", "n_comp", "is_alt")] +
                     [("frac_rep", "<f4"), ("pad", "<i4"), ("hash", "<u8")])
assert SMEM_DT.itemsize == 40 and SEQPAIR_DT.itemsize == 56 and REG_DT.itemsize == 56 and ALNREG_DT.itemsize == 96


class IndexDesc(C.Structure):
    _fields_ = [("ref_len", C.c_int64), ("count", C.c_int64 * 5), ("sentinel_index", C.c_int64),
                ("cp_occ", C.c_void_p), ("sa_ms_byte", C.c_void_p), ("sa_ls_word", C.c_void_p),
                ("ref_string", C.c_void_p), ("l_pac", C.c_int64), ("n_seqs", C.c_int32),
                ("ann_offset", C.c_void_p), ("ann_len", C.c_void_p), ("ann_is_alt", C.c_void_p),
                ("ann_name", C.c_void_p), ("ann_anno", C.c_void_p)]


class Opt(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "pen_clip5", "pen_clip3",
                                         "w", "zdrop", "min_seed_len", "split_width", "max_occ", "max_chain_gap",
                                         "min_chain_weight", "max_chain_extend")] + \
               [("max_mem_intv", C.c_int64), ("split_factor", C.c_float), ("mask_level", C.c_float),
                ("drop_ratio", C.c_float), ("mask_level_redun", C.c_float), ("mat", C.c_int8 * 25),
                ("pad", C.c_int8 * 3)]


class SamOpt(C.Structure):
    _fields_ = [("T", C.c_int32), ("flag", C.c_int32), ("max_XA_hits", C.c_int32), ("max_XA_hits_alt", C.c_int32),
                ("XA_drop_ratio", C.c_float), ("mapQ_coef_len", C.c_float), ("mapQ_coef_fac", C.c_int32),
                ("pen_unpaired", C.c_int32), ("max_ins", C.c_int32), ("max_matesw", C.c_int32), ("n_threads", C.c_int32),
                ("rescue_inline", C.c_int32), ("rg_id", C.c_char_p)]


class Fastq(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("pad", C.c_int32), ("n_bases", C.c_int64), ("enc", C.POINTER(C.c_uint8)),
                ("off", C.POINTER(C.c_int64)), ("len", C.POINTER(C.c_int32)), ("name", C.POINTER(C.c_char_p)),
                ("comment", C.POINTER(C.c_char_p)), ("qual", C.POINTER(C.c_char_p)), ("arena", C.c_void_p)]


class PeStat(C.Structure):
    _fields_ = [("low", C.c_int32), ("high", C.c_int32), ("failed", C.c_int32), ("pad", C.c_int32), ("avg", C.c_double), ("std", C.c_double)]


class ReadText(C.Structure):
    _fields_ = [("name", C.POINTER(C.c_char_p)), ("comment", C.POINTER(C.c_char_p)), ("qual", C.POINTER(C.c_char_p))]


class SwParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("o_del", "e_del", "o_ins", "e_ins", "zdrop", "end_bonus", "w_match",
                                         "w_mismatch")] + [("mat", C.c_int8 * 25), ("pad", C.c_int8 * 3)]


class Reads(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("enc", C.c_void_p), ("off", C.c_void_p), ("len", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_reads", "n_bases", "n_smem", "n_sa", "n_chain", "n_reg_raw", "n_reg",
                                         "n_ext", "n_lf", "n_sw_cells", "n_sw_tasks")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


EXPORTS = ["bm2_index_load", "bm2_index_free", "bm2_opt_init", "bm2_opt_fill_scmat", "bm2_create", "bm2_create_shared", "bm2_destroy",
           "bm2_last_error", "bm2_device_count", "bm2_set_stream_priority", "bm2_host_cpus", "bm2_host_alloc", "bm2_host_free", "bm2_bsw", "bm2_bsw_upload", "bm2_bsw_run", "bm2_bsw_download", "bm2_smem", "bm2_sal", "bm2_seed_chain_extend",
           "bm2_batch_upload", "bm2_batch_run", "bm2_batch_stats", "bm2_batch_download", "bm2_batch_kernel_ms", "bm2_batch_parts",
           "bm2_batch_fetch", "bm2_batch_finish", "bm2_batch_download_alnregs", "bm2_finish_regs_dev", "bm2_chunk_hits_sharded", "bm2_index_build", "bm2_sam_opt_init", "bm2_sam_se", "bm2_sam_pe", "bm2_fastq_parse", "bm2_fastq_parse_mt", "bm2_fastq_free", "bm2_ksw_align2", "bm2_ksw_align2_dev", "bm2_sam_pe_dev", "bm2_sam_se_dev", "bm2_sam_pe_dev_multi", "bm2_sam_se_dev_multi", "bm2_sam_cigar_stats", "bm2_gen_cigar", "bm2_gen_cigar_dev", "bm2_sam_header", "bm2_sam_rescue_stats"]

_lib = None


def host_cpus():
    """CPUs this process can really use (hardware threads capped by the cgroup CPU-time quota): bm2_host_cpus"""
    return int(lib().bm2_host_cpus())


class Pinned:
    """A page-locked host array (bm2_host_alloc): .a is a numpy view; close() frees it."""

    def __init__(self, n, dtype=np.uint8):
        L = lib()
        L.bm2_host_alloc.restype = C.c_void_p
        L.bm2_host_alloc.argtypes = [C.c_int64]
        L.bm2_host_free.argtypes = [C.c_void_p]
        L.bm2_host_free.restype = None
        dt = np.dtype(dtype)
        self.nbytes = max(int(n), 1) * dt.itemsize
        self.p = L.bm2_host_alloc(self.nbytes)
        if not self.p:
            raise MemoryError("bm2_host_alloc(%d): %s" % (self.nbytes, L.bm2_last_error()))
        self.a = np.frombuffer((C.c_uint8 * self.nbytes).from_address(self.p), dtype=dt)

    def close(self):
        if self.p:
            self.a = None
            lib().bm2_host_free(C.c_void_p(self.p))
            self.p = None


def build():
    """Compile libbm2.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libbm2.so is not built (run __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.bm2_last_error.restype = C.c_char_p
        L.bm2_create.restype = C.c_void_p
        L.bm2_create.argtypes = [C.c_int, C.c_void_p]
        L.bm2_destroy.argtypes = [C.c_void_p]
        L.bm2_index_load.argtypes = [C.c_char_p, C.POINTER(IndexDesc)]
        L.bm2_index_free.argtypes = [C.POINTER(IndexDesc)]
        L.bm2_opt_init.argtypes = [C.POINTER(Opt)]
        L.bm2_opt_fill_scmat.argtypes = [C.POINTER(Opt)]
        L.bm2_bsw.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                              C.POINTER(SwParams)]
        L.bm2_smem.argtypes = [C.c_void_p, C.POINTER(Reads), C.POINTER(Opt), C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.bm2_sal.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.bm2_seed_chain_extend.argtypes = [C.c_void_p, C.POINTER(Reads), C.POINTER(Opt), C.c_void_p, C.c_int64,
                                            C.c_void_p, C.POINTER(C.c_int64), C.POINTER(Stats)]
        L.bm2_batch_upload.argtypes = [C.c_void_p, C.POINTER(Reads)]
        L.bm2_batch_run.argtypes = [C.c_void_p, C.POINTER(Opt)]
        L.bm2_batch_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.bm2_batch_download.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]
        L.bm2_batch_kernel_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_void_p]
        L.bm2_batch_parts.argtypes = [C.c_void_p]
        L.bm2_batch_fetch.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.bm2_index_build.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.bm2_sam_opt_init.argtypes = [C.POINTER(SamOpt)]
        L.bm2_sam_opt_init.restype = None
        L.bm2_sam_se.argtypes = [C.POINTER(IndexDesc), C.POINTER(Opt), C.POINTER(SamOpt), C.POINTER(Reads), C.POINTER(ReadText),
                                 C.c_void_p, C.c_void_p, C.c_int64, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]
        L.bm2_sam_pe.argtypes = [C.POINTER(IndexDesc), C.POINTER(Opt), C.POINTER(SamOpt), C.POINTER(Reads), C.POINTER(ReadText),
                                 C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]
        _lib = L
    return _lib


class Bm2Error(RuntimeError):
    pass


def _chk(rc, what):
    if rc != BM2_OK:
        raise Bm2Error("%s failed (%d): %s" % (what, rc, lib().bm2_last_error().decode()))


def default_opt(**kw):
    o = Opt()
    lib().bm2_opt_init(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    lib().bm2_opt_fill_scmat(C.byref(o))
    return o


def sw_params(opt, end_bonus):
    p = SwParams()
    p.o_del, p.e_del, p.o_ins, p.e_ins, p.zdrop = opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, opt.zdrop
    p.end_bonus, p.w_match, p.w_mismatch = end_bonus, opt.a, opt.b
    for i in range(25):
        p.mat[i] = opt.mat[i]
    return p


def _reads_struct(enc, off, ln):
    enc = np.ascontiguousarray(enc, np.uint8)
    off = np.ascontiguousarray(off, np.int64)
    ln = np.ascontiguousarray(ln, np.int32)
    r = Reads(len(ln), enc.ctypes.data, off.ctypes.data, ln.ctypes.data)
    return r, (enc, off, ln)


class Index:
    """The host arrays of one index (bm2_index_load), loaded once and shared by the host-side entry points: every function below
    that takes `index_prefix` also accepts an Index or a Context (then nothing is read from disk again)."""

    def __init__(self, prefix):
        self.prefix = prefix
        self._desc = IndexDesc()
        _chk(lib().bm2_index_load(prefix.encode(), C.byref(self._desc)), "bm2_index_load")

    @property
    def l_pac(self):
        return self._desc.l_pac

    def close(self):
        if self._desc is not None:
            lib().bm2_index_free(C.byref(self._desc))
            self._desc = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class _DescOf:
    """with _DescOf(prefix | Index | Context) as d: ...   (loads and frees only when given a path)"""

    def __init__(self, x):
        self.x, self.own = x, None

    def __enter__(self):
        if isinstance(self.x, (str, bytes)):
            self.own = Index(self.x if isinstance(self.x, str) else self.x.decode())
            return self.own._desc
        if getattr(self.x, "_desc", None) is None:
            raise Bm2Error("this Context / Index holds no index")
        return self.x._desc

    def __exit__(self, *a):
        if self.own is not None:
            self.own.close()


class Context:
    """One per GPU; mirrors the lifetime of the reference's FMI_search + ref_string (fastmap.cpp:848-888)."""

    def __init__(self, device=0, index_prefix=None, share=None):
        """index_prefix: path (loaded here), or an Index (kept by the caller); share=<Context>: a second context on the same
        device that uses that context's index replica (bm2_create_shared)."""
        L = lib()
        self._desc = None
        self._own_desc = True
        self.h = None
        if share is not None:
            L.bm2_create_shared.restype = C.c_void_p
            L.bm2_create_shared.argtypes = [C.c_void_p]
            self._desc, self._own_desc = share._desc, False
            self.h = L.bm2_create_shared(share.h)
            if not self.h:
                raise Bm2Error("bm2_create_shared failed: " + L.bm2_last_error().decode())
            return
        if isinstance(index_prefix, Index):
            self._desc, self._own_desc = index_prefix._desc, False
            self.h = L.bm2_create(device, C.byref(self._desc))
            if not self.h:
                raise Bm2Error("bm2_create failed: " + L.bm2_last_error().decode())
            return
        if index_prefix is not None:
            self._desc = IndexDesc()
            _chk(L.bm2_index_load(index_prefix.encode(), C.byref(self._desc)), "bm2_index_load")
        self.h = L.bm2_create(device, C.byref(self._desc) if self._desc is not None else None)
        if not self.h:
            msg = L.bm2_last_error().decode()
            if self._desc is not None:
                L.bm2_index_free(C.byref(self._desc))
            raise Bm2Error("bm2_create failed: " + msg)

    @property
    def l_pac(self):
        return self._desc.l_pac if self._desc is not None else 0

    def close(self):
        if self.h:
            lib().bm2_destroy(self.h)
            self.h = None
        if self._desc is not None:
            if self._own_desc:
                lib().bm2_index_free(C.byref(self._desc))
            self._desc = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # S1
    def bsw(self, pairs, ref, qer, w, params):
        pairs = np.ascontiguousarray(pairs, SEQPAIR_DT)
        ref = np.ascontiguousarray(ref, np.uint8)
        qer = np.ascontiguousarray(qer, np.uint8)
        _chk(lib().bm2_bsw(self.h, pairs.ctypes.data, ref.ctypes.data, len(ref), qer.ctypes.data, len(qer), len(pairs), w,
                           C.byref(params)), "bm2_bsw")
        return pairs

    def bsw_upload(self, pairs, ref, qer):
        """S1 with the batch resident: upload once (bm2_bsw_upload), bsw_run any number of times, bsw_download."""
        pairs = np.ascontiguousarray(pairs, SEQPAIR_DT)
        ref = np.ascontiguousarray(ref, np.uint8)
        qer = np.ascontiguousarray(qer, np.uint8)
        L = lib()
        L.bm2_bsw_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32]
        _chk(L.bm2_bsw_upload(self.h, pairs.ctypes.data, ref.ctypes.data, len(ref), qer.ctypes.data, len(qer), len(pairs)), "bm2_bsw_upload")
        self._n_bsw = len(pairs)

    def bsw_run(self, w, params, count_cells=False):
        """-> (kernel ms from HIP events, DP cells or None); counting the cells costs an atomic per pair: not in a timed run."""
        L = lib()
        L.bm2_bsw_run.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int64)]
        ms, cells = C.c_float(0), C.c_int64(0)
        _chk(L.bm2_bsw_run(self.h, w, C.byref(params), C.byref(ms), C.byref(cells) if count_cells else None), "bm2_bsw_run")
        return ms.value, (cells.value if count_cells else None)

    def bsw_download(self):
        out = np.zeros(self._n_bsw, SEQPAIR_DT)
        L = lib()
        L.bm2_bsw_download.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        _chk(L.bm2_bsw_download(self.h, out.ctypes.data, len(out)), "bm2_bsw_download")
        return out

    # S2
    def smem(self, enc, off, ln, opt, cap=None):
        r, keep = _reads_struct(enc, off, ln)
        cap = cap or max(1024, 64 * len(keep[2]))
        while True:
            out = np.zeros(cap, SMEM_DT)
            n = C.c_int64(0)
            rc = lib().bm2_smem(self.h, C.byref(r), C.byref(opt), out.ctypes.data, cap, C.byref(n))
            if rc == BM2_ECAP:
                cap = int(n.value)
                continue
            _chk(rc, "bm2_smem")
            return out[:n.value]

    def sal(self, smems, max_occ, cap=None):
        smems = np.ascontiguousarray(smems, SMEM_DT)
        cap = cap or int(np.minimum(smems["s"], max_occ).sum()) + 16
        out = np.zeros(cap, np.int64)
        n = C.c_int64(0)
        _chk(lib().bm2_sal(self.h, smems.ctypes.data, len(smems), max_occ, out.ctypes.data, cap, C.byref(n)), "bm2_sal")
        return out[:n.value]

    # S3
    def seed_chain_extend(self, enc, off, ln, opt, cap=None):
        r, keep = _reads_struct(enc, off, ln)
        nr = len(keep[2])
        cap = cap or max(1024, 16 * nr)
        reg_off = np.zeros(nr + 1, np.int64)
        st = Stats()
        while True:
            regs = np.zeros(cap, REG_DT)
            n = C.c_int64(0)
            rc = lib().bm2_seed_chain_extend(self.h, C.byref(r), C.byref(opt), regs.ctypes.data, cap, reg_off.ctypes.data,
                                             C.byref(n), C.byref(st))
            if rc == BM2_ECAP:
                cap = int(n.value)
                continue
            _chk(rc, "bm2_seed_chain_extend")
            return regs[:n.value], reg_off, st.as_dict()

    # split S3 (device-resident timing)
    def batch_upload(self, enc, off, ln):
        r, keep = _reads_struct(enc, off, ln)
        _chk(lib().bm2_batch_upload(self.h, C.byref(r)), "bm2_batch_upload")
        self._n_reads = len(keep[2])

    def set_stream_priority(self, level):
        """main stream at the highest (level > 0) / lowest (< 0) / default hardware queue priority"""
        L = lib()
        L.bm2_set_stream_priority.argtypes = [C.c_void_p, C.c_int]
        _chk(L.bm2_set_stream_priority(self.h, level), "bm2_set_stream_priority")

    def batch_upload_chunk(self, chunk):
        """H2D of a FastqChunk (its arrays stay in the library's buffers)."""
        _chk(lib().bm2_batch_upload(self.h, C.byref(chunk.reads)), "bm2_batch_upload")
        self._n_reads = chunk.n_reads

    def finish_regs(self, reads, opt, regs, reg_off):
        """The tail of mem_kernel2_core (mem_sort_dedup_patch + ALT flag, bwamem.cpp:1154-1169) on the device for hits given as host
        arrays (bm2_finish_regs_dev).  reads: a FastqChunk or (enc, off, len) -> (alnregs ALNREG_DT, aln_off)."""
        if isinstance(reads, FastqChunk):
            r, n_reads = reads.reads, reads.n_reads
        else:
            r, keep = _reads_struct(*reads)
            n_reads = len(keep[2])
        regs = np.ascontiguousarray(regs, REG_DT)
        reg_off = np.ascontiguousarray(reg_off, np.int64)
        out = np.zeros(max(len(regs), 1), ALNREG_DT)
        out_off = np.zeros(n_reads + 1, np.int64)
        n = C.c_int64(0)
        L = lib()
        L.bm2_finish_regs_dev.argtypes = [C.c_void_p, C.POINTER(Opt), C.POINTER(Reads), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                          C.POINTER(C.c_int64)]
        _chk(L.bm2_finish_regs_dev(self.h, C.byref(opt), C.byref(r), regs.ctypes.data, reg_off.ctypes.data, out.ctypes.data, len(out),
                                   out_off.ctypes.data, C.byref(n)), "bm2_finish_regs_dev")
        return out[:n.value], out_off

    def batch_finish(self, opt):
        """mem_sort_dedup_patch + ALT flag on the regs of the last batch_run, in HBM (bm2_batch_finish)."""
        L = lib()
        L.bm2_batch_finish.argtypes = [C.c_void_p, C.POINTER(Opt)]
        _chk(L.bm2_batch_finish(self.h, C.byref(opt)), "bm2_batch_finish")

    def batch_download_alnregs(self, cap=None, out=None):
        """-> (alnregs, aln_off); out = a caller's ALNREG_DT array to fill when it is large enough (e.g. pinned_empty: no staging copy)"""
        nr = self._n_reads
        aln_off = np.empty(nr + 1, np.int64)
        L = lib()
        L.bm2_batch_download_alnregs.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]
        n = C.c_int64(0)
        rc = L.bm2_batch_download_alnregs(self.h, None, 0, aln_off.ctypes.data, C.byref(n))       # the count first: an exact, untouched buffer
        if rc not in (BM2_OK, BM2_ECAP):
            _chk(rc, "bm2_batch_download_alnregs")
        if out is None or len(out) < max(int(n.value), 1):
            out = np.empty(max(int(n.value), 1), ALNREG_DT)
        _chk(L.bm2_batch_download_alnregs(self.h, out.ctypes.data, len(out), aln_off.ctypes.data, C.byref(n)), "bm2_batch_download_alnregs")
        return out[:n.value], aln_off

    def sam(self, chunk, opt, so, alnregs, aln_off, n_processed=0, paired=True, out=None):
        """SAM alignment lines of a chunk (bm2_sam_pe_dev / bm2_sam_se_dev: rescue and CIGAR alignments as device batches) -> uint8 array."""
        L = lib()
        alnregs = np.ascontiguousarray(alnregs, ALNREG_DT)
        aln_off = np.ascontiguousarray(aln_off, np.int64)
        need = C.c_int64(0)
        cap = max(1 << 20, int(3 * (int(chunk.f.n_bases) + 200 * chunk.n_reads)))
        if out is not None and len(out) >= 1 << 20:
            cap = len(out)                                      # a caller's buffer, reused from chunk to chunk (no fresh pages to fault in)
        while True:
            buf = out if out is not None and len(out) == cap else np.empty(cap, np.uint8)
            if paired:
                rc = L.bm2_sam_pe_dev(C.c_void_p(self.h), C.byref(self._desc), C.byref(opt), C.byref(so), C.byref(chunk.reads), C.byref(chunk.text),
                                      C.c_void_p(alnregs.ctypes.data), C.c_void_p(aln_off.ctypes.data), C.c_int64(n_processed), None, None,
                                      C.c_void_p(buf.ctypes.data), C.c_int64(cap), C.byref(need))
            else:
                a = alnregs.copy()                                  # reordered in place
                rc = L.bm2_sam_se_dev(C.c_void_p(self.h), C.byref(self._desc), C.byref(opt), C.byref(so), C.byref(chunk.reads), C.byref(chunk.text),
                                      C.c_void_p(a.ctypes.data), C.c_void_p(aln_off.ctypes.data), C.c_int64(n_processed),
                                      C.c_void_p(buf.ctypes.data), C.c_int64(cap), C.byref(need))
            if rc == BM2_ECAP:
                cap = need.value + 16
                continue
            _chk(rc, "bm2_sam_pe_dev" if paired else "bm2_sam_se_dev")
            return buf[:need.value]                              # a uint8 view of the buffer the library wrote into (no copy); bytes(x) / x.tobytes() for text

    def batch_run(self, opt):
        _chk(lib().bm2_batch_run(self.h, C.byref(opt)), "bm2_batch_run")

    def batch_stats(self):
        st = Stats()
        _chk(lib().bm2_batch_stats(self.h, C.byref(st)), "bm2_batch_stats")
        return st.as_dict()

    def batch_download(self, cap=None):
        nr = self._n_reads
        cap = cap or max(1024, 16 * nr)
        reg_off = np.zeros(nr + 1, np.int64)
        while True:
            regs = np.zeros(cap, REG_DT)
            n = C.c_int64(0)
            rc = lib().bm2_batch_download(self.h, regs.ctypes.data, cap, reg_off.ctypes.data, C.byref(n))
            if rc == BM2_ECAP:
                cap = int(n.value)
                continue
            _chk(rc, "bm2_batch_download")
            return regs[:n.value], reg_off

    def batch_fetch(self, what, dtype):
        n = C.c_int64(0)
        rc = lib().bm2_batch_fetch(self.h, what.encode(), None, 0, C.byref(n))
        if rc not in (BM2_OK, BM2_ECAP):
            _chk(rc, "bm2_batch_fetch")
        dt = np.dtype(dtype)
        out = np.zeros(n.value // dt.itemsize, dt)
        if n.value:
            _chk(lib().bm2_batch_fetch(self.h, what.encode(), out.ctypes.data, n.value, C.byref(n)), "bm2_batch_fetch")
        return out

    def batch_kernel_ms(self):
        ms = (C.c_float * 32)()
        names = (C.c_char_p * 32)()
        n = C.c_int32(0)
        _chk(lib().bm2_batch_kernel_ms(self.h, ms, 32, C.byref(n), names), "bm2_batch_kernel_ms")
        return [(names[i].decode(), float(ms[i])) for i in range(n.value)]

    def batch_parts(self):
        """Parts the last uploaded chunk was cut into (each runs beside the others on streams of its own)."""
        return int(lib().bm2_batch_parts(self.h))


def chunk_hits_sharded(ctxs, reads, opt):
    """One chunk over several contexts (bm2_chunk_hits_sharded): reads = FastqChunk or (enc, off, len) -> (alnregs, aln_off)."""
    if isinstance(reads, FastqChunk):
        r, n_reads = reads.reads, reads.n_reads
    else:
        r, keep = _reads_struct(*reads)
        n_reads = len(keep[2])
    L = lib()
    L.bm2_chunk_hits_sharded.argtypes = [C.c_void_p, C.c_int, C.POINTER(Reads), C.POINTER(Opt), C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]
    hs = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    aln_off = np.zeros(n_reads + 1, np.int64)
    cap = max(1024, 4 * n_reads)
    while True:
        out = np.zeros(cap, ALNREG_DT)
        n = C.c_int64(0)
        rc = L.bm2_chunk_hits_sharded(hs, len(ctxs), C.byref(r), C.byref(opt), out.ctypes.data, cap, aln_off.ctypes.data, C.byref(n))
        if rc == BM2_ECAP:
            cap = int(n.value)                                  # (the parts run again: size generously)
            continue
        _chk(rc, "bm2_chunk_hits_sharded")
        return out[:n.value], aln_off


def ksw_align2(pairs, xtra, opt, ctx=None):
    """pairs: list of (query codes, target codes); xtra: list of ints -> int32 array [n, 7] (score, te, qe, score2, te2, tb, qb).
    ctx = a Context: run the device kernel (bm2_ksw_align2_dev) instead of the host one."""
    n = len(pairs)
    buf, q_off, q_len, t_off, t_len = [], [], [], [], []
    p = 0
    for q, t in pairs:
        q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
        q_off.append(p); q_len.append(len(q)); buf.append(q); p += len(q)
        t_off.append(p); t_len.append(len(t)); buf.append(t); p += len(t)
    seqs = np.concatenate(buf) if buf else np.zeros(1, np.uint8)
    q_off = np.array(q_off, np.int64); t_off = np.array(t_off, np.int64)
    q_len = np.array(q_len, np.int32); t_len = np.array(t_len, np.int32); xt = np.array(xtra, np.int32)
    out = np.zeros((max(n, 1), 7), np.int32)
    L = lib()
    if ctx is not None:
        L.bm2_ksw_align2_dev.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64] + [C.c_void_p] * 5 + [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        mat = (C.c_int8 * 25)(*opt.mat)
        _chk(L.bm2_ksw_align2_dev(ctx.h, n, seqs.ctypes.data, len(seqs), q_off.ctypes.data, q_len.ctypes.data, t_off.ctypes.data, t_len.ctypes.data,
                                  xt.ctypes.data, C.cast(mat, C.c_void_p), opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, out.ctypes.data), "bm2_ksw_align2_dev")
        return out[:n]
    L.bm2_ksw_align2.argtypes = [C.c_int32] + [C.c_void_p] * 6 + [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    mat = (C.c_int8 * 25)(*opt.mat)
    _chk(L.bm2_ksw_align2(n, seqs.ctypes.data, q_off.ctypes.data, q_len.ctypes.data, t_off.ctypes.data, t_len.ctypes.data, xt.ctypes.data,
                          C.cast(mat, C.c_void_p), opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, out.ctypes.data), "bm2_ksw_align2")
    return out[:n]


def sam_cigar_stats():
    """(planned, used, missed) CIGAR alignments of the last sam_se / sam_pe call that ran them as a batch."""
    v = [C.c_int64(0) for _ in range(3)]
    lib().bm2_sam_cigar_stats(*[C.byref(x) for x in v])
    return tuple(x.value for x in v)


def sam_rescue_stats():
    """(planned, used, missed) mate-rescue alignments of the last sam_pe call."""
    v = [C.c_int64(0) for _ in range(3)]
    lib().bm2_sam_rescue_stats(*[C.byref(x) for x in v])
    return tuple(x.value for x in v)


def sam_header(index_prefix, hdr_line=None):
    """@SQ lines (+ the caller's header lines) as bwa_print_sam_hdr prints them -> bytes."""
    L = lib()
    with _DescOf(index_prefix) as d:
        L.bm2_sam_header.argtypes = [C.POINTER(IndexDesc), C.c_char_p, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]
        need = C.c_int64(0)
        L.bm2_sam_header(C.byref(d), hdr_line, None, 0, C.byref(need))
        buf = C.create_string_buffer(need.value + 1)
        _chk(L.bm2_sam_header(C.byref(d), hdr_line, buf, need.value, C.byref(need)), "bm2_sam_header")
        return buf.raw[:need.value]


def gen_cigar(index_prefix, opt, tasks, ctx=None):
    """tasks: list of (query codes, rb, re, w) -> list of (score, NM, [cigar ops] or None, MD bytes).
    ctx = a Context created with this index: the device kernel (bm2_gen_cigar_dev) instead of the host code."""
    L = lib()
    with _DescOf(index_prefix) as d:
        n = len(tasks)
        qs = [np.ascontiguousarray(t[0], np.uint8) for t in tasks]
        q_len = np.array([len(q) for q in qs], np.int32)
        q_off = np.concatenate([[0], np.cumsum(q_len[:-1])]).astype(np.int64) if n else np.zeros(0, np.int64)
        seqs = np.concatenate(qs) if n else np.zeros(1, np.uint8)
        rb = np.array([t[1] for t in tasks], np.int64); re_ = np.array([t[2] for t in tasks], np.int64); w = np.array([t[3] for t in tasks], np.int32)
        score = np.zeros(max(n, 1), np.int32); nm = np.zeros(max(n, 1), np.int32); nc = np.zeros(max(n, 1), np.int32)
        c_off = np.zeros(max(n, 1), np.int64); m_off = np.zeros(max(n, 1), np.int64)
        cneed, mneed = C.c_int64(0), C.c_int64(0)
        ccap, mcap = 16 * max(n, 1), 64 * max(n, 1)
        L.bm2_gen_cigar.argtypes = [C.POINTER(IndexDesc), C.POINTER(Opt), C.c_int32] + [C.c_void_p] * 11 + [C.c_int64, C.POINTER(C.c_int64),
                                    C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        while True:
            cig = np.zeros(ccap, np.uint32); md = C.create_string_buffer(mcap)
            tail = (q_off.ctypes.data, q_len.ctypes.data, rb.ctypes.data, re_.ctypes.data, w.ctypes.data, score.ctypes.data, nm.ctypes.data,
                    nc.ctypes.data, c_off.ctypes.data, cig.ctypes.data, ccap, C.byref(cneed), m_off.ctypes.data, C.cast(md, C.c_void_p), mcap, C.byref(mneed))
            if ctx is not None:
                L.bm2_gen_cigar_dev.argtypes = [C.c_void_p, C.POINTER(Opt), C.c_int32, C.c_void_p, C.c_int64] + [C.c_void_p] * 10 + \
                                               [C.c_int64, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
                rc = L.bm2_gen_cigar_dev(ctx.h, C.byref(opt), n, seqs.ctypes.data, len(seqs), *tail)
            else:
                rc = L.bm2_gen_cigar(C.byref(d), C.byref(opt), n, seqs.ctypes.data, *tail)
            if rc == BM2_ECAP:
                ccap, mcap = max(ccap, cneed.value + 1), max(mcap, mneed.value + 1)
                continue
            _chk(rc, "bm2_gen_cigar")
            break
        out = []
        raw = md.raw
        for i in range(n):
            if nc[i] < 0:
                out.append((int(score[i]), int(nm[i]), None, b""))
            else:
                m = raw[m_off[i]:raw.index(b"\0", m_off[i])]
                out.append((int(score[i]), int(nm[i]), [int(x) for x in cig[c_off[i]:c_off[i] + nc[i]]], m))
        return out


def fastq_parse(text):
    """FASTA/FASTQ bytes -> (enc, off, len, names, comments, quals) with kseq / bseq_read semantics."""
    L = lib()
    f = Fastq()
    L.bm2_fastq_parse.argtypes = [C.c_char_p, C.c_int64, C.POINTER(Fastq)]
    L.bm2_fastq_free.argtypes = [C.POINTER(Fastq)]
    L.bm2_fastq_free.restype = None
    _chk(L.bm2_fastq_parse(text, len(text), C.byref(f)), "bm2_fastq_parse")
    try:
        n = f.n_reads
        enc = np.ctypeslib.as_array(f.enc, shape=(max(f.n_bases, 1),))[:f.n_bases].copy()
        off = np.ctypeslib.as_array(f.off, shape=(max(n, 1),))[:n].copy()
        ln = np.ctypeslib.as_array(f.len, shape=(max(n, 1),))[:n].copy()
        names = [f.name[i] for i in range(n)]
        comments = [f.comment[i] for i in range(n)]
        quals = [f.qual[i] for i in range(n)]
        return enc, off, ln, names, comments, quals
    finally:
        L.bm2_fastq_free(C.byref(f))


class FastqChunk:
    """A parsed chunk that stays in the library's arrays (bm2_fastq_parse_mt): numpy views of enc / off / len and the bm2_read_text
    pointers the SAM writer takes -- no per-read Python objects.  close() frees it."""

    def __init__(self, text1, text2=None, n_threads=0):
        L = lib()
        L.bm2_fastq_parse_mt.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, C.c_int, C.POINTER(Fastq)]
        L.bm2_fastq_free.argtypes = [C.POINTER(Fastq)]
        L.bm2_fastq_free.restype = None
        self.f = Fastq()
        _chk(L.bm2_fastq_parse_mt(text1, len(text1), text2, len(text2) if text2 is not None else 0, n_threads, C.byref(self.f)), "bm2_fastq_parse_mt")
        f = self.f
        n = f.n_reads
        self.n_reads = n
        self.enc = np.ctypeslib.as_array(f.enc, shape=(max(f.n_bases, 1),))[:f.n_bases]
        self.off = np.ctypeslib.as_array(f.off, shape=(max(n, 1),))[:n]
        self.len = np.ctypeslib.as_array(f.len, shape=(max(n, 1),))[:n]
        self.reads = Reads(n, C.cast(f.enc, C.c_void_p), C.cast(f.off, C.c_void_p), C.cast(f.len, C.c_void_p))
        self.text = ReadText(f.name, f.comment, f.qual)

    def names(self):
        return [self.f.name[i] for i in range(self.n_reads)]

    def close(self):
        if self.f is not None:
            self.enc = self.off = self.len = None
            lib().bm2_fastq_free(C.byref(self.f))
            self.f = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def default_sam_opt(**kw):
    o = SamOpt()
    lib().bm2_sam_opt_init(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def sam_pe(index_prefix, enc, off, ln, opt, alnregs, reg_off, names, quals=None, comments=None, sam_opt=None, n_processed=0, pes_in=None, ctx=None):
    """Paired-end SAM alignment lines (reads interleaved) -> (bytes, [4 PeStat]); pes_in = 4 PeStat to use instead of mem_pestat;
    ctx = a Context created with this index: the mate-rescue alignments run on the device (bm2_sam_pe_dev)."""
    return sam_se(index_prefix, enc, off, ln, opt, alnregs, reg_off, names, quals, comments, sam_opt, n_processed, paired=True, pes_in=pes_in, ctx=ctx)


def sam_se(index_prefix, enc, off, ln, opt, alnregs, reg_off, names, quals=None, comments=None, sam_opt=None, n_processed=0, paired=False,
           pes_in=None, ctx=None):
    """Single-end SAM alignment lines (host only, no GPU) from the alnregs of finish_regs -> bytes."""
    L = lib()
    with _DescOf(index_prefix) as d:
        r, keep = _reads_struct(enc, off, ln)
        n = len(keep[2])
        so = sam_opt if sam_opt is not None else default_sam_opt()

        def arr(v):
            if v is None:
                return None
            a = (C.c_char_p * n)()
            for i, x in enumerate(v):
                a[i] = None if x is None else (x if isinstance(x, bytes) else x.encode())
            return a
        nm, ql, cm = arr(names), arr(quals), arr(comments)
        t = ReadText(C.cast(nm, C.POINTER(C.c_char_p)), C.cast(cm, C.POINTER(C.c_char_p)) if cm is not None else None,
                     C.cast(ql, C.POINTER(C.c_char_p)) if ql is not None else None)
        reg_off = np.ascontiguousarray(reg_off, np.int64)
        need = C.c_int64(0)
        cap = max(1 << 20, int(3 * (np.asarray(ln, np.int64).sum() + 200 * n)))       # a generous first guess: ECAP means running the chunk again
        while True:
            a = np.ascontiguousarray(alnregs, ALNREG_DT).copy()          # reordered in place: every call gets fresh regs
            buf = C.create_string_buffer(cap)
            if paired:
                pes = (PeStat * 4)()
                args = (C.byref(d), C.byref(opt), C.byref(so), C.byref(r), C.byref(t), C.c_void_p(a.ctypes.data), C.c_void_p(reg_off.ctypes.data),
                        C.c_int64(n_processed), (PeStat * 4)(*pes_in) if pes_in is not None else None, pes, buf, C.c_int64(cap), C.byref(need))
                if isinstance(ctx, (list, tuple)):              # several contexts: the tail's device batches are cut over them (bm2_sam_pe_dev_multi)
                    hs = (C.c_void_p * len(ctx))(*[c.h for c in ctx])
                    rc = L.bm2_sam_pe_dev_multi(hs, C.c_int(len(ctx)), *args)
                else:
                    rc = L.bm2_sam_pe_dev(C.c_void_p(ctx.h), *args) if ctx is not None else L.bm2_sam_pe(*args)
            else:
                args = (C.byref(d), C.byref(opt), C.byref(so), C.byref(r), C.byref(t), C.c_void_p(a.ctypes.data), C.c_void_p(reg_off.ctypes.data),
                        C.c_int64(n_processed), buf, C.c_int64(cap), C.byref(need))
                if isinstance(ctx, (list, tuple)):
                    hs = (C.c_void_p * len(ctx))(*[c.h for c in ctx])
                    rc = L.bm2_sam_se_dev_multi(hs, C.c_int(len(ctx)), *args)
                else:
                    rc = L.bm2_sam_se_dev(C.c_void_p(ctx.h), *args) if ctx is not None else L.bm2_sam_se(*args)
            if rc == BM2_ECAP:
                cap = need.value + 16
                continue
            _chk(rc, "bm2_sam_pe" if paired else "bm2_sam_se")
            return (buf.raw[:need.value], list(pes)) if paired else buf.raw[:need.value]


def index_build(fasta, prefix=None, n_threads=0):
    """Multi-threaded, byte-identical equivalent of `bwa-mem2 index` (host only)."""
    _chk(lib().bm2_index_build(fasta.encode(), (prefix or fasta).encode(), n_threads), "bm2_index_build")
