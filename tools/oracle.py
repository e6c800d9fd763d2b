"""ctypes binding of oracle/_ref/libbm2oracle.so (the plain-C restatement).  TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import refio

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORADIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = os.path.join(_ORADIR, "_ref", "libbm2oracle.so")


class OraOpt(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "pen_clip5", "pen_clip3",
                                         "w", "zdrop", "min_seed_len", "split_width", "max_occ", "max_chain_gap",
                                         "min_chain_weight", "max_chain_extend")] + \
               [("max_mem_intv", C.c_int64), ("split_factor", C.c_float), ("mask_level", C.c_float),
                ("drop_ratio", C.c_float), ("mask_level_redun", C.c_float), ("mat", C.c_int8 * 25),
                ("pad", C.c_int8 * 3)]


class OraResult(C.Structure):
    _fields_ = [("n_smem", C.c_int64), ("smem", C.c_void_p), ("n_sa", C.c_int64), ("sa_coord", C.c_void_p),
                ("sa_cnt", C.c_void_p),
                ("n_chn0", C.c_int64), ("chn0", C.c_void_p), ("n_seed0", C.c_int64), ("seed0", C.c_void_p),
                ("n_chn1", C.c_int64), ("chn1", C.c_void_p), ("n_seed1", C.c_int64), ("seed1", C.c_void_p),
                ("n_regraw", C.c_int64), ("regraw", C.c_void_p), ("n_regprg", C.c_int64), ("regprg", C.c_void_p),
                ("n_pair", C.c_int64), ("pair", C.c_void_p),
                ("n_ext", C.c_int64), ("n_ext_sameblk", C.c_int64), ("n_lf", C.c_int64), ("n_sa_lookup", C.c_int64),
                ("n_sw_cells", C.c_int64), ("n_regfin", C.c_int64), ("regfin", C.c_void_p)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _ORADIR, "oracle"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.ora_index_load.restype = C.c_void_p
        _lib.ora_index_load.argtypes = [C.c_char_p]
        _lib.ora_index_free.argtypes = [C.c_void_p]
        _lib.ora_opt_init.argtypes = [C.POINTER(OraOpt)]
        _lib.ora_opt_fill_scmat.argtypes = [C.POINTER(OraOpt)]
        _lib.ora_run.restype = C.c_int
        _lib.ora_run.argtypes = [C.c_void_p, C.POINTER(OraOpt), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.POINTER(OraResult), C.c_int]
        _lib.ora_result_free.argtypes = [C.POINTER(OraResult)]
        _lib.ora_ksw_extend.restype = C.c_int
        _lib.ora_ksw_extend.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 8 + \
                                       [C.POINTER(C.c_int)] * 5 + [C.c_void_p]
        _lib.ora_ksw_extend_cls.restype = C.c_int
        _lib.ora_ksw_extend_cls.argtypes = _lib.ora_ksw_extend.argtypes + [C.c_int]
        _lib.ora_band_clamp.restype = C.c_int
        _lib.ora_band_clamp.argtypes = [C.c_int] * 9
        _lib.ora_pair_class.restype = C.c_int
        _lib.ora_pair_class.argtypes = [C.c_int] * 4
        _lib.ora_finish_regs.restype = C.c_int64
        _lib.ora_finish_regs.argtypes = [C.c_void_p, C.POINTER(OraOpt), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    return _lib


def default_opt(**kw):
    o = OraOpt()
    lib().ora_opt_init(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    lib().ora_opt_fill_scmat(C.byref(o))
    return o


def _arr(ptr, n, dt):
    if not ptr or n == 0:
        return np.zeros(0, dt)
    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dt, n).copy()


class Index:
    def __init__(self, prefix):
        self.h = lib().ora_index_load(prefix.encode())
        if not self.h:
            raise IOError("cannot load index " + prefix)

    def close(self):
        if self.h:
            lib().ora_index_free(self.h)
            self.h = None

    def run(self, enc, off, ln, opt=None, seeding_only=False):
        """-> dict of numpy arrays with the same tags as refdump + PAIR + counters."""
        opt = opt or default_opt()
        enc = np.ascontiguousarray(enc, np.uint8)
        off = np.ascontiguousarray(off, np.int64)
        ln = np.ascontiguousarray(ln, np.int32)
        r = OraResult()
        rc = lib().ora_run(self.h, C.byref(opt), len(ln), enc.ctypes.data, off.ctypes.data, ln.ctypes.data,
                           C.byref(r), int(seeding_only))
        if rc != 0:
            raise RuntimeError("ora_run failed: %d" % rc)
        out = {"SMEM": _arr(r.smem, r.n_smem, refio.SMEM_DT), "SACOORD": _arr(r.sa_coord, r.n_sa, np.dtype("<i8")),
               "SACNT": _arr(r.sa_cnt, len(ln), np.dtype("<i4")),
               "CHN0": _arr(r.chn0, r.n_chn0, refio.CHAIN_DT), "SEED0": _arr(r.seed0, r.n_seed0, refio.SEED_DT),
               "CHN1": _arr(r.chn1, r.n_chn1, refio.CHAIN_DT), "SEED1": _arr(r.seed1, r.n_seed1, refio.SEED_DT),
               "REGRAW": _arr(r.regraw, r.n_regraw, refio.REG_DT), "REGPRG": _arr(r.regprg, r.n_regprg, refio.REG_DT),
               "REGFIN": _arr(r.regfin, r.n_regfin, refio.REG_DT),
               "PAIR": _arr(r.pair, r.n_pair, refio.PAIR_DT),
               "counters": dict(n_ext=r.n_ext, n_ext_sameblk=r.n_ext_sameblk, n_lf=r.n_lf,
                                n_sa_lookup=r.n_sa_lookup, n_sw_cells=r.n_sw_cells)}
        lib().ora_result_free(C.byref(r))
        return out


def _finish(self, enc, off, ln, regs, opt=None):
    """Tail of mem_kernel2_core (mem_sort_dedup_patch + ALT flag) on REG_DT records grouped by read -> REG_DT records."""
    opt = opt or default_opt()
    enc = np.ascontiguousarray(enc, np.uint8)
    off = np.ascontiguousarray(off, np.int64)
    regs = np.ascontiguousarray(regs, refio.REG_DT)
    out = np.zeros(max(len(regs), 1), refio.REG_DT)
    n = lib().ora_finish_regs(self.h, C.byref(opt), len(ln), enc.ctypes.data, off.ctypes.data, regs.ctypes.data, len(regs), out.ctypes.data)
    return out[:n]


Index.finish_regs = _finish


def ksw_extend(query, target, opt, w, end_bonus, h0, cls=None):
    """-> (score, qle, tle, gtle, gscore, max_off); band clamp and Z-drop rule of the kernel class the pair runs in."""
    L = lib()
    q = np.ascontiguousarray(query, np.uint8)
    t = np.ascontiguousarray(target, np.uint8)
    if cls is None:
        cls = L.ora_pair_class(len(t), len(q), h0, opt.a)
    wc = L.ora_band_clamp(w, len(q), opt.a, end_bonus, opt.o_ins, opt.e_ins, opt.o_del, opt.e_del, cls)
    outs = [C.c_int() for _ in range(5)]
    mat = (C.c_int8 * 25)(*opt.mat)
    sc = L.ora_ksw_extend_cls(len(q), q.ctypes.data, len(t), t.ctypes.data, C.addressof(mat), opt.o_del, opt.e_del,
                              opt.o_ins, opt.e_ins, wc, end_bonus, opt.zdrop, h0, *[C.byref(x) for x in outs], None, cls)
    return (sc,) + tuple(x.value for x in outs)
