"""Parsers for test inputs/outputs: refdump section files, FASTQ -> 2-bit codes, index prefix helpers.

Test infrastructure (used by tests/, bench.py's cpu_baseline leg and tools/make_golden.py).
Record layouts mirror oracle/refdump.cpp and oracle/bm2_oracle.h.
"""
import numpy as np

SMEM_DT = np.dtype([("read", "<i4"), ("m", "<i4"), ("n", "<i4"), ("pad", "<i4"),
                    ("k", "<i8"), ("l", "<i8"), ("s", "<i8")])
CHAIN_DT = np.dtype([("read", "<i4"), ("n", "<i4"), ("rid", "<i4"), ("is_alt", "<i4"), ("pos", "<i8"),
                     ("frac_rep", "<f4"), ("w", "<i4"), ("kept", "<i4"), ("first", "<i4")])
SEED_DT = np.dtype([("rbeg", "<i8"), ("qbeg", "<i4"), ("len", "<i4"), ("score", "<i4"), ("pad", "<i4")])
REG_DT = np.dtype([("read", "<i4"), ("pad", "<i4"), ("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"),
                   ("rid", "<i4"), ("score", "<i4"), ("truesc", "<i4"), ("sub", "<i4"), ("alt_sc", "<i4"),
                   ("csub", "<i4"), ("sub_n", "<i4"), ("w", "<i4"), ("seedcov", "<i4"), ("secondary", "<i4"),
                   ("secondary_all", "<i4"), ("seedlen0", "<i4"), ("n_comp", "<i4"), ("is_alt", "<i4"),
                   ("frac_rep", "<f4"), ("pad2", "<i4")])
PAIR_DT = np.dtype([("read", "<i4"), ("reg", "<i4"), ("is_right", "<i4"), ("len1", "<i4"), ("len2", "<i4"),
                    ("h0", "<i4"), ("ref_pos", "<i8"), ("q_pos", "<i4"), ("w_used", "<i4"), ("score", "<i4"),
                    ("qle", "<i4"), ("tle", "<i4"), ("gtle", "<i4"), ("gscore", "<i4"), ("max_off", "<i4")])
assert SMEM_DT.itemsize == 40 and CHAIN_DT.itemsize == 40 and SEED_DT.itemsize == 24
assert REG_DT.itemsize == 96 and PAIR_DT.itemsize == 64

_SECTION_DT = {"SMEM": SMEM_DT, "SACOORD": np.dtype("<i8"), "SACNT": np.dtype("<i4"),
               "CHN0": CHAIN_DT, "SEED0": SEED_DT, "CHN1": CHAIN_DT, "SEED1": SEED_DT,
               "REGRAW": REG_DT, "REGPRG": REG_DT, "REGFIN": REG_DT, "COUNTS": np.dtype("<i8")}


def read_dump(path):
    """-> dict tag -> numpy array (see oracle/refdump.cpp for the section list)."""
    out = {}
    buf = open(path, "rb").read()
    p = 0
    while p < len(buf):
        tag = buf[p:p + 8].rstrip(b"\0").decode()
        nb = int(np.frombuffer(buf, "<i8", 1, p + 8)[0])
        dt = _SECTION_DT[tag]
        out[tag] = np.frombuffer(buf, dt, nb // dt.itemsize, p + 16).copy()
        p += 16 + nb
    return out


def write_dump(path, sections):
    with open(path, "wb") as f:
        for tag, arr in sections.items():
            f.write(tag.encode().ljust(8, b"\0"))
            b = np.ascontiguousarray(arr).tobytes()
            f.write(np.int64(len(b)).tobytes())
            f.write(b)


_NT4 = np.full(256, 4, dtype=np.uint8)     # nst_nt4_table (bntseq.cpp:38-55): ACGT/acgt -> 0..3, else 4
for _i, _c in enumerate("ACGT"):
    _NT4[ord(_c)] = _i
    _NT4[ord(_c.lower())] = _i


def read_fastq(path):
    """-> (names, list of uint8 code arrays) from a 4-line-record FASTQ or one-sequence-per-line file."""
    names, seqs = [], []
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    if lines and lines[0].startswith(b"@"):
        for i in range(0, len(lines) - 1, 4):
            if not lines[i]:
                continue
            names.append(lines[i][1:].split()[0].decode())
            seqs.append(_NT4[np.frombuffer(lines[i + 1], dtype=np.uint8)])
    else:
        for i, l in enumerate(lines):
            if l:
                names.append("r%d" % i)
                seqs.append(_NT4[np.frombuffer(l, dtype=np.uint8)])
    return names, seqs


def pack_reads(seqs):
    """list of code arrays (or 2-D array) -> (enc uint8[sum], off int64[n], len int32[n])."""
    if isinstance(seqs, np.ndarray) and seqs.ndim == 2:
        n, L = seqs.shape
        ln = np.full(n, L, dtype=np.int32)
        off = np.arange(n, dtype=np.int64) * L
        return np.ascontiguousarray(seqs.reshape(-1)), off, ln
    ln = np.array([len(s) for s in seqs], dtype=np.int32)
    off = np.zeros(len(seqs), dtype=np.int64)
    if len(seqs) > 1:
        off[1:] = np.cumsum(ln[:-1], dtype=np.int64)
    enc = np.concatenate(seqs).astype(np.uint8) if len(seqs) else np.zeros(0, np.uint8)
    return enc, off, ln
