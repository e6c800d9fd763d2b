"""SA coordinates at and beyond 2^32: the high byte of the sampled suffix array (sa_ms_byte, FMI_search.cpp:1202-1255) is zero on every
small test genome, so an index copy with that byte PATCHED (positive and negative int8 values: the reference's array is signed)
is looked up by the oracle and by the device's k_sal through bm2_sal.  The same index files go to both; SMEM intervals do not
depend on the suffix array, so the lookups are those of the real reads.  (The GRCh38-sized bench compares real coordinates beyond
2^32 with the compiled reference in its parity gate; this is the small, always-run version of the same check.)"""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from helpers import load_golden
from tools import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def patched_index(golden_dir, name, dst):
    """copy of the golden index `name` whose sa_ms_byte[i] = ((i * 7) % 11) - 3  (layout: FMI_search.cpp:415-457)"""
    for fn in os.listdir(golden_dir):
        if fn.startswith(name + ".fa"):
            shutil.copy(os.path.join(golden_dir, fn), os.path.join(dst, fn))
    pre = os.path.join(dst, name + ".fa")
    with open(pre + ".bwt.2bit.64", "r+b") as f:
        ref_len = int(np.frombuffer(f.read(8), "<i8")[0])
        nocc, nsa = (ref_len >> 6) + 1, (ref_len >> 3) + 1
        f.seek(8 + 5 * 8 + nocc * 64)
        ms = ((np.arange(nsa, dtype=np.int64) * 7) % 11 - 3).astype(np.int8)
        f.write(ms.tobytes())
    return pre


def expected(pre, enc, off, ln):
    ix = oracle.Index(pre)
    try:
        r = ix.run(enc, off, ln, seeding_only=True)
    finally:
        ix.close()
    return r["SMEM"], r["SACOORD"]


SCRIPT = r'''
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r); sys.path.insert(0, %(pkg)r)
import numpy as np, bm2
if %(lib)r:
    bm2.LIB_PATH = %(lib)r
from helpers import load_golden
from test_sa_high_byte import expected
pre0, enc, off, ln, d = load_golden(%(gold)r, "g20k_l76")
n = %(n)d
ln = ln[:n]; off = off[:n]; enc = enc[:int(off[-1] + ln[-1])]
pre = %(pre)r
smem, sa = expected(pre, enc, off, ln)
assert (np.abs(sa) >= (1 << 32)).sum() > len(sa) // 4 and (sa < 0).any(), "the patch must reach the lookups"
ctx = bm2.Context(0, pre)
opt = bm2.default_opt()
got_smem = ctx.smem(enc, off, ln, opt)
assert got_smem.tobytes() == smem.tobytes()
got = ctx.sal(got_smem, opt.max_occ)
assert len(got) == len(sa) and (got == sa).all(), (len(got), len(sa), int((got != sa).sum()) if len(got) == len(sa) else -1)
print("ok", len(sa), int((np.abs(sa) >= (1 << 32)).sum()))
'''


def _run(lib, pre, golden_dir, n):
    script = SCRIPT % dict(root=ROOT, tests=os.path.join(ROOT, "tests"), pkg=os.path.join(ROOT, "bwa-mem2_amd"), lib=lib, gold=golden_dir,
                           n=n, pre=pre)
    p = subprocess.run([sys.executable, "-c", script], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0 and p.stdout.startswith(b"ok"), p.stderr.decode()[-2000:]


def test_sa_high_byte_on_the_emulator(golden_dir, tmp_path, emu_lib):
    _run(emu_lib, patched_index(golden_dir, "g20k_l76", str(tmp_path)), golden_dir, 24)


@pytest.mark.gpu
def test_sa_high_byte_on_the_device(golden_dir, tmp_path):
    _run("", patched_index(golden_dir, "g20k_l76", str(tmp_path)), golden_dir, 303)
