"""The local SW of mate rescue (bm2_ksw_align2, host today, the seam of the next device kernel) against the reference's own
ksw_align2 called through oracle/_ref/refdump ksw: random pairs with planted local matches, both lane widths (KSW_XBYTE or
not), ties, second-best hits, several scorings.  Known answers: all seven result fields must agree."""
import subprocess

import numpy as np
import pytest

import bm2
from helpers import ref_binary
import helpers  # noqa: E402

KSW_XBYTE, KSW_XSTOP, KSW_XSUBO, KSW_XSTART = 0x10000, 0x20000, 0x40000, 0x80000


def _pairs(seed, n):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        ql = int(rng.integers(20, 260))
        tl = int(rng.integers(ql // 2, 900))
        q = rng.integers(0, 4, size=ql, dtype=np.uint8)
        t = rng.integers(0, 4, size=tl, dtype=np.uint8)
        for _ in range(int(rng.integers(0, 3))):                 # plant noisy copies of pieces of the query: best and second-best hits
            a = int(rng.integers(0, ql - 10)); l = int(rng.integers(10, ql - a + 1))
            piece = q[a:a + l].copy()
            mut = rng.random(l) < rng.choice([0.0, 0.03, 0.1])
            piece[mut] = (piece[mut] + 1) % 4
            if rng.random() < 0.3 and l > 20:                    # an indel
                c = int(rng.integers(5, l - 5)); piece = np.delete(piece, slice(c, c + int(rng.integers(1, 4))))
            p = int(rng.integers(0, max(tl - len(piece), 1)))
            t[p:p + len(piece)] = piece[:tl - p]
        if rng.random() < 0.05:
            q[int(rng.integers(0, ql))] = 4
        out.append((q, t))
    return out


@pytest.mark.parametrize("args,kw", [([], {}), (["-A", "2", "-B", "5", "-O", "7,8", "-E", "2,1"], dict(a=2, b=5, o_del=7, o_ins=8, e_del=2, e_ins=1)),
                                     (["-B", "1", "-O", "1,1", "-E", "1,1"], dict(b=1, o_del=1, o_ins=1))])
def test_ksw_align2_matches_reference(tmp_path, args, kw):
    dump = ref_binary("refdump")
    if dump is None:
        helpers.no_checker("oracle/_ref not built (make -C oracle ref)")
    opt = bm2.default_opt(**kw)
    pairs = _pairs(7 + len(args), 1500)
    xtra = []
    for q, t in pairs:
        x = KSW_XSUBO | KSW_XSTART | (19 * opt.a)
        if len(q) * opt.a < 250:
            x |= KSW_XBYTE
        xtra.append(x)
    pf, of = str(tmp_path / "pairs.txt"), str(tmp_path / "out.bin")
    with open(pf, "w") as f:
        for (q, t), x in zip(pairs, xtra):
            f.write("%d %s %s\n" % (x, "".join("ACGTN"[c] for c in q), "".join("ACGTN"[c] for c in t)))
    subprocess.check_call([dump] + args + ["ksw", pf, of], stderr=subprocess.DEVNULL)
    exp = np.fromfile(of, "<i4").reshape(-1, 7)
    got = bm2.ksw_align2(pairs, xtra, opt)
    assert len(exp) == len(got)
    bad = np.nonzero((exp != got).any(axis=1))[0]
    assert len(bad) == 0, "pair %d: ref %s ours %s (qlen %d, tlen %d, xtra %#x)" % (bad[0], exp[bad[0]], got[bad[0]], len(pairs[bad[0]][0]), len(pairs[bad[0]][1]), xtra[bad[0]])
    assert (exp[:, 0] >= 19 * opt.a).sum() > 300 and (exp[:, 3] > 0).sum() > 50       # the case has real hits and second-best hits
