"""bm2_index_load / bm2_fastq_parse on damaged input: an error code and a message naming the file, never a crash or a silent
half-loaded index (the reference aborts inside err_fread_noeof / bns_restore on the same files)."""
import ctypes as C
import os
import shutil

import pytest

import bm2

EXTS = (".bwt.2bit.64", ".0123", ".ann", ".amb", ".pac")


def _load(prefix):
    L = bm2.lib()
    d = bm2.IndexDesc()
    rc = L.bm2_index_load(prefix.encode(), C.byref(d))
    msg = L.bm2_last_error().decode()
    if rc == 0:
        L.bm2_index_free(C.byref(d))
    return rc, msg


def _copy(golden_dir, tmp_path, name):
    pre = str(tmp_path / name)
    for e in EXTS:
        shutil.copy(os.path.join(golden_dir, "g20k_l76.fa" + e), pre + e)
    return pre


def test_missing_index(tmp_path):
    rc, msg = _load(str(tmp_path / "nope"))
    assert rc == bm2.BM2_EIO and "cannot open" in msg and "nope.bwt.2bit.64" in msg


@pytest.mark.parametrize("ext", [".bwt.2bit.64", ".0123", ".ann"])
@pytest.mark.parametrize("how", ["half", "empty"])
def test_truncated_file(golden_dir, tmp_path, ext, how):
    pre = _copy(golden_dir, tmp_path, "t")
    size = os.path.getsize(pre + ext)
    with open(pre + ext, "r+b") as f:
        f.truncate(size // 2 if how == "half" else 0)
    rc, msg = _load(pre)
    assert rc == bm2.BM2_EIO and ("t" + ext) in msg, msg


@pytest.mark.parametrize("text,what", [("this is not an ann file\n", "first line"), ("99999999999 2 11\n0 a (null)\n0 5 0\n", "ref_len"),
                                       ("%L 1 11\n0 a (null)\n7 %L 0\n", "contiguous"), ("%L 2 11\n0 a (null)\n0 10 0\n0 b (null)\n10 10 0\n", "add up"),
                                       ("%L -4 11\n", "out of range")])
def test_inconsistent_ann(golden_dir, tmp_path, text, what):
    pre = _copy(golden_dir, tmp_path, "g")
    l_pac = int(open(pre + ".ann").read().split()[0])
    open(pre + ".ann", "w").write(text.replace("%L", str(l_pac)))
    rc, msg = _load(pre)
    assert rc == bm2.BM2_EIO and "g.ann is malformed" in msg and what in msg, msg


def test_intact_copy_loads(golden_dir, tmp_path):
    assert _load(_copy(golden_dir, tmp_path, "ok"))[0] == 0


def test_fastq_quality_of_a_different_length_is_an_error():
    # kseq_read returns -2 here and the reference stops reading the file without a word (kseq.h:199-201); a library must say so
    with pytest.raises(bm2.Bm2Error, match="quality string of a different length"):
        bm2.fastq_parse(b"@r0\nACGT\n+\nIIII\n@r1\nACGT\n+\nII\n")


def test_fastq_degenerate_records():
    r = bm2.fastq_parse(b"")
    assert len(r[2]) == 0
    r = bm2.fastq_parse(b"no header here\nACGT\n")
    assert len(r[2]) == 0
    r = bm2.fastq_parse(b"@\n\n+\n\n>y\nAC\nGT")                # empty name + empty read, then multi-line FASTA without a final newline
    assert list(r[2]) == [0, 4] and [bytes(x) for x in r[3]] == [b"", b"y"]
    r = bm2.fastq_parse(b"@r1\r\nACGT\r\n+\r\nIIII\r\n")
    assert list(r[2]) == [4] and bytes(r[5][0]) == b"IIII"
