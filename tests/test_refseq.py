"""bwa-mem2_amd/csrc/refseq.h (the index's reference string four bases per byte on the device, read through RefPtr) on the host: plain C++, so
g++ can check every access form the kernels use -- elements, views, four codes per load in both directions, the lane kernel's 28-row windows --
against the same codes held as bytes.  The device side of the same code runs in tests/test_device_sources_on_host.py (emulator, packed by
bm2_create's k_pack_ref) and in every `-m gpu` test that aligns reads."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packed_reference_reads_like_bytes(tmp_path):
    exe = str(tmp_path / "refseq_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "bwa-mem2_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "refseq_check.cpp"), "-o", exe])
    for n in ("100003", "29", "4096"):
        p = subprocess.run([exe, n], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and p.stdout.strip() == "ok", (n, p.stdout, p.stderr[-500:])


def test_counter_summary_of_steady_state_dispatches(tmp_path):
    """tools/rocpd_summary.py: the second half of each kernel's dispatches in a table of its own (the first batch of a process runs without the
    previous batch's statistics), and tools/pmc_to_profiles.py reads THAT table when it is there."""
    import sqlite3
    import sys
    db = str(tmp_path / "r.db")
    con = sqlite3.connect(db)
    con.execute("create table kernels (name text, start integer, end integer)")
    con.execute("create table counters_collection (dispatch_id integer, kernel_name text, counter_name text, value real)")
    for d in range(4):                                           # dispatches 0, 1: the cold step (ten times the counts); 2, 3: steady state
        con.execute("insert into kernels values ('k_x(int)', ?, ?)", (d * 100, d * 100 + 50))
        for dim in range(2):
            con.execute("insert into counters_collection values (?, 'k_x(int)', 'SQ_WAVES', ?)", (d, 1000.0 if d < 2 else 100.0))
    con.commit(); con.close()
    out = str(tmp_path / "s.md")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), db, out], stdout=subprocess.DEVNULL)
    txt = open(out).read()
    assert "| k_x | SQ_WAVES | 4400 | 8 |" in txt and "second half" in txt and "| k_x | SQ_WAVES | 400 | 4 |" in txt, txt
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_to_profiles
    pmc, calls = pmc_to_profiles.counters(out)
    assert pmc["k_x"]["SQ_WAVES"] == (400.0, 4) and calls["k_x"][0] == 4
