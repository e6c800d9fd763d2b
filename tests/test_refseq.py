"""RefPtr (bwa-mem2_amd/csrc/refseq.h): the word loaders of the device kernels -- nib8 (CIGAR kernels: eight bases per load) and load4 (extension kernels) --
against element-by-element access on the 2-bit and the byte storage, both directions, every alignment.  Plain C++: the header is host code too."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_word_loaders_equal_element_access(tmp_path):
    exe = str(tmp_path / "nib8_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "bwa-mem2_amd", "csrc"), os.path.join(ROOT, "tests", "native", "nib8_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith(" 0 differ"), out.stdout + out.stderr
