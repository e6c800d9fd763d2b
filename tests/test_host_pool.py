"""The host side's worker pool (bwa-mem2_amd/csrc/host_pool.h): a stress of its phase protocol from three calling threads, and its counting sort
against std::stable_sort.  Compiled with g++ here (the header is plain C++; the library builds it with hipcc)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pool_phases_and_counting_sort(tmp_path):
    exe = str(tmp_path / "pool_stress")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "bwa-mem2_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "pool_stress.cpp"), "-o", exe])
    for env in ({"BM2_POOL_SPIN_US": "100"}, {"BM2_TAIL_PIN": "0", "BM2_POOL_SPIN_US": "0"}, {}):       # (short polls: the box may have few cores)
        p = subprocess.run([exe, "1500" if not env else "4000", "12"], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert p.returncode == 0 and p.stdout.strip() == "ok", p.stderr[-500:]
