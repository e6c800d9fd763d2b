#!/usr/bin/env python3
"""Builds the device sources of bwa-mem2_amd/csrc for the host emulator (tools/emu/fakehip): TEST INFRASTRUCTURE.
    python tools/emu/build_emu.py <out_dir> [--csrc DIR]   ->  <out_dir>/libbm2_emu.so  (the C ABI of include/bm2.h, kernels on OS threads)
Source rewrites (on copies; the device sources are not touched): `extern __shared__ ATTR T name[];` becomes a pointer into the
emulator's LDS buffer; address_space(3) and the "v" register constraint of an empty asm mean nothing on the host."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HIP = ["bm2_api.hip", "bsw.hip", "smem.hip", "scan.hip", "chain.hip", "seedsw.hip", "extend.hip", "pipeline.hip", "matesw.hip", "cigar.hip", "finish.hip"]
HOST = ["index_io.cpp", "index_build.cpp", "sam_tail.cpp", "fastq_io.cpp"]


def rewrite(text):
    text = re.sub(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_][A-Za-z0-9_]*)\s+([A-Za-z_][A-Za-z0-9_]*)\[\];",
                  r"\1 *\2 = (\1 *)emu_dyn_lds;", text)
    text = text.replace("__attribute__((address_space(3)))", "")
    text = text.replace('"+v"', '"+r"').replace('"+s"', '"+r"')
    text = re.sub(r"__attribute__\(\(amdgpu_waves_per_eu\([^)]*\)\)\)", "", text)       # a register-allocation hint
    # collectives called from divergent lanes -> their one-lane-at-a-time versions (see the fake hip_runtime.h); the definition stays
    text = re.sub(r"(?<![A-Za-z_])wave_alloc<(\w+)>\(", r"emu_wave_alloc<\1>(", text)
    text = text.replace("static __device__ __forceinline__ int64_t emu_wave_alloc(", "static __device__ __forceinline__ int64_t wave_alloc(")
    text = re.sub(r"(?<![A-Za-z_])wave_alloc_exact\(", "emu_wave_alloc<1>(", text)                      # (one id per calling lane, atomically: what the GPU's ballot + one atomic gives)
    text = text.replace("static __device__ __forceinline__ int64_t emu_wave_alloc<1>(LdsPool *,", "static __device__ __forceinline__ int64_t wave_alloc_exact(LdsPool *,")
    # implicit lockstep: "lane 0 initialises the wave's LDS pool, then every lane uses it" needs no barrier on a GPU (one wavefront,
    # program order); OS threads need a rendezvous there.  (Sites are wave-uniform: the first statements of the kernels.)
    text = re.sub(r"(\n[ \t]*if \((?:\(threadIdx\.x & 63\) == 0|lane == 0)\) \{[^\n{}]*->pos = [^\n{}]*\}[ \t]*)(?=\n)", r"\1 emu_wsync();", text)
    return text


def build(out_dir, csrc=None, only=None):
    csrc = csrc or os.path.join(ROOT, "bwa-mem2_amd", "csrc")
    os.makedirs(out_dir, exist_ok=True)
    srcs = []
    for f in only or [f for f in HIP if os.path.exists(os.path.join(csrc, f))]:
        dst = os.path.join(out_dir, f.replace(".hip", "_emu.cpp"))
        with open(os.path.join(csrc, f)) as g:
            open(dst, "w").write(rewrite(g.read()))
        srcs.append(dst)
    if only is None:
        srcs += [os.path.join(csrc, f) for f in HOST]
    so = os.path.join(out_dir, "libbm2_emu.so")
    cxx = "/opt/rocm/lib/llvm/bin/clang++" if os.path.exists("/opt/rocm/lib/llvm/bin/clang++") else "g++"      # host mode: no HIP involved
    cmd = [cxx, "-O1", "-std=c++17", "-pthread", "-w", "-fPIC", "-shared", "-I", os.path.join(HERE, "fakehip"), "-I", csrc] + srcs + \
          [os.path.join(HERE, "emu_runtime.cpp"), "-o", so]
    subprocess.check_call(cmd)
    return so


if __name__ == "__main__":
    a = sys.argv[1:]
    csrc = None
    if "--csrc" in a:
        i = a.index("--csrc"); csrc = a[i + 1]; del a[i:i + 2]
    print(build(a[0] if a else "/tmp/bm2_emu", csrc))
