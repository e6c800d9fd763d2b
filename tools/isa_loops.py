#!/usr/bin/env python3
"""Disassemble kernels of the BUILT libbm2.so (the linked gfx950 code object, i.e. after the device LTO of -fgpu-rdc) and report the
loop structure around their work-item atomics.  Used to check that persistent work loops of the form
    for (;;) { it = atomicAdd(cur, lane == 0); hid = readfirstlane(it); if (hid >= n) break; ... }
compile to ONE loop whose every back edge passes the atomic (notes/NEXT.md: a structurised variant once hung the GPU).

    python tools/isa_loops.py [kernel-name-substring ...]       (default: the kernels with persistent work loops)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(so):
    d = tempfile.mkdtemp(prefix="bm2isa")
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    return subprocess.check_output([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], text=True)


def functions(asm):
    out, name, body = {}, None, []
    for l in asm.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
        if m:
            if name:
                out[name] = body
            name, body = m.group(1), []
        elif name:
            body.append(l)
    if name:
        out[name] = body
    return out


def analyse(body):
    """-> (n_instructions, [atomic positions], [(branch position, target position)] back edges)"""
    addr, ins = [], []
    for l in body:
        m = re.match(r"^\s*(\S.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if m:
            ins.append(m.group(1)); addr.append(int(m.group(2), 16))
    pos = {a: i for i, a in enumerate(addr)}
    atom = [i for i, t in enumerate(ins) if "atomic" in t]
    back = []
    for i, t in enumerate(ins):
        m = re.match(r"s_c?branch\S*\s+(\d+)", t) or re.match(r"s_c?branch\S*\s+(-\d+)", t)
        if m:
            tgt = addr[i] + 4 + 4 * (int(m.group(1)) if int(m.group(1)) < 32768 else int(m.group(1)) - 65536)
            if tgt in pos and pos[tgt] <= i:
                back.append((i, pos[tgt]))
    return len(ins), atom, back


def main():
    want = sys.argv[1:] or ["k_bwd_heavy", "k_postfilter_heavy", "k_chain_heavy", "k_finish"]
    fns = functions(disassemble(os.path.join(ROOT, "bwa-mem2_amd", "libbm2.so")))
    rc = 0
    for name, body in fns.items():
        if not any(w in name for w in want):
            continue
        n, atom, back = analyse(body)
        print("%s: %d instructions, atomics at %s" % (name, n, atom))
        loops = sorted(set(t for _, t in back))
        for t in loops:
            ends = [b for b, tt in back if tt == t]
            inside = [a for a in atom if t <= a <= max(ends)]
            print("   loop [%d, %d]  back edges from %s  atomics inside: %s" % (t, max(ends), ends, inside))
    return rc


if __name__ == "__main__":
    sys.exit(main())
