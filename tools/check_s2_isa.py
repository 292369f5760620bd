#!/usr/bin/env python3
"""Check the `_s2` kernels of lib/libddt.so at the instruction level.

`_s2` keeps the level-0/1 node records of four trees in SGPRs, fetched by four inline-asm `s_load_dwordx4` one sub-group
ahead (ddt_kernels.hip: top_issue / top_wait).  hipcc does not know that those registers are still being written when the
asm statement returns: it is free to copy, spill or reuse them before the kernel's own `s_waitcnt lgkmcnt(0)`.  It does not
in the shipped kernels -- this script proves it for the binary that was actually built: between the issue of a set and the
first full lgkm wait after it, no instruction may read or write any SGPR of the set.

Usage: check_s2_isa.py [path/to/libddt.so]   (exit status 1 and a listing on a violation)
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="s2isa_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = []
        for f in sorted(os.listdir(tmp)):
            if "gfx950" in f:
                text.append(subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout)
        return "\n".join(text)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def kernels(dis):
    """yield (symbol, [instruction text, ...]) per function of the disassembly"""
    name, body = None, []
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
            continue
        if name and "\t" in line:
            ins = line.split("//")[0].strip()
            if ins:
                body.append(ins)
    if name:
        yield name, body


def check_kernel(name, body):
    """returns (number of record sets issued, [violations])"""
    sets, bad = 0, []
    pending = {}  # sgpr -> index of the s_load that writes it
    for i, ins in enumerate(body):
        op = ins.split()[0]
        if op == "s_waitcnt" and "lgkmcnt(0)" in ins:
            pending.clear()
            continue
        touched = sregs(ins)
        if op == "s_load_dwordx4":
            dst = sregs(ins.split(",")[0])
            src = touched - dst
            hit = (dst | src) & set(pending)
            if hit:
                bad.append((i, ins, sorted(hit)))
            # the asm's loads come in fours off one base pointer; the compiler's own s_loads are waited for by its own waits
            for r in dst:
                pending[r] = i
            sets += 1
            continue
        if op.startswith("s_load") or op.startswith("s_buffer_load"):
            # compiler-issued scalar loads (kernel arguments): tracked the same way, hipcc waits before it uses them
            for r in sregs(ins.split(",")[0]):
                pending[r] = i
            continue
        hit = touched & set(pending)
        if hit:
            bad.append((i, ins, sorted(hit)))
    return sets, bad


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "distributed-decisiontrees_amd", "lib", "libddt.so")
    dis = disassemble(lib)
    n_kernels, n_sets, failed = 0, 0, 0
    for name, body in kernels(dis):
        if "score_q16_kernel" not in name:
            continue
        sets, bad = check_kernel(name, body)
        if sets < 8:  # kernels without _s2 only have the compiler's argument loads
            continue
        n_kernels += 1
        n_sets += sets
        if bad:
            failed += 1
            print(f"{name}: {len(bad)} instruction(s) touch SGPRs with a scalar load in flight", file=sys.stderr)
            for i, ins, regs in bad[:8]:
                print(f"  #{i}: {ins}   <- s{regs}", file=sys.stderr)
    print(f"{n_kernels} _s2 kernels, {n_sets} s_load_dwordx4 checked, {failed} with violations")
    return 1 if failed or n_kernels == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
