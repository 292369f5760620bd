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


def find_objdump():
    """llvm-objdump of the ROCm toolchain: $ROCM_PATH, what `hipconfig --rocmpath` says, /opt/rocm, then PATH"""
    roots = [os.environ.get("ROCM_PATH"), os.environ.get("HIP_PATH")]
    try:
        roots.append(subprocess.run(["hipconfig", "--rocmpath"], capture_output=True, text=True, timeout=20).stdout.strip())
    except Exception:
        pass
    roots.append("/opt/rocm")
    for r in roots:
        if r:
            for sub in ("lib/llvm/bin", "llvm/bin", "bin"):
                cand = os.path.join(r, sub, "llvm-objdump")
                if os.path.exists(cand):
                    return cand
    return shutil.which("llvm-objdump")


OBJDUMP = find_objdump() or "/opt/rocm/lib/llvm/bin/llvm-objdump"
EXIT_CANNOT_RUN = 3  # no disassembler: nothing was checked (distinct from 1 = a violation)

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
    """yield (symbol, [(address, instruction text), ...]) per function of the disassembly"""
    name, body = None, []
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
            continue
        if name and "\t" in line:
            ins, _, tail = line.partition("//")
            ins = ins.strip()
            m = re.match(r"\s*([0-9A-Fa-f]+):", tail)
            if ins:
                body.append((int(m.group(1), 16) if m else None, ins))
    if name:
        yield name, body


def check_kernel(name, body):
    """returns (number of s_load_dwordx4 seen, [violations]).  A may-analysis over the control-flow graph: an SGPR is pending
    from a scalar load that writes it until the next `s_waitcnt lgkmcnt(0)` on EVERY path; an instruction that reads or writes
    a pending SGPR is a violation.  `body` = [(address, text)] (addresses may be None in hand-made straight-line listings)."""
    body = [x if isinstance(x, tuple) else (None, x) for x in body]
    n = len(body)
    index_of = {a: i for i, (a, _) in enumerate(body) if a is not None}

    def target(i):
        a, ins = body[i]
        if a is None or i + 1 >= n or body[i + 1][0] is None:
            return None
        simm = int(ins.split()[1])
        if simm >= 0x8000:
            simm -= 0x10000
        return index_of.get(body[i + 1][0] + 4 * simm)

    # basic blocks
    leaders = {0}
    for i, (_, ins) in enumerate(body):
        op = ins.split()[0]
        if op == "s_branch" or op.startswith("s_cbranch"):
            t = target(i)
            if t is not None:
                leaders.add(t)
            if i + 1 < n:
                leaders.add(i + 1)
        elif op == "s_endpgm" and i + 1 < n:
            leaders.add(i + 1)
    starts = sorted(leaders)
    block_of = {}
    for b, st in enumerate(starts):
        for i in range(st, starts[b + 1] if b + 1 < len(starts) else n):
            block_of[i] = b
    succs = [[] for _ in starts]
    for b, st in enumerate(starts):
        end = (starts[b + 1] if b + 1 < len(starts) else n) - 1
        op = body[end][1].split()[0]
        if op == "s_endpgm":
            continue
        if op == "s_branch" or op.startswith("s_cbranch"):
            t = target(end)
            if t is not None:
                succs[b].append(block_of[t])
            if op == "s_branch":
                continue
        if end + 1 < n:
            succs[b].append(block_of[end + 1])

    sets = sum(1 for _, ins in body if ins.split()[0] == "s_load_dwordx4")
    pend_in = [set() for _ in starts]
    bad = {}

    def run_block(b, record):
        pending = set(pend_in[b])
        st = starts[b]
        end = starts[b + 1] if b + 1 < len(starts) else n
        for i in range(st, end):
            ins = body[i][1]
            op = ins.split()[0]
            if op == "s_waitcnt" and "lgkmcnt(0)" in ins:
                pending.clear()
                continue
            touched = sregs(ins)
            if op.startswith("s_load") or op.startswith("s_buffer_load"):
                dst = sregs(ins.split(",")[0])
                hit = touched & pending
                if hit and record:
                    bad[i] = (i, ins, sorted(hit))
                pending |= dst
                continue
            hit = touched & pending
            if hit and record:
                bad[i] = (i, ins, sorted(hit))
        return pending

    work = list(range(len(starts)))
    while work:
        b = work.pop()
        out = run_block(b, False)
        for s2 in succs[b]:
            if not out <= pend_in[s2]:
                pend_in[s2] |= out
                work.append(s2)
    for b in range(len(starts)):
        run_block(b, True)
    return sets, [bad[i] for i in sorted(bad)]


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "distributed-decisiontrees_amd", "lib", "libddt.so")
    if not os.path.exists(OBJDUMP):
        print(f"check_s2_isa: no llvm-objdump found ({OBJDUMP}): the _s2 kernels of {lib} were NOT checked", file=sys.stderr)
        return EXIT_CANNOT_RUN
    try:
        dis = disassemble(lib)
    except (OSError, subprocess.CalledProcessError) as ex:
        print(f"check_s2_isa: could not disassemble {lib}: {ex}: the _s2 kernels were NOT checked", file=sys.stderr)
        return EXIT_CANNOT_RUN
    n_kernels, n_sets, failed = 0, 0, 0
    for name, body in kernels(dis):
        if "score_q16_kernel" not in name and "score_q16p_kernel" not in name:
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
