#!/usr/bin/env python3
"""Check the chunk-DMA waits of lib/libddt.so at the instruction level.

The kernels stage model chunks with `global_load_lds_dwordx4` (global -> LDS DMA through M0, issued from inline asm: hipcc does not
track it) and publish them with an s_barrier.  Two kinds of wait stand in front of such a barrier:

* `s_waitcnt vmcnt(0)`: everything has landed -- always safe;
* `s_waitcnt vmcnt(N)`, N > 0 (csrc/ddt_deep.hip wait_for_dma): vector-memory operations return in order, so the DMA has landed once at
  most N operations are outstanding PROVIDED at least N operations were issued behind it.  That holds for the source (every step issues
  all of its gathers, valid or not) -- but only the instruction stream hipcc actually emitted decides: a gather that was dropped, merged,
  narrowed away or made conditional leaves fewer than N behind the DMA, the wait returns early and the walk reads a half-written chunk.

This script proves the invariant for the binary that was built.  Per kernel, a forward data-flow analysis over the control-flow graph
carries "a DMA may be in flight, and at least c vector-memory operations were issued behind the youngest one" (minimum over all paths).
The waits that are MEANT to cover a DMA are the ones a barrier relies on: the last `s_waitcnt vmcnt(N)` in front of an `s_barrier` with no
vector-memory operation issued in between (hipcc's own counted waits stand in front of the instruction that consumes a gathered register
and are followed by more gathers; they may well leave a DMA in flight: double buffering).  For each of them with N > 0: c >= N on every
path that reaches it with a DMA in flight, else the wait can return with the chunk half written -- a
violation.  c > N is reported as "stricter than needed" (correct; the barrier then also waits for c - N gathers).

Usage: check_dma_waits.py [path/to/libddt.so]   (exit status 1 and a listing on a violation, 3 if no disassembler is installed)
"""
import importlib.util
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("check_s2_isa", os.path.join(ROOT, "tools", "check_s2_isa.py"))
s2 = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(s2)

CAP = 64  # vmcnt is a 6-bit counter: more operations behind the DMA than that make no difference
VMEM = ("buffer_load", "buffer_store", "buffer_atomic", "global_load", "global_store", "global_atomic", "flat_load", "flat_store", "flat_atomic",
        "scratch_load", "scratch_store")
VMCNT = re.compile(r"vmcnt\((\d+)\)")


def _cfg(body):
    """basic blocks of [(address, text)]: (starts, succs) -- the same construction as check_s2_isa.check_kernel"""
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
    return starts, succs


def _merge(a, b):
    """state = None (no DMA in flight) or the least number of operations issued behind the youngest DMA; the merge keeps the worse"""
    if a is None:
        return b
    if b is None:
        return a
    return min(a, b)


def check_kernel(name, body):
    """returns (DMA instructions, barrier waits [(index, N, least operations behind the youngest DMA or None)], violations [(index, text, why)])"""
    body = [x if isinstance(x, tuple) else (None, x) for x in body]
    n = len(body)
    if n == 0:
        return 0, [], []
    starts, succs = _cfg(body)
    dmas = sum(1 for _, ins in body if ins.split()[0].startswith("global_load_lds"))
    # state = (c or None, the waits a barrier reached now would rely on)
    state_in = [(None, frozenset())] * len(starts)
    reached = [False] * len(starts)
    reached[0] = True
    at_wait, relied = {}, set()

    def run_block(b, record):
        st, lw = state_in[b]
        end = starts[b + 1] if b + 1 < len(starts) else n
        for i in range(starts[b], end):
            ins = body[i][1]
            op = ins.split()[0]
            if op.startswith("global_load_lds") or (op.startswith("buffer_load") and " lds" in ins):
                st, lw = 0, frozenset()  # the youngest DMA: nothing behind it yet
            elif op.startswith(VMEM):
                lw = frozenset()
                if st is not None:
                    st = min(st + 1, CAP)
            elif op == "s_waitcnt":
                m = VMCNT.search(ins)
                if m:
                    N = int(m.group(1))
                    if record:
                        at_wait[i] = (i, N, st)
                    lw = frozenset([i])
                    if st is not None and st >= N:
                        st = None  # at most N outstanding and >= N issued behind the DMA: it has landed
            elif op == "s_barrier":
                if record:
                    relied.update(lw)
        return st, lw

    work = [0]
    while work:
        b = work.pop()
        out = run_block(b, False)
        for s in succs[b]:
            new = (_merge(state_in[s][0], out[0]), state_in[s][1] | out[1]) if reached[s] else out
            if not reached[s] or new != state_in[s]:
                reached[s] = True
                state_in[s] = new
                work.append(s)
    for b in range(len(starts)):
        if reached[b]:
            run_block(b, True)
    waits, bad = {}, {}
    for i in sorted(relied):
        _, N, st = at_wait[i]
        if N == 0:
            continue
        waits[i] = (i, N, st)
        if st is not None and st < N:
            bad[i] = (i, body[i][1], f"a barrier relies on this s_waitcnt vmcnt({N}), but on some path only {st} vector-memory operation(s) were issued behind the "
                                     "chunk DMA: the wait can return with the DMA in flight")
    return dmas, [waits[i] for i in sorted(waits)], [bad[i] for i in sorted(bad)]


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "distributed-decisiontrees_amd", "lib", "libddt.so")
    if not os.path.exists(s2.OBJDUMP):
        print(f"check_dma_waits: no llvm-objdump found ({s2.OBJDUMP}): the DMA waits of {lib} were NOT checked", file=sys.stderr)
        return s2.EXIT_CANNOT_RUN
    try:
        dis = s2.disassemble(lib)
    except (OSError, s2.subprocess.CalledProcessError) as ex:
        print(f"check_dma_waits: could not disassemble {lib}: {ex}: the DMA waits were NOT checked", file=sys.stderr)
        return s2.EXIT_CANNOT_RUN
    n_kernels = n_dma = n_counted = n_strict = failed = n_deep = n_relied = 0
    for name, body in s2.kernels(dis):
        dmas, waits, bad = check_kernel(name, body)
        if dmas == 0:
            continue
        n_kernels += 1
        n_dma += dmas
        n_relied += len(waits)
        covering = [w for w in waits if w[2] is not None]  # (None: hipcc's own full wait inside the step has already covered the DMA on every path)
        n_counted += len(covering)
        n_strict += sum(1 for w in covering if w[1] < w[2] < CAP)
        if "score_q16d_kernel" in name:
            n_deep += 1
            if len(waits) < 2 and not bad:  # both chunk barriers of the deep kernels' loop wait with a count: the check must have seen them
                bad = [(0, "", f"only {len(waits)} counted barrier wait(s) found: the check did not see the kernel's chunk barriers")]
        if bad:
            failed += 1
            print(f"{name}: {len(bad)} violation(s)", file=sys.stderr)
            for i, ins, why in bad[:8]:
                print(f"  #{i}: {ins}   <- {why}", file=sys.stderr)
    print(f"{n_kernels} kernels with global->LDS DMA ({n_deep} deep), {n_dma} DMA instructions, {n_relied} counted waits (vmcnt(N), N > 0) that a barrier relies "
          f"on, {n_counted} of them with a DMA possibly in flight ({n_strict} stricter than needed), {failed} kernels with violations")
    return 1 if failed or n_kernels == 0 or n_deep == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
