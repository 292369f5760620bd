"""The `_s2` kernels keep node records in SGPRs that inline-asm scalar loads fill one sub-group ahead; hipcc does not know the
registers are still in flight.  tools/check_s2_isa.py proves on the built binary that nothing touches them before the kernel's
own full wait; this test runs it on lib/libddt.so, and on a hand-made listing with a copy in flight (the checker must see it)."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_s2_isa", os.path.join(ROOT, "tools", "check_s2_isa.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

LIB = os.path.join(ROOT, "distributed-decisiontrees_amd", "lib", "libddt.so")


@pytest.mark.skipif(not os.path.exists(chk.OBJDUMP), reason="llvm-objdump not installed")
def test_built_s2_kernels_never_touch_records_in_flight():
    assert os.path.exists(LIB), "build the library first (__graft_entry__.build())"
    dis = chk.disassemble(LIB)
    seen = 0
    for name, body in chk.kernels(dis):
        if "score_q16_kernel" not in name and "score_q16p_kernel" not in name:
            continue
        sets, bad = chk.check_kernel(name, body)
        if sets < 8:
            continue
        seen += 1
        assert not bad, f"{name}: {bad[:4]}"
    assert seen >= 4  # depth 5, 6, 7 and the depth-8 _gl_s2 kernel


def test_checker_sees_a_copy_in_flight():
    ok = ["s_load_dwordx4 s[20:23], s[4:5], 0x0", "v_add_u32_e32 v1, v2, v3", "s_waitcnt lgkmcnt(0)", "s_mov_b64 s[8:9], s[20:21]"]
    assert chk.check_kernel("k", ok) == (1, [])
    bad = ["s_load_dwordx4 s[20:23], s[4:5], 0x0", "s_mov_b64 s[8:9], s[22:23]", "s_waitcnt lgkmcnt(0)"]
    sets, found = chk.check_kernel("k", bad)
    assert sets == 1 and found and found[0][2] == [22, 23]
    partial = ["s_load_dwordx4 s[20:23], s[4:5], 0x0", "s_waitcnt lgkmcnt(1)", "v_mov_b32_e32 v0, s21", "s_waitcnt vmcnt(0) lgkmcnt(0)"]
    assert chk.check_kernel("k", partial)[1], "a counted wait does not cover scalar loads (they return out of order)"


def test_checker_follows_branches():
    # 0x00 load; 0x08 branch over the wait to 0x14; 0x0c wait; 0x10 (fall-through use, fine); 0x14 use with the load in flight
    listing = [(0x00, "s_load_dwordx4 s[20:23], s[4:5], 0x0"), (0x08, "s_cbranch_scc1 2"), (0x0C, "s_waitcnt lgkmcnt(0)"),
               (0x10, "s_mov_b32 s1, s20"), (0x14, "s_mov_b32 s2, s21"), (0x18, "s_endpgm")]
    sets, found = chk.check_kernel("k", listing)
    assert sets == 1 and [f[0] for f in found] == [4]  # only the instruction the branch reaches without the wait
    # a loop: the load at the bottom is still in flight at the top of the next iteration
    loop = [(0x00, "s_mov_b32 s1, s20"), (0x04, "s_waitcnt lgkmcnt(0)"), (0x08, "s_load_dwordx4 s[20:23], s[4:5], 0x0"),
            (0x10, "s_cbranch_scc1 65531"), (0x14, "s_endpgm")]
    assert [f[0] for f in chk.check_kernel("k", loop)[1]] == [0]


# ---- tools/check_dma_waits.py: the counted waits in front of the chunk barriers -------------------------------------
spec2 = importlib.util.spec_from_file_location("check_dma_waits", os.path.join(ROOT, "tools", "check_dma_waits.py"))
dma = importlib.util.module_from_spec(spec2)
spec2.loader.exec_module(dma)

GATHER = "buffer_load_dwordx4 v[2:5], v1, s[8:11], s2 offen"
DMA = "global_load_lds_dwordx4 v[42:43], off"


@pytest.mark.skipif(not os.path.exists(chk.OBJDUMP), reason="llvm-objdump not installed")
def test_built_kernels_counted_barrier_waits_cover_their_dma():
    assert os.path.exists(LIB), "build the library first (__graft_entry__.build())"
    dis = chk.disassemble(LIB)
    deep = relied = 0
    for name, body in chk.kernels(dis):
        dmas, waits, bad = dma.check_kernel(name, body)
        if dmas == 0:
            continue
        assert not bad, f"{name}: {bad[:4]}"
        if "score_q16d_kernel" in name:
            deep += 1
            assert len(waits) >= 2, (name, waits)   # both chunk barriers of the loop wait with a count: the check has seen them
            relied += len(waits)
    assert deep >= 8 and relied >= 16


def test_checker_counts_the_gathers_behind_a_dma():
    ok = [DMA] + [GATHER] * 8 + ["s_waitcnt vmcnt(8)", "s_add_i32 s2, s22, 1", "s_barrier", "ds_read_b32 v1, v2"]
    dmas, waits, bad = dma.check_kernel("k", ok)
    assert dmas == 1 and waits == [(9, 8, 8)] and not bad
    # one gather dropped (or narrowed away / merged): only 7 behind the DMA, vmcnt(8) can return with the DMA in flight
    short = [DMA] + [GATHER] * 7 + ["s_waitcnt vmcnt(8)", "s_barrier"]
    _, waits, bad = dma.check_kernel("k", short)
    assert waits == [(8, 8, 7)] and bad and bad[0][0] == 8
    # more gathers than the count: correct, the barrier merely also waits for one gather
    _, waits, bad = dma.check_kernel("k", [DMA] + [GATHER] * 9 + ["s_waitcnt vmcnt(8)", "s_barrier"])
    assert waits == [(10, 8, 9)] and not bad
    # a counted wait of hipcc's own (a gathered register is consumed, more gathers follow) may leave the DMA in flight: not a barrier wait
    own = [DMA] + [GATHER] * 2 + ["s_waitcnt vmcnt(4)", "v_add_u32_e32 v9, v2, v3", GATHER, "s_waitcnt vmcnt(0)", "s_barrier"]
    _, waits, bad = dma.check_kernel("k", own)
    assert waits == [] and not bad
    # a full wait in between has covered the DMA: the counted wait in front of the barrier has nothing left to cover
    covered = [DMA, GATHER, "s_waitcnt vmcnt(0)"] + [GATHER] * 4 + ["s_waitcnt vmcnt(4)", "s_barrier"]
    _, waits, bad = dma.check_kernel("k", covered)
    assert waits == [(7, 4, None)] and not bad


def test_checker_sees_a_gather_made_conditional():
    # the deliberately broken build of VERDICT r5 item 5: ONE of the four gathers behind the DMA sits under a branch -- on the path that
    # skips it only three operations stand behind the DMA, and vmcnt(4) can return with the chunk half written
    listing = [(0x00, DMA), (0x08, GATHER), (0x10, GATHER), (0x18, GATHER), (0x20, "s_cbranch_execz 2"), (0x24, GATHER), (0x2C, "s_waitcnt vmcnt(4)"),
               (0x30, "s_barrier"), (0x34, "s_endpgm")]
    _, waits, bad = dma.check_kernel("k", listing)
    assert waits == [(6, 4, 3)] and [b[0] for b in bad] == [6]
    fixed = [(0x00, DMA), (0x08, GATHER), (0x10, GATHER), (0x18, GATHER), (0x20, GATHER), (0x28, "s_waitcnt vmcnt(4)"), (0x2C, "s_barrier"), (0x30, "s_endpgm")]
    assert not dma.check_kernel("k", fixed)[2]
    # a loop: the DMA is issued behind the barrier, four gathers per iteration, the counted wait at the top of the next iteration; the FIRST
    # iteration is covered by a full wait in front of the loop (csrc/ddt_deep.hip run())
    loop = [(0x00, DMA), (0x08, "s_waitcnt vmcnt(0)"), (0x0C, "s_barrier"), (0x10, DMA), (0x18, GATHER), (0x20, GATHER), (0x28, GATHER), (0x30, GATHER),
            (0x38, "s_waitcnt vmcnt(4)"), (0x3C, "s_cbranch_scc1 65523"), (0x40, "s_endpgm")]   # back to 0x0C
    _, waits, bad = dma.check_kernel("k", loop)
    assert waits == [(8, 4, 4)] and not bad
    # ... without that full wait the path "entry -> counted wait" would exist only if the wait stood at the loop's top: the checker follows paths, not values
    top = [(0x00, DMA), (0x08, "s_waitcnt vmcnt(4)"), (0x0C, "s_barrier"), (0x10, DMA), (0x18, GATHER), (0x20, GATHER), (0x28, GATHER), (0x30, GATHER),
           (0x38, "s_cbranch_scc1 65523"), (0x3C, "s_endpgm")]   # back to 0x08
    assert dma.check_kernel("k", top)[2]
