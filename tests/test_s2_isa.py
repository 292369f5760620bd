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
