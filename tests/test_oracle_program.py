"""The PROGRAMMING side of the reference -- parameter registers, PU control word, where the model and tuple streams land
in the PU memories, the per-tree base offsets, EMPTY slots -- pinned to the reference's own RTL.

tests/golden/csr_rtl_vectors.npz and program_rtl_vectors.npz were produced by tests/golden/make_program_golden.py, which
EXECUTES the text of EngineCSR.sv:146-308, the EngineCSR -> Core wiring of DTInference.sv, Core.sv:380,
core/DTPU.sv:304-354,379-399,429-447,459-460,512-567 (procedural-Verilog interpreter), core/Mem1in2out.v,
core/dualport_mem.v and the elaborated generate blocks of core/PipelinedMUX.sv, and walks every (tuple, tree slot) through
the traversal datapath of DTPU.sv:579-760 with all memory reads going through those wrappers.

Held to it here:
  * the oracle's reading of the wire format (orc_leaves: word n of a tree's weights lines = node n, leaves after the
    2^D - 1 internal nodes, 16-bit feature-index entry n, feature j of a tuple, per-tree stride = lines per tree,
    little-endian lines) -- bit for bit, on streams whose padding words are junk;
  * EMPTY slots read as +0;
  * the PRODUCT's CSR codec (ddt_csr_encode_ex of libddt.so, host-only code): the block it emits today is the block the
    RTL was executed on, and the registers / PU parameters the RTL derived from it are the run parameters.
Four defects of the published RTL that the script surfaced are asserted as recorded (documented, not replicated)."""
import ctypes as C
import os

import numpy as np
import pytest

import ddt
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def prog():
    return np.load(os.path.join(GOLD, "program_rtl_vectors.npz"))


@pytest.fixture(scope="module")
def csr():
    return np.load(os.path.join(GOLD, "csr_rtl_vectors.npz"))


def _cases(v):
    """Slice the concatenated blobs back into per-case arrays."""
    pos = {k: 0 for k in ("wlines", "flines", "tuples", "out", "instr")}
    for i in range(len(v["D"])):
        g = {k: int(v[k][i]) for k in ("pu_id", "D", "F", "slots", "K", "wl", "fl", "tl", "missing", "n_tuples")}
        size = {"wlines": g["K"] * g["wl"] * 4, "flines": g["K"] * g["fl"] * 8, "tuples": g["n_tuples"] * g["tl"] * 4,
                "out": g["n_tuples"] * g["slots"], "instr": g["n_tuples"] * g["slots"]}
        for k, n in size.items():
            g[k] = v[k][pos[k]:pos[k] + n]
            pos[k] += n
        yield g
    assert all(pos[k] == len(v[k]) for k in pos)


def _oracle_leaves(K, D, F, wl, fl, missing, wlines, flines, tuples):
    p = O.Params(K, D, F, missing, wl, fl, 0, 1)
    m = O.Model(p, wlines, flines)
    return np.stack([O.leaves(m, row) for row in tuples.reshape(-1, (F + 3) // 4 * 4)])


def test_the_oracle_reads_the_streams_the_way_the_programmed_pu_does(prog):
    walks = 0
    for g in _cases(prog):
        rtl = g["out"].reshape(g["n_tuples"], g["slots"])
        want = _oracle_leaves(g["K"], g["D"], g["F"], g["wl"], g["fl"], g["missing"], g["wlines"], g["flines"], g["tuples"])
        assert np.array_equal(rtl[:, :g["K"]], want), (g["pu_id"], g["D"], g["F"])
        assert not rtl[:, g["K"]:].any(), "slots >= local_num_trees are EMPTY: leaf +0 (DTPU.sv:544,587,760)"
        walks += rtl.size
    assert walks >= 200


def test_padded_strides_and_junk_padding_are_covered(prog):
    gs = list(_cases(prog))
    assert any(g["wl"] > ((1 << (g["D"] + 1)) - 1 + 3) // 4 for g in gs) and any(g["fl"] > ((1 << g["D"]) - 1 + 7) // 8 for g in gs)
    assert any(g["K"] < g["slots"] for g in gs) and any(g["D"] == 8 and g["K"] == 15 for g in gs)
    g = gs[2]   # junk, not zeros, behind the last leaf of every tree
    nodes = (1 << (g["D"] + 1)) - 1
    assert g["wlines"].reshape(g["K"], -1)[:, nodes:].all()


def test_instruction_offsets_advance_by_the_control_words_strides(prog):
    tob, tub = (int(x) for x in prog["instr_fields"])     # TREE_OFFSET_BITS, TUPLE_OFFSET_BITS (DTPU.sv:84-86,554)
    for g in _cases(prog):
        ins = g["instr"].reshape(g["n_tuples"], g["slots"]).astype(np.uint64)
        w_off, f_off = ins & np.uint64((1 << tob) - 1), (ins >> np.uint64(tob)) & np.uint64((1 << tob) - 1)
        t_off = (ins >> np.uint64(2 * tob)) & np.uint64((1 << tub) - 1)
        empty, last = (ins >> np.uint64(2 * tob + tub)) & np.uint64(1), (ins >> np.uint64(2 * tob + tub + 1)) & np.uint64(1)
        k = np.arange(g["slots"], dtype=np.uint64)
        assert (w_off == k * np.uint64(g["wl"])).all() and (f_off == k * np.uint64(g["fl"])).all()
        assert (t_off == (np.arange(g["n_tuples"], dtype=np.uint64) * np.uint64(g["tl"]))[:, None]).all()   # ring: tuple k at line k * tl
        assert (empty == (k >= g["K"])).all() and (last == (k == g["slots"] - 1)).all()


def test_recorded_defects_of_the_published_rtl(prog):
    # (3) an idle cycle carrying the PU's id writes the feature-index memory (TFI_wen has no valid qualifier, DTPU.sv:343)
    assert int(prog["idle_tfi_advance"][0]) == 1
    # (4) without a tuple line on the input the weights read address is the programming pointer (DTPU.sv:599)
    assert int(prog["idle_read_hits_prog_addr"][0]) == 1
    # (5) a PU given its full 16 trees counts 16 mod 16 = 0 and flags every slot EMPTY (DTPU.sv:115,316,544)
    assert int(prog["full_pu_local_num_trees"][0]) == 0 and int(prog["full_pu_all_zero"][0]) == 1
    # (1) with the published wiring (stride = lines per tree - 1) tree k is looked up k lines too early: slot 0 still agrees
    #     with the oracle, later slots do not
    tob = int(prog["instr_fields"][0])
    wl, fl = int(prog["quirk_wl"][0]), int(prog["quirk_fl"][0])
    ins = prog["quirk_instr"].astype(np.uint64)
    assert (ins & np.uint64((1 << tob) - 1)).tolist() == [k * (wl - 1) for k in range(4)]
    want = _oracle_leaves(4, 3, 8, wl, fl, 0x7FC00000, prog["quirk_wlines"], prog["quirk_flines"], prog["quirk_tuples"])
    got = prog["quirk_out"].reshape(1, 4)
    assert got[0, 0] == want[0, 0] and not np.array_equal(got, want)


def test_scores_fill_result_lines_four_to_a_line_in_tuple_order(prog):
    """ResultsCombiner.sv:131-160,193 executed: word k of result line L is score 4L + k -- the fp32 array in tuple order that
    ddt_score returns (and ddt_cli writes, padded to whole lines) read as little-endian 128-bit lines; an unfinished line is not
    emitted (the reference needs N % 4 == 0, the library does not)."""
    scores, lines = prog["result_scores"], prog["result_lines"]
    assert lines.shape == (len(scores) // 4, 4)
    assert np.array_equal(lines.reshape(-1), scores[:lines.size])


def test_accumulator_with_its_control_is_sequential_from_three_cycles_apart(prog):
    """core/FPAggregator.v executed as a whole (clocked block by the interpreter, 2-stage adder and delay as pipelines, the absent
    quick_fifo as a show-ahead FIFO): values that arrive >= 3 cycles apart are summed strictly in arrival order -- orc_aggregate,
    acc <- x + acc, the order the oracle and the engine implement.  Closer together the published module is defective (#8 of
    the table in profiles/EXPERIMENTS.md): it pops every 2 cycles, a sum re-enters the adder after 3, and the output is the interleaved chain that
    holds the last value -- there is no other "reference order" hiding there, just lost addends."""
    lens, spacings, out = prog["agg_lengths"], [int(x) for x in prog["agg_spacings"]], prog["agg_out"]
    L = O.lib()
    pos = 0
    for i, n in enumerate(int(x) for x in lens):
        v = np.ascontiguousarray(prog["agg_values"][pos:pos + n])
        pos += n
        want = L.orc_aggregate(v.ctypes.data, n)
        chain = np.ascontiguousarray(v[(n - 1) % 2::2])                       # the values of the last one's parity, in order
        lost = L.orc_aggregate(chain.ctypes.data, len(chain))
        for j, sp in enumerate(spacings):
            if sp >= 3:
                assert int(out[i, j]) == want, (n, sp)
            else:
                assert int(out[i, j]) == lost, (n, sp)
                assert n < 3 or lost != want


# ---------------------------------------------------------------------------------------------------- the CSR chain
COLS = ("T", "D", "F", "C", "missing", "wl", "fl", "n", "devices", "mode", "index")


def _rows(csr):
    return [dict(zip(COLS, (int(x) for x in row))) for row in csr["cases"]]


def test_the_codec_still_emits_the_blocks_the_rtl_was_executed_on(csr):
    L = ddt.lib()
    for r, words in zip(_rows(csr), csr["csr"]):
        p = ddt.Params()
        p.num_trees, p.num_levels, p.num_features, p.missing_bits = r["T"], r["D"], r["F"], r["missing"]
        p.weights_lines_per_tree, p.findex_lines_per_tree, p.clusters_per_tuple = r["wl"], r["fl"], r["C"]
        buf = (C.c_uint64 * 12)()
        assert L.ddt_csr_encode_ex(C.byref(p), r["n"], r["devices"], r["mode"], r["index"], C.byref(buf)) == 0
        assert [int(x) for x in buf] == [int(x) for x in words], r
    assert len(csr["cases"]) >= 40


def test_rtl_registers_derived_from_the_codec_blocks_are_the_run_parameters(csr):
    for i, r in enumerate(_rows(csr)):
        reg = lambda name: int(csr["reg_" + name][i])
        tl = (r["F"] + 3) // 4
        assert reg("tree_weights_numcls_minus_one") == r["wl"] - 1 and reg("tree_feature_index_numcls_minus_one") == r["fl"] - 1
        assert reg("tuple_numcls") == tl and reg("tuple_numcls_minus_one") == tl - 1
        assert reg("missing_value") == r["missing"] and reg("num_levels_per_tree_minus_one") == (r["D"] - 1) & 0xF
        assert reg("num_clusters_per_tuple") == r["C"] and reg("num_clusters_per_tuple_minus_one") == r["C"] - 1
        assert reg("total_num_weights_cls") == r["T"] * r["wl"] and reg("total_num_trees_cls") == r["T"] * (r["wl"] + r["fl"])
        assert reg("numDevs_minus_one") == r["devices"] - 1
        rows_mode, multi = r["mode"] == 1, r["devices"] > 1
        per_dev = r["T"] if rows_mode else -(-r["T"] // r["devices"])
        slots = -(-(-(-per_dev // 8)) // r["C"])                       # ceil(ceil(trees / 8 PUs) / C clusters) tree slots per PU
        assert reg("num_trees_per_pu_minus_one") == slots - 1
        if per_dev * r["wl"] - 1 <= 0xFFFF and per_dev * r["fl"] <= 0xFFFF:   # the per-device line counters (PCIeReceiver.sv:241-264)
            assert reg("numcls_local_weights_minus_one") == per_dev * r["wl"] - 1
            assert reg("numcls_local_findexes_minus_one") == per_dev * r["fl"] - 1
        # one model replica per C clusters, the first tuple on clusters 0..C-1 (the schedule test starts from these values)
        assert reg("prog_schedule") == sum(1 << k for k in range(0, 8, r["C"])) and reg("proc_schedule") == (1 << r["C"]) - 1
        # mode flags (EngineCSR.sv:194-205)
        assert reg("host_node") == int(r["index"] == 0) and reg("pcie_receiver_enabled") == int(r["index"] == 0)
        assert reg("last_node") == int(r["index"] == r["devices"] - 1) and reg("multiple_nodes") == int(multi)
        assert reg("broadcast_data") == int(multi and not rows_mode) and reg("aggreg_enabled") == int(multi and not rows_mode)
        assert reg("broadcast_trees") == int(multi and rows_mode) and reg("data_distributed") == 0
        assert reg("total_results_numcls") == (r["n"] + 3) // 4 or (rows_mode and multi and r["index"] > 0)
        for d in range(20):
            assert reg(f"devices_list__{d}") == (d if d < r["devices"] else 0), (r, d)
        nxt = 0 if r["index"] == r["devices"] - 1 else r["index"] + 1
        assert reg("broadcast_address") == nxt and reg("results_address") == nxt


def test_pu_parameters_decoded_from_the_control_word(csr):
    """Core.sv:380 -> DTPU.sv:438-446, fed through the wiring of DTInference.sv as published."""
    wiring = dict(x.split("=") for x in csr["core_wiring"])
    # defect (1): the stride ports of Core are wired to the *_minus_one registers (DTInference.sv:505-506)
    assert wiring["tree_weights_numcls"] == "tree_weights_numcls_minus_one"
    assert wiring["tree_feature_index_numcls"] == "tree_feature_index_numcls_minus_one"
    assert wiring["tuple_numcls"] == "tuple_numcls" and wiring["missing_value"] == "missing_value"
    for i, r in enumerate(_rows(csr)):
        pu = lambda name: int(csr["pu_" + name][i])
        assert pu("LastLevelIndex") == (r["D"] - 1) & 0xF and pu("MissingFeatureValue") == r["missing"]
        assert pu("tuple_numlines") == (r["F"] + 3) // 4 and pu("PartialTrees") == 0
        assert pu("num_trees_per_pu_minus_one") == int(csr["reg_num_trees_per_pu_minus_one"][i]) & 0xF   # 4 bits in the PU (DTPU.sv:70,439)
        assert pu("num_lines_per_tree_weights") == (r["wl"] - 1) & 0x3FF       # the published wiring: lines - 1, 10 bits (DTPU.sv:442)
        assert pu("num_lines_per_tree_findex") == (r["fl"] - 1) & 0x3FF


def test_product_decoder_agrees_with_the_rtl_on_random_register_writes(csr):
    L = ddt.lib()
    checked = 0
    for i, words in enumerate(csr["rand_csr"]):
        reg = lambda name: int(csr["rand_" + name][i])
        buf = (C.c_uint64 * 12)(*[int(x) for x in words])
        p, n, nd, mode, flags = ddt.Params(), C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        ids = (C.c_uint8 * 20)()
        rc = L.ddt_csr_decode_ex(C.byref(buf), C.byref(p), C.byref(n), C.byref(nd), C.byref(mode), C.byref(flags), ids)
        # the id slices and the flag byte do not depend on the block being a consistent parameter set
        want_ids = [reg(f"devices_list__{d}") for d in range(20)]
        want_flags = (reg("data_distributed") | reg("host_node") << 1 | reg("broadcast_data") << 2 | reg("broadcast_trees") << 3 |
                      reg("aggreg_enabled") << 4 | reg("multiple_nodes") << 5 | reg("pcie_receiver_enabled") << 6 | reg("last_node") << 7)
        if rc == 0:
            assert list(ids) == want_ids and flags.value == want_flags
            assert p.missing_bits == reg("missing_value") and p.clusters_per_tuple == reg("num_clusters_per_tuple")
            checked += 1
        # the RTL's own slices, recomputed here from the words (EngineCSR.sv:250-296): what ddt_csr_decode_ex must mirror
        for d in range(20):
            assert want_ids[d] == (int(words[8 + d // 8]) >> (8 * (d % 8))) & 0x1F
        assert want_flags == int(words[1]) & 0xFF
        assert reg("tree_weights_numcls_minus_one") == ((int(words[4]) >> 16) - 1) & 0xFFFF
        assert reg("num_levels_per_tree_minus_one") == ((int(words[5]) >> 32) - 1) & 0xF
    assert checked >= 0
