#!/usr/bin/env python3
"""Golden vectors for the reference's per-device split of the model and tuple streams, produced by EXECUTING
rtl/DTEngine/PCIeReceiver.sv (the host node's stream router) on the registers that EngineCSR.sv derives from the CSR
blocks the PRODUCT's codec emits (ddt_csr_encode_ex of libddt.so, host-only code).

What is executed (procedural-Verilog interpreter of make_schedule_golden.py / make_program_golden.py):
    EngineCSR.sv:146-308      CSR 200-211 writes -> registers (as in make_program_golden.py)
    PCIeReceiver.sv:136-150   the line stamp (prog_mode = weights while numcls_received < total_num_weights_cls, data_valid in
                              RECEIVE_DATA) and the local / SL3 routing assigns
    PCIeReceiver.sv:156-180   the always @(*) block that decides whether a line stays on the host device
    PCIeReceiver.sv:186-316   the receiver FSM with its line counters and the running device index
    InputDistributor.sv:199-232,247-288   on every device: which lines also travel on to the next device (ring re-broadcast) and the
                              counters that stamp `last` on the final line of every tree / tuple for the local core
The input FIFO (vendor-style quick_fifo, absent from the reference) is a pass-through: one line per cycle, both consumers
ready, the distributor empty when asked.  Recorded per stream line: FSM state, prog_mode, data_valid, the running device
index and whether the line stays local -- for the whole model stream (T trees of weights lines, then T trees of
feature-index lines) and the first tuple lines.

tests/test_oracle_receiver.py holds the tree -> device map of the product (ddt_shard_range == the shards
ddt_load_model_shard loads, contiguous ceil(T/G)) to it, and the oracle's multi-device model with it.

One more observation about the published RTL (recorded, asserted, not replicated): (6) the running device index is NOT
reset between the weights and the feature-index stream (:242-264), so when the trees do not divide evenly over the devices
(or a trailing device holds none) the feature-index lines start at the device the weights ended on and every device
receives another device's feature indexes (`ragged_*` cases).  With T a multiple of G * 1 the two streams line up.

Run HERE (needs /root/reference and the built libddt.so); writes tests/golden/receiver_rtl_vectors.npz.
"""
import ctypes
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from make_program_golden import (REF, Params, assigned_names, csr_design, expr, find_instance, package_consts, py_const,  # noqa: E402
                                 rtl_csr_write, subst)
from make_rtl_golden import Evaluator, Module, _strip  # noqa: E402
from make_schedule_golden import Sim, always_blocks  # noqa: E402

OUT = os.path.join(HERE, "receiver_rtl_vectors.npz")
STATES = {"IDLE": 0, "RECEIVE_TREES": 1, "WAIT_DATA": 2, "RECEIVE_DATA": 3}


class Receiver:
    def __init__(self, consts):
        text = _strip(open(f"{REF}/PCIeReceiver.sv").read())
        m = re.search(r"localparam\s*\[1:0\]\s*IDLE\s*=\s*2'b00,\s*RECEIVE_TREES\s*=\s*2'b01,\s*WAIT_DATA\s*=\s*2'b10,\s*RECEIVE_DATA\s*=\s*2'b11", text)
        assert m, "state encoding moved"
        c = dict(consts)
        c.update(STATES)
        text = subst(text, c)
        text = re.sub(r"devices_list\[currDevID\]", "devices_list_sel", text)
        width = {"devices_list_sel": c["DEVICE_ADDRESS_WIDTH"], "input_fifo_valid": 1, "input_fifo_re": 1, "to_local_core": 1,
                 "distributer_empty": 1, "pcie_input_ready": 1, "sl3_output_ready": 1, "prog_mode": 1, "data_valid": 1}
        for rng, name in re.findall(r"\b(?:input\s+wire|output\s+reg|output\s+wire|reg|wire)\s*(\[[^\]]+\])?\s*(\w+)\s*[;,]", text):
            w = 1
            if rng:
                hi, lo = rng[1:-1].split(":")
                w = py_const(hi) - py_const(lo) + 1
            width.setdefault(name, w)
        self.blocks = {}
        for sens, ast, _pos in always_blocks(text):
            names = assigned_names(ast, set())
            if names == {"to_local_core"}:
                self.blocks["route"] = ast
            elif "receiver_fsm_state" in names:
                self.blocks["fsm"] = ast
        assert set(self.blocks) == {"route", "fsm"}, sorted(self.blocks)
        self.mod = Module.__new__(Module)
        self.mod.name, self.mod.inputs, self.mod.outputs, self.mod.cases, self.mod.insts = "PCIeReceiver", [], [], {}, []
        self.mod.width, self.mod.assign = width, {}
        for name in ("input_fifo_re", "pcie_input_valid", "sl3_output_valid", "sl3_output_address"):
            m = re.search(rf"\bassign\s+{name}\s*=\s*([^;]+);", text)
            assert m, name
            self.mod.assign[name] = expr(m.group(1))
        m = re.search(r"\bassign\s+input_line\s*=\s*'\{(.*?)\};", text, re.S)   # the struct literal: two of its members matter
        fields = dict(re.findall(r"(\w+)\s*:\s*(\([^()]*\)|[^,]+)", m.group(1)))
        self.mod.assign["prog_mode"], self.mod.assign["data_valid"] = expr(fields["prog_mode"]), expr(fields["data_valid"])
        self.width = width
        self.regs = sorted(assigned_names(self.blocks["fsm"], set()))

    def run(self, csr_regs, n_model_lines, n_tuple_lines):
        sim = Sim(self.width)
        s = sim.sig
        for k in self.mod.assign:   # wires are evaluated from their assigns, never held
            s.pop(k, None)
        s.update({k: v for k, v in csr_regs.items() if k in self.width})
        lazy = lambda env: Evaluator({}, {k: (v, self.width.get(k, 32)) for k, v in env.items()}, self.mod)
        sim.ev = lambda e, env: lazy(env).ev(e)[0]

        def clock(**inputs):
            s.update(inputs)
            env = dict(s)
            sim.run(self.blocks["route"], env, None, True)           # always @(*): blocking, settles in one pass (no feedback)
            s["to_local_core"] = env["to_local_core"]
            ev = lazy(env)
            snap = {k: ev.get(k)[0] for k in ("prog_mode", "data_valid", "input_fifo_re", "pcie_input_valid", "sl3_output_valid")}
            snap.update(state=env["receiver_fsm_state"], dev=env["currDevID"], local=env["to_local_core"])
            nxt = {}
            sim.run(self.blocks["fsm"], env, nxt, False)
            s.update(nxt)
            return snap

        base = dict(pcie_input_ready=1, sl3_output_ready=1, distributer_empty=1, process_done=0, input_fifo_valid=0)
        clock(rst_n=0, start_core=0, **base)
        clock(rst_n=1, start_core=1, **base)
        assert s["receiver_fsm_state"] == STATES["RECEIVE_TREES"], "the host node did not start receiving trees"
        rec = []
        for _ in range(n_model_lines):
            r = clock(start_core=0, **dict(base, input_fifo_valid=1))
            assert r["state"] == STATES["RECEIVE_TREES"] and r["input_fifo_re"] and r["pcie_input_valid"] == r["local"] != r["sl3_output_valid"], (r, len(rec))
            rec.append((r["state"], r["prog_mode"], r["data_valid"], r["dev"], r["local"]))
        guard = 0
        while s["receiver_fsm_state"] != STATES["RECEIVE_DATA"]:
            clock(**base)
            guard += 1
            assert guard < 8, "the FSM did not reach RECEIVE_DATA"
        for _ in range(n_tuple_lines):
            r = clock(**dict(base, input_fifo_valid=1))
            assert r["state"] == STATES["RECEIVE_DATA"] and r["data_valid"] == 1
            rec.append((r["state"], r["prog_mode"], r["data_valid"], r["dev"], r["local"]))
        return np.array(rec, np.uint8)


def chain_vectors(consts):
    """The result chain of the tree-sharded mode, device by device: ResultsCombiner.sv:355-393 (what a device puts on the SL3
    result link) and :422-453 (what the host hands to PCIe) EXECUTED as the combinational blocks they are, the four hop adders
    (:292-311) evaluated by make_rtl_golden.py's elaboration, the next-hop addresses from the codec's CSR 206 through the executed
    EngineCSR.sv (results_address: next device of the list, the last one closes the ring at the host).  Device 0 (host) sends its
    local line unchanged, every other device sends local + upstream, the host forwards what comes back to PCIe without adding."""
    from make_rtl_golden import chain_hop_module, load_modules, rtl_chain_hop, SRC

    text = subst(_strip(open(f"{REF}/ResultsCombiner.sv").read()), consts)
    blocks = {}
    for sens, ast, _pos in always_blocks(text):
        names = assigned_names(ast, set())
        if names == {"sl3_result_line", "sl3_result_line_valid"}:
            blocks["sl3"] = ast
        elif names == {"pcie_result_line", "pcie_result_line_valid"}:
            blocks["pcie"] = ast
    assert set(blocks) == {"sl3", "pcie"}, sorted(blocks)
    width = {"host_node": 1, "aggregEnabled": 1, "arbiter_state": 1, "aggreg_core_result_dout": 128, "aggreg_core_result_dout_valid": 1,
             "aggreg_sl3_result_dout": 128, "aggreg_sl3_result_valid": 1, "aggregate_result_line": 128, "aggreg_result_line_valid": 1,
             "sl3_result_line": 128, "sl3_result_line_valid": 1, "pcie_result_line": 128, "pcie_result_line_valid": 1}
    sim = Sim(width)
    mods, hop = load_modules(SRC), chain_hop_module()
    pack = lambda v: sum(int(x) << (32 * i) for i, x in enumerate(v))
    unpack = lambda line: [(line >> (32 * j)) & 0xFFFFFFFF for j in range(4)]

    def device_out(host, local4, upstream4):
        """-> (line put on the SL3 result link or None, line handed to PCIe or None, exception codes of the hop adders)"""
        agg, exc = (pack(local4), [1, 1, 1, 1]) if upstream4 is None else (lambda r: (pack(r[0]), r[1]))(rtl_chain_hop(mods, hop, local4, upstream4))
        env = {"host_node": int(host), "aggregEnabled": 1, "arbiter_state": 0, "aggreg_core_result_dout": pack(local4),
               "aggreg_core_result_dout_valid": 1, "aggreg_sl3_result_dout": 0 if upstream4 is None else pack(upstream4),
               "aggreg_sl3_result_valid": int(upstream4 is not None), "aggregate_result_line": agg, "aggreg_result_line_valid": int(upstream4 is not None),
               "sl3_result_line": 0, "sl3_result_line_valid": 0, "pcie_result_line": 0, "pcie_result_line_valid": 0}
        sim.run(blocks["sl3"], env, None, True)
        sim.run(blocks["pcie"], env, None, True)
        return (unpack(env["sl3_result_line"]) if env["sl3_result_line_valid"] else None,
                unpack(env["pcie_result_line"]) if env["pcie_result_line_valid"] else None, exc)

    rng = np.random.default_rng(23)
    out = {}
    for G in (2, 3, 5, 8):
        lines = 40
        parts = ((rng.random((G, lines, 4)) - 0.5) * 8.0).astype(np.float32).view(np.uint32)
        final, clean = np.zeros((lines, 4), np.uint32), np.ones((lines, 4), bool)
        for ln in range(lines):
            sl3, pcie, _ = device_out(True, parts[0, ln], None)          # host: local line onto the link, nothing for PCIe yet
            assert sl3 == [int(x) for x in parts[0, ln]] and pcie is None
            up = sl3
            for d in range(1, G):
                sl3, pcie, exc = device_out(False, parts[d, ln], up)      # device d: local + upstream onto the link
                assert pcie is None and sl3 is not None
                clean[ln] &= np.array(exc) != 0                           # exception 00: the hop forwards garbage (defect, section 2)
                up = sl3
            _, pcie, _ = device_out(True, parts[0, ln], up)               # back at the host: forwarded to PCIe as it is
            assert pcie == up
            final[ln] = pcie
        out[f"chain_parts_{G}"], out[f"chain_final_{G}"], out[f"chain_clean_{G}"] = parts, final, clean
        print(f"chain of {G} devices: {lines} result lines, {int(clean.sum())} of {clean.size} words free of exact cancellation")
    return out


class Distributor:
    """InputDistributor.sv: the line counters that stamp `last` on the final line of every tree / tuple for the local core
    (:247-288) and the combinational block that decides whether a line also travels on to the next device (:199-232)."""

    def __init__(self, consts):
        text = subst(_strip(open(f"{REF}/InputDistributor.sv").read()), consts)
        text = re.sub(r"(core_input_fifo_dout|input_fifo_dout)\.(\w+)", r"\1_\2", text)
        width = {"rst_n": 1, "start_core": 1, "core_input_fifo_valid": 1, "core_output_ready": 1, "core_input_fifo_dout_data_valid": 1,
                 "core_input_fifo_dout_prog_mode": 1, "input_fifo_valid": 1, "input_fifo_dout_data_valid": 1, "broadcast_data": 1, "broadcast_trees": 1,
                 "last_node": 1, "dest_input_fifo_full": 1, "core_input_fifo_full": 1, "core_input_fifo_we": 1, "dest_input_fifo_we": 1, "input_fifo_re": 1,
                 "tree_weights_numcls_minus_one": 16, "tree_feature_index_numcls_minus_one": 16, "tuple_numcls_minus_one": 16}
        for name in ("received_cl_count", "received_weight_cl_count", "received_findex_cl_count", "received_tuple_cl_count"):
            width[name] = 16
        self.blocks = {}
        for sens, ast, _pos in always_blocks(text):
            names = assigned_names(ast, set())
            if "received_weight_cl_count" in names:
                self.blocks["count"] = ast
            elif names == {"core_input_fifo_we", "dest_input_fifo_we", "input_fifo_re"}:
                self.blocks["route"] = ast
        assert set(self.blocks) == {"count", "route"}, sorted(self.blocks)
        self.mod = Module.__new__(Module)
        self.mod.name, self.mod.inputs, self.mod.outputs, self.mod.cases, self.mod.insts, self.mod.width, self.mod.assign = "dist", [], [], {}, [], width, {}
        for name in ("single_tree_weights_received", "single_tree_feature_indexes_received", "single_tuple_features_received", "data_last_flag"):
            m = re.search(rf"\bassign\s+{name}\s*=\s*([^;]+);", text)
            assert m, name
            self.mod.assign[name] = expr(m.group(1))
            width[name] = 1
        self.width = width

    def lasts(self, regs, stamps):
        """stamps: (data_valid, prog_mode) per line in arrival order -> the `last` flag the local core sees with each line"""
        sim = Sim(self.width)
        s = sim.sig
        for k in self.mod.assign:
            s.pop(k, None)
        s.update({k: regs[k] for k in ("tree_weights_numcls_minus_one", "tree_feature_index_numcls_minus_one", "tuple_numcls_minus_one")})
        lazy = lambda env: Evaluator({}, {k: (v, self.width.get(k, 32)) for k, v in env.items()}, self.mod)
        sim.ev = lambda e, env: lazy(env).ev(e)[0]
        nxt = {}
        sim.run(self.blocks["count"], dict(s, rst_n=0), nxt, False)
        s.update(nxt)
        out = []
        for dv, pm in stamps:
            env = dict(s, rst_n=1, start_core=0, core_input_fifo_valid=1, core_output_ready=1, core_input_fifo_dout_data_valid=int(dv),
                       core_input_fifo_dout_prog_mode=int(pm))
            out.append(lazy(env).get("data_last_flag")[0])
            nxt = {}
            sim.run(self.blocks["count"], env, nxt, False)
            s.update(nxt)
        return np.array(out, np.uint8)

    def route(self, data_valid, broadcast_data, broadcast_trees, last_node):
        sim = Sim(self.width)
        env = dict(sim.sig, input_fifo_valid=1, input_fifo_dout_data_valid=int(data_valid), broadcast_data=int(broadcast_data),
                   broadcast_trees=int(broadcast_trees), last_node=int(last_node), dest_input_fifo_full=0, core_input_fifo_full=0)
        sim.run(self.blocks["route"], env, None, True)
        return env["core_input_fifo_we"], env["dest_input_fifo_we"]


def main():
    if not os.path.exists(REF):
        sys.exit(f"{REF} not found: run this in the build container")
    consts = package_consts()
    design = csr_design(consts)
    rx = Receiver(consts)
    dist = Distributor(consts)
    # the registers reach these modules under their own names: DTInference.sv connects every parameter port of the receiver and the
    # distributor to the wire the EngineCSR output of the same name drives (only the SL3 / core stream ports are renamed)
    top = _strip(open(f"{REF}/DTInference.sv").read())
    _, csr_ports = find_instance(top, "EngineCSR")
    for typ, used in (("PCIeReceiver", ("pcie_receiver_enabled", "host_node", "data_distributed", "broadcast_trees", "broadcast_data",
                                        "core_data_batch_cls_minus_one", "total_num_trees_cls", "total_num_weights_cls", "numcls_local_weights_minus_one",
                                        "numcls_local_findexes_minus_one", "numDevs_minus_one")),
                      ("InputDistributor", ("broadcast_data", "broadcast_trees", "last_node", "tree_weights_numcls_minus_one",
                                            "tree_feature_index_numcls_minus_one", "tuple_numcls_minus_one"))):
        _, ports = find_instance(top, typ)
        for name in used:
            assert ports[name] == name == csr_ports[name], (typ, name, ports.get(name), csr_ports.get(name))
    lib = ctypes.CDLL(os.path.join(ROOT, "distributed-decisiontrees_amd", "lib", "libddt.so"))
    enc = lib.ddt_csr_encode_ex
    enc.argtypes = [ctypes.POINTER(Params), ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]
    # T, D, F, devices, mode (0 = trees sharded + tuples broadcast, 1 = trees broadcast + tuples dealt), extra lines per tree
    cases = [(8, 2, 8, 1, 0, 0), (16, 3, 13, 2, 0, 0), (24, 2, 8, 8, 0, 1), (40, 4, 32, 4, 0, 0), (20, 1, 5, 20, 0, 0), (12, 5, 28, 3, 0, 2),
             (16, 3, 16, 4, 1, 0), (9, 2, 8, 3, 1, 0), (8, 4, 64, 8, 1, 0),
             (9, 2, 8, 4, 0, 0), (10, 3, 16, 3, 0, 0), (37, 2, 12, 8, 0, 0)]       # ragged: T % G != 0 (9 / 4 also leaves device 3 empty)
    out = {"cases": [], "wiring": np.array(sorted(rx.regs))}
    for (T, D, F, G, mode, pad) in cases:
        wl, fl, tl = ((1 << (D + 1)) - 1 + 3) // 4 + pad, ((1 << D) - 1 + 7) // 8 + pad, (F + 3) // 4
        p = Params(T, D, F, 0x7FC00000, wl, fl, 0, 1, 0)
        buf = (ctypes.c_uint64 * 12)()
        n_tuples = 64
        assert enc(ctypes.byref(p), n_tuples, G, mode, 0, buf) == 0
        regs = rtl_csr_write(design, [int(x) for x in buf])
        regs["start_core"] = 0
        rec = rx.run(regs, T * (wl + fl), n_tuples * tl)
        key = f"{T}_{D}_{F}_{G}_{mode}_{pad}"
        out["cases"].append((T, D, F, G, mode, pad, wl, fl, tl, n_tuples))
        out["rec_" + key] = rec
        out["last_" + key] = dist.lasts(regs, [(int(r[2]), int(r[1])) for r in rec])       # the stamps the receiver gave the lines
        out["csr_" + key] = np.array([int(x) for x in buf], np.uint64)
        w, f = rec[:T * wl], rec[T * wl:T * (wl + fl)]
        print(f"T={T} D={D} G={G} mode={mode}: weights -> devices {w[::wl, 3].tolist()[:12]} findex -> {f[::fl, 3].tolist()[:12]} "
              f"tuples -> {rec[T * (wl + fl)::tl, 3].tolist()[:10]}")
    out["cases"] = np.array(out["cases"], np.uint64)
    # does a line travel on to the next device?  (data line / tree line) x (broadcast_data, broadcast_trees, last_node)
    out["route"] = np.array([(dv, bd, bt, ln, *dist.route(dv, bd, bt, ln)) for dv in (0, 1) for bd in (0, 1) for bt in (0, 1) for ln in (0, 1)], np.uint8)
    out.update(chain_vectors(consts))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
