#!/usr/bin/env python3
"""Golden vectors for the PROGRAMMING side of the reference, produced by EXECUTING THE REFERENCE'S OWN RTL TEXT.

make_rtl_golden.py pins the traversal datapath with the three PU memories modelled as flat arrays that the script itself
fills ("word n of the tree at line*4 + lane"); make_schedule_golden.py pins which cluster / PU a tree is sent to.  What
sits between the two -- how the host's parameter registers become the PU control word, how the model and tuple streams
land in the PU memories, and which base offsets the per-tree instructions carry -- is executed here:

  part 1 (csr_rtl_vectors.npz)      CSR 200-211 writes -> engine registers -> PU control word -> PU parameter registers
      EngineCSR.sv:146-308     the "Write SoftRegs" block, run by the procedural interpreter on CSR blocks that the
                               PRODUCT's codec emits (ddt_csr_encode_ex of libddt.so, host-only code) and on random blocks
      DTInference.sv           the port connections EngineCSR -> Core (this is where the `_minus_one` stride quirk lives)
      Core.sv:380              the control-word concatenation, with Core's declared port widths
      core/DTPU.sv:429-447     the PU's decode of the control word
  part 2 (program_rtl_vectors.npz)  one DTPU programmed and driven line by line
      core/DTPU.sv:304-354     write enables and write pointers of the weights / feature-index memories, local_num_trees
      core/DTPU.sv:379-399     tuple lines into the features ring, tuple_offset
      core/DTPU.sv:459-460,512-567  one instruction per tree slot: base offsets advance by the control word's strides,
                               EMPTY flag for slots >= local_num_trees (both FIFOs modelled as queues: pass-through)
      core/Mem1in2out.v, core/dualport_mem.v   line address / word offset split around the vendor RAMs (the RAM IP itself
                               is absent from the reference: modelled as an array of lines at the address its wrapper passes)
      core/PipelinedMUX.sv     the word select, ELABORATED from its generate blocks for the instance parameters
      core/FPAggregator.v      the accumulator WITH its control, cycle by cycle (part 5): from which input spacing it is sequential
      ResultsCombiner.sv:131-160,193  scores -> 128-bit result lines (four to a line, word k = score 4L + k)
      core/DTPU.sv:579-760     the walk itself (datapath evaluator of make_rtl_golden.py), every memory read going
                               through the wrappers above -- no address or word order is asserted by this script

Stimulus: the lines arrive back to back (what the input FIFO delivers while it is non-empty).  Observations about the
published RTL made while writing this (recorded in the vectors, asserted by tests/test_oracle_program.py, documented in
profiles/EXPERIMENTS.md, last table; none is replicated by oracle or engine):
  (3) TFI_wen = ~mode[0] & ~mode[1] & (pu == PU_ID) has no valid qualifier (DTPU.sv:343): "feature-index line" and "idle"
      are the same encoding, so EVERY idle cycle -- and every line addressed to a disabled cluster, Core.sv:467 -- whose pu
      field equals the PU's id writes the feature-index memory and advances its pointer (`idle_tfi_advance`)
  (4) the weights read address is muxed by mode[0] (DTPU.sv:599): a walk step taken in a cycle without a tuple line on the
      input reads the PROGRAMMING pointer's line instead of the node's (`idle_read_hits_prog_addr`)
  (5) local_num_trees is a 4-bit counter (DTPU.sv:70,96,115,316) compared with the slot index (:544): a PU that is given its
      full 16 trees counts 16 mod 16 = 0 and flags EVERY slot EMPTY -- all its leaves read as 0 (`full_pu_*`); 15 trees work
  (1) the stride fields of the control word receive lines-per-tree MINUS ONE (DTInference.sv:505-506): part 1
      shows it from the executed wiring, part 2 records the mis-strided instruction offsets (`quirk_*`)
The vectors themselves use the evident intent: idle cycles carry another PU's id, tuple lines stream during the walks,
stride = lines per tree.

Run HERE (needs /root/reference and the built libddt.so); writes tests/golden/{csr,program}_rtl_vectors.npz:
    python tests/golden/make_program_golden.py
"""
import ctypes
import os
import re
import sys
from collections import deque

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from make_rtl_golden import Evaluator, Module, Parser, _match, _strip, dtpu_module  # noqa: E402
from make_schedule_golden import Sim, StmtParser, always_blocks, tokenize  # noqa: E402

REF = "/root/reference/rtl/DTEngine"
OUT_CSR = os.path.join(HERE, "csr_rtl_vectors.npz")
OUT_PROG = os.path.join(HERE, "program_rtl_vectors.npz")


def expr(text):
    return Parser(tokenize(text)).parse()


def py_const(text, consts=None):
    """Value of an elaboration-time constant expression (parameters, loop bounds, array bounds)."""
    s = text
    for k in sorted(consts or {}, key=len, reverse=True):
        s = re.sub(rf"\b{k}\b", str(consts[k]), s)
    if not re.fullmatch(r"[\d\s()+\-*/%<>=!]+", s):
        raise ValueError(f"not a constant expression: {text!r} -> {s!r}")
    return int(eval(s.replace("/", "//")))


def ev_const(ast, vals):
    return Evaluator(None, dict(vals)).ev(ast)[0]


def package_consts():
    consts = {}
    for path in (f"{REF}/../NetTypes.sv", f"{REF}/common/DTEngine_Types.sv"):
        for name, e in re.findall(r"\bparameter\s+(?:\[[^\]]+\]\s*)?(\w+)\s*=\s*([^;,]+)[;,]", _strip(open(path).read())):
            try:
                consts[name] = ev_const(expr(e), {k: (v, 32) for k, v in consts.items()})
            except Exception:
                pass
    return consts


def module_body(path, name):
    m = re.search(rf"\bmodule\s+{name}\b(.*?)\bendmodule\b", _strip(open(path).read()), re.S)
    assert m, (path, name)
    return m.group(1)


def subst(text, consts):
    """Constants -> literals; `.NAME (...)` port / parameter names of instances are left alone."""
    for k in sorted(consts, key=len, reverse=True):
        text = re.sub(rf"(?<![.\w]){k}\b", str(consts[k]), text)
    return text


def instance_ports(text, start):
    """{port: expression text} of the `( .a(x), .b(y) )` list that opens at text[start] == '('."""
    end = _match(text, start + 1, r"\(", r"\)")
    body, ports, pos = text[start + 1:end - 1], {}, 0
    for m in re.finditer(r"\.(\w+)\s*\(", body):
        if m.start() < pos:
            continue
        close = _match(body, m.end(), r"\(", r"\)")
        ports[m.group(1)] = body[m.end():close - 1].strip()
        pos = close
    return ports, end


def find_instance(text, typ, name=None):
    """(parameter dict, port dict) of `typ #( .P(v), ... ) name ( .port(sig), ... );` (the #() part is optional)."""
    for m in re.finditer(rf"\b{typ}\b\s*", text):
        pos, params = m.end(), {}
        if text[pos] == "#":
            pos = text.index("(", pos)
            params, pos = instance_ports(text, pos)
        m2 = re.compile(r"\s*(\w+)\s*\(").match(text, pos)
        if not m2 or (name and m2.group(1) != name):
            continue
        ports, _ = instance_ports(text, m2.end() - 1)
        return params, ports
    raise KeyError((typ, name))


# ------------------------------------------------------------------------------------------ generate elaboration
def elaborate(text, out):
    """Walk module-level / generate-level items: unroll for loops, choose if branches, collect always / assign / decl."""
    pos, n = 0, len(text)
    at = lambda rx: re.compile(rx).match(text, pos)
    while True:
        pos = re.compile(r"\s*").match(text, pos).end()
        if pos >= n:
            return
        m = at(r"(generate|endgenerate)\b|(genvar|localparam|parameter|integer)\b[^;]*;")
        if m:
            pos = m.end()
            continue
        m = at(r"for\s*\(")
        if m:
            close = _match(text, m.end(), r"\(", r"\)")
            init, cond, step = [x.strip() for x in text[m.end():close - 1].split(";")]
            var, lo = [x.strip() for x in init.split("=")]
            hi = re.fullmatch(rf"{var}\s*<\s*(.+)", cond).group(1)
            st = re.fullmatch(rf"{var}\s*=\s*{var}\s*\+\s*(.+)", step).group(1)
            b = re.compile(r"\s*begin\s*(?::\s*\w+)?").match(text, close)
            end = _match(text, b.end(), r"\bbegin\b", r"\bend\b")
            body = text[b.end():end - 3]
            for v in range(py_const(lo), py_const(hi), py_const(st)):
                elaborate(re.sub(rf"\b{var}\b", f"({v})", body), out)
            pos = end
            continue
        if at(r"if\s*\("):
            taken = False
            while True:
                m = at(r"if\s*\(")
                cond = None
                if m:
                    close = _match(text, m.end(), r"\(", r"\)")
                    cond, pos = text[m.end():close - 1], close
                b = at(r"\s*begin\s*(?::\s*\w+)?")
                end = _match(text, b.end(), r"\bbegin\b", r"\bend\b")
                if not taken and (cond is None or py_const(cond)):
                    elaborate(text[b.end():end - 3], out)
                    taken = True
                pos = end
                m = at(r"\s*else\b\s*")
                if not m:
                    break
                pos = m.end()
            continue
        m = at(r"always\s*@\s*\(([^)]*)\)\s*begin\b")
        if m:
            end = _match(text, m.end(), r"\bbegin\b", r"\bend\b")
            out.append(("always", m.group(1).strip(), text[m.end():end - 3]))
            pos = end
            continue
        m = at(r"assign\s+([^;]+);")
        if m:
            out.append(("assign", m.group(1)))
            pos = m.end()
            continue
        m = at(r"(reg|wire)\b\s*(\[[^\]]+\])?\s*(\w+)\s*((?:\[[^\]]+\]\s*)*);")
        if m:
            out.append(("decl", m.group(2), m.group(3), m.group(4)))
            pos = m.end()
            continue
        raise SyntaxError(f"cannot elaborate: {text[pos:pos + 80]!r}")


class Elaborated:
    """A small combinational module from elaborated items: arrays flattened (run-time selects become ?: chains), clocked
    registers treated as wires (the steady state of a feed-forward pipeline with its inputs held)."""

    def __init__(self, items, port_widths):
        self.width, self.dims = dict(port_widths), {}
        for it in items:
            if it[0] == "decl":
                _, rng, name, unpacked = it
                w = 1
                if rng:
                    hi, lo = rng[1:-1].split(":")
                    w = py_const(hi) - py_const(lo) + 1
                d = []
                for g in re.findall(r"\[([^\]]+)\]", unpacked):
                    a, b = [py_const(x) for x in g.split(":")]
                    d.append(range(min(a, b), max(a, b) + 1))
                self.dims[name] = d
                self.width[name] = w
        self.m = Module.__new__(Module)
        self.m.name, self.m.inputs, self.m.outputs, self.m.cases, self.m.insts = "elab", [], [], {}, []
        self.m.assign, self.m.width = {}, self.width
        for it in items:
            if it[0] == "assign":
                lhs, rhs = it[1].split("=", 1)
                self.drive(self.flat(lhs.strip()), expr(self.flat(rhs)))
            elif it[0] == "always":
                self.collect(StmtParser(tokenize("begin " + self.flat(it[2]) + " end")).stmt())

    def drive(self, name, ast):
        assert re.fullmatch(r"\w+", name), name
        assert name not in self.m.assign, f"{name} driven twice"
        self.m.assign[name] = ast

    def collect(self, st):
        if st[0] == "block":
            for s in st[1]:
                self.collect(s)
        elif st[0] == "if":  # only reset tests occur: resolved with rst_n = 1
            c = ev_const(st[1], {"rst_n": (1, 1)})
            if c:
                self.collect(st[2])
            elif st[3] is not None:
                self.collect(st[3])
        else:
            _, _op, (name, hi, lo), rhs = st
            assert hi is None, st
            self.drive(name, rhs)

    def flat(self, text):
        """NAME[i][j][k] -> NAME__i__j__k; a non-constant index becomes a ?: chain over the declared range."""
        if not self.dims:
            return text
        pat = re.compile(r"\b(" + "|".join(sorted(self.dims, key=len, reverse=True)) + r")\s*\[")
        out, pos = [], 0
        while True:
            m = pat.search(text, pos)
            if not m:
                out.append(text[pos:])
                return "".join(out)
            out.append(text[pos:m.start()])
            name, p, idx = m.group(1), m.end() - 1, []
            for _ in self.dims[name]:
                assert text[p] == "[", text[p:p + 40]
                q = _match(text, p + 1, r"\[", r"\]")
                inner = self.flat(text[p + 1:q - 1])
                try:
                    idx.append(py_const(inner))
                except ValueError:
                    idx.append(inner)
                p = q
                while p < len(text) and text[p].isspace():
                    p += 1
            out.append(self.build(name, "", idx, 0))
            pos = p

    def build(self, name, suffix, idx, k):
        if k == len(idx):
            full = name + suffix
            self.width.setdefault(full, self.width[name])
            return full
        if isinstance(idx[k], int):
            return self.build(name, f"{suffix}__{idx[k]}", idx, k + 1)
        chain = None
        for v in reversed(self.dims[name][k]):
            br = self.build(name, f"{suffix}__{v}", idx, k + 1)
            chain = br if chain is None else f"(({idx[k]}) == {v} ? {br} : {chain})"
        return chain

    def get(self, out, **inputs):
        vals = {k: (v, self.width[k]) for k, v in inputs.items()}
        vals["rst_n"] = (1, 1)
        return Evaluator({}, vals, self.m).get(out)[0]


def elaborate_mux(params):
    """core/PipelinedMUX.sv for one instance's parameters -> f(line, addr) = word."""
    body = module_body(f"{REF}/core/PipelinedMUX.sv", "PipelinedMUX")
    consts = {k: int(v) for k, v in re.findall(r"\bparameter\s+(\w+)\s*=\s*(\d+)", body)}
    consts.update(params)
    for name, e in re.findall(r"\blocalparam\s+(\w+)\s*=\s*([^;]+);", body):
        consts[name] = py_const(e, consts)
    text = subst(body[body.index(");") + 2:], consts)
    items = []
    elaborate(text, items)
    el = Elaborated(items, {"din": consts["DATA_WIDTH"], "addr": consts["ADDR_WIDTH"], "dout": consts["WORD_WIDTH"], "rst_n": 1})
    return lambda line, addr: el.get("dout", din=line, addr=addr)


# ------------------------------------------------------------------------------------------ memory wrappers
class LineMemory:
    """core/Mem1in2out.v or core/dualport_mem.v around its (absent) vendor RAM: the RAM is an array of lines at whatever
    address expression the wrapper passes to it; the word offset register and the PipelinedMUX come from the text."""

    def __init__(self, path, module, ram, params):
        body = module_body(path, module)
        consts = {k: int(v) for k, v in re.findall(r"\bparameter\s+(\w+)\s*=\s*(\d+)", body)}
        consts.update(params)
        text = subst(body, consts)
        self.width = {}
        for rng, name in re.findall(r"\binput\s+wire\s*(\[[^\]]+\])?\s*(\w+)", text):
            w = 1
            if rng:
                hi, lo = rng[1:-1].split(":")
                w = py_const(hi) - py_const(lo) + 1
            self.width[name] = w
        _, rp = find_instance(text, ram)
        self.ram = {k: expr(v) for k, v in rp.items()}
        self.d1 = {lhs: expr(rhs) for lhs, rhs in re.findall(r"\b(\w+_d1)\s*<=\s*([^;]+);", text)}
        self.out = {}
        for name in re.findall(r"\bPipelinedMUX\b[^;]*?\)\s*(\w+)\s*\(", text):
            mp, ports = find_instance(text, "PipelinedMUX", name)
            mux = elaborate_mux({k: py_const(v) for k, v in mp.items()})
            q = [k for k, v in rp.items() if v == ports["din"]]
            assert len(q) == 1, (ports, rp)
            addr_port = {"q_a": "address_a", "q_b": "address_b", "q": "rdaddress"}[q[0]]
            self.out[ports["dout"]] = (self.ram[addr_port], self.d1[ports["addr"]], mux)
        self.lines = {}

    def vals(self, ports):
        return {k: (v, self.width[k]) for k, v in ports.items()}

    def write(self, **ports):
        v = self.vals(ports)
        wren = self.ram.get("wren_a", self.ram.get("wren"))
        if ev_const(wren, v):
            addr = ev_const(self.ram.get("address_a", self.ram.get("wraddress")), v)
            self.lines[addr] = ev_const(self.ram.get("data_a", self.ram.get("data")), v)
            return addr
        return None

    def read(self, dout, **ports):
        v = self.vals(ports)
        line_addr, sel, mux = self.out[dout]
        return mux(self.lines.get(ev_const(line_addr, v), 0), ev_const(sel, v))


# ------------------------------------------------------------------------------------------ one DTPU
def assigned_names(ast, acc):
    if ast[0] == "block":
        for s in ast[1]:
            assigned_names(s, acc)
    elif ast[0] == "if":
        assigned_names(ast[2], acc)
        if ast[3] is not None:
            assigned_names(ast[3], acc)
    elif ast[0] == "case":
        for _, body in ast[2]:
            assigned_names(body, acc)
    else:
        acc.add(ast[2][0])
    return acc


class LazySim(Sim):
    """Procedural blocks whose expressions may read continuous assigns of the module (evaluated on demand)."""

    def __init__(self, width, mod):
        super().__init__(width)
        self.mod = mod

    def ev(self, e, env):
        return Evaluator({}, {k: (v, self.width.get(k, 32)) for k, v in env.items()}, self.mod).ev(e)[0]


class PU:
    def __init__(self, pu_id):
        self.mod, self.consts = dtpu_module(pu_id)
        c = self.consts
        text = subst(_strip(open(f"{REF}/core/DTPU.sv").read()), c)
        self.text = text
        w = self.mod.width
        w.update({"data_line_in": 128, "data_line_in_valid": 1, "data_line_in_last": 1, "data_line_in_ctrl": 1,
                  "data_line_in_mode": 2, "data_line_in_pu": 3, "rst_n": 1, "clk": 1})
        want = {"prog": "tree_prog_addr", "tfi": "TFI_wr_addr", "feat": "features_wr_addr", "ctrl": "num_lines_per_tree_weights",
                "issue": "curr_tree_w_offset"}
        self.blocks = {}
        for sens, ast, _pos in always_blocks(text):
            names = assigned_names(ast, set())
            for key, sig in want.items():
                if sig in names:
                    assert key not in self.blocks, key
                    self.blocks[key] = ast
        assert set(self.blocks) == set(want), sorted(self.blocks)
        self.state = set()
        for ast in self.blocks.values():
            assigned_names(ast, self.state)
        # instruction FIFO input (DTPU.sv:554) and the instance wiring of the three memories
        m = re.search(r"\)\s*TupleInstrctionFIFO\s*\(", text)
        ports, _ = instance_ports(text, m.end() - 1)
        self.mod.assign["TupleInstrctionFIFO_din"] = expr(ports["din"])
        w["TupleInstrctionFIFO_din"] = c["INSTRUCTION_WIDTH"]
        self.sim = LazySim(w, self.mod)
        self.sig = {k: 0 for k in self.state}
        self.sig.update({"time_stamp": 0, "TupleInstrctionFIFO_full": 0, "delayed_instruction_valid_f": 0, "delayed_instruction_o": 0,
                         "tuple_old_enough": 0, "tree_instruction_valid": 0, "tuple_instruction": 0})
        self.sig.update({k[len("NEXT__"):]: 0 for k in self.mod.assign if k.startswith("NEXT__")})   # traversal registers at reset
        self.mem, self.conn = {}, {}
        for typ, inst, ram, path in (("Mem1in2out", "WeightsMem", "bramin1out2", "Mem1in2out.v"),
                                     ("DualPortMem", "TreeFeatureIndex_Mem", "Qdualport_mem", "dualport_mem.v"),
                                     ("DualPortMem", "SamplesFeatures_Mem", "Qdualport_mem", "dualport_mem.v")):
            params, ports = find_instance(text, typ, inst)
            self.mem[inst] = LineMemory(f"{REF}/core/{path}", typ, ram, {k: py_const(v) for k, v in params.items()})
            self.conn[inst] = {k: expr(v) for k, v in ports.items() if k not in ("clk", "rst_n") and not k.startswith(("dout", "valid_out"))}
        self.delayed, self.instructions = deque(), []

    def wire(self, name, env=None):
        env = self.sig if env is None else env
        return Evaluator({}, {k: (v, self.sim.width.get(k, 32)) for k, v in env.items()}, self.mod).get(name)[0]

    def port_vals(self, inst, env, names):
        ev = Evaluator({}, {k: (v, self.sim.width.get(k, 32)) for k, v in env.items()}, self.mod)
        return {p: ev.ev(self.conn[inst][p])[0] for p in names}

    def clock(self, rst_n=1, line=0, valid=0, last=0, ctrl=0, mode=0, pu=0, **extra):
        s = self.sig
        s.update({"rst_n": rst_n, "data_line_in": line, "data_line_in_valid": valid, "data_line_in_last": last,
                  "data_line_in_ctrl": ctrl, "data_line_in_mode": mode, "data_line_in_pu": pu})
        s.update(extra)
        env = dict(s)
        wrote = {}
        if rst_n:
            wrote["w"] = self.mem["WeightsMem"].write(**self.port_vals("WeightsMem", env, ("we", "wraddr", "din")))
            wrote["f"] = self.mem["TreeFeatureIndex_Mem"].write(**self.port_vals("TreeFeatureIndex_Mem", env, ("we", "waddr", "din")))
            wrote["x"] = self.mem["SamplesFeatures_Mem"].write(**self.port_vals("SamplesFeatures_Mem", env, ("we", "waddr", "din")))
            if self.wire("delayed_instruction_we", env):
                self.delayed.append(self.wire("delayed_instruction_i", env))
            if self.wire("tuple_instruction_we", env):
                self.instructions.append(self.wire("TupleInstrctionFIFO_din", env))
            pop = self.wire("delayed_instruction_re", env) and s["delayed_instruction_valid_f"]
        nxt = {}
        for ast in self.blocks.values():
            self.sim.run(ast, env, nxt, False)
        s.update(nxt)
        if rst_n and pop:
            self.delayed.popleft()
        return wrote

    def issue_all(self):
        """Drain the delayed-instruction queue: one instruction per tree slot and tuple (DTPU.sv:512-567)."""
        guard = 0
        while self.delayed:
            self.clock(pu=(self.consts["PU_ID"] + 1) % 8, delayed_instruction_valid_f=1, delayed_instruction_o=self.delayed[0],
                       tuple_old_enough=1)
            guard += 1
            assert guard < 100000
        self.clock(pu=(self.consts["PU_ID"] + 1) % 8, delayed_instruction_valid_f=0, tuple_old_enough=0)

    def walk(self, instr, mode0=1, probe=None):
        """One instruction through the traversal datapath (DTPU.sv:579-760); memory reads through the wrappers."""
        m, w = self.mod, self.sim.width
        fixed = {k: (self.sig[k], w[k]) for k in ("LastLevelIndex", "MissingFeatureValue", "PartialTrees", "tree_prog_addr")}
        fixed.update({"tuple_instruction": (instr, self.consts["INSTRUCTION_WIDTH"]), "tree_instruction_valid": (0, 1),
                      "data_line_in_mode": (mode0, 2)})
        regs = {k[len("NEXT__"):]: (0, m.width[k]) for k in m.assign if k.startswith("NEXT__")}
        W, FI, X = self.mem["WeightsMem"], self.mem["TreeFeatureIndex_Mem"], self.mem["SamplesFeatures_Mem"]
        cw, cf, cx = self.conn["WeightsMem"], self.conn["TreeFeatureIndex_Mem"], self.conn["SamplesFeatures_Mem"]
        for _ in range(17):
            base = dict(fixed)
            base.update(regs)
            a = Evaluator({}, dict(base), m)
            wraddr = a.ev(cw["wraddr"])[0]
            if probe is not None:
                probe.append((wraddr, a.get("tree_w_node_addr_s1")[0]))
            base["TWM_weight_data"] = (W.read("dout1", wraddr=wraddr, raddr=0), 32)
            base["TFI_rd_data"] = (FI.read("dout", raddr=a.ev(cf["raddr"])[0]), 16)
            b = Evaluator({}, dict(base), m)
            base["features_rd_data"] = (X.read("dout", raddr=b.ev(cx["raddr"])[0]), 32)
            c = Evaluator({}, dict(base), m)
            go_out = c.get("goToOutput")[0]
            regs = {k[len("NEXT__"):]: c.get(k) for k in m.assign if k.startswith("NEXT__")}
            regs["tree_instruction_valid"] = (1 - go_out, 1)
            if go_out:
                d = dict(fixed)
                d.update(regs)
                raddr = Evaluator({}, d, m).ev(cw["raddr"])[0]
                leaf = W.read("dout2", raddr=raddr, wraddr=0)
                return 0 if regs["tree_instruction_type_EMPTY"][0] else leaf
        raise RuntimeError("the walk did not end")


def pack_line(words, bits):
    v = 0
    for k, x in enumerate(words):
        v |= int(x) << (bits * k)
    return v


def control_word(core_ctrl, core_width, **fields):
    return ev_const(core_ctrl, {k: (v, core_width[k]) for k, v in fields.items()})


def core_control_expr(consts):
    core = subst(_strip(open(f"{REF}/Core.sv").read()), consts)
    m = re.search(r"data_line_distr\[0\]\[0\]\s*<=\s*(\{\s*24'b0[^;]+);", core)
    assert m, "Core.sv:380 moved"
    width = {}
    for rng, name in re.findall(r"\binput\s+wire\s*(\[[^\]]+\])?\s*(\w+)", core):
        w = 1
        if rng:
            hi, lo = rng[1:-1].split(":")
            w = py_const(hi) - py_const(lo) + 1
        width[name] = w
    return expr(m.group(1)), width


def run_pu_case(rng, core_ctrl, core_width, pu_id, D, F, slots, K, pad_w, pad_f, n_tuples, stride_minus=0):
    nint, nleaf = (1 << D) - 1, 1 << D
    wl, fl, tl = (nint + nleaf + 3) // 4 + pad_w, (nint + 7) // 8 + pad_f, (F + 3) // 4
    missing = 0x7FC00000
    pu = PU(pu_id)
    other = (pu_id + 3) % 8
    pu.clock(rst_n=0, pu=other)
    cw = control_word(core_ctrl, core_width, tuple_numcls=tl, missing_value=missing, tree_feature_index_numcls=fl - stride_minus,
                      tree_weights_numcls=wl - stride_minus, num_levels_per_tree_minus_one=D - 1, num_trees_per_pu_minus_one=slots - 1)
    pu.clock(line=cw, ctrl=1, pu=other)
    thr = (rng.random((K, nint)).astype(np.float32) * 2 - 1).view(np.uint32)
    leaf = ((rng.random((K, nleaf)) - 0.5).astype(np.float32)).view(np.uint32)
    ent = (rng.integers(0, F, (K, nint)) | (rng.integers(0, 2, (K, nint)) << 13)).astype(np.uint16)
    wlines = rng.integers(1, 1 << 32, (K, wl * 4), dtype=np.uint64).astype(np.uint32)   # padding words are junk, not zeros
    wlines[:, :nint], wlines[:, nint:nint + nleaf] = thr, leaf
    flines = rng.integers(1, 1 << 16, (K, fl * 8), dtype=np.uint64).astype(np.uint16) & np.uint16(0x27FF)
    flines[:, :nint] = ent
    waddrs, faddrs = [], []
    # weights lines (mode = {prog_mode = 1, data_valid = 0}), a foreign PU's lines in between
    for t in range(K):
        for ln in range(wl):
            r = pu.clock(line=pack_line(wlines[t, 4 * ln:4 * ln + 4], 32), last=int(ln == wl - 1), mode=2, pu=pu_id)
            waddrs.append(r["w"])
            assert r["f"] is None and r["x"] is None
        r = pu.clock(line=(1 << 128) - 1, last=1, mode=2, pu=other)
        assert r["w"] is None and r["f"] is None
    for t in range(K):
        for ln in range(fl):
            r = pu.clock(line=pack_line(flines[t, 8 * ln:8 * ln + 8], 16), last=int(ln == fl - 1), mode=0, pu=pu_id)
            faddrs.append(r["f"])
            assert r["w"] is None
        r = pu.clock(line=(1 << 128) - 1, last=1, mode=0, pu=other)
        assert r["f"] is None
    assert pu.sig["local_num_trees"] == K % 16      # a 4-bit counter (DTPU.sv:70,96,115): observation (5) in the docstring
    tuples = (rng.random((n_tuples, tl * 4)).astype(np.float32) * 2 - 1).view(np.uint32)
    tuples[rng.random(tuples.shape) < 0.08] = missing
    tuples[:, F:] = 0
    for t in range(n_tuples):
        for ln in range(tl):
            pu.clock(line=pack_line(tuples[t, 4 * ln:4 * ln + 4], 32), valid=1, last=int(ln == tl - 1), mode=1, pu=other)
    pu.issue_all()
    assert len(pu.instructions) == n_tuples * slots, (len(pu.instructions), n_tuples, slots)
    out = np.zeros((n_tuples, slots), np.uint32)
    for i, ins in enumerate(pu.instructions):
        out[i // slots, i % slots] = pu.walk(ins)
    return dict(pu=pu, wl=wl, fl=fl, tl=tl, missing=missing, wlines=wlines, flines=flines, tuples=tuples, out=out,
                waddrs=waddrs, faddrs=faddrs, instr=np.array(pu.instructions, np.uint64), ctrl=cw)


def program_vectors(consts):
    core_ctrl, core_width = core_control_expr(consts)
    rng = np.random.default_rng(41)
    cases = [  # pu_id, D, F, slots, K programmed trees, extra weights / findex lines per tree, tuples
        (3, 1, 5, 2, 2, 0, 0, 4), (1, 2, 8, 4, 3, 0, 0, 4), (5, 3, 13, 3, 3, 1, 0, 4), (2, 4, 16, 5, 4, 0, 2, 4),
        (6, 5, 28, 4, 4, 0, 0, 3), (7, 6, 32, 16, 9, 0, 0, 3), (4, 7, 20, 6, 6, 3, 1, 3), (3, 8, 32, 16, 15, 0, 0, 2),
        (1, 8, 32, 16, 5, 0, 0, 2), (2, 3, 64, 8, 8, 0, 0, 3)]
    rec = {k: [] for k in ("pu_id", "D", "F", "slots", "K", "wl", "fl", "tl", "missing", "n_tuples", "ctrl_lo", "ctrl_hi")}
    blobs = {k: [] for k in ("wlines", "flines", "tuples", "out", "instr")}
    for (pid, D, F, slots, K, pw, pf, nt) in cases:
        r = run_pu_case(rng, core_ctrl, core_width, pid, D, F, slots, K, pw, pf, nt)
        # line L of the PU's stream lands on line L of the memory, trees back to back
        assert r["waddrs"] == list(range(K * r["wl"])) and r["faddrs"] == list(range(K * r["fl"]))
        for k, v in (("pu_id", pid), ("D", D), ("F", F), ("slots", slots), ("K", K), ("wl", r["wl"]), ("fl", r["fl"]), ("tl", r["tl"]),
                     ("missing", r["missing"]), ("n_tuples", nt), ("ctrl_lo", r["ctrl"] & (2**64 - 1)), ("ctrl_hi", r["ctrl"] >> 64)):
            rec[k].append(v)
        for k in blobs:
            blobs[k].append(np.asarray(r[k]).reshape(-1))
        print(f"PU {pid} D={D} F={F} slots={slots} programmed={K} wl={r['wl']} fl={r['fl']}: {nt * slots} walks, "
              f"{int((r['out'][:, K:] == 0).all())} empty-slots-zero")
    out = {k: np.array(v, np.uint64) for k, v in rec.items()}
    out.update({k: np.concatenate(v) for k, v in blobs.items()})
    # ---- observations about the published RTL (see the module docstring)
    pu = PU(3)
    pu.clock(rst_n=0, pu=0)
    before = pu.sig["TFI_wr_addr"]
    pu.clock(pu=3)                                            # an idle cycle that carries this PU's id
    out["idle_tfi_advance"] = np.array([pu.sig["TFI_wr_addr"] - before], np.int64)
    r = run_pu_case(np.random.default_rng(5), core_ctrl, core_width, 3, 3, 8, 2, 2, 0, 0, 1)
    probe = []
    r["pu"].walk(int(r["instr"][1]), mode0=0, probe=probe)    # slot 1: its node address differs from the programming pointer
    out["idle_read_hits_prog_addr"] = np.array([int(all(a == (r["pu"].sig["tree_prog_addr"] << 2) for a, _ in probe) and
                                                    any(a != b for a, b in probe))], np.int64)
    q = run_pu_case(np.random.default_rng(5), core_ctrl, core_width, 3, 3, 8, 4, 4, 0, 0, 1, stride_minus=1)
    out["quirk_instr"], out["quirk_wl"], out["quirk_fl"] = q["instr"], np.array([q["wl"]], np.uint64), np.array([q["fl"]], np.uint64)
    out["quirk_out"] = q["out"].reshape(-1)
    out["quirk_wlines"], out["quirk_flines"], out["quirk_tuples"] = q["wlines"].reshape(-1), q["flines"].reshape(-1), q["tuples"].reshape(-1)
    f = run_pu_case(np.random.default_rng(6), core_ctrl, core_width, 2, 2, 6, 16, 16, 0, 0, 1)   # a PU holding the full 16 trees
    out["full_pu_local_num_trees"] = np.array([f["pu"].sig["local_num_trees"]], np.int64)
    out["full_pu_all_zero"] = np.array([int((f["out"] == 0).all())], np.int64)
    out["instr_fields"] = np.array([consts_pu["TREE_OFFSET_BITS"], consts_pu["TUPLE_OFFSET_BITS"]], np.uint64)
    out["result_scores"], out["result_lines"] = result_line_vectors(consts)
    out.update(aggregator_timing_vectors())
    np.savez_compressed(OUT_PROG, **out)
    print(f"wrote {OUT_PROG}: idle_tfi_advance={out['idle_tfi_advance'][0]} idle_read_hits_prog_addr={out['idle_read_hits_prog_addr'][0]}")


# ------------------------------------------------------------------------------------------ part 1: the CSR chain
class Params(ctypes.Structure):   # include/ddt.h ddt_params
    _fields_ = [(n, ctypes.c_uint32) for n in ("num_trees", "num_levels", "num_features", "missing_bits", "weights_lines_per_tree",
                                               "findex_lines_per_tree", "cmp_mode", "clusters_per_tuple", "sum_mode")] + [
                                                   ("reserved", ctypes.c_uint32 * 3)]


def csr_design(consts):
    text = subst(_strip(open(f"{REF}/EngineCSR.sv").read()), consts)
    text = re.sub(r"softreg_req\.(\w+)", r"softreg_req_\1", text)
    text = re.sub(r"devices_list\[(\d+)\]", r"devices_list__\1", text)
    text = re.sub(r"for\s*\(i = 0;[^)]*\)\s*begin\s*devices_list\[i\]\s*<=\s*0;\s*end", "", text)   # reset loop: registers start at 0 anyway
    width = {"softreg_req_valid": 1, "softreg_req_isWrite": 1, "softreg_req_addr": 32, "softreg_req_data": 64, "rst_n": 1}
    for rng, name in re.findall(r"\b(?:output\s+)?reg\s*(\[[^\]]+\])?\s*(\w+)", text):
        w = 1
        if rng:
            hi, lo = rng[1:-1].split(":")
            w = py_const(hi) - py_const(lo) + 1
        width.setdefault(name, w)
    for k in range(consts["NUM_FPGA_DEVICES"]):
        width[f"devices_list__{k}"] = consts["DEVICE_ADDRESS_WIDTH"]
    block = None
    for sens, ast, _pos in always_blocks(text):
        if "prog_schedule" in assigned_names(ast, set()):
            assert block is None
            block = ast
    assert block is not None, "the Write SoftRegs block did not parse"
    return block, width


def rtl_csr_write(design, csr_words):
    block, width = design
    sim = Sim(width)
    regs = sorted(assigned_names(block, set()))
    s = sim.sig
    nxt = {}
    s.update({"rst_n": 0, "softreg_req_valid": 0, "softreg_req_isWrite": 0, "softreg_req_addr": 0, "softreg_req_data": 0})
    sim.run(block, dict(s), nxt, False)
    s.update(nxt)
    for k, wd in enumerate(csr_words):
        nxt = {}
        s.update({"rst_n": 1, "softreg_req_valid": 1, "softreg_req_isWrite": 1, "softreg_req_addr": 200 + k, "softreg_req_data": int(wd)})
        sim.run(block, dict(s), nxt, False)
        s.update(nxt)
    return {r: s[r] for r in regs}


def core_wiring():
    """Core port -> EngineCSR output, through the wires of DTInference.sv (both instances' port lists)."""
    text = _strip(open(f"{REF}/DTInference.sv").read())
    _, csr_ports = find_instance(text, "EngineCSR")
    wire_to_reg = {v: k for k, v in csr_ports.items() if re.fullmatch(r"\w+", v)}
    _, core_ports = find_instance(text, "Core")
    return {p: wire_to_reg[v] for p, v in core_ports.items() if v in wire_to_reg}


def csr_vectors(consts):
    lib = ctypes.CDLL(os.path.join(ROOT, "distributed-decisiontrees_amd", "lib", "libddt.so"))
    enc = lib.ddt_csr_encode_ex
    enc.argtypes = [ctypes.POINTER(Params), ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]
    design = csr_design(consts)
    wiring = core_wiring()
    core_ctrl, core_width = core_control_expr(consts)
    pu = PU(3)
    cases = []  # T, D, F, C, pad_w, pad_f, n_tuples, devices, mode, device_index
    for (T, D, F, C) in ((8, 4, 16, 1), (100, 6, 28, 1), (1000, 8, 32, 8), (37, 8, 32, 4), (512, 5, 64, 2), (1024, 8, 2048, 8), (16, 11, 7, 2),
                         (64, 1, 1, 1), (24, 16, 40, 1)):
        for (nd, mode, idx) in ((1, 0, 0), (8, 0, 0), (8, 0, 7), (9, 1, 4), (20, 0, 19), (20, 1, 0)):
            cases.append((T, D, F, C, (T + D) % 3, (T + F) % 2, 1000 * T + 3, nd, mode, idx))
    rows, csrs, regs_all, pu_all = [], [], [], []
    for (T, D, F, C, pw, pf, n, nd, mode, idx) in cases:
        p = Params(T, D, F, 0x7FC00000 ^ (T * 2654435761 & 0xFFFF), ((1 << (D + 1)) - 1 + 3) // 4 + pw, ((1 << D) - 1 + 7) // 8 + pf, 0, C, 0)
        buf = (ctypes.c_uint64 * 12)()
        rc = enc(ctypes.byref(p), n, nd, mode, idx, buf)
        if rc:   # e.g. per-device line counters overflow: the codec refuses, nothing to decode
            continue
        words = [int(x) for x in buf]
        regs = rtl_csr_write(design, words)
        core_in = {port: regs[reg] for port, reg in wiring.items() if port in core_width and reg in regs}
        cw = control_word(core_ctrl, core_width, **{k: core_in[k] for k in ("tuple_numcls", "missing_value", "tree_feature_index_numcls",
                                                                            "tree_weights_numcls", "num_levels_per_tree_minus_one",
                                                                            "num_trees_per_pu_minus_one")})
        pu.clock(rst_n=0, pu=0)
        pu.clock(line=cw, ctrl=1, pu=0)
        pu_regs = {k: pu.sig[k] for k in ("num_trees_per_pu_minus_one", "PartialTrees", "LastLevelIndex", "num_lines_per_tree_weights",
                                          "num_lines_per_tree_findex", "MissingFeatureValue", "tuple_numlines")}
        rows.append((T, D, F, C, p.missing_bits, p.weights_lines_per_tree, p.findex_lines_per_tree, n, nd, mode, idx))
        csrs.append(words)
        regs_all.append(regs)
        pu_all.append(pu_regs)
    rng = np.random.default_rng(77)
    rand = rng.integers(0, 1 << 63, (64, 12), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (64, 12), dtype=np.uint64)
    rand_regs = [rtl_csr_write(design, [int(x) for x in row]) for row in rand]
    out = {"cases": np.array(rows, np.uint64), "csr": np.array(csrs, np.uint64), "rand_csr": rand,
           "core_wiring": np.array([f"{k}={v}" for k, v in sorted(wiring.items())])}
    for name in regs_all[0]:
        out["reg_" + name] = np.array([r[name] for r in regs_all], np.uint64)
        out["rand_" + name] = np.array([r[name] for r in rand_regs], np.uint64)
    for name in pu_all[0]:
        out["pu_" + name] = np.array([r[name] for r in pu_all], np.uint64)
    np.savez_compressed(OUT_CSR, **out)
    print(f"wrote {OUT_CSR}: {len(rows)} codec blocks + {len(rand)} random blocks, {len(regs_all[0])} registers; "
          f"Core stride ports wired to {wiring['tree_weights_numcls']} / {wiring['tree_feature_index_numcls']}")


# ------------------------------------------------------------------------------------------ part 3: result lines (A14)
def result_line_vectors(consts):
    """ResultsCombiner.sv:131-160: local scores are collected four to a 128-bit line, word k of the line = score 4L + k; the
    line enters the result FIFO as {w3, w2, w1, w0} (:193).  Executed with the FIFO never full."""
    text = subst(_strip(open(f"{REF}/ResultsCombiner.sv").read()), consts)
    m = re.search(r"\.din\s*\(\s*\{local_core_result_line\[3\], local_core_result_line\[2\], local_core_result_line\[1\], local_core_result_line\[0\]\}\s*\)", text)
    assert m, "the result FIFO no longer takes {line[3], line[2], line[1], line[0]}"
    text = re.sub(r"for\s*\(i = 0; i < 4; i=i\+1\)\s*begin\s*local_core_result_line\[\s*i\s*\]\s*<=\s*32'b0;\s*end", "", text)
    # the run-time word select on the left-hand side, spelled out per word
    text, n = re.subn(r"local_core_result_line\[\s*curr_word\s*\]\s*<=\s*local_core_result;",
                      " ".join(f"if (curr_word == {k}) local_core_result_line__{k} <= local_core_result;" for k in range(4)), text)
    assert n == 1
    block = None
    for sens, ast, _pos in always_blocks(text):
        if "curr_word" in assigned_names(ast, set()):
            block = ast
    assert block is not None
    fill = expr(re.search(r"\bassign\s+fill_local_core_result_line\s*=\s*([^;]+);", text).group(1))
    width = {"rst_n": 1, "local_core_result_valid": 1, "local_core_result": 32, "aggreg_core_result_full": 1, "curr_word": 2,
             "local_core_result_line_filled": 1, "fill_local_core_result_line": 1}
    width.update({f"local_core_result_line__{k}": 32 for k in range(4)})
    mod = Module.__new__(Module)
    mod.name, mod.inputs, mod.outputs, mod.cases, mod.insts, mod.width, mod.assign = "rc", [], [], {}, [], width, {"fill_local_core_result_line": fill}
    sim = LazySim(width, mod)
    s = sim.sig
    s.pop("fill_local_core_result_line")

    lines = []

    def clock(**inp):
        s.update(inp)
        if s.get("local_core_result_line_filled"):                                    # the result FIFO's write enable, this cycle
            lines.append(sum(s[f"local_core_result_line__{w}"] << (32 * w) for w in range(4)))   # din = {w3, w2, w1, w0}
        env, nxt = dict(s), {}
        sim.run(block, env, nxt, False)
        s.update(nxt)

    rng = np.random.default_rng(9)
    scores = rng.integers(1, 1 << 32, 23, dtype=np.uint64).astype(np.uint32)
    clock(rst_n=0, local_core_result_valid=0, local_core_result=0, aggreg_core_result_full=0)
    for k, sc in enumerate(scores):
        clock(rst_n=1, local_core_result_valid=1, local_core_result=int(sc), aggreg_core_result_full=0)
        if k % 3 == 2:
            clock(local_core_result_valid=0, local_core_result=0xDEADBEEF)            # a gap in the score stream
    clock(local_core_result_valid=0)
    return scores, np.array([[(ln >> (32 * w)) & 0xFFFFFFFF for w in range(4)] for ln in lines], np.uint32)


# ------------------------------------------------------------------------------------------ part 5: the accumulator's control
def aggregator_timing_vectors():
    """core/FPAggregator.v as a WHOLE -- its clocked block (latency counter, running sum, output rule) executed by the interpreter,
    the 2-cycle `delay` of valid / last and the 2-stage FloPoCo adder (evaluated from its source) as two-deep pipelines -- with the
    one module the reference does not contain, `quick_fifo`, modelled as a show-ahead FIFO (valid = not empty, `re` pops): the only
    semantics under which the module's `re = ready` does not lose data.  Values arrive `spacing` cycles apart.
    Result (recorded, asserted by tests/test_oracle_program.py): from 3 cycles apart the module computes the strictly sequential
    sum the datapath vectors pin (acc <- x + acc); 1 or 2 cycles apart it pops a value every 2 cycles while a sum re-enters the
    adder after 3, keeps two interleaved chains and outputs only the one that holds the last value."""
    from make_rtl_golden import SRC, load_modules, rtl_add

    text = _strip(open(f"{REF}/core/FPAggregator.v").read())
    m = re.search(r"parameter\s+FP_ADDER_LATENCY\s*=\s*(\d+)", text)
    assert m and int(m.group(1)) == 2
    text = re.sub(r"\bFP_ADDER_LATENCY\b", m.group(1), text)
    blk = [ast for _s, ast, _p in always_blocks(text) if "prev_aggreg_value" in assigned_names(ast, set())]
    assert len(blk) == 1
    width = {"rst_n": 1, "prev_aggreg_value": 34, "fpadder_latency_count": 4, "aggreg_out_valid_d1": 1, "aggreg_out_d1": 32, "aggregator_ready": 1,
             "aggreg_in_fifo_valid": 1, "fp_in_valid_delayed": 1, "fp_in_last_delayed": 1, "aggreg_value": 34, "aggreg_out_fifo_almfull": 1,
             "aggreg_in_fifo_dout": 33}
    ready_e = expr(re.search(r"assign\s+aggregator_ready\s*=\s*([^;]+);", text).group(1))
    in_a = expr(re.search(r"assign\s+input_A\s*=\s*([^;]+);", text).group(1))
    assert re.search(r"assign\s+aggreg_in_fifo_re\s*=\s*aggregator_ready\s*;", text) and re.search(r"\.DELAY_CYCLES\(2\)", text)
    mods = load_modules(SRC)

    def run(values, spacing):
        sim = Sim(width)
        s = sim.sig
        fifo, outs = deque(), []
        adder, dly = deque([None, None]), deque([(0, 0), (0, 0)])
        nxt = {}
        sim.run(blk[0], dict(s, rst_n=0), nxt, False)
        s.update(nxt)
        s["rst_n"] = 1
        arrivals = {k * spacing: (int(v), int(k == len(values) - 1)) for k, v in enumerate(values)}
        for c in range(len(values) * max(spacing, 3) + 12):
            if c in arrivals:
                fifo.append(arrivals[c])
            valid = int(len(fifo) > 0)
            val, last = fifo[0] if fifo else (0, 0)
            ready = sim.ev(ready_e, dict(s)) & 1
            x_in = sim.ev(in_a, {"aggreg_in_fifo_dout": (last << 32) | val})
            r = adder[0]
            dv, dl = dly[0]
            env = dict(s, aggregator_ready=ready, aggreg_in_fifo_valid=valid, fp_in_valid_delayed=dv, fp_in_last_delayed=dl,
                       aggreg_value=rtl_add(mods, r[0], r[1]) if r else 0)
            nxt = {}
            sim.run(blk[0], env, nxt, False)
            adder.popleft()
            adder.append((x_in, s["prev_aggreg_value"]))              # X = new value, Y = running sum, sampled this cycle
            dly.popleft()
            dly.append((valid & ready, last))
            if ready and valid:
                fifo.popleft()
            s.update(nxt)
            if s["aggreg_out_valid_d1"]:
                outs.append(s["aggreg_out_d1"])
        assert len(outs) == 1 and not fifo
        return outs[0]

    rng = np.random.default_rng(3)
    seqs = [(rng.random(n).astype(np.float32) * 4 - 2).view(np.uint32) for n in (2, 5, 8, 9, 16)]
    spacings = (1, 2, 3, 4, 7)
    out = np.array([[run(v, sp) for sp in spacings] for v in seqs], np.uint32)
    return {"agg_values": np.concatenate(seqs), "agg_lengths": np.array([len(v) for v in seqs], np.uint32), "agg_spacings": np.array(spacings, np.uint32),
            "agg_out": out}


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit(f"{REF} not found: run this in the build container")
    consts = package_consts()
    _m, consts_pu = dtpu_module(0)
    program_vectors(consts)
    csr_vectors(consts)
