#!/usr/bin/env python3
"""Golden vectors for the reference's tree -> cluster / PU schedule, produced by EXECUTING THE REFERENCE'S OWN RTL TEXT.

What fixes the ORDER in which the reference adds leaf values -- and therefore what "bit-exact fp32 sums" means -- is a
small piece of sequential control in rtl/DTEngine/Core.sv:

  * the always @(*) block that chooses `schedule_to_shift` / `shift_enable` / `shift_count`      (Core.sv:291-319)
  * the rotate-left-shifter it drives                                                             (core/RLS.v:36-62)
  * the core FSM that raises init_w / init_idx / init_p                                           (Core.sv:167-245)
  * the clocked block that stamps every line with the cluster-enable mask and the PU number       (Core.sv:323-372)
  * the clocked block that walks the clusters of a tuple when their partial sums are added up     (Core.sv:503-541)

No HDL simulator exists in this image, so this script is a small interpreter for the procedural Verilog subset those
five blocks use (begin/end, if/else, case, blocking and non-blocking assignments, part selects on the left-hand side),
built on the expression parser / evaluator of make_rtl_golden.py.  The blocks are cut out of the source text by their
position, constants come from common/DTEngine_Types.sv, and the design is stepped cycle by cycle: all clocked blocks
read the pre-edge values, non-blocking updates commit together, the combinational block and the continuous assigns it
interacts with are iterated to a fixed point.  Stimulus = what the input FIFO would deliver: T trees of weights lines,
T trees of feature-index lines, then tuples, one line per cycle, with every cluster ready.

ONE documented repair is applied to the text (SURVEY.md section 8a "known defects" #2): in the FSM's IDLE arm the
published source tests `start_core` inside the else-branch of `if(~rst_n | start_core)`, so the IDLE -> PROG_MODE edge
can never be taken and init_w / init_idx / init_p would stay 0 forever (every group of trees would then be written to
the same clusters).  The delayed copy `start_core_d1` exists one line away and is what the control pulse uses
(Core.sv:159-166,340); the script substitutes it in that ONE statement and asserts the substitution matched exactly once.

Recorded per case (clusters per tuple C in {1,2,4,8}, T a multiple of 8 -- the RTL's PU / group counters only line up
for whole groups of 8 trees): for every tree the cluster-enable mask and PU number its weights lines and its
feature-index lines were stamped with, for every tuple its cluster mask, and the sequence of clusters whose partial sums
enter the final accumulator together with the `last` flag.

Run HERE (needs /root/reference); writes tests/golden/schedule_rtl_vectors.npz:
    python tests/golden/make_schedule_golden.py
tests/test_oracle_schedule.py rebuilds the summation order from these placements and holds oracle/ddt_oracle.c
(orc_reduce_device: tree i -> PU i % 8, group g = i / 8 -> cluster g % C, slot g / C, clusters added 0..C-1) to it.
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_rtl_golden import Evaluator, Parser  # noqa: E402  (expression AST + evaluator)

REF = "/root/reference/rtl/DTEngine"
OUT = os.path.join(HERE, "schedule_rtl_vectors.npz")

TOK = re.compile(r"\s*(?:(\d+)\s*'\s*([bBhHdD])\s*([0-9a-fA-F_]+)|(\d+)|([A-Za-z_$][A-Za-z_0-9$]*)|"
                 r"(<=|>=|==|!=|&&|\|\||<<|>>|.))", re.S)


def tokenize(text):
    out, pos = [], 0
    text = text.strip()
    while pos < len(text):
        m = TOK.match(text, pos)
        if m.group(1):
            out.append(("lit", int(m.group(3).replace("_", ""), {"b": 2, "h": 16, "d": 10}[m.group(2).lower()]), int(m.group(1))))
        elif m.group(4):
            out.append(("lit", int(m.group(4)), 32))
        elif m.group(5):
            out.append(("id", m.group(5)))
        else:
            out.append(("op", m.group(6)))
        pos = m.end()
    return out


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


class StmtParser(Parser):
    """Statements of a procedural block -> tuples; expressions are handed to the inherited expression parser."""

    def is_id(self, name):
        return self.peek() == ("id", name)

    def lhs(self):
        tok = self.take()
        assert tok[0] == "id", tok
        if self.peek() == ("op", "["):
            self.take()
            hi = self.ternary()
            lo = hi
            if self.peek() == ("op", ":"):
                self.take()
                lo = self.ternary()
            self.take("]")
            return (tok[1], hi, lo)
        return (tok[1], None, None)

    def stmt(self):
        if self.is_id("begin"):
            self.take()
            body = []
            while not self.is_id("end"):
                body.append(self.stmt())
            self.take()
            return ("block", body)
        if self.is_id("if"):
            self.take()
            self.take("(")
            cond = self.ternary()
            self.take(")")
            then = self.stmt()
            els = None
            if self.is_id("else"):
                self.take()
                els = self.stmt()
            return ("if", cond, then, els)
        if self.is_id("case"):
            self.take()
            self.take("(")
            sel = self.ternary()
            self.take(")")
            arms, default = [], None
            while not self.is_id("endcase"):
                if self.is_id("default"):
                    self.take()
                    self.take(":")
                    default = self.stmt()
                    continue
                labels = [self.ternary()]
                while self.peek() == ("op", ","):
                    self.take()
                    labels.append(self.ternary())
                self.take(":")
                arms.append((labels, self.stmt()))
            self.take()
            return ("case", sel, arms, default)
        target = self.lhs()
        op = self.take()
        assert op in (("op", "<="), ("op", "=")), op
        rhs = self.ternary()
        self.take(";")
        return ("assign", op[1], target, rhs)


def always_blocks(text):
    """[(sensitivity text, statement AST)] of every `always` block of a module text that parses."""
    out = []
    for m in re.finditer(r"\balways\s*@\s*\(([^)]*)\)", text):
        toks = tokenize(text[m.end():m.end() + 12000])
        try:
            out.append((m.group(1).strip(), StmtParser(toks).stmt(), m.start()))
        except (SyntaxError, AssertionError, IndexError):
            pass  # blocks this subset cannot express are not needed here
    return out


class Sim:
    def __init__(self, width):
        self.width = dict(width)
        self.sig = {k: 0 for k in width}

    def ev(self, expr, env):
        v = Evaluator(None, {k: (val, self.width.get(k, 32)) for k, val in env.items()}).ev(expr)[0]
        return v

    def run(self, stmt, env, out, blocking):
        """Execute a statement.  blocking: assignments update `env` at once; else they accumulate in `out`."""
        k = stmt[0]
        if k == "block":
            for s in stmt[1]:
                self.run(s, env, out, blocking)
        elif k == "if":
            if self.ev(stmt[1], env):
                self.run(stmt[2], env, out, blocking)
            elif stmt[3] is not None:
                self.run(stmt[3], env, out, blocking)
        elif k == "case":
            sel = self.ev(stmt[1], env)
            for labels, body in stmt[2]:
                if any(self.ev(x, env) == sel for x in labels):
                    self.run(body, env, out, blocking)
                    return
            if stmt[3] is not None:
                self.run(stmt[3], env, out, blocking)
        else:
            _, _op, (name, hi, lo), rhs = stmt
            w = self.width[name]
            val = self.ev(rhs, env)
            dst = env if blocking else out
            if hi is None:
                dst[name] = val & ((1 << w) - 1)
            else:
                h, l = self.ev(hi, env), self.ev(lo, env)
                cur = dst.get(name, env[name])
                mask = ((1 << (h - l + 1)) - 1) << l
                dst[name] = (cur & ~mask) | ((val << l) & mask)


def load_design():
    types = strip_comments(open(f"{REF}/common/DTEngine_Types.sv").read())
    consts = {}
    for name, expr in re.findall(r"\bparameter\s+(\w+)\s*=\s*([^;,]+);", types):
        try:
            consts[name] = Evaluator(None, {k: (v, 32) for k, v in consts.items()}).ev(Parser(tokenize(expr)).parse())[0]
        except Exception:
            pass
    core = strip_comments(open(f"{REF}/Core.sv").read())
    rls = strip_comments(open(f"{REF}/core/RLS.v").read())
    consts.update({"IDLE": 0, "PROG_MODE": 1, "PROCESS_MODE": 2, "ENGINE_DONE": 3})
    m = re.search(r"localparam\s*\[1:0\]\s*IDLE\s*=\s*2'b00,\s*PROG_MODE\s*=\s*2'b01,\s*PROCESS_MODE\s*=\s*2'b10", core)
    assert m, "state encoding moved"
    consts["DATA_LINE_DISTR_LEVELS"] = 3
    assert consts["NUM_DTPU_CLUSTERS"] == 8 and consts["NUM_PUS_PER_CLUSTER"] == 8
    # the ONE repair (see the module docstring)
    fixed, n = re.subn(r"(IDLE:\s*begin\s*started\s*<=\s*0;\s*if\(\s*)start_core(\s*\)\s*begin\s*core_fsm_state\s*<=\s*PROG_MODE;)",
                       r"\1start_core_d1\2", core)
    assert n == 1, "the IDLE arm of the core FSM no longer looks as documented"
    core = fixed
    # struct members and the [0][0] root of the distribution tree become plain names; array element selects by the
    # running cluster index become one input / a bit select
    core = re.sub(r"InDataFIFO_dout\.(\w+)", r"InDataFIFO_dout_\1", core)
    core = re.sub(r"(data_line_distr\w*)\[0\]\[0\]", r"\1_0_0", core)
    core = core.replace("partial_aggregation_out[curr_cluster]", "partial_aggregation_out_sel")
    rls = rls.replace("DATA_WIDTH_BITS", "3").replace("DATA_WIDTH", "8")
    for k in sorted(consts, key=len, reverse=True):
        core = re.sub(rf"\b{k}\b", str(consts[k]), core)
        rls = re.sub(rf"\b{k}\b", str(consts[k]), rls)
    width = {}
    for text in (core, rls):
        for rng, names in re.findall(r"\b(?:input|output)?\s*(?:wire|reg)\s*(\[[^\]]+\])?\s*([\w\s,]+?)\s*[;,)\[]", text):
            w = 1
            if rng:
                hi, lo = rng[1:-1].split(":")
                w = int(eval(hi.replace("**", "^").replace("^", "**"))) - int(eval(lo)) + 1
            for nm in [x.strip() for x in names.split(",") if x.strip()]:
                width.setdefault(nm, w)
    width.update({"InDataFIFO_dout_data_valid": 1, "InDataFIFO_dout_last": 1, "InDataFIFO_dout_prog_mode": 1, "InDataFIFO_dout_data": 128,
                  "data_line_distr_valid_0_0": 1, "data_line_distr_last_0_0": 1, "data_line_distr_ctrl_0_0": 1, "data_line_distr_mode_0_0": 2,
                  "data_line_distr_en_0_0": 8, "data_line_distr_pu_0_0": 3, "data_line_distr_0_0": 128, "partial_aggregation_out_sel": 32,
                  "partial_aggregation_out_valid": 8, "shifted_data": 8, "data_in": 8, "start_core": 1, "rst_n": 1, "clk": 1})
    blocks = {}
    for sens, ast, pos in always_blocks(core):
        src = core[pos:pos + 6000]
        head = src[:src.find("end") + 3] if False else src
        if sens == "*" and "schedule_to_shift" in head[:900] and "shift_enable" in head[:400]:
            blocks.setdefault("comb", ast)
        elif "posedge" in sens and re.match(r"always\s*@\s*\([^)]*\)\s*begin\s*if\s*\(\s*~rst_n\s*\|\s*start_core\s*\)", head):
            blocks.setdefault("fsm", ast)
        elif "posedge" in sens and "curr_pu" in head[:700] and "tuples_passed" in head[:700]:
            blocks.setdefault("stamp", ast)
        elif "posedge" in sens and "tuple_cluster_offset" in head[:500] and "partial_leaf_aggreg_value" in head[:300]:
            blocks.setdefault("walk", ast)
        elif "posedge" in sens and re.match(r"always\s*@\s*\([^)]*\)\s*begin\s*if\s*\(\s*~rst_n\s*\)\s*begin\s*start_core_d1", head):
            blocks.setdefault("d1", ast)
    assert set(blocks) == {"comb", "fsm", "stamp", "walk", "d1"}, sorted(blocks)
    rb = [b for b in always_blocks(rls) if "posedge" in b[0]]
    blocks["rls"] = rb[0][1]  # the NUM_DTPU_CLUSTERS == 8 arm of the generate-if comes first
    assigns = {}
    for name in ("target_clusters_ready", "curr_cluster", "curr_cluster_valid", "InDataFIFO_re"):
        m = re.search(rf"\bassign\s+{name}\s*=\s*([^;]+);", core)
        assert m, name
        assigns[name] = Parser(tokenize(m.group(1))).parse()
    for nm in ("target_clusters_ready", "curr_cluster_valid", "InDataFIFO_re"):
        width[nm] = 1
    width["curr_cluster"] = 3
    return consts, width, blocks, assigns


def simulate(design, C, T, n_tuples, wl=2, fl=1, tl=2):
    consts, width, blocks, assigns = design
    sim = Sim(width)
    s = sim.sig
    prog = 0
    for k in range(0, 8, C):
        prog |= 1 << k                      # one model replica per C clusters (the host's CSR 204 value, ddt_csr_encode)
    s.update({"prog_schedule": prog, "proc_schedule": (1 << C) - 1, "num_clusters_per_tuple": C,
              "num_clusters_per_tuple_minus_one": C - 1, "clusters_ready": 0xFF, "aggregator_ready": 1,
              "partial_aggregation_out_valid": 0, "tuple_out_data_ready": 1})
    PW, PF = consts["TREE_WEIGHTS_PROG"], consts["TREE_FEATURE_INDEX_PROG"]
    stream = []
    for mode, lines in ((PW, wl), (PF, fl)):
        for _ in range(T):
            for ln in range(lines):
                stream.append((0, int(ln == lines - 1), mode))
    for _ in range(n_tuples):
        for ln in range(tl):
            stream.append((1, int(ln == tl - 1), 0))
    stamps, results = [], []

    def settle(env):
        for _ in range(3):  # comb block <-> continuous assigns: fixed point
            sim.run(blocks["comb"], env, None, True)
            for nm, ex in assigns.items():
                env[nm] = sim.ev(ex, env) & ((1 << width[nm]) - 1)
        env["data_in"] = env["schedule_to_shift"]
        env["shifted_schedule"] = env["shifted_data"]

    def clock(env):
        settle(env)
        nxt = {}
        for name in ("fsm", "stamp", "walk", "d1", "rls"):
            sim.run(blocks[name], env, nxt, False)
        env.update(nxt)
        env["shifted_schedule"] = env["shifted_data"]

    # reset, then the start pulse
    s.update({"rst_n": 0, "start_core": 0, "InDataFIFO_valid_out": 0})
    clock(s)
    s.update({"rst_n": 1, "start_core": 1})
    clock(s)
    s["start_core"] = 0
    clock(s)
    clock(s)
    assert s["core_fsm_state"] == consts["PROG_MODE"], "the FSM did not leave IDLE"
    for (dv, last, mode) in stream:
        s.update({"InDataFIFO_valid_out": 1, "InDataFIFO_dout_data_valid": dv, "InDataFIFO_dout_last": last, "InDataFIFO_dout_prog_mode": mode})
        clock(s)
        # what the distribution tree now carries towards the clusters for THIS line
        assert s["data_line_distr_last_0_0"] == last and (s["data_line_distr_mode_0_0"] & 1) == dv
        if last:
            stamps.append((dv, mode, s["data_line_distr_en_0_0"], s["data_line_distr_pu_0_0"]))
    s["InDataFIFO_valid_out"] = 0
    clock(s)
    # result side: every cluster has its partial sum ready; watch which one the accumulator takes and when `last` is raised
    s["partial_aggregation_out_valid"] = 0xFF
    for _ in range(n_tuples * C):
        settle(s)
        results.append(s["curr_cluster"])
        clock(s)
        results[-1] = (results[-1], s["partial_leaf_aggreg_value_last"])
    w = [(en, pu) for dv, mode, en, pu in stamps if not dv and mode == PW]
    f = [(en, pu) for dv, mode, en, pu in stamps if not dv and mode == PF]
    t = [en for dv, mode, en, pu in stamps if dv]
    assert len(w) == T and len(f) == T and len(t) == n_tuples
    return np.array(w, np.uint8), np.array(f, np.uint8), np.array(t, np.uint8), np.array(results, np.uint8)


def main():
    design = load_design()
    cases = [(C, T) for C in (1, 2, 4, 8) for T in (8, 24, 64, 128 * C if C < 8 else 256)]
    out = {"cases": np.array(cases, np.uint32)}
    for C, T in cases:
        w, f, t, r = simulate(design, C, T, 19)
        out[f"w_{C}_{T}"], out[f"f_{C}_{T}"], out[f"t_{C}_{T}"], out[f"r_{C}_{T}"] = w, f, t, r
        print(f"C={C} T={T}: first tree masks {[bin(x)[2:].zfill(8) for x in w[::8, 0][:5]]} PUs {w[:9, 1].tolist()} "
              f"tuple masks {[bin(x)[2:].zfill(8) for x in t[:4]]} result order {r[:2 * C, 0].tolist()}")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
