"""Generates tests/golden/*.npz -- small frozen input/output vectors for the scoring path.

The reference (FPGA RTL) ships no golden vectors, so these are produced by the CPU
oracle (oracle/ddt_oracle.c) after it passed its known-answer tests; they freeze today's semantics so
that (a) the oracle cannot drift silently and (b) the GPU tests have fixtures that do not need the
oracle at run time.  Re-run:  python tests/golden/make_golden.py   (from the repo root)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # name, T, D, F, dist, cmp_mode, clusters, rows
    ("cfg1_t8_d4_f16", 8, 4, 16, 0, 0, None, 1000),          # BASELINE config 1 (plumbing case)
    ("mix_t100_d6_f28_missing", 100, 6, 28, 1, 0, None, 777),  # config-2 shape, negatives + missing values
    ("mix_t37_d8_f32_ieee_c4", 37, 8, 32, 1, 1, 4, 515),       # depth 8, IEEE compare extension, C=4
    ("deep_t5_d11_f64", 5, 11, 64, 1, 0, 2, 130),              # generic-kernel territory
]


def main():
    for name, T, D, F, dist, cmp_mode, clusters, rows in CASES:
        m = O.gen_model(T, D, F, dist=dist, cmp_mode=cmp_mode, clusters=clusters)
        x = O.gen_tuples(11, rows, F, dist=dist, missing_bits=m.params.missing_bits)
        ref = O.score(m, x, sum_mode=O.SUM_REF_FLOPOCO)
        f64 = O.score(m, x, sum_mode=O.SUM_F64_SEQ)
        ref2 = O.score(m, x, sum_mode=O.SUM_REF_FLOPOCO, n_devices=2)
        p = m.params
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            params=np.array([p.num_trees, p.num_levels, p.num_features, p.missing_bits, p.weights_lines_per_tree,
                             p.findex_lines_per_tree, p.cmp_mode, p.clusters_per_tuple], np.uint32),
            wlines=m.wlines, flines=m.flines, tuples=x,
            score_ref_bits=ref.view(np.uint32), score_f64_bits=f64.view(np.uint32),
            score_ref_2dev_bits=ref2.view(np.uint32))
        print(name, "rows", rows, "mean", float(ref.mean()))


if __name__ == "__main__":
    main()
