#!/usr/bin/env python3
"""What a collective's CUs cost the scoring kernel: the rank's shard scored on a stream whose CU mask leaves out k CUs
(hipExtStreamCreateWithCUMask), k = 0, 8, 16, 32, 64 spread evenly over the chip.  RCCL's all-reduce kernels occupy CUs while the
next chunk is being scored (csrc/ddt_comm.cpp pipelines them); on a one-GPU box the masked stream stands in for that loss.
Prints one JSON object; feeds ddt/perf_model.py (Mi355x.rccl_cus / tree_sharded_ms)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "distributed-decisiontrees_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--trees", type=int, default=1000)
    ap.add_argument("--shard-of", type=int, default=8)
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--masked", default="0,8,16,32,64")
    ap.add_argument("--one-xcd", action="store_true", help="take all masked CUs from XCD 0 (at most 24) instead of k / 8 from every XCD")
    args = ap.parse_args()
    import torch

    import ddt

    hip = C.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    L = ddt.lib()
    eng = ddt.Engine(0)
    eng.set_option("variant", args.variant)
    w, f = ddt.synth_model(args.trees, 8, 32, 0)
    eng.load_model(ddt.make_params(args.trees, 8, 32), w, f, min(3, args.shard_of - 1), args.shard_of)
    info = eng.info()
    cus = info.num_cus
    tuples = eng.synth_tuples_device(0, args.rows, 32, 0)
    out = torch.empty(args.rows, dtype=torch.float32, device=tuples.device)
    eng.set_option("reserve_rows", args.rows)
    eng.score_device(tuples, out=out)
    torch.cuda.synchronize()
    res = {"trees_on_this_shard": int(info.tree_end - info.tree_begin), "rows": args.rows, "kernel": info.variant_name.decode(), "cus": cus,
           "masked_from": "XCD 0 only" if args.one_xcd else "every XCD alike", "runs": []}
    for k in [int(v) for v in args.masked.split(",")]:
        words = (cus + 31) // 32
        mask = [0xFFFFFFFF] * words
        if k:
            # CU index = XCD + 8 * (CU inside the XCD) (measured round 4: every 32nd CU off = 8 CUs of ONE XCD, 1.31x slower; a mask
            # that empties an XCD is ignored).  even: k / 8 CUs off in every XCD, like a collective's workgroups, which the dispatcher
            # deals round-robin over the XCDs; uneven (--one-xcd): all k in XCD 0
            for i in range(k):
                cu = (i % 8) + 8 * (i // 8) if not args.one_xcd else 8 * i
                mask[cu // 32] &= ~(1 << (cu % 32))
        arr = (C.c_uint32 * words)(*mask)
        s = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), words, arr)
        if rc:
            res["runs"].append({"masked_cus": k, "error": rc})
            continue
        ms = []
        for it in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = L.ddt_score_device(eng._h, tuples.data_ptr(), args.rows, out.data_ptr(), s)
            hip.hipStreamSynchronize(s)
            ms.append((time.perf_counter() - t0) * 1e3)
        hip.hipStreamDestroy(s)
        res["runs"].append({"masked_cus": k, "ms": round(min(ms[1:]), 4), "rc": rc})
    base = next((r["ms"] for r in res["runs"] if r.get("masked_cus") == 0 and "ms" in r), None)
    if base:
        for r in res["runs"]:
            if "ms" in r:
                r["slowdown"] = round(r["ms"] / base, 4)
                r["ideal"] = round(cus / (cus - r["masked_cus"]), 4)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
