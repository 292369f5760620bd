"""Worker of tests/test_comm_two_ranks_gpu.py: ONE rank of a two-rank job whose ranks share cuda:0.

usage: two_rank_worker.py <rank> <id file> <out dir> <job>
The two processes run ddt_comm_create(engine, rank, 2, id) through the real RCCL -- the first execution of csrc/ddt_comm.cpp's
pipeline with a real peer that a 1-GPU box allows, IF RCCL accepts two ranks on one device.  The worker only records what
happened (JSON status + the raw results); the parent compares with the oracle.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "distributed-decisiontrees_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    rank, id_file, out_dir, job = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4]
    status = {"rank": rank, "job": job, "stage": "start"}

    def save():
        with open(os.path.join(out_dir, f"status{rank}.json"), "w") as fh:
            json.dump(status, fh)

    save()
    import torch

    import ddt

    T, D, F, rows, dist = (int(v) for v in os.environ.get("DDT_TWO_RANK_SHAPE", "1000,8,32,40000,1").split(","))
    eng = ddt.Engine(0)
    w, f = ddt.synth_model(T, D, F, dist)
    if job == "rows":
        eng.load_model(ddt.make_params(T, D, F), w, f, 0, 1)   # replicas: every rank holds the whole model
    else:
        eng.load_model(ddt.make_params(T, D, F), w, f, rank, 2)  # PCIeReceiver.sv:241-264: contiguous shard `rank` of 2
    if rank == 0:
        uid = ddt.comm_unique_id()
        with open(id_file + ".tmp", "wb") as fh:
            fh.write(uid)
        os.replace(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            if time.time() - t0 > 60:
                status["stage"] = "no id"
                save()
                return 3
            time.sleep(0.05)
        with open(id_file, "rb") as fh:
            uid = fh.read()
    status["stage"] = "comm_create"
    save()
    try:
        comm = ddt.Comm(eng, rank, 2, uid)
    except ddt.DDTError as e:  # RCCL refused (e.g. two ranks on one device)
        status.update(stage="refused", error=str(e))
        save()
        return 0
    status["stage"] = "created"
    save()
    d = eng.synth_tuples_device(0, rows, F, dist)
    comm.set_option("chunk_rows", 9000)  # several chunks + a ragged tail: the chunk pipeline really overlaps
    res = {}
    if job == "rows":
        res["rows"] = comm.score_rowsharded(d)
    else:
        res["chain"] = comm.score_sharded(d, combine=ddt.COMBINE_CHAIN)
        res["allreduce"] = comm.score_sharded(d, combine=ddt.COMBINE_ALLREDUCE)
        res["chain2"] = comm.score_sharded(d, combine=ddt.COMBINE_CHAIN)  # back to back: workspace slots reused
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"result{rank}.npz"), **{k: v.cpu().numpy() for k, v in res.items()})
    status.update(stage="done", kernel=eng.info().variant_name.decode())
    save()
    comm.close()
    eng.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
