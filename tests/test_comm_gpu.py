"""The multi-GPU job behind the C-ABI (include/ddt.h ddt_comm_* / ddt_group_*; csrc/ddt_comm.cpp): RCCL calls issued from
C++, chunk pipeline on two HIP streams.  One GPU is all a gpurun box has, so these tests run the REAL code path --
ncclCommInitRank / ncclCommInitAll, ncclAllReduce, grouped ncclSend/ncclRecv, ncclAllGather, the chain add, the
event choreography between the caller's stream and the comm stream -- in a ONE-rank communicator, where every
collective is the identity: results must equal the plain single-engine call bit for bit, for every chunking.  The
multi-rank arithmetic (shard boundaries, chain order) is covered by tests/test_sharded_gloo.py on CPU and by the
virtual-rank tests of test_gpu_parity.py; the driver's 8-GPU bench is what runs this path with real peers."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
import ddt

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def eng():
    e = ddt.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def comm(eng):
    c = ddt.Comm(eng, 0, 1, ddt.comm_unique_id())
    yield c
    c.close()


@pytest.mark.parametrize("T,D,F,rows,dist", [(1000, 8, 32, 5000, 0), (100, 6, 28, 3001, 1), (37, 8, 20, 1029, 1)])
def test_one_rank_sharded_job_equals_plain_call(eng, comm, T, D, F, rows, dist):
    import torch

    m = O.gen_model(T, D, F, dist)
    x = O.gen_tuples(0, rows, F, dist)
    want = O.score(m, x)
    eng.load_model(ddt.make_params(T, D, F), m.wlines, m.flines, 0, 1)
    d = torch.from_numpy(x.view(np.int32)).cuda()
    for chunk in (12_500_000, 1024, 1000, 7):  # one chunk; several; ragged tail; many tiny chunks
        comm.set_option("chunk_rows", chunk)
        for combine in (ddt.COMBINE_ALLREDUCE, ddt.COMBINE_CHAIN):
            got = comm.score_sharded(d, combine=combine)
            torch.cuda.synchronize()
            assert np.array_equal(_bits(got.cpu().numpy()), _bits(want)), (chunk, combine)
        got = comm.score_rowsharded(d)
        torch.cuda.synchronize()
        assert np.array_equal(_bits(got.cpu().numpy()), _bits(want))
    comm.set_option("chunk_rows", 12_500_000)


def test_one_rank_sharded_sparse_and_back_to_back_calls(eng, comm):
    import torch

    s = O.gen_sparse_model(24, 13, 20, 4, 650, 1)
    x = O.gen_tuples(0, 4000, 20, 1)
    want = O.score_sparse(s, x)
    eng.load_model_sparse(ddt.make_sparse_params(24, 13, 20), s.node_lines, s.first)
    d = torch.from_numpy(x.view(np.int32)).cuda()
    comm.set_option("chunk_rows", 900)
    outs = [comm.score_sharded(d, combine=c) for c in (1, 0, 1, 1, 0)]  # workspace slots reused across calls without a host sync
    torch.cuda.synchronize()
    for o in outs:
        assert np.array_equal(_bits(o.cpu().numpy()), _bits(want))
    comm.set_option("chunk_rows", 12_500_000)


def test_one_rank_sharded_classes(eng, comm):
    import torch

    T, D, F, K, rows = 90, 6, 16, 3, 2500
    m = O.gen_model(T, D, F, 1)
    x = O.gen_tuples(0, rows, F, 1)
    labels, cs = O.classify(m, x, K)
    eng.load_model_multiclass(ddt.make_params(T, D, F, clusters=1), m.wlines, m.flines, K, True, 0, 1)
    d = torch.from_numpy(x.view(np.int32)).cuda()
    for chunk in (12_500_000, 700):
        comm.set_option("chunk_rows", chunk)
        for combine in (0, 1):
            gl, gs = comm.classify_sharded(d, combine=combine)
            torch.cuda.synchronize()
            assert np.array_equal(gl.cpu().numpy(), labels) and np.array_equal(_bits(gs.cpu().numpy()), _bits(cs)), (chunk, combine)
    comm.set_option("chunk_rows", 12_500_000)
    with pytest.raises(ddt.DDTError):  # the scalar call refuses a multi-class model
        comm.score_sharded(d)


def test_one_rank_hybrid_job(eng):
    """ddt_comm_create_hybrid on the real RCCL: ncclCommInitRank + ncclCommSplit with one rank (one row group of one tree shard), the
    hybrid calls on device and host buffers, the refusals, ddt_comm_abort.  Every collective is the identity: results == the plain call."""
    import torch

    T, D, F, rows = 300, 8, 32, 5003
    m = O.gen_model(T, D, F, 0)
    x = O.gen_tuples(0, rows, F, 0)
    want = O.score(m, x)
    eng.load_model(ddt.make_params(T, D, F), m.wlines, m.flines, 0, 1)
    c = ddt.Comm(eng, 0, 1, ddt.comm_unique_id(), tree_ranks=1)
    lay = c.layout()
    assert (lay.n_ranks, lay.tree_ranks, lay.row_groups, lay.row_group) == (1, 1, 1, 0)
    d = torch.from_numpy(x.view(np.int32)).cuda()
    for chunk in (12_500_000, 1024, 700):
        c.set_option("chunk_rows", chunk)
        for combine in (ddt.COMBINE_ALLREDUCE, ddt.COMBINE_CHAIN):
            for gather in (True, False):
                got = c.score_hybrid(d, combine=combine, gather=gather)
                torch.cuda.synchronize()
                assert np.array_equal(_bits(got.cpu().numpy()), _bits(want)), (chunk, combine, gather)
    assert np.array_equal(_bits(c.score(x)), _bits(want))                    # host buffers through the hybrid communicator
    with pytest.raises(ddt.DDTError):
        c.score_sharded(d)                                                    # the tree-sharded call refuses a hybrid communicator
    plain = ddt.Comm(eng, 0, 1, ddt.comm_unique_id())
    with pytest.raises(ddt.DDTError):
        plain.score_hybrid(d)                                                 # ... and the hybrid call a plain one
    plain.close()
    c.abort()
    with pytest.raises(ddt.DDTError):
        c.score_hybrid(d)                                                     # dead after ddt_comm_abort
    c.close()
    g = ddt.Group([0], tree_ranks=1)                                          # the single-process form
    g.load_model(ddt.make_params(T, D, F), m.wlines, m.flines)
    assert np.array_equal(_bits(g.score(x)), _bits(want))
    g.close()


def test_group_of_one_device_host_buffers(eng):
    m = O.gen_model(200, 8, 32, 0)
    x = O.gen_tuples(0, 30_000, 32, 0)
    want = O.score(m, x)
    g = ddt.Group([0])
    g.load_model(ddt.make_params(200, 8, 32), m.wlines, m.flines)
    for combine in (0, 1):
        assert np.array_equal(_bits(g.score(x, combine=combine)), _bits(want))
    s = O.gen_sparse_model(16, 12, 20, 4, 600, 1)
    xs = O.gen_tuples(0, 3000, 20, 1)
    g.load_model_sparse(ddt.make_sparse_params(16, 12, 20), s.node_lines, s.first)
    assert np.array_equal(_bits(g.score(xs)), _bits(O.score_sparse(s, xs)))
    # a multi-class model through the group: labels + per-class sums; more rows than one super-chunk would need no change
    T, D, F, K, rows = 60, 6, 16, 3, 2100
    m = O.gen_model(T, D, F, 1)
    xc = O.gen_tuples(0, rows, F, 1)
    labels, cs = O.classify(m, xc, K)
    g.load_model_multiclass(ddt.make_params(T, D, F, clusters=1), m.wlines, m.flines, K, True)
    for combine in (0, 1):
        gl, gs = g.classify(xc, combine=combine)
        assert np.array_equal(gl, labels) and np.array_equal(_bits(gs), _bits(cs)), combine
    with pytest.raises(ddt.DDTError):  # the scalar call refuses a multi-class model
        g.score(xc)
    g.close()
    with pytest.raises(ddt.DDTError):  # no such device
        ddt.Group([0, 63])


def test_cli_runs_the_multi_gpu_job_without_python(tmp_path):
    """`ddt_cli score --devices 1`: C++ host -> ddt_group_* -> RCCL, no Python in the scoring process."""
    pre = str(tmp_path / "job")
    T, D, F, n = 120, 6, 28, 4099
    subprocess.check_call([ddt.CLI_PATH, "gen", "--trees", str(T), "--levels", str(D), "--features", str(F), "--rows", str(n),
                           "--dist", "1", "--prefix", pre])
    want = O.score(O.gen_model(T, D, F, dist=1), O.gen_tuples(0, n, F, dist=1))
    # one invocation: a fresh process pays for loading /opt/rocm's librccl (hundreds of MB) before ncclCommInitAll; the chain
    # combine runs through the same C++ path in-process above
    for combine in ("allreduce",):
        out = subprocess.check_output([ddt.CLI_PATH, "score", "--csr", pre + ".csr", "--weights", pre + ".weights", "--findex", pre + ".findex",
                                       "--tuples", pre + ".tuples", "--out", pre + ".results", "--devices", "1", "--combine", combine],
                                      env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")).decode()
        assert f"scored {n} tuples on 1 device(s)" in out and "RCCL" in out
        res = np.fromfile(pre + ".results", np.float32)
        assert np.array_equal(res[:n].view(np.uint32), want.view(np.uint32))
