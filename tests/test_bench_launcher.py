"""bench.py launches its own ranks when `--gpus N` (N > 1) arrives without a launcher around it (`python bench.py --gpus 8`
used to exit with an error).  CPU part: the command it builds, and that the re-exec really happens (on this GPU-less box every rank then
stops at "needs a GPU": the launcher path ran, rc != 0, no JSON line).  GPU part (1-GPU box, two gloo ranks sharing the GPU): one valid line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launcher_command_line():
    import bench

    cmd = bench.self_launch_command(4, ["--gpus", "4", "--steps", "3"], port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3"]
    port = int(bench.self_launch_command(2, [])[bench.self_launch_command(2, []).index("--master-port") + 1])
    assert 1024 < port < 65536  # a free port picked at launch time


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="the GPU-less half: ranks stop at 'needs a GPU'")
def test_gpus_2_without_world_size_becomes_the_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--rows", "1000", "--trees", "16"],
                       capture_output=True, env=env, timeout=600)
    err = r.stderr.decode()
    assert "without WORLD_SIZE: launching" in err and "torch.distributed.run" in err   # it became the launcher ...
    assert "needs a GPU" in err and r.returncode != 0                                  # ... and its ranks ran bench.py's own checks
    assert r.stdout.decode().strip() == ""                                              # no half-printed line


@pytest.mark.gpu
def test_gpus_2_self_launched_on_one_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--rows", "2000000", "--steps", "2", "--warmup", "1",
                        "--no-other-modes", "--cpu-seconds", "0.5"], capture_output=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["unit"] == "Mtuples/s" and j["scaling"] == "strong"
    assert abs(j["value"] - 2000000 / j["ms_per_step"] / 1e3) / j["value"] < 1e-3
    # two real ranks (gloo, one GPU): the combined result of the timed job against the oracle's 2-device model, the oracle timed on rank 0
    par = j["parity"]
    assert par["required"] == "within_tolerance" and par["within_tolerance"] is True and par["rows_checked"] > 0, par
    assert j["cpu_baseline"]["value"] > 0 and "wait behind a barrier" in j["cpu_baseline"]["sample"]


def test_hybrid_legs_geometry():
    """other_modes' hybrid legs: tree groups of 2 / 4 consecutive ranks that leave >= 2 row groups; row slices in whole 1024-tuple tiles
    that cover every tuple exactly once."""
    import bench
    import ddt

    sys.path.insert(0, os.path.join(ROOT, "distributed-decisiontrees_amd"))
    assert bench.hybrid_tree_groups(8) == [2, 4] and bench.hybrid_tree_groups(4) == [2] and bench.hybrid_tree_groups(2) == []
    assert bench.hybrid_tree_groups(1) == [] and bench.hybrid_tree_groups(6) == [2]
    for n, Gr in ((100_000_000, 4), (100_000_000, 2), (10_000_001, 4), (3000, 2), (1000, 4)):
        cuts = [ddt.hybrid_rows(n, Gr, rg) for rg in range(Gr)]      # the C-ABI's own split (ddt_hybrid_rows, host-only)
        assert cuts[0][0] == 0 and cuts[-1][1] == n and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        assert all(lo % 1024 == 0 for lo, _ in cuts if lo < n)
