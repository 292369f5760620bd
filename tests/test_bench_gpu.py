"""bench.py's contract on a GPU box, at sizes that take seconds: ONE JSON line on stdout with the driver's keys, the
`roofline` / `cpu_baseline` / `parity` / `streamed` objects at N = 1, and the multi-GPU branch (the C++ chunk pipeline and its
collectives in a one-rank communicator, --force-collectives) incl. the `scaling_detail` leg the driver's N > 1 runs print."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"}


def _bench(*args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29611")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                          # the driver wants exactly one line on stdout
    return json.loads(lines[0])


def test_default_line_has_the_contract_keys_and_the_added_objects():
    j = _bench("--rows", "300000", "--trees", "300", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0.3")
    assert KEYS <= set(j) and j["n_gpus"] == 1 and j["unit"] == "Mtuples/s" and j["dtype"] == "f32" and j["vs_baseline"] is None
    assert j["value"] > 0 and abs(j["value"] - 300000 / j["ms_per_step"] / 1e3) / j["value"] < 1e-3
    ro = j["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-4
    assert ro["kernel_ms"] > 0 and ro["kernel_ms"] + ro["prepass_ms"] <= j["ms_per_step"] * 1.05
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] >= 1 and j["cpu_baseline"]["value"] > 0
    assert j["parity"]["bit_exact"] is True and j["parity"]["rows_checked"] > 0
    assert j["streamed"]["bit_exact_vs_resident"] is True


@pytest.mark.parametrize("extra", [[], ["--combine", "chain"], ["--taper", "1", "--chunk-rows", "70000"]])
def test_multi_gpu_branch_in_a_one_rank_communicator(extra):
    j = _bench("--rows", "300000", "--trees", "200", "--steps", "2", "--warmup", "1", "--force-collectives", "--no-cpu-baseline", "--no-streamed",
               *extra)
    assert KEYS <= set(j) and j["n_gpus"] == 1 and j["value"] > 0
    assert j["config"]["collectives"].startswith("C-ABI") and j["config"]["combine"] in ("allreduce", "chain")
    d = j["scaling_detail"]
    assert "error" not in d, d
    assert d["shard_compute_only_ms"] > 0 and d["trees_on_this_rank"] == 200
    assert abs(d["combine_overhead_ms"] - (j["ms_per_step"] - d["shard_compute_only_ms"])) < 1e-3
