#!/bin/bash
# session r06_s53: rank_kernel with eight searches per lane and pass (DDT_RANK_ILP=8) against four, config 6 alternating on one box; parity of the ILP-8 form
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s53; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], d["ms_per_step"], r["scoring_launches_per_step"], r["prepass_ms"], r["kernel_ms"], d.get("parity"))'
for i in 1 2 3; do
  ( timeout 600 python bench.py --config 6 --no-cpu-baseline --no-streamed ) > $O/cfg6_ilp4_$i.log 2> $O/cfg6_ilp4_$i.err; tail -1 $O/cfg6_ilp4_$i.log | python -c "$P" ilp4
  ( DDT_RANK_ILP=8 timeout 600 python bench.py --config 6 --no-cpu-baseline --no-streamed ) > $O/cfg6_ilp8_$i.log 2> $O/cfg6_ilp8_$i.err; tail -1 $O/cfg6_ilp8_$i.log | python -c "$P" ilp8
done
( DDT_RANK_ILP=8 timeout 900 python -m pytest tests/test_q16.py tests/test_q16_deep.py tests/test_fuzz_gpu.py tests/test_sparse.py tests/test_graph_capture.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "Extension modules" ) > $O/gpu_tests_ilp8.log; grep -n "passed\|failed" $O/gpu_tests_ilp8.log | tail -1
( DDT_RANK_ILP=8 timeout 300 python tools/run_shape.py --trees 512 --levels 12 --features 60 --wide-features 200 --rows 4000000 ) 2>&1 | tail -1
( timeout 300 python tools/run_shape.py --trees 512 --levels 12 --features 60 --wide-features 200 --rows 4000000 ) 2>&1 | tail -1
