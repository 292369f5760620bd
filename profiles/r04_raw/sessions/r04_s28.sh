#!/bin/bash
# config 1: L2 <-> fabric counters of the stream kernel with phased and with direct result stores (reads / writes to the fabric, reads in flight)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s28; rm -rf "$OUT"; mkdir -p "$OUT"
for mode in phased direct; do
  opt=""; [ $mode = direct ] && opt="--opt stream_res_tiles=1"
  S="python $GRAFT_REPO_ROOT/tools/run_shape.py --trees 8 --levels 4 --features 16 --rows 200000000 --reps 6 $opt"
  i=0
  for ctrs in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_CYCLE_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    ( cd /tmp && timeout 200 rocprofv3 --pmc $ctrs -d $OUT/${mode}_pmc$i -o pmc -- $S ) > $OUT/${mode}_pmc$i.log 2>&1; echo "$mode pass $i rc=$?"
  done
  python tools/pmc_dump.py $OUT/${mode}_pmc* > $OUT/${mode}_counters.json
done
python - <<'PY'
import json,os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_s28'
for mode in ('phased','direct'):
    d=json.load(open(f'{out}/{mode}_counters.json'))
    for k,v in d.items():
        if 'score_stream' in k:
            print(mode, {c: round(v[c],1) for c in v if c.startswith(('TCC','GRBM'))}, 'ns', v.get('avg_ns_under_pmc',{}).get('TCC_CYCLE_sum'))
PY
