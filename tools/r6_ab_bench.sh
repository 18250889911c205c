#!/bin/bash
# A/B of library builds on the captured step: tools/r6_ab_bench.sh "<libs>" "<workloads>" [reps]   (lib "main" = the shipped one)
LIBS=${1:-"main"}; WLS=${2:-"power_hint_8"}; REPS=${3:-2}
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6ab; mkdir -p $O; rm -f $O/bench.txt
for W in $WLS; do for R in $(seq $REPS); do for L in $LIBS; do
  F=hint_amd/lib/libhint_amd_$L.so; [ "$L" = "main" ] && F=hint_amd/lib/libhint_amd.so
  echo -n "$L $W " >> $O/bench.txt
  HINT_AMD_LIB=$PWD/$F python bench.py --workload $W --no-cpu-baseline --no-other-workloads --no-module-path --no-live-traffic 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']), round(r['ms_per_step']*1e3,2), r['rep_ms'])" >> $O/bench.txt
done; done; done
cat $O/bench.txt
