#!/bin/bash
# A/B of environments on the captured step (shipped library): tools/r6_ab_env.sh "<VAR=val ...;VAR=val ...>" "<workloads>" [reps]
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6ab; mkdir -p $O; rm -f $O/env.txt
IFS=';' read -ra ENVS <<< "$1"
for W in $2; do for R in $(seq ${3:-2}); do for E in "${ENVS[@]}"; do
  echo -n "[$E] $W " >> $O/env.txt
  env $E python bench.py --workload $W --no-cpu-baseline --no-other-workloads --no-module-path --no-live-traffic 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']), round(r['ms_per_step']*1e3,2), r['rep_min_ms'], r['rep_max_ms'])" >> $O/env.txt
done; done; done
cat $O/env.txt
