#!/bin/bash
# tools/r6_env.sh "<VAR=val ...;VAR=val ...>" "<workloads>" [steps]: time_legs under different environments (shipped library)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6env; mkdir -p $O; rm -f $O/env.txt
IFS=';' read -ra ENVS <<< "$1"
for W in $2; do for E in "${ENVS[@]}"; do
  echo -n "[$E] " >> $O/env.txt; env $E python tools/time_legs.py $W ${3:-40} 2>&1 | tail -1 >> $O/env.txt
done; done
cat $O/env.txt
