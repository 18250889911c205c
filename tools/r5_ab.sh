#!/bin/bash
# A/B of library builds: tools/r5_ab.sh "<libs>" "<workloads>" [reps] [pytest args]   (lib "main" = the shipped one)
LIBS=${1:-"base main"}; WLS=${2:-"power_hint_8"}; REPS=${3:-2}; PYT=${4:-}
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5ab; mkdir -p $O; rm -f $O/legs.txt
if [ -n "$PYT" ]; then python -m pytest $PYT -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt; fi
for W in $WLS; do for R in $(seq $REPS); do for L in $LIBS; do
  F=hint_amd/lib/libhint_amd_$L.so; [ "$L" = "main" ] && F=hint_amd/lib/libhint_amd.so
  echo -n "$L " >> $O/legs.txt; HINT_AMD_LIB=$PWD/$F python tools/time_legs.py $W 40 2>&1 | tail -1 | sed "s|$PWD/hint_amd/lib/||" >> $O/legs.txt
done; done; done
cat $O/legs.txt
