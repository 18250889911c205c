#!/bin/bash
# per-kernel average durations of a workload's un-captured steps: tools/r5_kstats.sh <workload> [steps] [ENV=val ...]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
W=${1:-power_hint_8}; N=${2:-20}; shift; shift
O=gpurun_out/r5k; rm -rf $O; mkdir -p $O
for E in "$@"; do export "$E"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o k -- python3 tools/steps.py $W $N > $O/log.txt 2>&1
F=$(find $O/st -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
for r in list(csv.DictReader(open("$F")))[:10]:
    print(r["Name"].split("(")[0][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
rm -rf $O/st
