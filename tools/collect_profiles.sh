#!/bin/bash
# Copy what tools/refresh_profiles.sh left under gpurun_out/<round>/ into profiles/ (tracked).
R=${1:-r04}
cd "$(dirname "$0")/.."
O=gpurun_out/$R
tail -1 $O/bench.json | python -m json.tool > profiles/${R}_bench.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) profiles/${R}_kernel_stats.csv
python -c "
import json,sys
rows=[json.loads(l) for l in open('$O/workloads.jsonl') if l.startswith('{')]
json.dump(rows, open('profiles/${R}_workloads.json','w'), indent=1)"
python tools/pmc_summary.py profiles/${R}_pmc_summary.json 8 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ $O/pmc_MIX
for D in $O/pmc_*_FETCH_SIZE; do
  W=$(basename $D); W=${W#pmc_}; W=${W%_FETCH_SIZE}
  [ -d "$D" ] || continue
  NB=$(python -c "import bench; c=dict(bench.WORKLOADS, **bench.CONDITIONAL)['$W']; print(c['n_blocks'])")
  python tools/pmc_summary.py profiles/${R}_pmc_$W.json $NB $O/pmc_${W}_FETCH_SIZE $O/pmc_${W}_WRITE_SIZE $O/pmc_${W}_SQ $O/pmc_${W}_MIX
  F=$(find $O/stats_$W -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F profiles/${R}_kernel_stats_$W.csv
done
