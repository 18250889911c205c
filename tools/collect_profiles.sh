#!/bin/bash
# Copy what tools/refresh_profiles.sh left under gpurun_out/<round>/ into profiles/ (tracked).
R=${1:-r03}
cd "$(dirname "$0")/.."
O=gpurun_out/$R
tail -1 $O/bench.json | python -m json.tool > profiles/${R}_bench.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) profiles/${R}_kernel_stats.csv
python -c "
import json,sys
rows=[json.loads(l) for l in open('$O/workloads.jsonl') if l.startswith('{')]
json.dump(rows, open('profiles/${R}_workloads.json','w'), indent=1)"
python tools/pmc_summary.py profiles/${R}_pmc_summary.json 8 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ $O/pmc_MIX
