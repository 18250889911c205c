#!/bin/bash
# full GPU suite + the driver's bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5b; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -2
python bench.py $@ > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('value',r['value'],'ms',r['ms_per_step'],'roof',r.get('roofline',{}).get('frac'), r.get('roofline',{}).get('avg_launch_us'))
print({k:round(v,1) for k,v in r.get('kernels_in_step_us',{}).items()})
for k,v in (r.get('other_workloads') or {}).items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('nll_rel_err'))
"
