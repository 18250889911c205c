#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5c; mkdir -p $O
python -m pytest tests/test_gpu_conditional.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|Error|error" $O/pytest.txt | tail -8
HINT_COND_LEGACY=1 python bench.py --no-cpu-baseline --workload conditional_hint_4_full 2>$O/legacy.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('legacy', r['value'], r['ms_per_step'])"
python bench.py --no-cpu-baseline --workload conditional_hint_4_full 2>$O/new.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('new', r['value'], r['ms_per_step'], r.get('nll_check'))"
tail -3 $O/new.err
