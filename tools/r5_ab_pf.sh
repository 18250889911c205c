#!/bin/bash
# round 5: the new large-batch parity cases + A/B of the general backward's L2 warm-up (now wired: HINT_PF=0 vs on)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5pf; mkdir -p $O
python -m pytest tests/test_gpu_chain_workloads.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for W in miniboone_hint_10 plus_hint_4 plus_hint_4_big; do
  for PF in 0 1 0 1; do
    echo -n "PF=$PF " >> $O/legs.txt; HINT_PF=$PF python tools/time_legs.py $W 30 2>&1 | tail -1 >> $O/legs.txt
  done
done
cat $O/legs.txt
