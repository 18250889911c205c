#!/bin/bash
# Re-collect the files under profiles/ for one round on the GPU box (run through gpurun):
#   tools/refresh_profiles.sh r03        -> gpurun_out/r03/{bench.json,stats/,pmc_*/,workloads.jsonl}
# then, back in the container:  tools/collect_profiles.sh r03   (copies the summaries into profiles/)
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o final -- python3 bench.py --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_FETCH_SIZE -- python3 tools/steps.py > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_WRITE_SIZE -- python3 tools/steps.py > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_SQ -- python3 tools/steps.py > $O/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH --output-format csv -d $O/pmc_MIX -- python3 tools/steps.py > $O/pmc4.log 2>&1
rm -f $O/workloads.jsonl
for W in gas_hint_8 miniboone_hint_10 plus_hint_4 power_hint_4 plus_hint_4_big; do python bench.py --no-cpu-baseline --workload $W 2>/dev/null | tail -1 >> $O/workloads.jsonl; done
python bench.py --workload conditional_hint_4_full --steps 50 --warmup 5 2>/dev/null | tail -1 >> $O/workloads.jsonl
# the headline workload at larger batches (row pairs: two 16-row tiles per workgroup on one weight stream)
for B in 8192 16384; do python bench.py --no-cpu-baseline --batch $B 2>/dev/null | tail -1 >> $O/workloads.jsonl; done
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
tail -1 $O/bench.json | cut -c1-300
