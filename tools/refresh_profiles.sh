#!/bin/bash
# Re-collect the files under profiles/ for one round on the GPU box (run through gpurun):
#   tools/refresh_profiles.sh r04 [pmc workloads ...]   -> gpurun_out/r04/{bench.json,stats/,pmc_*/,workloads.jsonl}
# then, back in the container:  tools/collect_profiles.sh r04   (copies the summaries into profiles/)
# Counter passes: the headline workload always; further workloads (default: miniboone_hint_10 plus_hint_4
# conditional_hint_4_full) get FETCH_SIZE / WRITE_SIZE / SQ passes of their own -> pmc_<workload>_<pass>/.
R=${1:-r04}; shift
PMC_WL=${@:-miniboone_hint_10 plus_hint_4 conditional_hint_4_full}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o final -- python3 bench.py --no-cpu-baseline --no-other-workloads > $O/stats.log 2>&1
SQ1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_FETCH_SIZE -- python3 tools/steps.py > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_WRITE_SIZE -- python3 tools/steps.py > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/pmc_SQ -- python3 tools/steps.py > $O/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/pmc_MIX -- python3 tools/steps.py > $O/pmc4.log 2>&1
for W in $PMC_WL; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$W -o final -- python3 tools/steps.py $W 12 > $O/stats_$W.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${W}_FETCH_SIZE -- python3 tools/steps.py $W 8 > $O/pmc_${W}_1.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${W}_WRITE_SIZE -- python3 tools/steps.py $W 8 > $O/pmc_${W}_2.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/pmc_${W}_SQ -- python3 tools/steps.py $W 8 > $O/pmc_${W}_3.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/pmc_${W}_MIX -- python3 tools/steps.py $W 8 > $O/pmc_${W}_4.log 2>&1
done
rm -f $O/workloads.jsonl
for W in power_hint_4 plus_hint_4_big; do python bench.py --no-cpu-baseline --no-other-workloads --workload $W 2>/dev/null | tail -1 >> $O/workloads.jsonl; done
# the headline workload at larger batches (row pairs: two 16-row tiles per workgroup on one weight stream)
for B in 8192 16384; do python bench.py --no-cpu-baseline --batch $B 2>/dev/null | tail -1 >> $O/workloads.jsonl; done
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
tail -1 $O/bench.json | cut -c1-300
