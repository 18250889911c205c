#!/bin/bash
# Re-collect the files under profiles/ for one round on the GPU box (run through gpurun):
#   tools/refresh_profiles.sh r05 [pmc workloads ...]   -> gpurun_out/r05/summary/{r05_bench.json,r05_kernel_stats*.csv,r05_pmc_*.json,r05_workloads.json}
# (the raw rocprofv3 output is summarised on the box and removed: gpurun copies back at most 64 MiB)
# then, back in the container:  cp gpurun_out/r05/summary/* profiles/
# Counter passes: the headline workload always; further workloads (default: gas_hint_8 miniboone_hint_10 plus_hint_4
# conditional_hint_4_full - every BASELINE config) get FETCH_SIZE / WRITE_SIZE / SQ passes of their own.
R=${1:-r06}; shift
export GIT_REV=${GIT_REV:-unknown}       # (.git does not travel to the GPU box: pass the commit the tree is at, tools/pmc_summary.py records it)
PMC_WL=${@:-gas_hint_8 miniboone_hint_10 plus_hint_4 conditional_hint_4_full}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R; S=$O/summary; mkdir -p $S
python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | python -m json.tool > $S/${R}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o final -- python3 bench.py --no-cpu-baseline --no-other-workloads --no-module-path --no-live-traffic > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $S/${R}_kernel_stats.csv
SQ1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH"
passes() {   # $1 = directory prefix, $2.. = arguments of tools/steps.py
  P=$1; shift
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${P}_FETCH_SIZE -- python3 tools/steps.py $@ > ${P}_1.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${P}_WRITE_SIZE -- python3 tools/steps.py $@ > ${P}_2.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d ${P}_SQ -- python3 tools/steps.py $@ > ${P}_3.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d ${P}_MIX -- python3 tools/steps.py $@ > ${P}_4.log 2>&1
  for L in ${P}_[1-4].log; do grep -q "^ok " $L || { echo "FAILED: $L"; tail -3 $L; }; done
}
passes $O/pmc power_hint_8 30
python tools/pmc_summary.py $S/${R}_pmc_summary.json 8 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ $O/pmc_MIX > $O/pmc_summary.log 2>&1
for W in $PMC_WL; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$W -o final -- python3 tools/steps.py $W 12 > $O/stats_$W.log 2>&1
  F=$(find $O/stats_$W -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $S/${R}_kernel_stats_$W.csv
  passes $O/pmc_$W $W 8
  NB=$(python -c "import bench; c=dict(bench.WORKLOADS, **bench.CONDITIONAL)['$W']; print(c['n_blocks'])")
  python tools/pmc_summary.py $S/${R}_pmc_$W.json $NB $O/pmc_${W}_FETCH_SIZE $O/pmc_${W}_WRITE_SIZE $O/pmc_${W}_SQ $O/pmc_${W}_MIX > $O/pmc_summary_$W.log 2>&1
done
rm -f $O/workloads.jsonl
for W in power_hint_4 plus_hint_4_big; do python bench.py --no-cpu-baseline --no-other-workloads --no-live-traffic --workload $W 2>/dev/null | tail -1 >> $O/workloads.jsonl; done
# the N > 1 code path on ONE GPU (no multi-GPU node behind gpurun): a one-rank RCCL group, the gradient all-reduce captured in
# the step's graph and the separate clamp + Adam launch instead of the optimizer inside the slab reduction - what leaving the
# fused-optimizer path costs, tracked until a node exists
HSA_ENABLE_IPC_MODE_LEGACY=0 HINT_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --no-cpu-baseline --no-other-workloads 2>$O/force_dist.err | tail -1 | python -c "
import json,sys
l=sys.stdin.read().strip()
if l.startswith('{'):
    r=json.loads(l); r['note']='HINT_FORCE_DIST=1: one-rank RCCL group on one GPU (all-reduce captured + separate hint_adam_kernel)'; print(json.dumps(r))
" >> $O/workloads.jsonl
# the headline workload at larger batches (row pairs: two 16-row tiles per workgroup on one weight stream)
for B in 8192 16384; do python bench.py --no-cpu-baseline --batch $B 2>/dev/null | tail -1 >> $O/workloads.jsonl; done
python -c "
import json
rows=[json.loads(l) for l in open('$O/workloads.jsonl') if l.startswith('{')]
json.dump(rows, open('$S/${R}_workloads.json','w'), indent=1)"
# keep the summaries and the logs, drop the raw traces
find $O -mindepth 1 -maxdepth 1 -type d ! -name summary -exec rm -rf {} +
du -sh $O; ls $S
tail -1 $O/bench.json | cut -c1-200
