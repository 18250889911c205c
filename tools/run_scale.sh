#!/bin/bash
# The multi-GPU sweep of the bench contract, for a node with 8 MI355X (not runnable from the 1-GPU gpurun boxes):
#   tools/run_scale.sh [weak|strong] [steps] [warmup]   ->  gpurun_out/scale_<mode>.jsonl, one JSON line per N
#   tools/run_scale.sh --gpus N                         ->  dry path: only checks that N devices are visible (exit 0 / 1)
# weak: 4096 rows per GPU (the reference's batch per device); strong: 4096 rows in total, split over the ranks
# (below 4096 rows per GPU the grid no longer fills the chip: DESIGN.md section 7).
# Fails fast - non-zero exit, nothing launched, no re-exec - when fewer devices are visible than a run needs
# (torch.cuda.device_count() does not initialise the GPU).
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
visible() { python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0; }
need() { local have; have=$(visible); if [ "$have" -lt "$1" ]; then echo "run_scale.sh: $1 GPUs needed, $have visible" >&2; exit 1; fi; }
if [ "$1" = "--gpus" ]; then need "${2:-8}"; echo "run_scale.sh: ${2:-8} GPUs visible"; exit 0; fi
MODE=${1:-weak}; STEPS=${2:-200}; WARMUP=${3:-20}
need 8
mkdir -p gpurun_out
OUT=gpurun_out/scale_$MODE.jsonl; rm -f $OUT
for N in 1 2 4 8; do
  if [ $N -eq 1 ]; then
    python bench.py --gpus 1 --steps $STEPS --warmup $WARMUP --scaling $MODE --no-cpu-baseline --no-other-workloads --no-module-path --no-live-traffic | tail -1 >> $OUT
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py --gpus $N --steps $STEPS --warmup $WARMUP --scaling $MODE --no-cpu-baseline | tail -1 >> $OUT
  fi
done
python - <<PY
import json
rows = [json.loads(l) for l in open("$OUT") if l.startswith("{")]
base = rows[0]["value"]
for r in rows:
    n = r["n_gpus"]
    print(f"N={n}: {r['value']:.0f} samples/s, {r['ms_per_step']:.3f} ms/step, x{r['value'] / base:.2f} of N=1, {r['config'].get('gradient_allreduce')}")
PY
