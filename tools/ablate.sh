#!/bin/bash
# Timing-only ablation builds (results are wrong by construction) through the bench: what each piece of
# the block kernels costs inside a real step.
#   for v in ABLATE_MFMA ABLATE_WLOAD ABLATE_EPI ABLATE_AREAD ABLATE_OUTER SKIP_COLSUM SKIP_WSCOPY SKIP_COUPLE SKIP_GEMM NO_BARRIER; do
#       make -C hint_amd/csrc OUT=../lib/libhint_amd_ab_$v.so EXTRA=-DHINT_$v; done
#   gpurun -- 'bash tools/ablate.sh'
run() { python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); k=r[\"kernels_in_step_us\"]; print(\"$1\", round(r[\"ms_per_step\"],4), [round(v,1) for v in k.values()])"; }
run shipped
for f in hint_amd/lib/libhint_amd_ab_*.so; do v=${f##*_ab_}; v=${v%.so}; HINT_AMD_LIB=$PWD/$f run $v; done
run shipped
