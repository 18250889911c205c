#!/bin/bash
# Same-box A/B of builds of the library (clock/box variance between gpurun calls is ~2%):
#   git stash; make -C hint_amd/csrc OUT=../lib/libhint_amd_base.so; git stash pop; make -C hint_amd/csrc
#   gpurun -- 'bash tools/ab.sh [extra bench.py args]'      (AB_LIBS="base x y": variants libhint_amd_<name>.so)
run() { python bench.py --no-cpu-baseline "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); k=r[\"kernels_in_step_us\"]; print(\"$1\", round(r[\"ms_per_step\"],4), [round(v,1) for v in k.values()])"; }
for i in 1 2 3; do
for L in ${AB_LIBS:-base}; do HINT_AMD_LIB=$PWD/hint_amd/lib/libhint_amd_$L.so run $L "$@"; done
run new "$@"
done
