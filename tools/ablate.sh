run() { python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); k=r[\"kernels_in_step_us\"]; print(\"$1\", round(r[\"ms_per_step\"],4), [round(v,1) for v in k.values()])"; }
run new
for v in WLOAD MFMA EPI AREAD NOBAR SKIPGEMM; do HINT_AMD_LIB=$PWD/hint_amd/lib/libhint_amd_ab_$v.so run $v; done
run new
