#!/bin/bash
# A/B timing of library builds on the GPU box:  tools/ab_legs.sh "<lib suffixes>" "<workloads>" [steps]
#   every hint_amd/lib/libhint_amd_<suffix>.so (suffix "" = the shipped one) x every workload -> gpurun_out/ab_legs.txt
LIBS=${1:-"v0 v1"}; WLS=${2:-"power_hint_8"}; N=${3:-30}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for W in $WLS; do for L in $LIBS; do
  F=hint_amd/lib/libhint_amd_$L.so; [ "$L" = "main" ] && F=hint_amd/lib/libhint_amd.so
  HINT_AMD_LIB=$PWD/$F python tools/time_legs.py $W $N 2>&1 | grep -v "amdgpu.ids" | tail -1 | sed "s|$PWD/hint_amd/lib/||" >> gpurun_out/ab_legs.txt
done; done
cat gpurun_out/ab_legs.txt
