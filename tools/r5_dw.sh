#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5p
S=$PWD/hint_amd/lib/libhint_amd_stamps.so
for W in $@; do DW_WAVES=2 HINT_AMD_LIB=$S python tools/stamps_dw.py $W 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5p/dw_$W.txt; done
