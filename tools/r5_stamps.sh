#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5p
S=$PWD/hint_amd/lib/libhint_amd_stamps.so
HINT_AMD_LIB=$S python tools/stamps_wl.py ${1:-power_hint_8} 3 > gpurun_out/r5p/stamps_${1:-power_hint_8}_${2:-x}.txt 2>&1
grep -E "total|==|grp [0-9] " gpurun_out/r5p/stamps_${1:-power_hint_8}_${2:-x}.txt | head -60
