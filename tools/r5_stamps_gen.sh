#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5p
S=$PWD/hint_amd/lib/libhint_amd_stamps.so
for W in $@; do HINT_AMD_LIB=$S python tools/stamps.py $W 3 > gpurun_out/r5p/gstamps_$W.txt 2>&1; done
head -100 gpurun_out/r5p/gstamps_$1.txt
