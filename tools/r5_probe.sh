#!/bin/bash
# round-5 first probe: stamps of the wave-local kernels + leg timings of every workload on today's box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5p
O=gpurun_out/r5p
S=$PWD/hint_amd/lib/libhint_amd_stamps.so
HINT_AMD_LIB=$S python tools/stamps_wl.py power_hint_8 3 > $O/stamps_power.txt 2>&1
WIDTHS=16,16,16,16 HINT_AMD_LIB=$S python tools/stamps_wl.py power_hint_8 3 > $O/stamps_power_w16.txt 2>&1
for W in power_hint_8 gas_hint_8 miniboone_hint_10 plus_hint_4; do
  python tools/time_legs.py $W 30 2>&1 | tail -1 >> $O/legs.txt
done
WIDTHS=16,16,16,16 python tools/time_legs.py power_hint_8 30 2>&1 | tail -1 >> $O/legs.txt
cat $O/legs.txt
