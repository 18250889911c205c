#!/bin/bash
# round 6's randomised sweeps on the final build (seeds 81.. : not the ones pytest runs); the deep / wide trees a second time with
# the lean-wide groups at the widest the kernels take (HINT_LEANW_MAX=28: the instances the default limit of 12 never runs)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6f; mkdir -p $O
( timeout 1500 python tools/fuzz_parity.py 400 ${S0:-81} > $O/parity.txt 2>&1; tail -3 $O/parity.txt )
( timeout 900 python tools/fuzz_parity.py 150 ${S1:-82} deep > $O/deep.txt 2>&1; tail -3 $O/deep.txt )
( HINT_LEANW_MAX=28 timeout 900 python tools/fuzz_parity.py 150 ${S1:-82} deep > $O/deep28.txt 2>&1; tail -3 $O/deep28.txt )
( timeout 900 python tools/fuzz_parity.py 200 ${S2:-83} lean > $O/lean.txt 2>&1; tail -3 $O/lean.txt )
( timeout 600 python tools/fuzz_parity.py 40 ${S3:-84} wide > $O/wide.txt 2>&1; tail -3 $O/wide.txt )
( HINT_LEANW_MAX=28 timeout 600 python tools/fuzz_parity.py 40 ${S3:-84} wide > $O/wide28.txt 2>&1; tail -3 $O/wide28.txt )
( timeout 900 python tools/fuzz_flow.py 120 ${S4:-85} > $O/flow.txt 2>&1; tail -3 $O/flow.txt )
( timeout 900 python tools/fuzz_inverse_grad.py 80 ${S5:-86} > $O/invgrad.txt 2>&1; tail -3 $O/invgrad.txt )
( timeout 600 python tools/edge_cases.py > $O/edge.txt 2>&1; tail -3 $O/edge.txt )
( timeout 900 python tools/big_batch.py > $O/big.txt 2>&1; tail -3 $O/big.txt )
