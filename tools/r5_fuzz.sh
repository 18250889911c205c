#!/bin/bash
# round 5's randomised sweeps on the final build (seeds 50.. : not the ones pytest runs)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5f; mkdir -p $O
( timeout 1500 python tools/fuzz_parity.py 400 ${S0:-51} > $O/parity.txt 2>&1; tail -3 $O/parity.txt ) 
( timeout 900 python tools/fuzz_parity.py 150 ${S1:-52} deep > $O/deep.txt 2>&1; tail -3 $O/deep.txt )
( timeout 900 python tools/fuzz_parity.py 200 ${S2:-53} lean > $O/lean.txt 2>&1; tail -3 $O/lean.txt )
( timeout 600 python tools/fuzz_parity.py 40 ${S3:-54} wide > $O/wide.txt 2>&1; tail -3 $O/wide.txt )
( timeout 900 python tools/fuzz_flow.py 120 ${S4:-55} > $O/flow.txt 2>&1; tail -3 $O/flow.txt )
( timeout 900 python tools/fuzz_inverse_grad.py 80 ${S5:-56} > $O/invgrad.txt 2>&1; tail -3 $O/invgrad.txt )
( timeout 600 python tools/edge_cases.py > $O/edge.txt 2>&1; tail -3 $O/edge.txt )
( timeout 900 python tools/big_batch.py > $O/big.txt 2>&1; tail -3 $O/big.txt )
