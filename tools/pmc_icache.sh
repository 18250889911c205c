# instruction-mix counter passes over tools/steps.py (diagnostic; results under gpurun_out/)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES"; do
i=$((i+1))
rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_mix$i -o m -- python3 tools/steps.py > gpurun_out/pmc_mix$i.log 2>&1
done
