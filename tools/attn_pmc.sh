cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export ATTN_AB_CHILD=1 N=3
for v in 0 1; do
  export UV_ATTN_W3=$v TAG=w3_$v
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/pmc_a_$v --output-format csv -- python3 tools/attn_ab.py > gpurun_out/pmc_a_$v.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d gpurun_out/pmc_b_$v --output-format csv -- python3 tools/attn_ab.py > gpurun_out/pmc_b_$v.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU -d gpurun_out/pmc_c_$v --output-format csv -- python3 tools/attn_ab.py > gpurun_out/pmc_c_$v.log 2>&1
done
ls gpurun_out/pmc_a_0 | head
