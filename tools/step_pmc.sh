# Runs ON the GPU box: SQ / GRBM counters of every kernel of a 2-block denoise step (effective clock and MFMA-pipe occupancy per kernel).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_step
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_step -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-vae --no-pipeline-path --no-ranker --layers 4 > gpurun_out/pmc_step.log 2>&1
find gpurun_out/pmc_step -name "*.csv" ! -name "*counter_collection.csv" ! -name "*kernel_trace.csv" -delete
du -sh gpurun_out/pmc_step
