# Runs ON the GPU box: effective clock and matrix-pipe occupancy of the VAE's convolution kernels (full-size decode; usage: vae_pmc_clock.sh [f16x3|fp32|bf16x6]).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/vae_clk
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/vae_clk -- python3 tools/vae_trace.py decode ${1:-f16x3} > gpurun_out/vae_clk.log 2>&1
find gpurun_out/vae_clk -name "*.csv" ! -name "*counter_collection.csv" -delete
tail -2 gpurun_out/vae_clk.log
