# Runs ON the GPU box: rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of a full-size VAE encode + decode.
# usage: vae_profiles.sh [fp32|bf16x6|f16x3] [both|decode|encode]
PREC=${1:-fp32}; WHAT=${2:-both}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/vae_kt gpurun_out/vae_fetch gpurun_out/vae_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vae_kt -- python3 tools/vae_trace.py $WHAT $PREC > gpurun_out/vae_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/vae_fetch -- python3 tools/vae_trace.py $WHAT $PREC > gpurun_out/vae_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/vae_write -- python3 tools/vae_trace.py $WHAT $PREC > gpurun_out/vae_write.log 2>&1
cat gpurun_out/vae_kt.log | tail -3
find gpurun_out/vae_kt -name "*.csv" ! -name "*kernel_stats.csv" ! -name "*kernel_trace.csv" -delete
find gpurun_out/vae_fetch gpurun_out/vae_write -name "*.csv" ! -name "*counter_collection.csv" -delete
du -sh gpurun_out/vae_*
