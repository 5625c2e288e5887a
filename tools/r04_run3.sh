cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -k "f16x3 or default_workload_vae or graph_runner" 2>&1 | tail -8 > gpurun_out/r04_t3.log
python3 tools/default_shape_tune.py > gpurun_out/r04_default_tune.log 2>&1
rm -rf gpurun_out/vae_kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vae_kt -- python3 tools/vae_trace.py decode f16x3 > gpurun_out/vae_kt_f16x3.log 2>&1
find gpurun_out/vae_kt -name "*.csv" ! -name "*kernel_trace.csv" -delete
tail -5 gpurun_out/r04_t3.log; tail -30 gpurun_out/r04_default_tune.log; tail -2 gpurun_out/vae_kt_f16x3.log
