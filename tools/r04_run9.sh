cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/vae_kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vae_kt -- python3 tools/vae_trace.py decode f16x3 > gpurun_out/vae_kt_f16x3.log 2>&1
find gpurun_out/vae_kt -name "*.csv" ! -name "*kernel_trace.csv" -delete
grep decode gpurun_out/vae_kt_f16x3.log
python -m pytest tests -m gpu -q -k "vae" 2>&1 | tail -4
