cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -k "conv3d_kernel_geometries or phase_upsample or f16x3_precision or bf16x6_precision or full_width" 2>&1 | tail -8 > gpurun_out/r04_t5.log
rm -rf gpurun_out/vae_kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vae_kt -- python3 tools/vae_halo_ab.py f16x3 1 > gpurun_out/vae_kt_ab.log 2>&1
find gpurun_out/vae_kt -name "*.csv" ! -name "*kernel_trace.csv" -delete
tail -5 gpurun_out/r04_t5.log; grep decode gpurun_out/vae_kt_ab.log
