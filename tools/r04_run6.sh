cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -k "conv3d or f16 or vae" 2>&1 | tail -8 > gpurun_out/r04_t6.log
python3 tools/vae_trace.py both f16x3 > gpurun_out/r04_vae_f16x3.log 2>&1
python3 tools/vae_trace.py both bf16x6 > gpurun_out/r04_vae_bf16x6.log 2>&1
tail -5 gpurun_out/r04_t6.log; grep -h 'code\|: ' gpurun_out/r04_vae_f16x3.log gpurun_out/r04_vae_bf16x6.log | grep -v amdgpu
