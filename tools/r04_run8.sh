cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -k "conv3d or f16 or vae" 2>&1 | tail -8 > gpurun_out/r04_t8.log
python3 tools/vae_halo_ab.py f16x3 2 > gpurun_out/r04_vae_halo_ab_f16x3_b.log 2>&1
tail -5 gpurun_out/r04_t8.log; tail -2 gpurun_out/r04_vae_halo_ab_f16x3_b.log
