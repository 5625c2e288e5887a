cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do UV_GEMM_PERSIST=$v python bench.py --steps 8 --warmup 2 --no-vae --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160; done
