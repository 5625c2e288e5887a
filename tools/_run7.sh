cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "gemm" 2>&1 | tail -4
python bench.py --steps 6 --warmup 2 --no-vae --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
