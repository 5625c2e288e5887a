cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "gemm" 2>&1 | tail -4
timeout 900 python tools/gemm_bench.py --cfgs 0,8 --shapes 0,1,2,3,4,11,12 --rounds 5 --check 2>&1 | tail -22
