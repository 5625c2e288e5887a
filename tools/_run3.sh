cd $GRAFT_REPO_ROOT
for nb in 30 32 0 30 32 0; do UV_ATTN12_BLOCKS=$nb B=2 N=20 python tools/attn_bench.py 2>&1 | tail -1; done
B=1 N=20 python tools/attn_bench.py 2>&1 | tail -1
UV_ATTN12_BLOCKS=30 B=1 N=20 python tools/attn_bench.py 2>&1 | tail -1
LK=512 B=2 N=20 python tools/attn_bench.py 2>&1 | tail -1
python -m pytest tests -m gpu -x -q -k "flash or attention or context_cache or checkpoint_directory or replicas or rccl or second_gpu or sampler_traj or 50_step or cfg_parallel or sequence_parallel_forward" 2>&1 | tail -8
python bench.py --steps 6 --warmup 2 2>&1 | tail -2
