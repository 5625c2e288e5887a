cd $GRAFT_REPO_ROOT
for i in 1 2; do B=2 N=20 python tools/attn_bench.py 2>&1 | tail -1; done
B=1 N=20 python tools/attn_bench.py 2>&1 | tail -1
python tools/glue_bench.py 2>&1 | tail -6
python tools/graph_bench.py 2>&1 | tail -12
python tools/ranker_bench.py 2>&1 | tail -8
python -m pytest tests -m gpu -x -q -k "flash or attention or rmsnorm or sampler_traj or 50_step or dit_tiny or batched or siglip or full_size_attention or stack_equals" 2>&1 | tail -8
python bench.py --steps 6 --warmup 2 --no-vae --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
