# Runs ON the GPU box: everything behind the round's committed profiles (kernel trace + PMC passes of the bench step, the VAE
# passes, the default bench line, graph / ranker / glue benches).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh
bash tools/vae_profiles.sh
( time python3 bench.py ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python3 tools/graph_bench.py > gpurun_out/graph_bench.log 2>&1
python3 tools/ranker_bench.py > gpurun_out/ranker_bench.log 2>&1
python3 tools/glue_bench.py > gpurun_out/glue_bench.log 2>&1
python3 tools/gemm_bench.py --cfgs 0,8 --shapes 0,1,2,3,4,11,12 --rounds 5 > gpurun_out/gemm_bench.log 2>&1
tail -3 gpurun_out/bench_default.err
