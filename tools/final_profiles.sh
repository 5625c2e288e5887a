# Runs ON the GPU box: everything behind the round's committed profiles (kernel trace + PMC passes of the bench step, the VAE
# passes in the default f16x3 mode, the default bench line, the end-to-end clip times, the default-workload tuning sweep).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh
bash tools/vae_profiles.sh f16x3 both
( time python3 bench.py ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python3 tools/end_to_end.py > gpurun_out/end_to_end.log 2>&1
python3 tools/default_shape_tune.py > gpurun_out/default_shape_tune.log 2>&1
tail -3 gpurun_out/bench_default.err; grep frames gpurun_out/end_to_end.log
