# Runs ON the GPU box: everything behind the round's committed profiles (GPU test log, kernel trace + PMC passes of the bench step, the VAE
# passes in the default f16x3 mode, the default bench line, the end-to-end clip times, the default-workload tuning sweep).
# usage: final_profiles.sh [quick]   (quick: tests, DiT profiles and the bench line only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/pytest_gpu_full.log 2>&1; tail -14 gpurun_out/pytest_gpu_full.log > gpurun_out/pytest_gpu.log
bash tools/collect_profiles.sh
( time python3 bench.py ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
if [ "$1" != "quick" ]; then
  bash tools/vae_profiles.sh f16x3 both
  python3 tools/end_to_end.py > gpurun_out/end_to_end.log 2>&1
  python3 tools/default_shape_tune.py > gpurun_out/default_shape_tune.log 2>&1
fi
tail -3 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/bench_default.err
