# Runs ON the GPU box (gpurun): rocprofv3 kernel trace of the bench command and the two PMC passes for HBM-side traffic.
# Output under gpurun_out/prof_*; summarise with tools/prof_summary.py and commit the summaries under profiles/.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-stress-shape --no-default-shape --no-pipeline-path --no-ranker > gpurun_out/prof_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vae --no-stress-shape --no-default-shape --no-pipeline-path --no-ranker --layers 2 > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-vae --no-stress-shape --no-default-shape --no-pipeline-path --no-ranker --layers 2 > gpurun_out/prof_write.log 2>&1
tail -1 gpurun_out/prof_kt.log | cut -c1-200
# keep the merged output small: the kernel trace CSV of the stats run, the two counter CSVs
find gpurun_out/prof_kt -name "*.csv" ! -name "*kernel_trace.csv" ! -name "*kernel_stats.csv" -delete
find gpurun_out/prof_fetch gpurun_out/prof_write -name "*.csv" ! -name "*counter_collection.csv" -delete
du -sh gpurun_out/prof_*
