cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i 'WRREQ\|WRITE_SIZE\|WRITE_REQ' | head -20 > gpurun_out/r04_counters.txt
rm -rf gpurun_out/attn_wr
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/attn_wr/w -- python3 tools/diag/attn_write_probe.py > gpurun_out/attn_wr_w.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/attn_wr/f -- python3 tools/diag/attn_write_probe.py > gpurun_out/attn_wr_f.log 2>&1
find gpurun_out/attn_wr -name "*.csv" ! -name "*counter_collection.csv" -delete
python3 - <<'PY'
import csv, glob
for f in sorted(glob.glob('gpurun_out/attn_wr/*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'flash_attn' in r['Kernel_Name']:
            print(f.split('/')[2], r['Counter_Name'], r['Counter_Value'], r['Grid_Size'])
PY
head -5 gpurun_out/r04_counters.txt
