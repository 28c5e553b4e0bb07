#!/bin/bash
# rocprofv3 kernel stats of tools/step_time.py (args passed through), summary printed
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof; rm -rf $OUT; mkdir -p $OUT
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $GRAFT_REPO_ROOT/tools/step_time.py "$@" > $OUT/log 2>&1
grep -v amdgpu.ids $OUT/log | tail -2
cut -d, -f1-4 $OUT/k_kernel_stats.csv | grep -v "at::native\|rocclr" | sed 's/(anonymous namespace):://g; s/(KParams)//g; s/(KParams, int)//g' | head -12
python3 - <<'PY'
import csv, os, collections
f = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof/k_kernel_trace.csv'
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last full step: find the last 'finalize' and walk back to the preceding 'setup'
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'finalize_kernel' in n]
end = idx[-2]
start = max(i for i in range(end) if 'setup_bin_kernel' in names[i])
t0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:end + 1]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:40]
    print(f"  {n:42s} start {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us")
PY
