#!/bin/bash
# Fit-iteration times + kernel launches per step of the device fitters (rocprofv3 --kernel-trace counts ALL kernels of a step, torch's
# included).  bash tools/fit_round.sh <tag>   -> gpurun_out/fit_<tag>/
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/fit_$TAG; mkdir -p $O
python tools/fit_times.py 2>&1 | grep -v amdgpu.ids | tee $O/fit_times.txt
cd /tmp && export TMPDIR=/tmp
for w in depth rgb multi8; do
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -o k -- python $GRAFT_REPO_ROOT/tools/fit_times.py $w > $O/trace_$w.log 2>&1
  python - <<PY
import csv
rows = list(csv.DictReader(open("$O/trace_$w/k_kernel_stats.csv")))
calls = sum(int(r["Calls"]) for r in rows)
steps = 5 + 30 + 30 + 8  # warm-up + step_device + step + profiled steps of tools/fit_times.py
ours = sum(int(r["Calls"]) for r in rows if "anonymous namespace" in r["Name"])
print("$w: %d kernel launches in %d steps = %.1f per step (%.1f of them the library's)" % (calls, steps, calls / steps, ours / steps))
for r in rows[:6]:
    print("   %6s x %8.1f us  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:90]))
PY
done 2>&1 | tee $O/launches.txt
