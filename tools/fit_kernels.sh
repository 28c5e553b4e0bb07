#!/bin/bash
# Kernel by kernel: what ONE REPLAYED ITERATION of the device fitters consists of (rocprofv3 --kernel-trace of `tools/fit_times.py <fitter>
# --graph-only`: 5 eager iterations, then 50 graph replays).  bash tools/fit_kernels.sh [tag]  -> gpurun_out/fitk_<tag>/
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/fitk_$TAG; mkdir -p $O
python tools/fit_times.py 2>&1 | grep -v amdgpu.ids | tee $O/fit_times.txt
cd /tmp && export TMPDIR=/tmp
for w in rgb multi8 depth; do
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g_$w -o k -- python $GRAFT_REPO_ROOT/tools/fit_times.py $w --graph-only > $O/g_$w.log 2>&1
  python - <<PY
import csv, re
rows = sorted(csv.DictReader(open("$O/g_$w/k_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
first = [i for i, nm in enumerate(names) if "fit_pose_project_kernel" in nm]
a, b = first[-2], first[-1]  # the last complete replay
t0 = int(rows[a]["Start_Timestamp"])
short = lambda nm: re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", nm)[:100]
print("$w: %d kernels in one replayed iteration, %.1f us from the first kernel's start to the next iteration's" % (b - a, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
end = None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("  %8.1f  %7.1f us  (gap %5.1f)  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0 if end is None else (s - end) / 1e3, short(r["Kernel_Name"])))
    end = e
PY
done 2>&1 | tee $O/kernels.txt
