#!/bin/bash
# Round profile: bench JSON, rocprofv3 kernel stats, and HBM traffic from PMC counters (separate passes, counters +
# --kernel-trace only).  Run on the GPU box:  bash tools/profile_round.sh <tag>
TAG=${1:-r01}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json; echo
cd /tmp && export TMPDIR=/tmp
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-view --no-other-configs --no-parity-check > $OUT/stats.log 2>&1
timeout -k 5 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-single-view --no-other-configs --no-parity-check > $OUT/fetch.log 2>&1
timeout -k 5 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-single-view --no-other-configs --no-parity-check > $OUT/write.log 2>&1
python - <<PY
import csv, glob, json, collections
out = "$OUT"
# bench.py's kernel groups (one hipEvent interval each) <- kernels of the library
groups = {"setup_bin_kernel": ["setup_bin_kernel"], "raster_fwd_kernel": ["tile_scan_kernel", "raster_fwd_fast_kernel", "raster_fwd_kernel"],
          "fill_kernel (forward-only calls: not part of the fit step)": ["fill_kernel"],
          "raster_bwd_kernel": ["raster_bwd_fast_kernel", "raster_bwd_edge_kernel", "raster_bwd_kernel"], "finalize_kernel": ["finalize_kernel"]}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("fetch", "write"):
    for f in glob.glob(f"{out}/{c}/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            for g, members in groups.items():
                hit = [m for m in members if m + "<" in name or m + "(" in name]
                if hit:
                    acc[(g, hit[0])][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    break
traffic = {}
for (g, k), d in acc.items():
    fetch_kb = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1)
    write_kb = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
    # MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B -> doubled; WRITE_SIZE as is
    t = traffic.setdefault(g, {"bytes_per_launch": 0.0, "parts": {}, "correction": "2 x FETCH_SIZE (gfx950 128-B requests counted as 64 B) + WRITE_SIZE, x 1024"})
    part = (2 * fetch_kb + write_kb) * 1024
    t["parts"][k] = {"bytes_per_launch": part, "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb}
    t["bytes_per_launch"] += part
# where the figures are from: bench.py copies this into roofline.traffic_source (the commit: tools/gpu_round.sh leaves it in .profile_commit)
import os
commit = open("$GRAFT_REPO_ROOT/.profile_commit").read().strip() if os.path.exists("$GRAFT_REPO_ROOT/.profile_commit") else None
traffic["_source"] = {"profile": "$TAG", "commit": commit, "command": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python bench.py --steps 4 --warmup 1", "workload": "the headline workload of bench.py"}
json.dump(traffic, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(traffic))
PY
cut -c1-160 $OUT/stats/k_kernel_stats.csv | head -8
