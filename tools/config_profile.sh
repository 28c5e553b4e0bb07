#!/bin/bash
# Per-kernel profile of ONE of the other BASELINE configurations (tools/config_times.py --only <name>): rocprofv3 kernel stats,
# HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction), SQ + TCC atomic counters.
#   bash tools/config_profile.sh <tag> "<name filter>"      e.g.  r04b_config4 "configs[4] shape, 8"
TAG=$1; ONLY=$2
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/config_times.py --only "$ONLY" 2>&1 | grep -v amdgpu.ids > $OUT/times.txt; cat $OUT/times.txt
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python $R/tools/config_times.py --only "$ONLY" > $OUT/stats.log 2>&1
pass() { n=$1; shift; timeout -k 5 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o p -- python $R/tools/config_times.py --only "$ONLY" > $OUT/$n.log 2>&1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES
pass tcc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum
python - $OUT <<'PY'
import csv, glob, sys, collections, re, json
out = sys.argv[1]
res = collections.defaultdict(dict)
for n in ("fetch", "write", "sq1", "tcc"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(f"{out}/{n}/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(raster_\w+|setup_bin_kernel|finalize_kernel|tile_scan_kernel|fill_kernel)", row["Kernel_Name"])
            if not m: continue
            agg[m.group(1)][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(m.group(1), row["Counter_Name"])] += 1
    for k, d in agg.items():
        for c, v in d.items():
            res[k][c] = v / cnt[(k, c)]
for k, d in res.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch (2 x FETCH_SIZE + WRITE_SIZE, KB -> B)"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
json.dump(res, open(f"{out}/counters.json", "w"), indent=1)
for k, d in res.items():
    print(k, {c: round(v) for c, v in d.items()})
PY
cut -c1-150 $OUT/stats/k_kernel_stats.csv | head -12
