#!/bin/bash
# bash tools/pmc_lib1.sh <lib|-> COUNTER...   : counters per kernel for one library variant (bounded)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; lib=$1; shift
LIBARG=""; [ "$lib" != "-" ] && LIBARG="--lib $R/$lib"
OUT=$R/gpurun_out/pmc1/$(basename $lib .so); rm -rf $OUT; mkdir -p $OUT
timeout -k 5 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -o p -- python $R/tools/step_time.py $LIBARG --steps 2 > $OUT/log 2>&1
grep "Mpixel" $OUT/log
python - $OUT <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f'{out}/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        m = re.search(r'(raster_\w+|setup_bin_kernel|finalize_kernel|tile_scan_kernel|fill_kernel)', row['Kernel_Name'])
        if not m: continue
        k = m.group(1)
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k,row['Counter_Name'])] += 1
for k, d in agg.items():
    print(k, {c: round(v / cnt[(k,c)]) for c, v in d.items()})
PY
