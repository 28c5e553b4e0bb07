# one PMC pass with the counters given as arguments (counters + --kernel-trace only); per-kernel averages
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-single-view $BENCH_ARGS > $OUT/log 2>&1
python - <<'PY'
import csv, glob, os, collections, re
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc1'
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f'{out}/c/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        m = re.search(r'(raster_\w+|setup_bin_kernel|finalize_kernel)', row['Kernel_Name'])
        if not m: continue
        k = m.group(1)
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k, row['Counter_Name'])] += 1
for k, d in agg.items():
    print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
if not agg:
    print(open(out + '/log').read()[-1500:])
PY
