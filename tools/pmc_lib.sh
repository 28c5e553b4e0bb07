#!/bin/bash
# Instruction counters of the kernels for library variants:  bash tools/pmc_lib.sh <lib1.so> <lib2.so> ...   ("-" = the product)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  name=$(basename $lib .so); OUT=$R/gpurun_out/pmclib/$name; mkdir -p $OUT
  LIBARG=""; [ "$lib" != "-" ] && LIBARG="--lib $R/$lib"
  python $R/tools/step_time.py $LIBARG --steps 20 2>&1 | grep -v amdgpu.ids
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT -o p -- python $R/tools/step_time.py $LIBARG --steps 2 > $OUT/log 2>&1
  python - $OUT $name <<'PY'
import csv, glob, sys, collections, re
out, name = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f'{out}/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        m = re.search(r'(raster_\w+|setup_bin_kernel|finalize_kernel|tile_scan_kernel|fill_kernel)', row['Kernel_Name'])
        if not m: continue
        k = m.group(1)
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k,row['Counter_Name'])] += 1
for k, d in agg.items():
    print(name, k, {c.replace('SQ_',''): round(v / cnt[(k,c)]) for c, v in d.items()})
PY
done
