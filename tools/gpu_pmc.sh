#!/bin/bash
# SQ counters of the bench's kernels (counters + --kernel-trace only), per-tile phase trace of the fused forward
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc; mkdir -p $O
if [ -n "$TESTS" ]; then timeout 600 python -m pytest tests -q -m gpu -x -k "$TESTS" 2>&1 | tail -3; fi
true
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
run() { timeout -k 5 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$NAME -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-single-view $BENCH_ARGS > $OUT/$NAME.log 2>&1; }
NAME=sq1; run SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
NAME=sq2; run SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
NAME=sq3; run GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64
python - <<'PY'
import csv, glob, os, collections, re
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc'
for name in ('sq1','sq2','sq3'):
    files = glob.glob(f'{out}/{name}/*counter_collection.csv')
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in files:
        for row in csv.DictReader(open(f)):
            m = re.search(r'(raster_\w+|setup_bin_kernel|finalize_kernel|tile_scan_kernel|fill_kernel)', row['Kernel_Name']); k = m.group(1) if m else ''
            if not k: continue
            agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k,row['Counter_Name'])] += 1
    for k, d in agg.items():
        print(name, k, {c: round(v / cnt[(k,c)]) for c, v in d.items()})
PY
