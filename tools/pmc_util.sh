#!/bin/bash
# Issue / wait counters of the fit step's kernels (two --pmc passes with --kernel-trace only):  bash tools/pmc_util.sh [lib.so]   ("-" or nothing = the product)
# What the forward raster is bound by: vector instructions x 4 cycles against the busy cycles of the SIMDs, and the share of a wavefront's life it waits for an issue slot.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
lib=${1:--}
name=$([ "$lib" = "-" ] && echo product || basename $lib .so); OUT=$R/gpurun_out/pmcutil/$name; mkdir -p $OUT
LIBARG=""; [ "$lib" != "-" ] && LIBARG="--lib $R/$lib"
python $R/tools/step_time.py $LIBARG --steps 20 2>&1 | grep -v amdgpu.ids
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/a -o p -- python $R/tools/step_time.py $LIBARG --steps 2 > $OUT/log_a 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/b -o p -- python $R/tools/step_time.py $LIBARG --steps 2 > $OUT/log_b 2>&1
python - $OUT $name <<'PY'
import csv, glob, sys, collections, re
out, name = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f'{out}/*/*counter_collection.csv') + glob.glob(f'{out}/*/*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        m = re.search(r'(raster_\w+|setup_bin_kernel|finalize_kernel|tile_scan_kernel)', row['Kernel_Name'])
        if not m: continue
        k = m.group(1)
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k,row['Counter_Name'])] += 1
for k, d in agg.items():
    v = {c.replace('SQ_',''): v / cnt[(k,c)] for c, v in d.items()}
    print(name, k, {c: round(x) for c, x in v.items()})
    if 'WAVE_CYCLES' in v and v['WAVE_CYCLES'] > 0:
        print(f"   per launch: VALU {v.get('INSTS_VALU',0)/1e6:.2f} M, SALU {v.get('INSTS_SALU',0)/1e6:.2f} M, LDS {v.get('INSTS_LDS',0)/1e6:.2f} M, lanes active per VALU instruction {v.get('THREAD_CYCLES_VALU',0)/max(v.get('INSTS_VALU',1),1):.1f} of 64;"
              f" waiting for anything {v.get('WAIT_ANY',0)/v['WAVE_CYCLES']:.2f} of a wavefront's life, for an issue slot {v.get('WAIT_INST_ANY',0)/v['WAVE_CYCLES']:.2f}; vector units busy {v.get('ACTIVE_INST_VALU',0)*4/1024/max(v.get('BUSY_CYCLES',1)/32,1):.2f} of the kernel (ACTIVE_INST_VALU quad-cycles over 1 024 SIMDs against BUSY_CYCLES over 32 shader engines)")
PY
