#!/bin/bash
# PC sampling of the bench workload (beta feature of rocprofv3; bounded):  bash tools/pcsamp.sh [method] [lib]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pcsamp; rm -rf $OUT; mkdir -p $OUT
M=${1:-stochastic}; U=cycles; I=65536
[ "$M" = host_trap ] && U=time && I=1
LIBARG=""; [ -n "$2" ] && LIBARG="--lib $R/$2"
timeout -k 5 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M --pc-sampling-unit $U --pc-sampling-interval $I --kernel-trace --output-format csv -d $OUT -o s -- python $R/tools/step_time.py $LIBARG --steps 200 > $OUT/log 2>&1
echo rc=$?
ls -la $OUT | head; tail -3 $OUT/log | cut -c1-300
for f in $OUT/*pc_sampling*.csv; do head -3 $f | cut -c1-400; wc -l $f; done
