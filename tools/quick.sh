#!/bin/bash
# quick GPU check: parity tests (optional: pass "notest" to skip), bench line at sigma 1 and per-kernel rocprof averages
cd $GRAFT_REPO_ROOT
if [ "$1" != "notest" ]; then python -m pytest tests -x -q -m gpu 2>&1 | tail -3; fi
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-single-view --no-other-configs --no-parity-check 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4), {k[:10]:round(v['avg_ms'],4) for k,v in d['roofline']['per_kernel'].items()})"
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/st
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/st -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-view --no-other-configs --no-parity-check > /dev/null 2>&1
cut -d, -f1-4 $GRAFT_REPO_ROOT/gpurun_out/st/k_kernel_stats.csv | grep -v "at::native\|rocclr" | sed 's/(anonymous namespace):://g; s/(KParams)//g' | head -8
