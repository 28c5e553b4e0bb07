#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
( timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
V=tools/variants
bash tools/ab3.sh "" "--lib $V/libdeodr_hip_nodeal.so" "--lib $V/libdeodr_hip_fillall.so" "--lib $V/libdeodr_hip_fill31.so" "--lib $V/libdeodr_hip_tilediv6.so" "--views 1" "--views 1 --lib $V/libdeodr_hip_nodeal.so" "--views 16" "--views 16 --lib $V/libdeodr_hip_nodeal.so" > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench20.json 2> $O/bench20.err; python -c "
import json; d=json.load(open('$O/bench20.json')); print(d['ms_per_step'], d['warmup'], d['roofline']['step_ms_by_stamps'], {k:(round(v['avg_ms']*1e3,1),v['launches'],round(v['avg_ms_events']*1e3,1)) for k,v in d['roofline']['per_kernel'].items()}, d['single_view']['ms_eager'])"
tail -3 $O/bench20.err
