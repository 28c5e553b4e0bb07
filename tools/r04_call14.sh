#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
V=tools/variants
bash tools/ab3.sh "" "--lib $V/libdeodr_hip_filldeal2.so" "--lib $V/libdeodr_hip_tilediv3.so" "--lib $V/libdeodr_hip_tilediv6.so" > $O/ab.txt 2>&1; cat $O/ab.txt
python tools/config_times.py --only "configs[4] shape, 8" 2>&1 | grep -v amdgpu.ids > $O/c4.txt
python tools/config_times.py --only "configs[4] shape, 8" --lib $V/libdeodr_hip_abl1048576.so 2>&1 | grep -v amdgpu.ids >> $O/c4.txt
python tools/config_times.py --only "configs[1]" 2>&1 | grep -v amdgpu.ids >> $O/c4.txt
python tools/config_times.py --only "configs[1]" --lib $V/libdeodr_hip_abl1048576.so 2>&1 | grep -v amdgpu.ids >> $O/c4.txt
cat $O/c4.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench20.json 2> $O/bench20.err; python -c "
import json; d=json.load(open('$O/bench20.json')); print(d['ms_per_step'], d['warmup'], {k:(round(v['avg_ms']*1e3,1),v['launches']) for k,v in d['roofline']['per_kernel'].items()}, d['single_view'])"
