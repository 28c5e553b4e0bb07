#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/b2; mkdir -p $O
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for v in "" fwd4 fwd6; do
  if [ -z "$v" ]; then timeout 300 python tools/step_time.py; else timeout 300 python tools/step_time.py --lib tools/variants/libdeodr_hip_$v.so; fi
done 2>&1 | grep -v amdgpu.ids | tee $O/variants.log
timeout 300 python tools/step_time.py --views 1 2>&1 | grep -v amdgpu.ids | tee -a $O/variants.log
timeout 300 python tools/fwd_trace.py --lib tools/variants/libdeodr_hip_fwdtrace.so 2>&1 | grep -v amdgpu.ids > $O/fwd_trace.log; cat $O/fwd_trace.log
timeout 300 python tools/wave_trace.py --lib tools/variants/libdeodr_hip_wavetrace.so 2>&1 | grep -v amdgpu.ids > $O/wave_trace.log; grep -A9 raster_fwd $O/wave_trace.log
