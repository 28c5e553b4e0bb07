#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/b6; mkdir -p $O
( timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -30 ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 300 python tools/step_time.py 2>&1 | grep -v amdgpu.ids | tee $O/variants.log
timeout 300 python tools/step_time.py --views 1 2>&1 | grep -v amdgpu.ids | tee -a $O/variants.log
timeout 300 python tools/fwd_trace.py --lib tools/variants/libdeodr_hip_fwdtrace.so 2>&1 | grep -v amdgpu.ids > $O/fwd_trace.log; cat $O/fwd_trace.log
