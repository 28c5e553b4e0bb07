#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/b5; mkdir -p $O
( timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -30 ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
for v in "" fwd5 fwd6; do
  if [ -z "$v" ]; then timeout 300 python tools/step_time.py; else timeout 300 python tools/step_time.py --lib tools/variants/libdeodr_hip_$v.so; fi
done 2>&1 | grep -v amdgpu.ids | tee $O/variants.log
timeout 300 python tools/step_time.py --views 1 2>&1 | grep -v amdgpu.ids | tee -a $O/variants.log
timeout 300 python tools/config_times.py 2>&1 | grep -v amdgpu.ids | tee $O/configs.log
