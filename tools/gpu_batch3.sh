#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/b3; mkdir -p $O
( timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -30 ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
for v in "" fwd4 abl128 abl4 abl8 abl140; do
  if [ -z "$v" ]; then timeout 300 python tools/step_time.py; else timeout 300 python tools/step_time.py --lib tools/variants/libdeodr_hip_$v.so; fi
done 2>&1 | grep -v amdgpu.ids | tee $O/variants.log
