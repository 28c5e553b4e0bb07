#!/bin/bash
# round 4, GPU call 3: set-up with one slot request per tile and workgroup (A/B against per-pair requests), finalize at 4 waves, L2-scope atomics probe
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
( timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 ) > $O/pytest.log 2>&1
tail -2 $O/pytest.log
timeout 200 tools/probes/l2_atomic_probe > $O/l2_atomics.txt 2>&1; cat $O/l2_atomics.txt
V=tools/variants
bash tools/ab3.sh "" "--lib $V/libdeodr_hip_notable.so" "--lib $V/libdeodr_hip_fin4.so" "--lib $V/libdeodr_hip_base.so" > $O/ab8.txt 2>&1; cat $O/ab8.txt
bash tools/ab3.sh "--views 2" "--views 2 --lib $V/libdeodr_hip_base.so" "--views 4" "--views 4 --lib $V/libdeodr_hip_base.so" "--views 16" "--views 16 --lib $V/libdeodr_hip_base.so" > $O/abviews.txt 2>&1; cat $O/abviews.txt
python tools/config_times.py 2>&1 | grep -v amdgpu.ids > $O/configs.txt; cat $O/configs.txt
