#!/bin/bash
# round 4, GPU call 5: finalize under the forward raster (A/B against the same code with finalize_kernel as a launch of its own)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
( timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ) > $O/pytest.log 2>&1
tail -8 $O/pytest.log
V=tools/variants
bash tools/ab3.sh "" "--lib $V/libdeodr_hip_nofin.so" "--views 1" "--views 1 --lib $V/libdeodr_hip_nofin.so" "--views 4" "--views 4 --lib $V/libdeodr_hip_nofin.so" > $O/ab.txt 2>&1; cat $O/ab.txt
python tools/config_times.py 2>&1 | grep -v amdgpu.ids > $O/configs.txt; cat $O/configs.txt
