#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q; mkdir -p $O
( timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
V=tools/variants
bash tools/ab3.sh "" "--lib $V/libdeodr_hip_nofin.so" "--views 1" "--views 1 --lib $V/libdeodr_hip_nofin.so" "--views 4" "--views 4 --lib $V/libdeodr_hip_nofin.so" "--views 16" "--views 16 --lib $V/libdeodr_hip_nofin.so" > $O/ab.txt 2>&1; cat $O/ab.txt
python tools/wave_trace.py --lib $V/libdeodr_hip_wavetrace.so 2>&1 | grep -E "finalize" -A8 | grep -v amdgpu > $O/wave8.txt; cat $O/wave8.txt
python tools/wave_trace.py --lib $V/libdeodr_hip_wavetrace.so --views 1 2>&1 | grep -E "finalize" -A8 | grep -v amdgpu > $O/wave1.txt; cat $O/wave1.txt
