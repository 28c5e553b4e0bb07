#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
( timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
V=tools/variants
bash tools/ab3.sh "" "--lib $V/libdeodr_hip_nofin.so" "--views 1" "--views 1 --lib $V/libdeodr_hip_nofin.so" "--views 4" "--views 4 --lib $V/libdeodr_hip_nofin.so" "--views 16" "--views 16 --lib $V/libdeodr_hip_nofin.so" > $O/ab.txt 2>&1; cat $O/ab.txt
