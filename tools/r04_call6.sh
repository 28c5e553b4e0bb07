#!/bin/bash
# round 4, GPU call 6: where the merged forward + finalize loses its time (probe variants; norole / nowait give WRONG gradients)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
V=tools/variants
bash tools/ab3.sh "" "--lib $V/libdeodr_hip_nofin.so" "--lib $V/libdeodr_hip_norole.so" "--lib $V/libdeodr_hip_nowait.so" "--views 1" "--views 1 --lib $V/libdeodr_hip_nofin.so" "--views 1 --lib $V/libdeodr_hip_norole.so" "--views 1 --lib $V/libdeodr_hip_nowait.so" > $O/ab.txt 2>&1; cat $O/ab.txt
