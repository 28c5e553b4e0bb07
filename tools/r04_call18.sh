#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p; mkdir -p $O
V=tools/variants
bash tools/ab3.sh "" "--lib $V/libdeodr_hip_hs3.so" "--lib $V/libdeodr_hip_hs6.so" "--lib $V/libdeodr_hip_td7.so" > $O/ab8.txt 2>&1; cat $O/ab8.txt
