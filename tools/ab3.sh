#!/bin/bash
# A/B on ONE box: every argument is a quoted argument string for tools/step_time.py ("" = the product); two interleaved rounds
#   bash tools/ab3.sh "" "--views 1" "--lib tools/variants/libdeodr_hip_x.so"
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for a in "$@"; do
  python tools/step_time.py $a 2>&1 | grep -v amdgpu.ids
done; done
