#!/bin/bash
# A/B of library variants on one box: bash tools/ab.sh [views] lib1 lib2 ...   ("-" = the product); two rounds, interleaved
cd $GRAFT_REPO_ROOT
V=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  LIBARG=""; [ "$lib" != "-" ] && LIBARG="--lib $lib"
  python tools/step_time.py $LIBARG --views $V 2>&1 | grep -v amdgpu.ids
done; done
