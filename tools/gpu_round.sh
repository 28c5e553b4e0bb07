#!/bin/bash
# One GPU call per milestone: parity suite, then the round profile (bench JSON, rocprofv3 stats, PMC traffic).
# bash tools/gpu_round.sh <tag> [notest]
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
O=gpurun_out/round_$TAG; mkdir -p $O
if [ "$2" != "notest" ]; then
  ( timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -30 ) > $O/pytest.log 2>&1
  tail -4 $O/pytest.log
fi
timeout 900 bash tools/profile_round.sh $TAG
