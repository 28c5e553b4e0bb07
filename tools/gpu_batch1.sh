#!/bin/bash
# first GPU batch of round 2: parity suite, bench, variants A/B, phase + wave traces, round profile
cd $GRAFT_REPO_ROOT
O=gpurun_out/b1; mkdir -p $O
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json; tail -3 $O/bench.err
for v in "" fwd4 fwd6; do
  if [ -z "$v" ]; then timeout 300 python tools/step_time.py; else timeout 300 python tools/step_time.py --lib tools/variants/libdeodr_hip_$v.so; fi
done 2>&1 | tee $O/variants.log
timeout 300 python tools/step_time.py --views 1 2>&1 | tee -a $O/variants.log
timeout 300 python tools/fwd_trace.py --lib tools/variants/libdeodr_hip_fwdtrace.so > $O/fwd_trace.log 2>&1; cat $O/fwd_trace.log
timeout 300 python tools/wave_trace.py --lib tools/variants/libdeodr_hip_wavetrace.so > $O/wave_trace.log 2>&1; cat $O/wave_trace.log
timeout 900 bash tools/profile_round.sh r02a 2>&1 | tail -12
