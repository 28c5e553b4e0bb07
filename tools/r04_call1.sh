#!/bin/bash
# round 4, GPU call 1: baseline of HEAD (suite, bench line with the new hygiene fields), hardware probes, wave timeline
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
( timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 ) > $O/pytest.log 2>&1
tail -2 $O/pytest.log
timeout 120 tools/probes/order_atomic_probe > $O/probe.txt 2>&1; cat $O/probe.txt
( python tools/step_time.py; python tools/step_time.py --views 1; python tools/step_time.py; python tools/step_time.py --views 4 ) 2>&1 | grep -v amdgpu.ids > $O/step.txt; cat $O/step.txt
timeout 300 python tools/wave_trace.py --lib tools/variants/libdeodr_hip_wavetrace.so > $O/wave.txt 2>&1; tail -40 $O/wave.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
