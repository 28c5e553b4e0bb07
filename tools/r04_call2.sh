#!/bin/bash
# round 4, GPU call 2: finalize with the LDS vertex table (A/B against round 3's library), any-order + counter probe, configs[4] profile
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
( timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 ) > $O/pytest.log 2>&1
tail -2 $O/pytest.log
timeout 200 tools/probes/anyorder_sync_probe > $O/anyorder.txt 2>&1; cat $O/anyorder.txt
bash tools/ab3.sh "" "--lib tools/variants/libdeodr_hip_base.so" "--views 1" "--views 1 --lib tools/variants/libdeodr_hip_base.so" > $O/ab.txt 2>&1; cat $O/ab.txt
python tools/config_times.py 2>&1 | grep -v amdgpu.ids > $O/configs.txt; cat $O/configs.txt
python tools/config_times.py --lib tools/variants/libdeodr_hip_base.so 2>&1 | grep -v amdgpu.ids > $O/configs_base.txt; cat $O/configs_base.txt
bash tools/config_profile.sh r04b_config4 "configs[4] shape, 8" 2>&1 | tail -30
