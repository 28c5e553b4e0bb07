#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04n; mkdir -p $O
V=tools/variants
bash tools/ab3.sh "" "--lib $V/libdeodr_hip_tilediv5.so" "--lib $V/libdeodr_hip_tilediv6.so" "--lib $V/libdeodr_hip_tilediv8.so" "--lib $V/libdeodr_hip_td6hs2.so" "--lib $V/libdeodr_hip_td6hs8.so" > $O/ab8.txt 2>&1; cat $O/ab8.txt
for n in 1 2 4 16; do bash tools/ab3.sh "--views $n" "--views $n --lib $V/libdeodr_hip_tilediv6.so" "--views $n --lib $V/libdeodr_hip_tilediv8.so"; done > $O/abviews.txt 2>&1; cat $O/abviews.txt
python tools/config_times.py --only "configs[3]" 2>&1 | grep -v amdgpu.ids > $O/c3.txt
python tools/config_times.py --only "configs[3]" --lib $V/libdeodr_hip_tilediv6.so 2>&1 | grep -v amdgpu.ids >> $O/c3.txt
python tools/config_times.py --only "configs[3]" --lib $V/libdeodr_hip_tilediv8.so 2>&1 | grep -v amdgpu.ids >> $O/c3.txt; cat $O/c3.txt
