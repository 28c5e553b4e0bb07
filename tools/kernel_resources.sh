#!/bin/bash
# Register / scratch / occupancy of every kernel as the compiler reports them (no GPU needed).
cd "$(dirname "$0")/../deodr_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics --cuda-device-only -c -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" dr_kernels.hip 2>&1 | grep "remark:" | sed 's/ \[-Rpass.*//; s/.*remark: *//' |
  awk -F': ' '/^Function Name/ {name=$2} /^TotalSGPRs/ {sg=$2} /^VGPRs:/ {v=$2} /^ScratchSize/ {s=$2} /^Occupancy/ {o=$2} /^VGPRs Spill/ {sp=$2} /^LDS Size/ {print name, "vgpr", v, "sgpr", sg, "scratch", s, "vspill", sp, "occ", o, "lds", $2}' |
  while read n rest; do echo "$(echo $n | c++filt | sed 's/(anonymous namespace):://g; s/((anonymous namespace)::KParams)//; s/(KParams)//; s/^void //') $rest"; done
