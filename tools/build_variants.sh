#!/bin/bash
# Variants of libdeodr_hip.so for the measurement tools (built here, they travel to the GPU box with the tree; *.so is
# git-ignored).  The product library is the one __graft_entry__.build() makes; nothing loads these unless a tool is given --lib.
#   fwdtrace   -DDR_FWD_TRACE   per-tile phase counters of the fused forward   (tools/fwd_trace.py)
#   wavetrace  -DDR_WAVE_TRACE  start / end of every wave                       (tools/wave_trace.py)
#   tiletrace  -DDR_TILE_TRACE  per-tile counters of the adjoint's edge kernel  (tools/tile_trace.py)
#   fwdN       -DDR_FWD_WAVES=N the staged forward compiled for N waves / SIMD  (tools/step_time.py --lib)
#   ablM       -DDR_ABLATE=M    ablation masks of the fused forward (4 no frame stores, 8 no fill stores, 128 no accumulator atomics)
cd "$(dirname "$0")/../deodr_amd/csrc" || exit 1
OUT=../../tools/variants
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics"
build() { /opt/rocm/bin/hipcc $FLAGS $2 -o $OUT/libdeodr_hip_$1.so dr_kernels.hip & }
for v in ${@:-fwdtrace wavetrace tiletrace fwd4 fwd6}; do
  case $v in
    fwdtrace) build $v -DDR_FWD_TRACE ;;
    wavetrace) build $v -DDR_WAVE_TRACE ;;
    tiletrace) build $v -DDR_TILE_TRACE ;;
    fwd*) build $v -DDR_FWD_WAVES=${v#fwd} ;;
    abl*) build $v "-DDR_ABLATE=${v#abl}" ;;
    *) build $v "$EXTRA" ;;
  esac
done
wait
ls -la $OUT
