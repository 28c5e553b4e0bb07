#!/bin/bash
# Variants of libdeodr_hip.so for the measurement tools (built here, they travel to the GPU box with the tree; *.so is
# git-ignored).  The product library is the one __graft_entry__.build() makes; nothing loads these unless a tool is given --lib.
#
# The per-tile phase counters, the ablation masks and the matrix-core owner reduction are NOT in the product sources: they are
# tools/variants/instrumentation.patch (a diff of deodr_amd/csrc against its instrumented twin), applied to a scratch copy of the
# sources for the variants that need it.  After a change of the product sources that touches the patched places, re-create the
# patch from a fixed-up copy (`diff -u a b` of the two directories) -- `patch` says which hunks no longer fit.
#   fwdtrace   -DDR_FWD_TRACE   per-tile phase counters of the fused forward   (tools/fwd_trace.py)            [patched sources]
#   tiletrace  -DDR_TILE_TRACE  per-tile counters of the adjoint's edge kernel  (tools/tile_trace.py)           [patched sources]
#   ablM       -DDR_ABLATE=M    ablation masks: 4 no frame stores of non-empty tiles, 128 no accumulator atomics of the owner adjoint,
#                               256 no owner adjoint in the fused forward, 512 no span arithmetic, 1024 no vertex-gradient atomics,
#                               2048 no binning, 4096 no record stores, 16384 spans twice, 1048576 no texture-gradient atomics,
#                               2097152 texture gradient dropped (no LDS adds, no scatter), 4194304 no texel loads / bilinear adjoint in the owner
#                               adjoint, 8388608 no LDS window for the texture gradient, 16777216 one tap in four added to the window,
#                               33554432 only the even lanes add to it  [patched]
#   mfma       -DDR_OWNER_MFMA=1 the owner reduction on the matrix cores (measured 14 us slower, round 2)        [patched sources]
#   fininfwd   -DDR_FIN_IN_FWD=1 finalize UNDER the forward raster (round 4: parity-green, not faster): the per-primitive adjoint algebra as
#                               workgroups of raster_fwd_fast_kernel gated block by block by device-side counters -- tools/variants/finalize_in_forward.patch
#                               (700 lines: set-up files primitives under screen blocks, the scan kernel builds work items, the walkers signal)  [its own patch]
#                               (an archived experiment: the patch fits the sources of commit 8628d5e, end of round 4 -- `git worktree add /tmp/r4 8628d5e` and
#                               run that tree's build_variants.sh; round 5 rewrote the places it touches)
#   classtrace                   {start, end} of every workgroup of the forward raster by class (fill / head walker / other walker), plain stores into a device
#                               table read by tools/wave_phase_probe.py -- tools/variants/forward_class_timeline.patch (round 5)                     [its own patch]
#   wavetrace  -DDR_WAVE_TRACE  start / end of every wave                       (tools/wave_trace.py)
#   fwdN       -DDR_FWD_WAVES=N the staged forward compiled for N waves / SIMD  (tools/step_time.py --lib)
#   (tools/variants/lean_many_walkers.patch: a round-5 walker experiment as a `git diff` against the sources of commit 824e91e -- apply from the root of a
#    checkout of that commit with `patch -p1`; measured and not adopted, profiles/r05y_ab_lean_and_many_edge_walkers.txt.  Its sibling, the tile body
#    compiled twice, is in the product since commit 3de053c: -DDR_ONE_BATCH_BODY=0 builds the neighbour)
#   <name>     EXTRA="-D..."    anything else: the product sources with the flags of $EXTRA
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$ROOT/tools/variants
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics"
PATCHED=""
patched_sources() { # a scratch copy of the sources with the instrumentation applied (once per invocation)
  if [ -z "$PATCHED" ]; then
    PATCHED=$(mktemp -d /tmp/deodr_instr.XXXXXX)
    mkdir -p $PATCHED/deodr_amd/csrc $PATCHED/include
    cp $ROOT/deodr_amd/csrc/*.h $ROOT/deodr_amd/csrc/*.hip $PATCHED/deodr_amd/csrc/
    cp $ROOT/include/*.h $PATCHED/include/
    (cd $PATCHED/deodr_amd/csrc && patch -p1 --no-backup-if-mismatch < $OUT/instrumentation.patch) || { echo "instrumentation.patch no longer fits the sources"; exit 1; }
  fi
}
build() { (cd $1 && /opt/rocm/bin/hipcc $FLAGS $3 -o $OUT/libdeodr_hip_$2.so dr_kernels.hip) & }
for v in ${@:-fwdtrace wavetrace tiletrace fwd4 fwd6}; do
  case $v in
    fwdtrace) patched_sources; build $PATCHED/deodr_amd/csrc $v -DDR_FWD_TRACE ;;
    tiletrace) patched_sources; build $PATCHED/deodr_amd/csrc $v -DDR_TILE_TRACE ;;
    abl*) patched_sources; build $PATCHED/deodr_amd/csrc $v "-DDR_ABLATE=${v#abl}" ;;
    mfma) patched_sources; build $PATCHED/deodr_amd/csrc $v -DDR_OWNER_MFMA=1 ;;
    fininfwd)
      FIF=$(mktemp -d /tmp/deodr_fif.XXXXXX); mkdir -p $FIF/deodr_amd/csrc $FIF/include
      cp $ROOT/deodr_amd/csrc/*.h $ROOT/deodr_amd/csrc/*.hip $FIF/deodr_amd/csrc/; cp $ROOT/include/*.h $FIF/include/
      (cd $FIF/deodr_amd/csrc && patch -p1 --no-backup-if-mismatch < $OUT/finalize_in_forward.patch) || { echo "finalize_in_forward.patch no longer fits the sources"; exit 1; }
      build $FIF/deodr_amd/csrc $v "-DDR_FIN_IN_FWD=1 -DDR_SPLIT_EDGES=0" ;; # (its block counters count one walker per tile: no split tiles)
    classtrace)
      CT=$(mktemp -d /tmp/deodr_ct.XXXXXX); mkdir -p $CT/deodr_amd/csrc $CT/include
      cp $ROOT/deodr_amd/csrc/*.h $ROOT/deodr_amd/csrc/*.hip $CT/deodr_amd/csrc/; cp $ROOT/include/*.h $CT/include/
      (cd $CT/deodr_amd/csrc && patch -p1 --no-backup-if-mismatch < $OUT/forward_class_timeline.patch) || { echo "forward_class_timeline.patch no longer fits the sources"; exit 1; }
      build $CT/deodr_amd/csrc $v "" ;;
    wavetrace) build $ROOT/deodr_amd/csrc $v -DDR_WAVE_TRACE ;;
    fwd*) build $ROOT/deodr_amd/csrc $v -DDR_FWD_WAVES=${v#fwd} ;;
    *) build $ROOT/deodr_amd/csrc $v "$EXTRA" ;;
  esac
done
wait
ls -la $OUT
