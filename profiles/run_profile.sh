#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's dominant kernel (render_kernel).
# Usage (on the GPU box, from the repo root):  bash profiles/run_profile.sh <tag> ["extra bench.py args"] [trace]
#   e.g.  bash profiles/run_profile.sh r01e_groups "--scene hexagons" trace     (third arg: durations only)
# Pass 1: --kernel-trace --stats (durations).  Passes 2..: --pmc counters, each in its own run
# (gpurun refuses --pmc combined with the trace domains that crash nodes on this pool).
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --steps ${PROFILE_STEPS:-10} --warmup ${PROFILE_WARMUP:-3} --cpu-seconds 0 --no-verify --no-one-shot --no-live-pmc ${2:-}"
# The scene kernel is compiled FIRST, by a plain run that leaves it in the JIT disk cache, and every profiled pass loads it from
# there: a hiprtc compile inside a process started under rocprofv3 produces a different code object (LABNOTES "Round 4"), and the
# evidence must describe the binary an unprofiled bench.py runs.  (The kernel id carries the code object's checksum: summarize.py
# refuses passes that disagree, bench.py refuses a summary of another binary.)
$BENCH --steps 1 --warmup 1 > "$OUT/precompile.log" 2>&1
echo "precompile exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
echo "trace exit $?"
if [ "${3:-}" = "trace" ]; then find "$OUT" -name "*.csv" | head; exit 0; fi
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F32" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  NAME=$(echo $SET | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc_$NAME" -- $BENCH > "$OUT/pmc_$NAME.log" 2>&1
  echo "pmc [$SET] exit $?"
done
find "$OUT" -name "*.csv" | head -50
