#!/bin/bash
# development: kernel time and HBM-side traffic (bench.py's live counters) of a scene under different JIT flags
#   bash tools/traffic_by_flags.sh "--workload C4" "" "-DRTC_WAVES_PER_SIMD=5" "-DRTC_WAVES_PER_SIMD=4"
export RTC_AMD_LIB=$PWD/ray_tracer_challenge_amd/librtc_amd_dev.so
S=$1; shift
for F in "$@"; do
  RTC_AMD_JIT_FLAGS="$F" python3 bench.py $S --steps 20 --warmup 5 --cpu-seconds 0 --no-one-shot --no-verify 2>/dev/null | python3 -c "
import sys, json
l = json.loads(sys.stdin.read()); r = l['roofline']
print('%-34s %-34s kernel %.4f ms  first %.4f  VALU %.1f M  HBM-side %.0f MB  frac %s  (%s)' % ('[$F]', l['config']['workload'][:34], r['kernel_ms'], l['schedule']['first_frame_kernel_ms'], (r['valu_wave_insts_per_launch'] or 0) / 1e6, (r['traffic'] or 0) / 1e6, r['frac'], (r['pmc_source'] or '')[:4]))"
done
