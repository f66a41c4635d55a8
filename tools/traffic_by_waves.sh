#!/bin/bash
# development: kernel time and HBM-side traffic (bench.py's live counters) of a tree scene at 6 / 5 / 4 waves per SIMD
#   bash tools/traffic_by_waves.sh "--scene hexagons --size 4096 --height 2048"
export RTC_AMD_LIB=$PWD/ray_tracer_challenge_amd/librtc_amd_dev.so
for W in 6 5 4; do
  RTC_AMD_TREE_WAVES=$W python3 bench.py $1 --steps 20 --warmup 5 --cpu-seconds 0 --no-one-shot --no-verify 2>/dev/null | python3 -c "
import sys, json
l = json.loads(sys.stdin.read()); r = l['roofline']
print('waves/SIMD $W  %-40s kernel %.4f ms  VALU %.1f M  HBM-side %.0f MB  frac %s  (%s)' % (l['config']['workload'][:40], r['kernel_ms'], (r['valu_wave_insts_per_launch'] or 0) / 1e6, (r['traffic'] or 0) / 1e6, r['frac'], (r['pmc_source'] or '')[:4]))"
done
