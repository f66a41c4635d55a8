#!/bin/bash
# development: instruction-cache / scalar-cache counters of one scene's kernel: bash tools/pmc_icache.sh <outdir> <scene> <size> [steps] [key=value ...]
export RTC_AMD_LIB="${RTC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/ray_tracer_challenge_amd/librtc_amd_dev.so}"
OUT=gpurun_out/${1:-pmc}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 -L > $OUT/counters.txt 2>&1
BENCH="python3 tools/time_scene.py $@"
$BENCH > $OUT/precompile.log 2>&1  # (compiled by a plain run first: under rocprofv3 the process holds another compiler -- LABNOTES "Round 4")
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/a -- $BENCH > $OUT/a.log 2>&1
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE --output-format csv -d $OUT/b -- $BENCH > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL --output-format csv -d $OUT/c -- $BENCH > $OUT/c.log 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY --output-format csv -d $OUT/d -- $BENCH > $OUT/d.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('$OUT/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'render_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v)/len(v) for k, v in agg.items()}
for k in sorted(m): print("%-28s %.5g" % (k, m[k]))
PY
