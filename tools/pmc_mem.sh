#!/bin/bash
# memory-instruction mix of the render kernel (development tool): bash tools/pmc_mem.sh <outdir> [bench args]
OUT=gpurun_out/${1:-pmcmem}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-verify $@"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_FLAT --output-format csv -d $OUT/a -- $BENCH > $OUT/a.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/b -- $BENCH > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('$OUT/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'render_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg): print("%-32s %.5g" % (k, sum(agg[k])/len(agg[k])))
PY
