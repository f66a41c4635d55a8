#!/bin/bash
# development: memory-side PMC probe of one scene:  bash tools/pmc_mem_scene.sh <outdir> <scene> <size> [steps]
OUT=gpurun_out/${1:-pmcm}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 tools/time_scene.py $@"
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM --output-format csv -d $OUT/a -- $BENCH > $OUT/a.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/b -- $BENCH > $OUT/b.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c -- $BENCH > $OUT/c.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/d -- $BENCH > $OUT/d.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('$OUT/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'render_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg): print("%-28s %.5g" % (k, sum(agg[k])/len(agg[k])))
PY
