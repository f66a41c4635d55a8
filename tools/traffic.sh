#!/bin/bash
# quick HBM-side traffic probe of bench.py's render kernel: bash tools/traffic.sh <outdir>
OUT=gpurun_out/${1:-traffic}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-verify --no-one-shot"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- $BENCH > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- $BENCH > $OUT/w.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('$OUT/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'render_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value'])); meta=(r['VGPR_Count'], r['Scratch_Size'], r['LDS_Block_Size'])
m = {k: sum(v)/len(v) for k, v in agg.items()}
print(m, meta, "HBM-side bytes/launch = %.3f GB" % ((2*m['FETCH_SIZE'] + m['WRITE_SIZE'])*1024/1e9))
PY
