#!/bin/bash
# the development switches (RTC_AMD_JIT_FLAGS, _BLOCK_S, ...) exist only in the development build of the library
export RTC_AMD_LIB="${RTC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/ray_tracer_challenge_amd/librtc_amd_dev.so}"
# quick PMC probe of one scene (development tool): bash tools/pmc_scene.sh <outdir> <scene> <size> [steps] [key=value ...]
OUT=gpurun_out/${1:-pmc}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 tools/time_scene.py $@"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -- $BENCH > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/b -- $BENCH > $OUT/b.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- $BENCH > $OUT/c.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('$OUT/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'render_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            meta = (r['VGPR_Count'], r['SGPR_Count'], r['Scratch_Size'])
m = {k: sum(v)/len(v) for k, v in agg.items()}
for k in sorted(m): print("%-28s %.5g" % (k, m[k]))
print("vgpr/sgpr/scratch", meta)
if 'GRBM_GUI_ACTIVE' in m and 'SQ_INSTS_VALU' in m:
    simd_cycles = m['GRBM_GUI_ACTIVE'] / 8 * 256 * 4
    print("VALU wave-instr per SIMD-cycle: %.3f  (cycles per VALU instr %.2f)" % (m['SQ_INSTS_VALU'] / simd_cycles, simd_cycles / m['SQ_INSTS_VALU']))
    print("wave-cycle split: active %.2f  wait_inst %.2f  wait_any %.2f" % (m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES'], m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES'], m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']))
PY
