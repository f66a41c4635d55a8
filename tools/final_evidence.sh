#!/bin/bash
# a round's final evidence set (bash tools/final_evidence.sh r04): profiles (kernel trace + PMC) of eight scenes, timelines, configuration table, whole-frame verification, bench line
TAG=${1:-r04}
mkdir -p gpurun_out/$TAG
bash profiles/run_all.sh $TAG > gpurun_out/$TAG/run_all.log 2>&1
for s in c3 c4 c5 hexagons mesh dragons reflect_refract first_textures; do python profiles/summarize.py ${TAG}_$s > gpurun_out/$TAG/summary_$s.json 2> gpurun_out/$TAG/summary_$s.err || echo "summarize $s failed"; done
cp profiles/${TAG}_* gpurun_out/$TAG/   # (summarize.py writes beside itself; only gpurun_out/ travels back)
echo profiles done
# 50 launches of the metric configuration under the kernel trace, and the line that run printed (its HIP-event kernel time beside the trace's)
PROFILE_STEPS=45 PROFILE_WARMUP=5 bash profiles/run_profile.sh ${TAG}_c3_50launches "" trace > gpurun_out/$TAG/run_50.log 2>&1
cp $(find gpurun_out/prof_${TAG}_c3_50launches/trace -name "*_kernel_stats.csv" | head -1) gpurun_out/$TAG/${TAG}_c3_50launches_kernel_stats.csv
grep '^{' gpurun_out/prof_${TAG}_c3_50launches/trace.log > gpurun_out/$TAG/${TAG}_c3_50launches_traced_bench_line.json
python - <<P
import csv, glob
f = glob.glob("gpurun_out/prof_${TAG}_c3_50launches/trace/*/*_kernel_trace.csv")[0]
d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "render_kernel" in r["Kernel_Name"] and int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) > 256]
print("50-launch trace: %d frames, mean %.4f ms, min %.4f, max %.4f" % (len(d), sum(d) / len(d) / 1e6, min(d) / 1e6, max(d) / 1e6))
P
for s in "reflect_refract 4096 2048" "first_textures 4096 2048" "mesh 2048 2048" "soft_shadows 4096 4096" "glass_and_mirror 4096 4096" "here_be_dragons 4000 1600"; do set -- $s
  python tools/wave_timeline.py --scene $1 --size $2 --height $3 2>&1 | grep -v amdgpu > gpurun_out/$TAG/timeline_$1.txt; done
echo timelines done
python tools/time_configs.py 2>&1 | grep -v amdgpu > gpurun_out/$TAG/time_configs.txt
echo table done
python bench.py > gpurun_out/$TAG/bench_line.json 2> gpurun_out/$TAG/bench.err
tail -c 300 gpurun_out/$TAG/bench_line.json
python tools/verify_configs.py 2>&1 | grep -v amdgpu > gpurun_out/$TAG/verify_configs.txt
tail -3 gpurun_out/$TAG/verify_configs.txt
