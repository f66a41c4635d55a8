#!/bin/bash
# round 3: the whole evidence set in one GPU call (profiles + timelines + config table + bench line); summaries are made on the box
TAG=${1:-r03}
mkdir -p gpurun_out/$TAG
bash profiles/run_all.sh $TAG > gpurun_out/$TAG/run_all.log 2>&1
for s in c3 c4 c5 hexagons mesh dragons reflect_refract first_textures; do python profiles/summarize.py ${TAG}_$s > gpurun_out/$TAG/summary_$s.json 2> gpurun_out/$TAG/summary_$s.err || echo "summarize $s failed"; done
cp profiles/${TAG}_* gpurun_out/$TAG/   # (summarize.py writes beside itself; only gpurun_out/ travels back)
if [ "$2" = "profiles-only" ]; then exit 0; fi
python tools/wave_timeline.py --scene reflect_refract --size 4096 --height 2048 2>&1 | grep -v amdgpu > gpurun_out/$TAG/timeline_reflect_refract.txt
python tools/wave_timeline.py --scene first_textures --size 4096 --height 2048 2>&1 | grep -v amdgpu > gpurun_out/$TAG/timeline_first_textures.txt
python tools/wave_timeline.py --scene mesh --size 2048 2>&1 | grep -v amdgpu > gpurun_out/$TAG/timeline_mesh.txt
python tools/wave_timeline.py --scene soft_shadows --size 4096 2>&1 | grep -v amdgpu > gpurun_out/$TAG/timeline_c3.txt
python tools/wave_timeline.py --scene glass_and_mirror --size 4096 2>&1 | grep -v amdgpu > gpurun_out/$TAG/timeline_c4.txt
python tools/time_configs.py > gpurun_out/$TAG/time_configs.txt 2>&1
python tools/verify_configs.py > gpurun_out/$TAG/verify_configs.txt 2>&1
python bench.py > gpurun_out/$TAG/bench_line.json 2> gpurun_out/$TAG/bench.err
tail -c 400 gpurun_out/$TAG/bench_line.json
