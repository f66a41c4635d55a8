#!/bin/bash
# round 3, second half of the evidence set: the kernel whose policy changed after the first half, timelines, the configuration table, whole-frame verification, the bench line
TAG=${1:-r03}
mkdir -p gpurun_out/$TAG
bash profiles/run_profile.sh ${TAG}_reflect_refract "--scene reflect_refract --size 4096 --height 2048" > gpurun_out/$TAG/run_rr.log 2>&1
python profiles/summarize.py ${TAG}_reflect_refract > gpurun_out/$TAG/summary_reflect_refract.json 2> gpurun_out/$TAG/summary_reflect_refract.err
cp profiles/${TAG}_reflect_refract_* gpurun_out/$TAG/
for s in "reflect_refract 4096 2048" "first_textures 4096 2048" "mesh 2048 2048" "soft_shadows 4096 4096" "glass_and_mirror 4096 4096"; do set -- $s
  python tools/wave_timeline.py --scene $1 --size $2 --height $3 2>&1 | grep -v amdgpu > gpurun_out/$TAG/timeline_$1.txt; done
python tools/time_configs.py 2>&1 | grep -v amdgpu > gpurun_out/$TAG/time_configs.txt
python tools/verify_configs.py 2>&1 | grep -v amdgpu > gpurun_out/$TAG/verify_configs.txt
python bench.py > gpurun_out/$TAG/bench_line.json 2> gpurun_out/$TAG/bench.err
tail -c 300 gpurun_out/$TAG/bench_line.json; tail -3 gpurun_out/$TAG/verify_configs.txt
