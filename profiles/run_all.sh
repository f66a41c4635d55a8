#!/bin/bash
# Collects the per-scene rocprofv3 evidence of one round (on the GPU box, from the repo root):
#   bash profiles/run_all.sh r03        -> gpurun_out/prof_r03_<scene>/ ; summarise each with profiles/summarize.py <tag>_<scene>
TAG=${1:-r04}
bash profiles/run_profile.sh ${TAG}_c3
bash profiles/run_profile.sh ${TAG}_c4 "--workload C4"
bash profiles/run_profile.sh ${TAG}_c5 "--workload C5"
bash profiles/run_profile.sh ${TAG}_hexagons "--scene hexagons --size 4096 --height 2048"
bash profiles/run_profile.sh ${TAG}_mesh "--scene mesh --size 2048"
bash profiles/run_profile.sh ${TAG}_dragons "--scene here_be_dragons --size 1000 --height 400"
bash profiles/run_profile.sh ${TAG}_reflect_refract "--scene reflect_refract --size 4096 --height 2048"
bash profiles/run_profile.sh ${TAG}_first_textures "--scene first_textures --size 4096 --height 2048"
