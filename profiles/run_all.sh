#!/bin/bash
# Collects the per-scene rocprofv3 evidence of one round (on the GPU box, from the repo root):
#   bash profiles/run_all.sh r02b        -> gpurun_out/prof_r02b_<scene>/ ; summarise each with profiles/summarize.py <tag>_<scene>
TAG=${1:-r02}
bash profiles/run_profile.sh ${TAG}_c3
bash profiles/run_profile.sh ${TAG}_c4 "--scene glass_and_mirror --size 4096"
bash profiles/run_profile.sh ${TAG}_c5 "--scene sphere_grid --size 8192"
bash profiles/run_profile.sh ${TAG}_hexagons "--scene hexagons --size 4096 --height 2048"
bash profiles/run_profile.sh ${TAG}_mesh "--scene mesh --size 2048"
bash profiles/run_profile.sh ${TAG}_dragons "--scene here_be_dragons --size 1000 --height 400"
