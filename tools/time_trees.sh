#!/bin/bash
# development: tree-kernel scenes, packet walk vs per-lane walk (scene-compiled kernels)
for sc in "sphere_grid 8192" "hexagons 4096" "grouped_grid 4096" "mesh 2048" "first_textures 4096"; do
  for w in packet lane; do RTC_AMD_TREE_WALK=$w python tools/time_scene.py $sc 5 2>&1 | grep -v amdgpu.ids || exit 1; done
done
