#!/bin/bash
# development: tree-kernel scenes with and without triangle pre-culling
for sc in "hexagons 4096" "mesh 2048" "mesh 512" "first_textures 4096" "here_be_dragons 1000x400" "here_be_dragons 4000x1600"; do
  for w in 0 1; do RTC_AMD_TRI_PRECULL=$w python tools/time_scene.py $sc 3 2>&1 | grep -v amdgpu.ids || exit 1; done
done
