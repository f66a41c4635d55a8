# the here_be_dragons demo binary on a mesh the size of a real scan (6 x 102 k triangles): ring hierarchy on / off
python -c "from ray_tracer_challenge_amd import scenes; print(scenes.dragon_stand_in_obj(320, 160), end='')" > /tmp/big.obj
for v in 1 0; do
  for size in 1000x400 4000x1600; do
    echo "RTC_AMD_CLUSTERS=$v $size"
    RTC_AMD_CLUSTERS=$v RTC_AMD_CLUSTER_STATS=1 ./demos/here_be_dragons /tmp/big.obj $size 2>&1 >/tmp/out_${v}_${size}.ppm | grep -v amdgpu.ids
    md5sum /tmp/out_${v}_${size}.ppm
  done
done
