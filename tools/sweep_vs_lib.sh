# every timed configuration: the library at HEAD against another build of it (RTC_AMD_LIB), same box, interleaved
OTHER=${1:-$PWD/tools/librtc_amd_r02f.so}
for cfg in "soft_shadows 1000 400" "single_sphere 1024 1024" "soft_shadows 4096 4096" "glass_and_mirror 4096 4096" "sphere_grid 8192 8192" "first_scene 4096 2048" "first_plane 4096 2048" "first_patterns 4096 2048" "reflect_refract 4096 2048" "hexagons 4096 2048" "first_textures 4096 2048" "skybox 4096 2048" "grouped_grid 4096 4096" "mesh 2048 2048" "mesh 512 384" "here_be_dragons 1000 400" "here_be_dragons 4000 1600"; do
set -- $cfg
python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 8 --rounds 3 "head" "other|RTC_AMD_LIB=$OTHER" | grep -v amdgpu || exit 1
done
