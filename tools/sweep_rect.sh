# scene rectangle on / off over the scenes it applies to
# the development switches (RTC_AMD_JIT_FLAGS, _BLOCK_S, ...) exist only in the development build of the library
export RTC_AMD_LIB="${RTC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/ray_tracer_challenge_amd/librtc_amd_dev.so}"
for cfg in "first_patterns 4096 2048" "first_plane 4096 2048" "first_scene 4096 2048" "skybox 4096 2048" "reflect_refract 4096 2048" "glass_and_mirror 4096 4096" "sphere_grid 8192 8192" "single_sphere 1024 1024"; do
set -- $cfg
python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 10 --rounds 3 "default" "whole_grid|RTC_AMD_SCENE_RECT=0" | grep -v amdgpu || exit 1
done
