for cfg in "mesh 512 384" "mesh 1024 1024" "mesh 2048 2048" "mesh 4096 4096" "here_be_dragons 1000 400" "here_be_dragons 2000 800" "here_be_dragons 4000 1600"; do
# the development switches (RTC_AMD_JIT_FLAGS, _BLOCK_S, ...) exist only in the development build of the library
export RTC_AMD_LIB="${RTC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/ray_tracer_challenge_amd/librtc_amd_dev.so}"
set -- $cfg
python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 6 --rounds 2 "default" "s1|RTC_AMD_BLOCK_S=1" "s2|RTC_AMD_BLOCK_S=2" "s3|RTC_AMD_BLOCK_S=3" || exit 1
done
