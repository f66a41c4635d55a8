# A/B of the ring hierarchy (RTC_AMD_CLUSTERS) on the mesh scenes
# the development switches (RTC_AMD_JIT_FLAGS, _BLOCK_S, ...) exist only in the development build of the library
export RTC_AMD_LIB="${RTC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/ray_tracer_challenge_amd/librtc_amd_dev.so}"
for cfg in "here_be_dragons 1000 400" "here_be_dragons 4000 1600"; do
set -- $cfg
python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 6 --rounds 2 "default" "no_clusters|RTC_AMD_CLUSTERS=0" \
  "s0|RTC_AMD_BLOCK_S=0" "s1|RTC_AMD_BLOCK_S=1" "s2|RTC_AMD_BLOCK_S=2" "s3|RTC_AMD_BLOCK_S=3" \
  "leaf16|RTC_AMD_CLUSTER_LEAF=16" "leaf16_s2|RTC_AMD_CLUSTER_LEAF=16|RTC_AMD_BLOCK_S=2" "leaf4_s1|RTC_AMD_CLUSTER_LEAF=4|RTC_AMD_BLOCK_S=1" "leaf4_s0|RTC_AMD_CLUSTER_LEAF=4|RTC_AMD_BLOCK_S=0" \
  "g6|RTC_AMD_CLUSTER_GMAX=0.6" "g95|RTC_AMD_CLUSTER_GMAX=0.95" || exit 1
done
