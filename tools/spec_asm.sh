#!/bin/bash
# the development switches (RTC_AMD_JIT_FLAGS, _BLOCK_S, ...) exist only in the development build of the library
export RTC_AMD_LIB="${RTC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/ray_tracer_challenge_amd/librtc_amd_dev.so}"
# development: compiles a scene-specialised kernel offline (same flags as rtc_device.hip jit_get) and prints its resource
# usage; the ISA is left in /tmp/k/<name>.s.   bash tools/spec_asm.sh <name> -DRTC_SPEC_LIST=0x500,0x501 -DRTC_SPEC_NOBJ=2 ...
NAME=$1; shift
mkdir -p /tmp/k && echo '#include "rtc_kernel_core.h"' > /tmp/k/spec.hip
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${RTC_CORE_DIR:-$ROOT/ray_tracer_challenge_amd/csrc}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I$SRC -I$ROOT/include \
  --cuda-device-only -S -o /tmp/k/$NAME.s /tmp/k/spec.hip "$@" -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "remark|error" | sed 's/.*remark: //' | head -20
grep -c "scratch_" /tmp/k/$NAME.s | sed 's/^/scratch instructions: /'
