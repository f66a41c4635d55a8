#!/bin/bash
# development: the library's HOST side under AddressSanitizer + UBSan (CPU only; GPU ASan is not available on this pool).
# Builds ray_tracer_challenge_amd/variants/librtc_amd_asan.so and runs the CPU test suite plus rtc_scene_validate (the
# whole flattening path: hierarchy, gates, triangle boxes, scene box) over the fuzz generator's worlds with it.
set -e
cd "$(dirname "$0")/.."
python tools/ab.py build asan="-Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -g"
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 RTC_AMD_LIB=$PWD/ray_tracer_challenge_amd/variants/librtc_amd_asan.so
python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider
python - <<'PY'
import sys
sys.path.insert(0, ".")
import ray_tracer_challenge_amd as P
from tests import test_gpu_fuzz as T
for seed in list(range(0, 120)) + list(range(4000, 4040)) + list(range(9000, 9040)):
    world, cam, depth = T._world(seed, P)
    world.validate(P.Camera(*cam))
    world.validate(None)
print("validate sweep clean")
from tests import wide_worlds as W   # the wide generator: thin scalings, far offsets, many objects, meshes -- every host-side guard of ERROR_BUDGET.md
for seed in range(0, 400):
    world, cam, depth, style = W.world(seed, P)
    world.validate(P.Camera(*cam))
    world.validate(None)
print("wide validate sweep clean")
PY
