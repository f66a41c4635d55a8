"""Builds librtc_amd.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m ray_tracer_challenge_amd.build

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the
arithmetic contract (the Rust reference never fuses a*b+c), not a tuning flag.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librtc_amd.so")
SOURCES = [os.path.join(CSRC, "rtc_device.hip"), os.path.join(CSRC, "rtc_host.cpp")]
HEADERS = [os.path.join(CSRC, "rtc_internal.h"), os.path.join(CSRC, "rtc_kernel_core.h"),
           os.path.join(ROOT, "include", "rtc.h")]

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",  # arithmetic contract, host and device
    # packed-f32 VALU (v_pk_mul/add_f32) is slower than two scalar ops on gfx950 for this kernel:
    # measured 10.25 ms -> 8.40 ms on C3 with identical output (profiles/README.md, A/B "noslp")
    "-fno-slp-vectorize",
    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc] + FLAGS + list(extra_flags) + ["-o", LIB] + SOURCES + ["-lhiprtc", "-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
