"""Hash of the sources a kernel profile depends on: the kernels, the device-tree builders that decide what they
traverse, the launch defaults (rt_abi.cpp) and the build flags.  tools/pmc_summary.py stores it in the profile; bench.py compares it."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["rt_amd/csrc/kernels.hip", "rt_amd/csrc/kernels.h", "rt_amd/csrc/rt_types.h", "rt_amd/csrc/rt_shared_math.h",
         "rt_amd/csrc/wide_bvh.cpp", "rt_amd/csrc/sah_build.cpp", "rt_amd/csrc/rt_abi.cpp", "rt_amd/csrc/Makefile"]


def kernel_source_hash():
    h = hashlib.sha256()
    for f in FILES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read() + b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())
