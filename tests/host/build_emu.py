"""Build the HOST emulation of the HIP library (test infrastructure): the unchanged sources of mortal_amd/csrc compiled by
clang++ against tests/host/emu/hip/hip_runtime.h (a fiber-based SIMT emulator) -> tests/host/_build/libmortal_amd_emu.so.
Only tests load it (tests/host/emu_pool.py); mortal_amd/ never does."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EXTRA = os.environ.get("EMU_EXTRA_FLAGS", "").split()  # e.g. -DSP_OPT=7: validate an experimental kernel variant on the host
OUT = os.path.join(HERE, "_build", "libmortal_amd_emu" + ("_" + "".join(c for c in "".join(EXTRA) if c.isalnum()) if EXTRA else "") + ".so")
CXX = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")


def build(force=False):
    csrc = os.path.join(ROOT, "mortal_amd", "csrc")
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(HERE, "emu", "hip", "hip_runtime.h"),
                                                                os.path.join(ROOT, "include", "mortal_amd.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cxx = CXX if os.path.exists(CXX) else "g++"
    subprocess.check_call([cxx, "-x", "c++", "-std=c++17", "-O2", "-g1", "-fPIC", "-shared", "-ffp-contract=off", "-DMJ_EMU",
                           "-Wno-unused-value", "-Wno-unknown-attributes", "-Wno-ignored-attributes", *EXTRA,
                           "-I", os.path.join(HERE, "emu"), "-o", OUT, os.path.join(csrc, "mj_capi.hip")])
    return OUT


if __name__ == "__main__":
    print(build(force=True))
