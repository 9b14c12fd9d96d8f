"""Builds tests/fake_rccl/libfake_rccl.so (see fake_rccl.cpp: test infrastructure, a stand-in for RCCL between processes that share
one GPU).  g++ against the HIP runtime; a second or two."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libfake_rccl.so")
SRC = os.path.join(HERE, "fake_rccl.cpp")


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", SRC, "-o", LIB,
                               "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-Wl,-rpath,/opt/rocm/lib"])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
