"""Build the oracle's C twin (oracle/_build/libshasta_oracle.so) with gcc.  Test infrastructure only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libshasta_oracle.so")
SRC = os.path.join(HERE, "voxelize_oracle.c")


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", LIB, SRC, "-lm"], check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
