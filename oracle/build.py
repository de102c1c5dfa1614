"""Builds the oracle's C restatement (test infrastructure; gcc, no GPU):
``python -m oracle.build`` -> ``oracle/_build/libfastlind.so``."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build() -> str:
    out_dir = os.path.join(_HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libfastlind.so")
    src = os.path.join(_HERE, "csrc", "fast_lindblad.c")
    subprocess.run(["gcc", "-O3", "-march=x86-64-v2", "-fopenmp", "-shared", "-fPIC", "-std=c11", src, "-o", out], check=True)
    return out


if __name__ == "__main__":
    print(build())
