"""Build libce_metrics_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
TARGET = os.path.join(_HERE, "libce_metrics_hip.so")


def build(force: bool = False, jobs: int = 4) -> str:
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    if not os.path.exists(TARGET):
        raise RuntimeError("hipcc did not produce " + TARGET)
    return TARGET


if __name__ == "__main__":
    print(build(force=True))
