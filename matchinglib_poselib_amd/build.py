"""In-tree build of libmlpl_hip.so for gfx950:  python -m matchinglib_poselib_amd.build"""
from __future__ import annotations

import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(jobs: int = 8, verbose: bool = False) -> str:
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, f"-j{jobs}", "ARCH=gfx950"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stdout.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libmlpl_hip.so failed")
    out = os.path.join(_HERE, "lib", "libmlpl_hip.so")
    assert os.path.exists(out), out
    return out


if __name__ == "__main__":
    print(build(verbose=True))
