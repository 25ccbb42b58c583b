"""Build libewn_hip.so (gfx950) in-tree with hipcc.  `python -m ewn_gym_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", "ewn_kernels.hip")]
DEPS = [os.path.join(HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(HERE, "csrc")))] + \
       [os.path.join(HERE, "..", "include", "ewn_hip.h")]
OUT = os.path.join(HERE, "lib", "libewn_hip.so")
# -ffp-contract=off: the fp64 heuristic and expectation sums must round exactly like the
# reference's Python floats (no FMA contraction); no fast-math anywhere.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc] + FLAGS + ["-o", OUT] + SRC
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
