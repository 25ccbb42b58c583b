"""Build libewn_hip.so (gfx950) in-tree with hipcc.  `python -m ewn_gym_amd.build`.

The library is several translation units (csrc/*.hip) compiled in parallel to objects and linked once: the kernel
template instantiations (board size x lanes per game x opponent x RNG kind) dominate the build time."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SRC = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
DEPS = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(HERE, "..", "include", "ewn_hip.h")]
OUT = os.path.join(HERE, "lib", "libewn_hip.so")
OBJ = os.path.join(HERE, "lib", "obj")
# -ffp-contract=off: the fp64 heuristic and expectation sums must round exactly like the
# reference's Python floats (no FMA contraction); no fast-math anywhere.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]
# per translation unit.  ewn_policy.hip: no SLP vectorisation -- it turns pairs of fp32 operations into packed-fp32 instructions
# (v_pk_add_f32 ...), which measured (tools/mfma_probe.hip) do not overlap with the bf16 matrix pipe the way plain VALU work does
FILE_FLAGS = {"ewn_policy.hip": ["-fno-slp-vectorize"]}


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    extra = os.environ.get("EWN_HIPCC_FLAGS", "").split()
    jobs = []
    for src in SRC:
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + extra + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((subprocess.Popen(cmd), obj, cmd))
    objs = []
    for p, obj, cmd in jobs:
        if p.wait() != 0:
            for q, _, _ in jobs:
                if q.poll() is None:
                    q.kill()
            raise subprocess.CalledProcessError(p.returncode, cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
