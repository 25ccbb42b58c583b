"""Build (hipcc, cross-compiles without a GPU) and run tools/valu_probe.hip; write profiles/r02/valu_probe.json.
  python tools/valu_probe.py --build          (container)   -> tools/bin/valu_probe
  python tools/valu_probe.py [--reps 2000]    (GPU box)     -> profiles/r02/valu_probe.json (also copied to gpurun_out/)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "bin", "valu_probe")


def build():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-Wno-unused-value", "-o", BIN,
                           os.path.join(ROOT, "tools", "valu_probe.hip")])
    return BIN


if __name__ == "__main__":
    if "--build" in sys.argv or not os.path.exists(BIN):
        build()
    if "--build" in sys.argv:
        sys.exit(0)
    reps = sys.argv[sys.argv.index("--reps") + 1] if "--reps" in sys.argv else "2000"
    txt = subprocess.check_output([BIN, reps]).decode()
    res = json.loads(txt)
    for d in (os.path.join(ROOT, "profiles", "r02"), os.path.join(ROOT, "gpurun_out")):
        os.makedirs(d, exist_ok=True)
        json.dump(res, open(os.path.join(d, "valu_probe.json"), "w"), indent=1)
    for k, v in res["classes"].items():
        print("%-18s w1 %.2f  w2 %.2f  w4 %.2f  w8 %.2f  issue cycles / wave-instruction / SIMD" % (
            k, v["w1"]["issue_cycles_per_inst"], v["w2"]["issue_cycles_per_inst"], v["w4"]["issue_cycles_per_inst"], v["w8"]["issue_cycles_per_inst"]))
