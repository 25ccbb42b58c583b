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
        print("%-18s " % k + "  ".join("w%d %.3g/s (%.2f ns/inst/SIMD, %d SIMDs, <=%d waves each, wave %.2f ticks/inst)" % (
            w, v["w%d" % w]["wave_insts_per_s"], v["w%d" % w]["ns_per_inst_per_simd"], v["w%d" % w]["simds_used"],
            v["w%d" % w]["max_waves_on_a_simd"], v["w%d" % w]["wave_ticks_per_inst"]) for w in (1, 2, 4, 8)))
