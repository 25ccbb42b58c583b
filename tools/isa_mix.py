"""Static instruction mix of one kernel of libewn_hip.so, for the VALU-issue roofline (bench.py roofline.valu_issue).

  python tools/isa_mix.py [--kernel SUBSTR] [--asm FILE] [--out profiles/r02/isa_mix.json]

Compiles ewn_kernels.hip to device assembly (hipcc -S --cuda-device-only, same flags as the build; ~50 s) unless --asm gives
an existing .s, finds the kernel whose mangled name contains SUBSTR, and counts its instructions by issue class.  Inside the
hot loop (the search's per-root body: the innermost s_cbranch back-edge region with the most instructions) the counts are
reported separately: that loop is >= 2/3 of the dynamic instruction stream.  The classes are the ones tools/valu_probe.hip
measures; `weighted_issue_cycles` combines both files."""
import argparse
import collections
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# mnemonic -> probe class whose measured issue cost prices it
F64 = ("v_add_f64", "v_mul_f64", "v_fma_f64", "v_max_f64", "v_min_f64")
CMP64 = re.compile(r"v_cmpx?_\w+_[fiu]64")
SHIFT64 = ("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")
MUL32 = ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32")


def classify(op):
    if not op.startswith("v_"):
        if op.startswith("s_"):
            return "salu"
        if op.startswith("ds_"):
            return "lds"
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            return "vmem"
        return "other"
    base = op
    for suf in ("_e32", "_e64", "_dpp", "_sdwa"):
        if base.endswith(suf):
            base = base[:-len(suf)]
    if op.endswith("_dpp"):
        return "v_mov_b32_dpp"
    if base in F64:
        return "v_add_f64"
    if CMP64.match(base):
        return "v_cmp_le_f64"
    if base in SHIFT64:
        return "v_lshlrev_b64"
    if base in MUL32:
        return "v_mul_lo_u32"
    if base.startswith("v_cvt") and "f64" in base:
        return "v_add_f64"
    if base in ("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32"):
        return "v_readlane"
    if base.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos", "v_div_")):
        return "v_trans"
    if base.startswith("v_cmp"):
        return "v_cmp_lt_u32"
    if base == "v_perm_b32":
        return "v_perm_b32"
    if base == "v_ffbh_u32":
        return "v_ffbh_u32"
    if base == "v_bcnt_u32_b32":
        return "v_bcnt_u32_b32"
    if base == "v_cndmask_b32":
        return "v_cndmask_b32"
    return "v_xor_b32"   # plain 32-bit integer / logic / move / 3-operand VALU


def source_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ewn_gym_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def kernel_body(lines, substr):
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and substr in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]) and ":" in l and not l.startswith("\t"):
            start = i
            name = l.split(":")[0]
            break
    if start is None:
        raise SystemExit("kernel %r not found" % substr)
    body = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        body.append(l)
    return name, body


def mix(body):
    """(total counts, hot-loop counts): the hot loop = the backward-branch region holding the most instructions"""
    labels, insts = {}, []
    for l in body:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m:
                labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        op = s.split()[0]
        insts.append((op, s))
    total = collections.Counter(classify(op) for op, _ in insts)
    best = (0, 0, 0)
    for i, (op, s) in enumerate(insts):
        if op.startswith("s_cbranch"):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= i and i - labels[tgt] > best[0]:
                # innermost preference: keep the largest region that does not contain a larger back edge of its own
                best = (i - labels[tgt], labels[tgt], i)
    loops = []
    for i, (op, s) in enumerate(insts):
        if op.startswith("s_cbranch"):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i))
    hot = collections.Counter()
    region = None
    if loops:
        # the back-edge region with the most VALU instructions
        def valu(lo, hi):
            return sum(1 for op, _ in insts[lo:hi + 1] if op.startswith("v_"))
        region = max(loops, key=lambda r: valu(*r))
        hot = collections.Counter(classify(op) for op, _ in insts[region[0]:region[1] + 1])
    return dict(total), dict(hot), len(insts), region


def class_rate(probe, cls_name):
    """best measured chip-wide rate (wave-instructions / s) of one class over 1..8 waves per SIMD"""
    c = probe["classes"][cls_name]
    return max(c[w]["wave_insts_per_s"] for w in ("w1", "w2", "w4", "w8"))


def peak_for_mix(counts, probe):
    """chip-wide VALU wave-instructions / s this mix could issue at if nothing but the issue port limited it:
    total / sum(n_c / rate_c), every class priced at its best measured rate.  A v_cndmask reads the mask the v_cmp before it
    wrote, so both are priced with the measured cmp+cndmask pair (a lone v_cndmask on a stale VCC measures 5x slower)."""
    cls = probe["classes"]
    alias = {"v_readlane": "v_mov_b32_dpp", "v_trans": "v_mul_lo_u32", "v_cndmask_b32": "mix_cmp_cndmask"}
    t = n_all = 0.0
    for c, n in counts.items():
        if not c.startswith("v_"):
            continue
        k = alias.get(c, c if c in cls else "v_xor_b32")
        t += n / class_rate(probe, k)
        n_all += n
    return n_all / t if t else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_step_d3ILi5ELi2ELi0ELi1E")
    ap.add_argument("--asm", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--probe", default=os.path.join(ROOT, "profiles", "r02", "valu_probe.json"))
    a = ap.parse_args()
    asm = a.asm
    if asm is None:
        asm = "/tmp/ewn_dev.s"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                               "--cuda-device-only", "-o", asm, os.path.join(ROOT, "ewn_gym_amd", "csrc", "ewn_kernels.hip")],
                              stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    out = {"source_hash": source_hash(), "kernels": {}}
    for sub in a.kernel.split(","):
        name, body = kernel_body(lines, sub)
        total, hot, n, region = mix(body)
        ent = {"static_instructions": n, "total": total, "hot_loop": hot,
               "hot_loop_instructions": (region[1] - region[0] + 1) if region else 0}
        if os.path.exists(a.probe):
            probe = json.load(open(a.probe))
            ent["peak_wave_insts_per_s"] = {"total": peak_for_mix(total, probe), "hot_loop": peak_for_mix(hot, probe)}
            ent["peak_note"] = ("VALU issue peak of this instruction mix on the whole chip, from per-class rates measured by "
                                "tools/valu_probe.hip (best of 1, 2, 4, 8 waves per SIMD)")
        out["kernels"][name] = ent
    txt = json.dumps(out, indent=1)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
