#!/usr/bin/env python3
"""Per-kernel resources of the shipping build from its device assembly: VGPRs, SGPRs, static LDS, scratch, instructions, FP64 MFMA
instructions, and the occupancy the registers allow (waves per SIMD = 512 / VGPRs, capped at 8).
usage: tools/kernel_resources.py [out.md]      (compiles r-vio_amd/csrc/rvio_hip.hip with -save-temps into a temporary directory)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "r-vio_amd", "csrc")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(.*", "", re.sub(r"^void ", "", o)) for o in out]


def main():
    with tempfile.TemporaryDirectory() as t:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-Wno-pass-failed",
                               os.path.join(CSRC, "rvio_hip.hip"), "-o", os.path.join(t, "lib.so"), "-save-temps=obj"], cwd=CSRC, stderr=subprocess.DEVNULL)
        s = open(os.path.join(t, "rvio_hip-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    rows = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        name, blk = m.group(1), m.group(2)
        g = lambda k: int(re.search(k + r"\s+(\d+)", blk).group(1))   # noqa: E731
        a = s.index("\n" + name + ":")
        body = s[a:s.index(".Lfunc_end", a)]
        ins = [ln.strip() for ln in body.split("\n") if ln.startswith("\t") and not ln.strip().startswith((".", ";"))]
        vg = g("next_free_vgpr")
        acc = re.search(r"\.amdhsa_accum_offset\s+(\d+)", blk)
        rows.append((name, vg, g("next_free_sgpr"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), len(ins),
                     sum(1 for i in ins if i.startswith("v_mfma_f64")), min(8, 512 // max(vg, 1)), int(acc.group(1)) if acc else 0))
    names = demangle([r[0] for r in rows])
    rows = sorted(zip(names, rows), key=lambda x: x[0])
    lines = ["# r05 — per-kernel resources of the shipping build (device assembly of `hipcc --offload-arch=gfx950 -O3`, `tools/kernel_resources.py`)", "",
             "VGPRs = `next_free_vgpr` (arch + accumulation registers of the unified file); waves / SIMD = min(8, 512 / VGPRs); LDS = the static part "
             "(kernels with dynamic LDS get the rest at launch); scratch = bytes per lane.", "",
             "| kernel | VGPRs | SGPRs | static LDS (B) | scratch (B) | instructions | `v_mfma_f64` | waves / SIMD by registers |", "|---|---|---|---|---|---|---|---|"]
    for n, r in rows:
        lines.append("| `%s` | %d | %d | %d | %d | %d | %d | %d |" % (n, r[1], r[2], r[3], r[4], r[5], r[6], r[7]))
    tot_mfma = sum(r[6] for _, r in rows)
    lines += ["", "%d kernels, %d `v_mfma_f64_16x16x4` instructions, %d kernels with scratch." % (len(rows), tot_mfma, sum(1 for _, r in rows if r[4] > 0))]
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(out)
    else:
        sys.stdout.write(out)


if __name__ == "__main__":
    main()
