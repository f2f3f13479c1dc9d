#!/usr/bin/env python
"""Register / scratch / occupancy table of the kernels of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_regs.py posepipeline_amd/csrc/conv_split.hip [name filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{ROOT}/include",
           f"-I{ROOT}/posepipeline_amd/csrc", "-ffp-contract=off", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
           "-x", "hip", "-c", src, "-o", "/dev/null"] + sys.argv[3:]
    t = subprocess.run(cmd, capture_output=True, text=True).stderr
    for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
        name = b.split("\n")[0].strip()
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = dn.replace("(anonymous namespace)::", "").replace("void ", "")
        if flt and flt not in dn:
            continue
        g = lambda k: re.search(k + r": (\d+)", b).group(1)
        print(f"{dn[:78]:80s} V{g('VGPRs'):>4s} A{g('AGPRs'):>4s} S{g('SGPRs'):>4s} scratch {g('ScratchSize .bytes/lane.'):>4s} occ {g('Occupancy .waves/SIMD.')}")


if __name__ == "__main__":
    main()
