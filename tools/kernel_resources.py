#!/usr/bin/env python3
"""Print per-kernel register / LDS / scratch usage of a device source (hipcc remarks, gfx950)."""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main():
    for src in sys.argv[1:]:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT/'include'}",
               f"-I{ROOT/'kfunca_amd/csrc/device'}", "-c", src, "-o", "/dev/null",
               "-Rpass-analysis=kernel-resource-usage"]
        out = subprocess.run(cmd, capture_output=True, text=True).stderr
        cur = None
        rows = []
        for line in out.splitlines():
            m = re.search(r"remark: (?:Function Name: |\s*)(.*?) \[-Rpass", line)
            if not m:
                continue
            body = m.group(1).strip()
            if "Function Name" in line:
                cur = {"name": subprocess.run(["c++filt", body], capture_output=True, text=True).stdout.strip()[:70]}
                rows.append(cur)
            elif cur is not None and ":" in body:
                k, v = body.split(":", 1)
                cur[k.strip()] = v.strip()
        print(f"== {src}")
        print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
        for r in rows:
            print(f"{r['name']:70s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('SGPRs','?'):>5s} "
                  f"{r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('Occupancy [waves/SIMD]','?'):>4s} "
                  f"{r.get('LDS Size [bytes/block]','?'):>7s}")


if __name__ == "__main__":
    main()
