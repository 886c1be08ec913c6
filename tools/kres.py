#!/usr/bin/env python3
"""Compile one .hip file for gfx950 and print a per-kernel resource table (VGPR/AGPR/spills/LDS/occupancy)."""
import re, subprocess, sys
src = sys.argv[1]
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        if "error" in line: print(line)
        continue
    body = m.group(1).strip()
    if body.startswith("Function Name:"):
        cur = body.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in body:
        k, v = body.split(":", 1); rows[cur][k.strip()] = v.strip()
def dem(n):
    try: return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()[:70]
    except Exception: return n[:70]
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'spill':>6s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
for n, r in rows.items():
    print(f"{dem(n):70s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('VGPRs Spill','?'):>6s} {r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('LDS Size [bytes/block]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>4s}")
