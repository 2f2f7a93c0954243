#!/usr/bin/env python
"""Register / scratch report of the kernels of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
   python tools/kres.py automatic-speech-recognition_amd/csrc/speller.hip [name-filter] [extra hipcc flags...]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: \s*(.*?) \[-Rpass-analysis", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void ", "")
    if flt and flt not in dem:
        continue
    print("%-58s VGPR %4s AGPR %3s spill %3s scratch %4s LDS %6s occ %s" % (dem[:58], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"),
          r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
