#!/usr/bin/env python
"""hipcc -Rpass-analysis=kernel-resource-usage of one csrc file as a table: VGPRs, AGPRs, spills, scratch, occupancy.
Usage: kernel_resources.py cobevt_amd/csrc/attention_resident.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kr.o"] + sys.argv[2:]
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: +Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE).stdout.decode().strip()
        cur = {"name": re.sub(r"cobevt::|\(anonymous namespace\)::|void ", "", name)[:70]}
        rows.append(cur)
        continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
    if "error" in line:
        print(line)
print("%-72s %5s %5s %6s %6s %7s %4s" % ("kernel", "VGPR", "AGPR", "vspill", "sspill", "scratch", "occ"))
for r in rows:
    print("%-72s %5d %5d %6d %6d %7d %4d" % (r["name"], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1),
                                             r.get("SGPRs Spill", -1), r.get("ScratchSize", -1), r.get("Occupancy", -1)))
