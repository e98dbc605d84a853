#!/usr/bin/env python3
"""Print one line per gfx950 kernel: VGPR/AGPR/SGPR, spills, scratch, LDS, occupancy (hipcc -Rpass-analysis)."""
import re, subprocess, sys, glob, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(root, "geo4d_amd/csrc/*.hip")))
for f in files:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{root}/include",
                          f"-I{root}/geo4d_amd/csrc", "-mllvm", "-amdgpu-mfma-vgpr-form", "-c", f, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (.+?) \[-Rpass", line)
        if not m: continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:90]}
        cur[k] = v
        if k.startswith("LDS Size"):
            print(f"{cur['name']:<92} V{cur.get('VGPRs','?'):>4} A{cur.get('AGPRs','?'):>3} S{cur.get('TotalSGPRs','?'):>4} "
                  f"spillV {cur.get('VGPRs Spill','?')} scratch {cur.get('ScratchSize [bytes/lane]','?')} occ {cur.get('Occupancy [waves/SIMD]','?')} lds {v}")
