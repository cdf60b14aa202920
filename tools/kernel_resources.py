"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (VGPR/SGPR/scratch/occupancy)."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
K_SCR = r"ScratchSize \[bytes/lane\]"
K_OCC = r"Occupancy \[waves/SIMD\]"
K_LDS = r"LDS Size \[bytes/block\]"
for b in blocks:
    name = b.split("\n")[0].strip()

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"

    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn).replace("void bposd::", "")
    print("%-58s VGPR %4s AGPR %3s SGPR %4s scratch %5s occ %2s lds %s" % (
        dn, g("VGPRs"), g("AGPRs"), g("SGPRs"), g(K_SCR), g(K_OCC), g(K_LDS)))
