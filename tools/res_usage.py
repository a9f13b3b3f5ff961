"""Register / spill / LDS report of hipcc's -Rpass-analysis=kernel-resource-usage output (stdin), one line per kernel
whose demangled name contains the filter (argv[1])."""
import re, subprocess, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for blk in sys.stdin.read().split("Function Name: ")[1:]:
    name = blk.split("\n")[0].strip()
    dn = subprocess.run(["c++filt", name.split()[0]], capture_output=True, text=True).stdout.strip()
    if flt not in dn:
        continue
    d = dict(re.findall(r"(VGPRs|VGPR Spill|SGPR Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)", blk))
    print(dn[dn.find("<"):dn.find(">") + 1], "vgpr", d.get("VGPRs"), "vspill", d.get("VGPR Spill"), "sspill", d.get("SGPR Spill"), "scratch",
          d.get("ScratchSize [bytes/lane]"), "lds", d.get("LDS Size [bytes/block]"), "occ", d.get("Occupancy [waves/SIMD]"))
