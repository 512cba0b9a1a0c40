"""Per-kernel register / spill summary of a hipcc -S listing.  usage: python tools/regs.py file.s [filter]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
pat = (r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?"
       r"\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)")
for m in re.finditer(pat, s):
    if flt in m.group(1):
        print("%-70s sgpr %3s (spill %2s)  vgpr %3s (spill %2s)" % (m.group(1)[:70], m.group(2), m.group(3), m.group(4), m.group(5)))
