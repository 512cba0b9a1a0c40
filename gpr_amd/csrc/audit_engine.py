#!/usr/bin/env python3
"""Build-time audit of the engine's compiled kernels (mfma_gemm.hip -> ISA listing).

The k-loop (engine_asm.inc) owns v88..v255 (fp64 kernels) / v80..v255 (fp32) and s84..s95; the HIP code around it is capped below them with
amdgpu_num_vgpr / amdgpu_num_sgpr.  Two things would silently break that contract and are checked here:
  * any compiler-generated instruction outside the ;;#ASMSTART / ;;#ASMEND blocks that names a register of ours
    (SGPR spills are parked in lanes of VGPRs chosen without regard to the cap);
  * static LDS: the loop addresses LDS absolutely from byte 0 of the workgroup's allocation, which is only the dynamic
    segment if the compiler places nothing of its own there (it promotes private arrays to LDS unless told not to:
    -mllvm -disable-promote-alloca-to-lds; a 1536-byte promoted array once overlaid the first A image);
  * (reported, not fatal) scratch use or register spills in these kernels: slow but safe as long as the rule above holds.
usage: audit_engine.py listing.s
"""
import re
import sys

text = open(sys.argv[1]).read()
bad = []
notes = set()
in_asm = False
kernel = None
vre = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
sre = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")
for ln in text.split("\n"):
    t = ln.strip()
    if t.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if t.startswith(";;#ASMEND"):
        in_asm = False
        continue
    m = re.match(r"^(_ZN6gprhip\w+):", ln)
    if m:
        kernel = m.group(1)
    if in_asm or not t or t.startswith(";") or t.startswith(".") or kernel is None:
        continue
    code = t.split(";")[0]
    for mm in vre.finditer(code):
        hi = int(mm.group(3) or mm.group(1))
        if hi >= (88 if "f64" in kernel else 80):
            bad.append("%s: compiler code touches %s: %s" % (kernel, mm.group(0), code))
    for mm in sre.finditer(code):
        hi = int(mm.group(3) or mm.group(1))
        if 84 <= hi <= 101:
            bad.append("%s: compiler code touches %s: %s" % (kernel, mm.group(0), code))
    # (SGPR spill lanes in a VGPR below the cap, or VGPR spills to scratch, are safe -- only slow; they are reported)
    if "v_writelane" in code or "v_readlane" in code or "scratch_" in code:
        notes.add("%s: spill code present (%s ...)" % (kernel, code.split()[0]))
for m in re.finditer(r"\.amdhsa_group_segment_fixed_size\s+(\d+)", text):
    if int(m.group(1)) != 0:
        bad.append("a kernel carries %s bytes of static LDS (compiler-placed): the loop's LDS map starts at byte 0" % m.group(1))
if bad:
    print("audit_engine: FAILED")
    for b in bad[:40]:
        print("  " + b)
    sys.exit(1)
for n in sorted(notes):
    print("audit_engine: note: " + n)
print("audit_engine: ok (%d kernels)" % len(set(re.findall(r"^(_ZN6gprhip\w+):", text, re.M))))
