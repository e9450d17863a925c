"""Build-time audit of attention_w64.s (hand-allocated accumulator registers, guide 5.7 item 4): no compiler-emitted v_accvgpr_* or AGPR
operand outside an inline-asm block, no VGPR spill, no scratch."""
import re
import sys

bad = []
in_asm = False
for n, line in enumerate(open(sys.argv[1]), 1):
    t = line.strip()
    if t.startswith(";;#ASMSTART"):
        in_asm = True
    elif t.startswith(";;#ASMEND"):
        in_asm = False
    elif not in_asm and t and not t.startswith((";", ".")):
        if "v_accvgpr" in t or re.search(r"\ba\[?\d", t.split(";")[0]):
            bad.append(f"{n}: compiler-emitted accumulator access: {t}")
    m = re.match(r"\.(vgpr_spill_count|private_segment_fixed_size):\s+(\d+)", t)
    if m and int(m.group(2)) != 0:
        bad.append(f"{n}: {t}")
if bad:
    print("attention_w64 audit FAILED:\n  " + "\n  ".join(bad[:20]), file=sys.stderr)
    sys.exit(1)
print("attention_w64 audit ok")
