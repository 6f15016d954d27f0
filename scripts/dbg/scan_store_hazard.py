"""Scan gfx950 assembly (hipcc -S --cuda-device-only) for a 128/96-bit buffer store with an SGPR soffset whose data registers are
overwritten by the very next instruction.  The ISA manual lists no wait state for that form and LLVM's hazard recognizer inserts
none (GCNHazardRecognizer::createsVALUHazard skips MUBUF stores whose soffset is a register); on MI355X the overwrite was
observed to reach memory (round 6, lstm_mfma.h v3: activations replaced by the next store's offset register)."""
import re, sys
pat = re.compile(r"\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(s\d+|\d+|0)\b")
kern = None; n = 0
lines = open(sys.argv[1]).read().split("\n")
for i, l in enumerate(lines):
    if l and not l.startswith(("\t", " ", ".", ";")) and l.endswith(":") and l.startswith("_Z"):
        kern = l[:-1]
    m = pat.match(l)
    if not m or not m.group(5).startswith("s"):
        continue
    lo, hi = int(m.group(2)), int(m.group(3))
    j = i + 1
    while j < len(lines) and (not lines[j].strip() or lines[j].strip().startswith((";", "."))):
        j += 1
    nxt = lines[j].strip()
    op = nxt.split()[0] if nxt else ""
    if not op.startswith("v_") and not op.startswith("ds_read") and not op.startswith("buffer_load") and not op.startswith("global_load"):
        continue
    # destination registers = first operand
    d = re.match(r"\S+\s+v\[(\d+):(\d+)\]|\S+\s+v(\d+)\b", nxt)
    if not d:
        continue
    dlo = int(d.group(1) or d.group(3)); dhi = int(d.group(2) or d.group(3))
    if dlo <= hi and dhi >= lo:
        n += 1
        print("%s\n  line %d: %s\n  line %d: %s" % (kern, i + 1, l.strip(), j + 1, nxt))
print("hazards:", n)
