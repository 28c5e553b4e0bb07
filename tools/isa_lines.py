"""Static instruction count of one kernel's ISA by source line (hipcc -S -gline-tables-only): which statements the instructions of a region
come from.   python tools/isa_lines.py kernel.s [first_line last_line of dr_forward.h delimiting a region, e.g. fwd_pair_tiles 848 1035]"""
import collections
import re
import sys

files = {}
cur = None
insts = []  # (file, line, mnemonic)
for ln in open(sys.argv[1]):
    s = ln.strip()
    m = re.match(r"\.file\s+(\d+)\s+\"[^\"]*\"\s+\"([^\"]+)\"", s)
    if m:
        files[int(m.group(1))] = m.group(2)
        continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
        continue
    insts.append((cur[0] if cur else 0, cur[1] if cur else 0, s.split()[0]))
print(len(insts), "instructions")
if len(sys.argv) > 3:
    fwd = next((k for k, v in files.items() if v.endswith("dr_forward.h")), 1)
    lo, hi = int(sys.argv[2]), int(sys.argv[3])
    idx = [i for i, (f, l, _) in enumerate(insts) if f == fwd and lo <= l <= hi]
    a, b = idx[0], idx[-1]
    print(f"region: instructions {a} .. {b} ({b - a + 1})")
    insts = insts[a : b + 1]
by = collections.Counter()
kinds = collections.Counter()
for f, l, m in insts:
    by[(files.get(f, str(f)), l)] += 1
    k = "valu_f64" if re.search(r"_f64", m) else ("valu" if m.startswith("v_") else ("salu" if m.startswith("s_") else ("lds" if m.startswith("ds_") else ("vmem" if m.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other"))))
    kinds[k] += 1
print(dict(kinds))
byfile = collections.Counter()
for (f, l), n in by.items():
    byfile[f] += n
print(dict(byfile))
for (f, l), n in sorted(by.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{n:5d}  {f}:{l}")
