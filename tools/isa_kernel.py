"""ISA of the library's device code and of one kernel of it (no GPU needed):
    python tools/isa_kernel.py build [out.s] [-DFLAG ...]      hipcc -S -gline-tables-only of dr_kernels.hip -> /tmp/isa/dr_kernels.s (~4 minutes)
    python tools/isa_kernel.py table [in.s]                    registers / spills / scratch of every kernel (the .amdhsa metadata)
    python tools/isa_kernel.py kernel <substring of the mangled name> [in.s] [out.s]
                                                               that kernel's text -> out.s (+ the .file table, for tools/isa_lines.py) and where its
                                                               scratch (spill) instructions are: the first one's position and the source lines around them
The headline instance of the forward raster: raster_fwd_fast_kernelIfLb1ELb0ELb0ELi4ELb1ELi0E"""
import collections, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = "/tmp/isa/dr_kernels.s"
cmd = sys.argv[1] if len(sys.argv) > 1 else "table"
if cmd == "build":
    rest = sys.argv[2:]
    out = rest[0] if rest and not rest[0].startswith("-") else DEFAULT
    flags = [a for a in rest if a.startswith("-")]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "-S", "-gline-tables-only",
                    "--cuda-device-only", *flags, "-o", out, "dr_kernels.hip"], check=True, cwd=os.path.join(ROOT, "deodr_amd", "csrc"))  # fmt: skip
elif cmd == "table":
    txt = open(sys.argv[2] if len(sys.argv) > 2 else DEFAULT).read()
    ks = re.findall(r"- \.agpr_count.*?(?=\n  - \.agpr_count|\namdhsa\.target)", txt, flags=re.S)
    rows = []
    for k in ks:
        g = lambda key: (re.search(r"\." + key + r":\s*(\S+)", k) or [None, None])[1]
        rows.append((g("name"), g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("sgpr_count"), g("group_segment_fixed_size")))
    dem = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
    for (n, v, s, p, sg, lds), d in zip(rows, dem):
        d = d.replace("(anonymous namespace)::", "").replace("((anonymous namespace)::KParams)", "").replace("void ", "")
        print(d[:100].ljust(100), "vgpr", v, "spilled", s, "scratch", p, "sgpr", sg, "lds", lds)
elif cmd == "kernel":
    want, src = sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else DEFAULT)
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(src), "kernel.s")
    o, files, body = None, [], []
    for ln in open(src):
        if ln.lstrip().startswith(".file"):
            files.append(ln)
        if o is None and re.match(r"^_Z\S*" + re.escape(want) + r"\S*:", ln):
            o = True
        if o:
            body.append(ln)
            if ln.startswith(".Lfunc_end"):
                break
    open(out, "w").writelines(body + files)
    fmap = {int(m.group(1)): m.group(2) for m in (re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', f) for f in files) if m}
    cur, idx, first, where = None, 0, None, collections.Counter()
    for ln in body:
        s = ln.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (fmap.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
            continue
        idx += 1
        if s.startswith("scratch_"):
            first = first or idx
            where[cur] += 1
    print(f"{out}: {idx} instructions, {sum(where.values())} scratch instructions, the first at {first}")
    for (f, l), n in where.most_common(25):
        print(f"{n:5d}  {os.path.basename(str(f))}:{l}")
