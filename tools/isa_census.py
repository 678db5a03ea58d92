"""Static instruction census of one kernel by SOURCE FUNCTION: `hipcc -S --offload-device-only -gline-tables-only` listing ->
instructions per function of lobsim_kernels.hip (everything is inlined into the kernel; the `.loc` line of an instruction is that of
the innermost inlined callee), split into VALU / SALU / LDS / VMEM / SMEM / branch+other.  Static counts: a straight-line phase
(prologue, op-list generation, features) executes about what is listed, the message routines are loops.
Usage: python tools/isa_census.py <listing.s> <kernel-name substring> [source]"""
import re, sys
from collections import defaultdict
from pathlib import Path
lst, sub = sys.argv[1], sys.argv[2]
src = Path(sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "lines" else Path(__file__).resolve().parent.parent / "rl4mm_amd/csrc/lobsim_kernels.hip").read_text().splitlines()
defs = []  # (line, name)
for i, s in enumerate(src, 1):
    m = re.match(r"^(?:template\s*<[^>]*>\s*)?(?:SDI|DI|static|__global__|inline)\b[^;(]*?\b(\w+)\s*\(", s)
    if m: defs.append((i, m.group(1)))
    elif re.match(r"^\s{4}(?:SDI|DI|static)\b[^;(]*?\b(\w+)\s*\(", s): defs.append((i, re.match(r"^\s{4}(?:SDI|DI|static)\b[^;(]*?\b(\w+)\s*\(", s).group(1)))
def fn_of(line):
    name = "?"
    for l, n in defs:
        if l <= line: name = n
        else: break
    return name
kind_of = lambda op: ("VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") and not re.match(r"s_(load|buffer_load|store|waitcnt|nop|branch|cbranch|barrier|endpgm|sleep|setprio|memtime|memrealtime|dcache)", op) else "SMEM" if re.match(r"s_(load|buffer_load|store|memtime|memrealtime)", op)
                     else "LDS" if op.startswith("ds_") else "VMEM" if re.match(r"(global|buffer|flat|scratch)_", op) else "other")
cnt = defaultdict(lambda: defaultdict(int)); per_line = defaultdict(int)
on, files, cur, site = False, {}, 0, 0
per_site = defaultdict(lambda: defaultdict(int))
for s in open(lst):
    if not on:
        if re.match(r"^_ZN\w+:", s) and sub in s.split(":")[0]: on = True
        continue
    if s.startswith(".Lfunc_end"): break
    if re.match(r"\s+\.loc\s", s):
        chain = [int(x) for x in re.findall(r"lobsim_kernels\.hip:(\d+)", s.split(";", 1)[1] if ";" in s else "")]
        chain = [x for x in chain if x > 0]
        if chain: cur, site = chain[0], (chain[-2] if len(chain) > 1 else chain[0])
        continue
    m = re.match(r"^\t([a-z_0-9]+)\b", s)
    if not m or s.startswith("\t."): continue
    op = m.group(1)
    f = fn_of(cur) if cur > 0 else "(other file)"
    cnt[f][kind_of(op)] += 1; per_line[cur] += 1; per_site[site][kind_of(op)] += 1
kinds = ["VALU", "SALU", "LDS", "VMEM", "SMEM", "other"]
tot = {k: sum(c[k] for c in cnt.values()) for k in kinds}
print(f"{'function':34s} " + " ".join(f"{k:>6s}" for k in kinds) + "   total")
for f, c in sorted(cnt.items(), key=lambda kv: -sum(kv[1].values())):
    print(f"{f:34s} " + " ".join(f"{c[k]:6d}" for k in kinds) + f"  {sum(c.values()):6d}")
print(f"{'TOTAL':34s} " + " ".join(f"{tot[k]:6d}" for k in kinds) + f"  {sum(tot.values()):6d}")
print("\nby call site in the kernel body (the statement of the outermost inlined frame):")
for l, c in sorted(per_site.items()):
    if sum(c.values()) >= 15: print(f"{l:6d} " + " ".join(f"{c[k]:6d}" for k in kinds) + f"  {sum(c.values()):6d}  {src[l-1].strip()[:110]}")
if "lines" in sys.argv:
    for l, n in sorted(per_line.items(), key=lambda kv: -kv[1])[:60]: print(f"{l:6d} {n:5d}  {src[l-1].strip()[:140] if l > 0 else ''}")
