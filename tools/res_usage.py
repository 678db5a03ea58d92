"""Per-kernel resource table from hipcc's -Rpass-analysis=kernel-resource-usage remarks (build container, no GPU).
usage: python tools/res_usage.py remarks.txt ...   (rl4mm_amd/_lib.py: build(verbose=True) prints the remarks to stderr)"""
import re, subprocess, sys

def parse(text):
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark: +([A-Za-z /\[\]]+?): +(\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    return rows

if __name__ == "__main__":
    for f in sys.argv[1:]:
        for r in parse(open(f).read()):
            name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
            print(f"{name:60s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>3s} SGPRspill {r.get('SGPRs Spill','?'):>4s} VGPRspill {r.get('VGPRs Spill','?'):>4s} "
                  f"scratch {r.get('ScratchSize [bytes/lane]','?'):>5s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} LDS {r.get('LDS Size [bytes/block]','?'):>6s}")
