"""Per-kernel code size / registers / scratch / call count from a `hipcc -S --offload-device-only` listing."""
import re
import sys

for path in sys.argv[1:]:
    s = open(path).read()
    print(path)
    for m in re.finditer(r"\n(_ZN4lobk\w+):\s*; @", s):
        name = m.group(1)
        if "kernel" not in name:
            continue
        tail = s[m.end():]
        end = tail.index("; codeLenInByte")
        body, meta = tail[:end], tail[end:end + 1500]
        g = lambda k: (re.search(k + r"[:=]\s*(\d+)", meta) or [None, "?"])[1]
        print("  %-52s code %6s sgpr %4s vgpr %4s scratch %4s calls %d" % (name[9:60], g("codeLenInByte "), g("NumSgprs"), g("NumVgprs"), g("ScratchSize"), body.count("s_swappc_b64")))
