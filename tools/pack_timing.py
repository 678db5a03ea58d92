"""How long the packer takes on a full LOBSTER-sized day ("a 4.3 M-row day packs in ~5 s").
CPU only:  python tools/pack_timing.py [msgs_per_s]   (default 183: one 09:30-16:00 synthetic day = 4.3 M rows)"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from rl4mm_amd.stream import pack_arrays  # noqa: E402
from rl4mm_amd.synth import generate_lobster  # noqa: E402

rate = float(sys.argv[1]) if len(sys.argv) > 1 else 183.0
t0 = time.perf_counter()
g = generate_lobster(20180221, 10, rate)
t1 = time.perf_counter()
ps = pack_arrays(g["time_s"], g["type"], g["id"], g["size"], g["price"], g["direction"], g["books"], "SYN", "2018-02-21", 10, "S")
t2 = time.perf_counter()
print(f"generated {len(g['type'])} LOBSTER rows in {t1 - t0:.1f} s; pack_arrays -> {ps.n_msgs} messages, {ps.n_snaps} snapshots in {t2 - t1:.1f} s")
