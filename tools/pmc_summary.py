"""Mean per dispatch of every counter in a rocprofv3 --pmc output directory, for kernels whose name contains a substring."""
import csv, glob, sys
from collections import defaultdict
d, sub = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "lob_step_kernel"
acc, n = defaultdict(float), defaultdict(int)
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print(" ".join(f"{k}={acc[k] / n[k]:.0f}" for k in sorted(acc)), f"dispatches={max(n.values()) if n else 0}")
