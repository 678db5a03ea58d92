"""Thread scaling of the CPU oracle on this host (the cpu_baseline leg of bench.py uses all cores)."""
import ctypes as C, os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench, oracle_lib as ol
from rl4mm_amd import _capi
print("affinity", len(os.sched_getaffinity(0)), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a", flush=True)
os.system("lscpu | grep -E 'Model name|Socket|Thread|NUMA node\\(s\\)|^CPU\\(s\\)'")
lib = ol.oracle_binding().lib
lib.oracle_set_threads.restype = None; lib.oracle_set_threads.argtypes = [C.c_int]
streams = bench.build_streams("random", 70)
per = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for th in (1, 8, 32, 64, 128, 256):
    if th > len(os.sched_getaffinity(0)): break
    lib.oracle_set_threads(th)
    n = per * th
    cpu = ol.oracle_sim(bench.workload_cfg(_capi, "random", n))
    cpu.add_stream(streams[0]); cpu.set_portfolio(0, 10**12)
    st, sd = bench.workload_starts("random", n, 1, 1)
    cpu.reset(st, sd)
    acts = np.random.default_rng(0).uniform(0, 10, size=(500, n, 4))
    t0 = time.perf_counter()
    out = ol.oracle_rollout(cpu, _capi.POLICY_GIVEN, [], 500, acts, want_actions=False)
    dt = time.perf_counter() - t0
    print(f"threads {th:4d} envs {n:6d}: {out['steps'].sum() / dt / 1e3:9.1f} k env-steps/s  ({out['steps'].sum() / dt / th / 1e3:6.1f} k per thread)", flush=True)
    cpu.close()
