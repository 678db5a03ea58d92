"""Diagnostic: per-phase share of the step kernel's wave cycles (s_memtime stamps, -DLOBSIM_PROFILE build).
Builds build/liblobsim_prof.so (never the product library) and runs the bench workload through it.
Read the SHARES, not the absolute cycles (stamps forbid overlaps the real kernel has)."""
import ctypes as C, subprocess, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from rl4mm_amd import _capi, _lib

so = ROOT / "gpurun_out" / "liblobsim_prof.so"
so.parent.mkdir(exist_ok=True)
_lib.compile_library(so, ["-DLOBSIM_PROFILE"])
b = _capi.Binding(so)
N = 4096
fixed = len(sys.argv) > 1 and sys.argv[1] == "fixed"
cfg = _capi.make_cfg(N, 10, bench.default_features(_capi), step_us=100_000, episode_us=1_800_000_000)
sim = _capi.Sim(b, cfg)
sim.add_stream(bench.build_stream(60)); sim.set_portfolio(0, 10**12)
starts = (36_000 + np.random.default_rng(1).integers(0, 60, size=N)) * 1_000_000
sim.reset(starts, 0)
dev = torch.device("cuda", 0)
K = 100
acts = np.random.default_rng(2).uniform(0, 10, size=(K, N, 4))
if fixed: acts[:] = np.array([1.0, 2.0, 1.0, 2.0])
acts = torch.from_numpy(acts).to(dev)
obs = torch.zeros((N, sim.obs_dim), dtype=torch.float64, device=dev)
reward = torch.zeros(N, dtype=torch.float64, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
out = np.zeros((N, 8), dtype=np.uint64)
b.lib.lobsim_prof_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]

for t in range(10): sim.step_device(acts[t].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
b.lib.lobsim_prof_read(sim.h, out.ctypes.data, 1)
for t in range(10, K): sim.step_device(acts[t].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
b.lib.lobsim_prof_read(sim.h, out.ctypes.data, 0)
steps = out[:, 7].astype(float)
names = ["load_env", "build_ops", "process_ops", "messages+resync+price", "features+reward", "outputs+store_env"]
tot = out[:, :6].sum(axis=1).astype(float) / steps
print(("fixed action" if fixed else "random actions"), f"mean cycles per env-step {tot.mean():.0f}; agent ops per step {(out[:,6]/steps).mean():.1f}")
print("  per-env mean cycles/step percentiles 1/50/99/max:", np.percentile(tot, [1, 50, 99, 100]).round(0).tolist(), "(a launch lasts as long as its slowest wave)")
for k, n in enumerate(names):
    c = out[:, k].astype(float) / steps
    print(f"  {n:26s} {c.mean():9.0f} cycles  {100*c.mean()/tot.mean():5.1f} %")

