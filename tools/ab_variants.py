"""A/B of build variants of the step kernel on the bench workload (diagnostic; builds into gpurun_out/, never the
product library).  usage: python tools/ab_variants.py name1:-DFLAG=.. name2:-DFLAG=.."""
import subprocess, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from rl4mm_amd import _capi, _lib

N, K = 4096, 200
dev = torch.device("cuda", 0)
stream = bench.build_stream(60)
starts = (36_000 + np.random.default_rng(1).integers(0, 60, size=N)) * 1_000_000
acts = torch.from_numpy(np.random.default_rng(2).uniform(0, 10, size=(K + 20, N, 4))).to(dev)
res = {}
variants = [(a.split(":") + ["", ""])[:3] for a in sys.argv[1:]] or [["default", "", ""]]  # name:flags:source
sims = {}
for name, flags, src in variants:
    so = ROOT / "gpurun_out" / f"liblobsim_{name}.so"
    so.parent.mkdir(exist_ok=True)
    _lib.compile_library(so, flags.split(), source=(ROOT / src) if src else None)
    cfg = _capi.make_cfg(N, 10, bench.default_features(_capi), step_us=100_000, episode_us=1_800_000_000)
    sim = _capi.Sim(_capi.Binding(so), cfg)
    sim.add_stream(stream); sim.set_portfolio(0, 10**12)
    t0 = time.perf_counter(); sim.reset(starts, 0); res[name] = {"reset_ms": 1e3 * (time.perf_counter() - t0), "runs": []}
    sims[name] = sim
obs = torch.zeros((N, 16), dtype=torch.float64, device=dev)
reward = torch.zeros(N, dtype=torch.float64, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
t = 0
for rep in range(3):  # interleaved A/B in one process
    for name, sim in sims.items():
        tt = 20 * rep  # every variant steps the same actions from its own (identical) state
        for i in range(5): sim.step_device(acts[(tt + i) % K].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
        sim.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(60): sim.step_device(acts[(tt + 5 + i) % K].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
        sim.sync()
        res[name]["runs"].append(1e6 * (time.perf_counter() - t0) / 60)
for name, r in res.items():
    print(f"{name:12s} reset {r['reset_ms']:8.1f} ms   step us/launch {np.round(r['runs'], 1).tolist()}  -> {N / np.median(r['runs']):.1f} M env-steps/s")
