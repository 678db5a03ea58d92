"""A/B of PREBUILT variants of the library on the bench workload: per-step launches and the multi-step rollout launch,
interleaved in one process.  usage: python tools/ab_libs.py name=path/to/lib.so ...   (build the variants in the build
container into build/variants/: they travel with the snapshot, the GPU box spends no time compiling)"""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from rl4mm_amd import _capi

import os
N, K = int(os.environ.get("AB_ENVS", 4096)), 200
dev = torch.device("cuda", 0)
stream = bench.build_stream(80)
starts = (36_000 + np.random.default_rng(1).integers(0, 60, size=N)) * 1_000_000
acts = torch.from_numpy(np.random.default_rng(2).uniform(0, 10, size=(K + 20, N, 4))).to(dev)
sims, res = {}, {}
for arg in sys.argv[1:]:
    name, path = arg.split("=")
    na = 0
    if "@" in path: path, na = path.split("@")[0], int(path.split("@")[1])  # name=lib.so@128: cfg.agent_orders
    kw = dict(per_step_reward=(_capi.R_ROLLING_SHARPE, 0, 0.0, 120, 60)) if os.environ.get("AB_REWARD") == "RS" else {}  # (RS: the EXTRAS kernel variants)
    if os.environ.get("AB_MAX_LEVELS"): kw["max_levels"] = int(os.environ["AB_MAX_LEVELS"])  # (slab pool size: 2 x 8 KiB per level and env)
    if na: kw["agent_orders"] = na
    cfg = _capi.make_cfg(N, 10, bench.default_features(_capi), step_us=100_000, episode_us=1_800_000_000, **kw)
    import subprocess
    missing = [n for n in _capi.PROTOTYPES if ("lobsim_" + n).encode() not in subprocess.run(["nm", "-D", str(ROOT / path)], capture_output=True).stdout]
    # (older builds lack the newest entry points; a library of round 4 is ABI 4: the cfg struct only renamed a reserved field since)
    sim = _capi.Sim(_capi.Binding(ROOT / path, skip=missing, check_abi=False), cfg)
    sim.add_stream(stream); sim.set_portfolio(0, 10**12)
    sim.reset(starts, 0)
    sims[name] = sim; res[name] = {"step": [], "roll": []}
obs = torch.zeros((N, 16), dtype=torch.float64, device=dev)
reward = torch.zeros(N, dtype=torch.float64, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
steps_taken = torch.zeros(N, dtype=torch.int32, device=dev)
for rep in range(int(os.environ.get("AB_REPS", 3))):
    for name, sim in sims.items():
        tt = 20 * rep
        for i in range(5): sim.step_device(acts[(tt + i) % K].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
        sim.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(60): sim.step_device(acts[(tt + 5 + i) % K].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
        sim.sync()
        res[name]["step"].append(1e6 * (time.perf_counter() - t0) / 60)
        t0 = time.perf_counter()
        sim.rollout_device(_capi.POLICY_GIVEN, [], 100, acts[:100].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0, steps_taken.data_ptr())
        sim.sync()
        res[name]["roll"].append(1e6 * (time.perf_counter() - t0) / 100)
inv = {n: s.portfolios()[0].sum() for n, s in sims.items()}
for name, r in res.items():
    print(f"{name:14s} step us {np.round(r['step'], 1).tolist()} -> {N / np.median(r['step']):6.1f} M/s | rollout us/step {np.round(r['roll'], 1).tolist()} -> {N / np.median(r['roll']):6.1f} M/s | inv checksum {inv[name]}")
