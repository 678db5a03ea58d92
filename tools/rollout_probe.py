"""A/B of build variants on the multi-step launch (lobsim_rollout_device, pre-drawn actions).
usage: python tools/rollout_probe.py name1:-DFLAG name2:..."""
import subprocess, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from rl4mm_amd import _capi, _lib

N, K = 4096, 300
dev = torch.device("cuda", 0)
stream = bench.build_stream(60)
starts = (36_000 + np.random.default_rng(1).integers(0, 60, size=N)) * 1_000_000
acts = torch.from_numpy(np.random.default_rng(2).uniform(0, 10, size=(K, N, 4))).to(dev)
for arg in sys.argv[1:] or ["default:"]:
    name, flags = (arg.split(":", 1) + [""])[:2]
    so = ROOT / "gpurun_out" / f"liblobsim_{name}.so"
    so.parent.mkdir(exist_ok=True)
    _lib.compile_library(so, flags.split())
    cfg = _capi.make_cfg(N, 10, bench.default_features(_capi), step_us=100_000, episode_us=1_800_000_000)
    sim = _capi.Sim(_capi.Binding(so), cfg)
    sim.add_stream(stream); sim.set_portfolio(0, 10**12)
    sim.reset(starts, 0)
    obs = torch.zeros((N, sim.obs_dim), dtype=torch.float64, device=dev)
    reward = torch.zeros(N, dtype=torch.float64, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
    ret = torch.zeros(N, dtype=torch.float64, device=dev); taken = torch.zeros(N, dtype=torch.int32, device=dev)
    ts = []
    for rep in range(3):
        sim.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
        sim.rollout_device(_capi.POLICY_GIVEN, [], 100, acts[rep * 100:].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), ret.data_ptr(), taken.data_ptr())
        sim.sync(); ts.append(time.perf_counter() - t0)
    print(f"{name:10s} 100-step launches: {[round(1e3 * t, 2) for t in ts]} ms -> {N * 100 / min(ts) / 1e6:.1f} M env-steps/s, return checksum {float(ret.sum().item()):.6e}", flush=True)
    sim.close()
