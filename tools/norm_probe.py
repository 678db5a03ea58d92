"""Step time with normalisation_on (z-score histories) for the default features, right after the 5-minute warm-up
(histories ~3000 values for the 0.1 s features).  usage: python tools/norm_probe.py [n_envs]"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from rl4mm_amd import _capi, _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
stream = bench.build_stream(60)
for norm in (0, 21004):
    feats = bench.default_features(_capi)
    for f in feats: f.norm_len = norm
    cfg = _capi.make_cfg(N, 10, feats, step_us=100_000, episode_us=1_800_000_000)
    sim = _lib.new_sim(cfg)
    sim.add_stream(stream); sim.set_portfolio(0, 10**12)
    starts = (36_000 + np.random.default_rng(1).integers(0, 60, size=N)) * 1_000_000
    t0 = time.perf_counter(); sim.reset(starts, 0); reset_s = time.perf_counter() - t0
    acts = torch.from_numpy(np.random.default_rng(2).uniform(0, 10, size=(120, N, 4))).to(dev)
    obs = torch.zeros((N, sim.obs_dim), dtype=torch.float64, device=dev)
    reward = torch.zeros(N, dtype=torch.float64, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
    for t in range(20): sim.step_device(acts[t].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
    sim.sync(); t0 = time.perf_counter()
    for t in range(20, 120): sim.step_device(acts[t].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
    sim.sync(); dt = (time.perf_counter() - t0) / 100
    print(f"normalisation {'on ' if norm else 'off'}: {N} envs, reset (3000 warm-up steps) {reset_s:.2f} s, step {dt * 1e6:.0f} us -> {N / dt / 1e6:.2f} M env-steps/s; finite obs share {float(torch.isfinite(obs).float().mean()):.2f}", flush=True)
    sim.close()
