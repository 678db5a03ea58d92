"""Experiment: the same 4096 envs as G independent handles (N/G envs each, own HIP stream), stepped round-robin.
Between two steps of one group there is no barrier with the other groups, so one group's stragglers (a launch lasts
as long as its busiest env) overlap with the other groups' next launch.  usage: python tools/group_probe.py [N] [K]"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from rl4mm_amd import _capi, _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda", 0)
stream = bench.build_stream(60)
rng = np.random.default_rng(1)
starts_all = (36_000 + rng.integers(0, 60, size=N)) * 1_000_000
acts_all = torch.from_numpy(np.random.default_rng(2).uniform(0, 10, size=(K + 50, N, 4))).to(dev)

for G in (1, 2, 4, 8):
    n = N // G
    sims, bufs, streams = [], [], []
    for g in range(G):
        cfg = _capi.make_cfg(n, 10, bench.default_features(_capi), step_us=100_000, episode_us=1_800_000_000)
        sim = _lib.new_sim(cfg)
        st = torch.cuda.Stream(device=dev)
        sim.set_stream(st.cuda_stream)
        sim.add_stream(stream); sim.set_portfolio(0, 10**12)
        sim.reset(starts_all[g * n:(g + 1) * n], 0)
        acts = acts_all[:, g * n:(g + 1) * n].contiguous()
        obs = torch.zeros((n, sim.obs_dim), dtype=torch.float32, device=dev)
        reward = torch.zeros(n, dtype=torch.float64, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev)
        ret = torch.zeros(n, dtype=torch.float64, device=dev)
        sims.append(sim); bufs.append((acts, obs, reward, done, ret)); streams.append(st)
    torch.cuda.synchronize()

    def run(t0, k):
        for t in range(t0, t0 + k):
            for g in range(G):
                a, o, r, d, ret = bufs[g]
                sims[g].step_device(a[t].data_ptr(), 0, 0, o.data_ptr(), r.data_ptr(), d.data_ptr(), ret.data_ptr())

    run(0, 50)
    for s in sims: s.sync()
    torch.cuda.synchronize()
    w0 = time.perf_counter()
    run(50, K)
    w1 = time.perf_counter()
    for s in sims: s.sync()
    torch.cuda.synchronize()
    w2 = time.perf_counter()
    total = sum(float(b[4].sum().item()) for b in bufs)
    print(f"G={G}: {N} envs, {K} steps: {1e3 * (w2 - w0) / K:.4f} ms/step -> {N * K / (w2 - w0) / 1e6:.1f} M env-steps/s "
          f"(enqueue {1e3 * (w1 - w0) / K:.4f} ms/step), return checksum {total:.6e}", flush=True)
    for s in sims: s.close()
