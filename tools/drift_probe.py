"""Diagnostic: what a drifting price costs.  4096 envs with the default features on a constructed stream that moves one tick every
K seconds (tests/drift_stream.py): the episode is advanced by multi-step launches, and at several points the per-step launch is
timed over 200 steps next to the level counts of a sample of envs (near ladder 64 levels per side, the rest in the far rings).
usage: python tools/drift_probe.py [seconds_per_tick=4] [n_envs=4096]"""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench
import drift_stream as ds
from rl4mm_amd import _capi, _lib

spt = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda", 0)
t0 = time.time()
ps = ds.drift_stream(5, ds.ramp_path(spt), duration_s=2_300.0, noise_per_s=180.0)  # the bench stream's message rate
print(f"stream: {ps.n_msgs} messages, {ps.n_snaps} snapshots ({time.time() - t0:.1f} s to build)", flush=True)
cfg = _capi.make_cfg(N, 10, bench.default_features(_capi), step_us=100_000, episode_us=1_800_000_000)
sim = _lib.new_sim(cfg)
sim.add_stream(ps); sim.set_portfolio(0, 10**12)
starts = (34_200 + 310 + np.random.default_rng(1).integers(0, 60, size=N)) * 1_000_000
sim.reset(starts, 0)
K, CH = 200, 2800
obs = torch.zeros((N, sim.obs_dim), dtype=torch.float64, device=dev)
reward = torch.zeros(N, dtype=torch.float64, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
steps = torch.zeros(N, dtype=torch.int32, device=dev)
g = torch.Generator(device=dev).manual_seed(3)
done_steps = 0
for leg in range(6):
    acts = torch.rand((K, N, 4), dtype=torch.float64, device=dev, generator=g) * 10.0
    torch.cuda.synchronize()
    for t in range(10): sim.step_device(acts[t].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
    sim.sync()
    t1 = time.perf_counter()
    for t in range(10, K): sim.step_device(acts[t].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
    sim.sync()
    us = (time.perf_counter() - t1) / (K - 10) * 1e6
    done_steps += K
    st = [sim.env_state(int(e)) for e in range(0, N, max(1, N // 32))]
    print(f"after {done_steps:6d} steps: {us:6.1f} us per step = {N / us:6.1f} M env-steps/s | buy levels {np.mean([s['n_buy_levels'] for s in st]):6.1f} sell levels {np.mean([s['n_sell_levels'] for s in st]):5.1f} | "
          f"envs with a status bit in the sample {sum(s['status'] != 0 for s in st)}", flush=True)
    if leg == 5: break
    big = torch.rand((CH, N, 4), dtype=torch.float64, device=dev, generator=g) * 10.0
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    sim.rollout_device(_capi.POLICY_GIVEN, [], CH, big.data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0, steps.data_ptr())
    sim.sync()
    done_steps += CH
    print(f"   ({CH} steps as one launch: {(time.perf_counter() - t1) / CH * 1e6:.1f} us per step)", flush=True)
