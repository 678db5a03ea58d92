"""Timing probe: separates fixed (synchronize) from per-step cost of the step launch loop."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from rl4mm_amd import _capi, _lib

N = 4096
dev = torch.device("cuda", 0)
stream = bench.build_stream(60)
cfg = _capi.make_cfg(N, 10, bench.default_features(_capi), step_us=100_000, episode_us=1_800_000_000)
sim = _lib.new_sim(cfg)
use_torch_stream = len(sys.argv) > 1 and sys.argv[1] == "torch"
if use_torch_stream:
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    sim.set_stream(torch.cuda.current_stream().cuda_stream)
sim.add_stream(stream); sim.set_portfolio(0, 10**12)
starts = (36_000 + np.random.default_rng(1).integers(0, 60, size=N)) * 1_000_000
sim.reset(starts, 0)
acts = torch.from_numpy(np.random.default_rng(2).uniform(0, 10, size=(600, N, 4))).to(dev)
obs = torch.zeros((N, sim.obs_dim), dtype=torch.float64, device=dev)
reward = torch.zeros(N, dtype=torch.float64, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
t = 0
def sync():
    t0 = time.perf_counter(); torch.cuda.synchronize(); sim.sync(); return (time.perf_counter() - t0) * 1e3
print("idle sync ms", sync(), sync())
for K in (1, 10, 100, 300):
    sync()
    t0 = time.perf_counter()
    for _ in range(K):
        sim.step_device(acts[t].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0); t += 1
    t1 = time.perf_counter()
    s = sync()
    t2 = time.perf_counter()
    print(f"K={K}: enqueue {1e3*(t1-t0):.2f} ms, final sync {s:.2f} ms, total/K {1e3*(t2-t0)/K:.3f} ms")
returns = torch.zeros(N, dtype=torch.float64, device=dev)
for variant in ("events", "add", "both"):
    K = 100
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    sync()
    t0 = time.perf_counter()
    for i in range(K):
        if variant in ("events", "both"): ev[i][0].record()
        sim.step_device(acts[t % 600].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0); t += 1
        if variant in ("events", "both"): ev[i][1].record()
        if variant in ("add", "both"): returns += reward
    t1 = time.perf_counter(); s = sync(); t2 = time.perf_counter()
    km = np.mean([a.elapsed_time(b) for a, b in ev]) if variant != "add" else 0
    print(f"{variant}: enqueue {1e3*(t1-t0):.2f} ms, total/K {1e3*(t2-t0)/K:.3f} ms, event kernel ms {km:.3f}")
