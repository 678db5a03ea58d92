"""Diagnostic: where do the ~90 us of fixed cost of a 20-step timed region go?  (host call times, event records, sync)"""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench
from rl4mm_amd import _capi, _lib

N, K = 4096, 20
dev = torch.device("cuda", 0)
cfg = bench.workload_cfg(_capi, "random", N)
sim = _lib.new_sim(cfg)
if "own" not in sys.argv:  # like bench.py: the kernels run on a torch-owned stream, so torch events bracket them
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    sim.set_stream(torch.cuda.current_stream().cuda_stream)
sim.add_stream(bench.build_streams("random", 60)[0]); sim.set_portfolio(0, 10**12)
starts, sids = bench.workload_starts("random", N, 1, 7)
sim.reset(starts, 0)
acts = torch.rand((400, N, 4), dtype=torch.float64, device=dev) * 10
obs = torch.zeros((N, sim.obs_dim), dtype=torch.float64, device=dev); obs32 = torch.zeros((N, sim.obs_dim), dtype=torch.float32, device=dev)
reward = torch.zeros(N, dtype=torch.float64, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev); ret = torch.zeros(N, dtype=torch.float64, device=dev)
ap = [acts[t].data_ptr() for t in range(400)]
op = (obs.data_ptr(), obs32.data_ptr(), reward.data_ptr(), done.data_ptr(), ret.data_ptr())
P = time.perf_counter
def region(k, events, t_off):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = P()
    if events: e0.record()
    t1 = P()
    for t in range(k): sim.step_device(ap[t_off + t], 0, *op)
    t2 = P()
    if events: e1.record()
    t3 = P()
    torch.cuda.synchronize()
    t4 = P()
    return (t4 - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6 / k, (t3 - t2) * 1e6, (t4 - t3) * 1e6, (e0.elapsed_time(e1) * 1e3 if events else 0.0)
big = torch.rand((4096, 4096), device=dev)
def spin(ms):  # keep the GPU busy without stepping the envs
    t0 = P()
    while (P() - t0) * 1e3 < ms: (big @ big).sum().item()
for mode in ("cold", "cold", "ev", "ev", "ev"):
    sim.reset(starts, 0)
    for t in range(5): sim.step_device(ap[t], 0, *op)
    torch.cuda.synchronize()
    if mode.startswith("spin"): spin(int(mode[4:]))
    if mode.startswith("sleep"): time.sleep(0.05)
    r = region(20, mode == "ev", 5)
    print(f"after reset + 5 steps, {mode}: 20 steps wall {r[0]:.0f} us ({r[0]/20:.2f}/step), record0 {r[1]:.1f} host/launch {r[2]:.2f} record1 {r[3]:.1f} final sync {r[4]:.0f} event span {r[5]:.0f}")
    r = region(20, False, 25)
    print(f"   next 20 steps: wall {r[0]:.0f} us ({r[0]/20:.2f}/step)")
