"""Diagnostic (not part of the suite): a long run of BASELINE configs[1] - 4096 envs, default features, random actions - on
the HIP library against the CPU oracle, EVERY env compared every `chunk` steps (inventory, cash, price bit for bit,
observations to 1e-12, the fill counters and resync counts at the end).  The suite's full-size test covers 300 steps; this
covers minutes of market time, per-step launches and multi-step launches alternating, so that rare paths (compactions of
deep queues, position-cache aliasing, outer-level resyncs, far levels) get thousands of chances.
usage: python tests/diag/soak.py [steps=6000] [chunk=250] [envs=4096] [workload=random|ad|deep50|teradactyl]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import bench
import oracle_lib as ol
from rl4mm_amd import _capi, _lib

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 250
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
wl = sys.argv[4] if len(sys.argv) > 4 else "random"
import os

# SOAK_LADDER=128: the 128-level kernels on a book that would fit 64; SOAK_STEP_US=50000: another step size (the generic, not the
# specialised kernels; the default features allow any step <= 0.1 s, HistoricalOrderbookEnvironment.py:398)
step_us = int(os.environ.get("SOAK_STEP_US", 100_000))
cfg = bench.workload_cfg(_capi, wl, n, ladder_levels=int(os.environ.get("SOAK_LADDER", 0)))  # (the bench's configuration of that workload; its episode is the 30 min default)
if step_us != 100_000:
    kw = dict(step_us=step_us, episode_us=bench.EPISODE_US, ladder_levels=int(os.environ.get("SOAK_LADDER", 0)))
    feats = bench.default_features(_capi)
    for k in (0, 5, 6): feats[k].update_us = step_us  # Spread, Inventory, EpisodeProportion tick with the step (get_default_features(step_size))
    feats[6].dparam = step_us / bench.EPISODE_US
    cfg = _capi.make_cfg(n, 10, feats, **kw)
streams = bench.build_streams(wl, steps * step_us // 1_000_000 + 20)
rng = np.random.Generator(np.random.PCG64(4242))
starts, sids = bench.workload_starts(wl, n, len(streams), 4242)
gpu, cpu = _lib.new_sim(cfg), ol.oracle_sim(cfg)
for s in (gpu, cpu):
    for st in streams:
        s.add_stream(st)
    s.set_portfolio(0, 10**12)
o_g, o_c = gpu.reset(starts, sids), cpu.reset(starts, sids)
assert np.allclose(o_g, o_c, rtol=1e-12, atol=1e-9)
dev = torch.device("cuda", 0)
D = gpu.obs_dim
obs = torch.zeros((n, D), dtype=torch.float64, device=dev)
reward = torch.zeros(n, dtype=torch.float64, device=dev)
done = torch.zeros(n, dtype=torch.uint8, device=dev)
ret = torch.zeros(n, dtype=torch.float64, device=dev)
taken = torch.zeros(n, dtype=torch.int32, device=dev)
ret_c = np.zeros(n)
tera = wl == "teradactyl"  # (the rule-based agent on the device: its action is worked out from the env's own observation, so a difference compounds)
policy, params = (_capi.POLICY_TERADACTYL, bench.TERADACTYL_PARAMS) if tera else (_capi.POLICY_GIVEN, [])
for c in range(steps // chunk):
    acts = None if tera else rng.uniform(0, 10, size=(chunk, n, 4))
    d_acts = None if tera else torch.from_numpy(acts).to(dev)
    if c % 2 == 0:  # per-step launches
        for t in range(chunk):
            if tera: gpu.rollout_device(policy, params, 1, 0, 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), ret.data_ptr(), 0)
            else: gpu.step_device(d_acts[t].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), ret.data_ptr())
    else:  # one multi-step launch
        gpu.rollout_device(policy, params, chunk, 0 if tera else d_acts.data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), ret.data_ptr(), taken.data_ptr())
    gpu.sync()
    out = ol.oracle_rollout(cpu, policy, params, chunk, acts, want_actions=False)
    ret_c += out["ret"]
    for a, b, what in zip(gpu.portfolios(), cpu.portfolios(), ("inventory", "cash", "price")):
        assert np.array_equal(a, b), f"chunk {c}: {what} differs in envs {np.nonzero(a != b)[0][:8]}"
    og = obs.cpu().numpy()
    assert np.allclose(og, out["obs"], rtol=1e-12, atol=1e-9), f"chunk {c}: obs differ by {np.abs(og - out['obs']).max()}"
    assert np.array_equal(done.cpu().numpy(), out["done"]), f"chunk {c}: done flags differ"
    assert np.allclose(ret.cpu().numpy(), ret_c, rtol=1e-9, atol=1e-6), f"chunk {c}: returns differ by {np.abs(ret.cpu().numpy() - ret_c).max()}"
    st = [gpu.env_state(e)["status"] for e in range(0, n, 64)]
    print(f"chunk {c + 1}/{steps // chunk} ({'steps' if c % 2 == 0 else 'rollout'}): {n} envs equal after {(c + 1) * chunk} steps; sampled status bits {sorted(set(st))}", flush=True)
keys = ("status", "now_us", "n_agent_orders", "n_resyncs", "n_buy_levels", "n_sell_levels", "agent_fill_count", "agent_buy_volume", "agent_sell_volume",
        "ext_buy_count", "ext_sell_count", "ext_buy_volume", "ext_sell_volume", "best_buy", "best_sell")  # fmt: skip
n_res = 0
for e in range(n):
    sg, sc = gpu.env_state(e), cpu.env_state(e)
    assert all(sg[k] == sc[k] for k in keys), (e, {k: (sg[k], sc[k]) for k in keys if sg[k] != sc[k]})
    n_res += sg["n_resyncs"]
for e in (0, n // 3, n - 1):
    assert np.array_equal(gpu.queues(e), cpu.queues(e)), f"env {e}: per-order book contents differ"
print(f"soak ok ({wl}): {n} envs x {steps} steps, every env equal to the oracle at every chunk boundary; {n_res} outer-level resyncs in total; high water (orders at one price, live agent orders, buy levels, sell levels) {ol.oracle_high_water(cpu).max(axis=0).tolist()}")
