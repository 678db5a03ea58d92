"""(GPU box) the deep-queue drifting day of tests/test_gpu_far.py::test_capacity_margins... one step at a time against the oracle:
the first env / step whose portfolio or book differs, its status, and the level that differs."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import drift_stream as ds, oracle_lib as ol
from rl4mm_amd import _capi, _lib

n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # > 1: steps run as chunk-step launches of the rollout kernel (oracle: oracle_rollout)
ps = ds.drift_stream(21, ds.ramp_path(6.0), duration_s=1_900.0, noise_per_s=150.0, target_depth=180)
F = _capi.make_feature
feats = [F(_capi.F_SPREAD, 0, 100_000, 0, 5000), F(_capi.F_INVENTORY, 0, 100_000, -1e6, 1e6), F(_capi.F_TRADE_VOLUME_IMBALANCE, 20, 100_000, -1, 1)]
cfg = _capi.make_cfg(n, 10, feats, step_us=100_000, episode_us=1_800_000_000)
gpu, cpu = _lib.new_sim(cfg), ol.oracle_sim(cfg)
starts = (34_205 + np.arange(n) % 50) * 1_000_000
for sim in (gpu, cpu):
    sim.add_stream(ps); sim.set_portfolio(0, 10**12)
assert np.array_equal(gpu.reset(starts, 0), cpu.reset(starts, 0))
rng = np.random.Generator(np.random.PCG64(5))
print("limits", gpu.limits())
import torch
obs, reward, done = torch.zeros((n, gpu.obs_dim), dtype=torch.float64).cuda(), torch.zeros(n, dtype=torch.float64).cuda(), torch.zeros(n, dtype=torch.uint8).cuda()
for t in range(0, 18000, chunk):
    if chunk == 1:
        a = rng.uniform(0, 10, size=(n, 4))
        og, rg, dg = gpu.step(a); oc, rc, dc = cpu.step(a)
    else:
        acts = rng.uniform(0, 10, size=(chunk, n, 4))
        d_acts = torch.from_numpy(acts).cuda()
        gpu.rollout_device(_capi.POLICY_GIVEN, [], chunk, d_acts.data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0, 0)
        gpu.sync()
        out = ol.oracle_rollout(cpu, _capi.POLICY_GIVEN, [], chunk, acts, want_actions=False)
        dg, dc = done.cpu().numpy(), out["done"]
    ig, cg, pg = gpu.portfolios(); ic, cc, pc = cpu.portfolios()
    bad = np.flatnonzero((ig != ic) | (cg != cc) | (pg != pc) | (dg != dc))
    if t % 500 == 0:
        st = gpu.env_state(0); print("step", t, "levels", st["n_buy_levels"], st["n_sell_levels"], "agent", st["n_agent_orders"], flush=True)
    check_books = chunk > 1 or t % 50 == 49
    if len(bad) == 0 and check_books:
        for e in range(0, n, 4 if chunk == 1 else 1):
            if not np.array_equal(gpu.queues(e), cpu.queues(e)):
                bad = np.array([e]); print("book differs (portfolios equal)")
                break
    if len(bad):
        e = int(bad[0])
        print("step", t, "envs", bad[:8], "status gpu", gpu.status()[bad[:8]], "cpu", cpu.status()[bad[:8]])
        print("gpu state", gpu.env_state(e)); print("cpu state", cpu.env_state(e))
        qg, qc = gpu.queues(e), cpu.queues(e)
        print("queue shapes", qg.shape, qc.shape)
        kg = {(int(r[0]), int(r[1])) for r in qg}; kc = {(int(r[0]), int(r[1])) for r in qc}
        print("levels only on gpu", sorted(kg - kc)[:5], "only on cpu", sorted(kc - kg)[:5])
        for key in sorted(kg & kc):
            lg = qg[(qg[:, 0] == key[0]) & (qg[:, 1] == key[1])]; lc = qc[(qc[:, 0] == key[0]) & (qc[:, 1] == key[1])]
            if lg.shape != lc.shape or not np.array_equal(lg, lc):
                print("level", key, "gpu orders", len(lg), "cpu orders", len(lc), "vol", lg[:, 4].sum(), lc[:, 4].sum())
                m = min(len(lg), len(lc))
                d = np.flatnonzero((lg[:m] != lc[:m]).any(axis=1))
                print(" first differing rows", d[:5]); 
                for i in d[:3]: print("   gpu", lg[i], "cpu", lc[i])
                break
        sys.exit(1)
print("no difference")
