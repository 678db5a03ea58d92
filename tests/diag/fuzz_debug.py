"""Debug helper for tests/diag/config_fuzz.py: replays one seed and prints, at the first reset whose observations differ, the differing entry of every reset env (both sides)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tests" / "diag"))
import config_fuzz as cf
import oracle_lib as ol
from rl4mm_amd import _lib
seed, n = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 96
cfg, stream, starts, amax, ep_steps, desc, r = cf.draw_case(seed, n)
print(desc)
gpu, cpu = _lib.new_sim(cfg), ol.oracle_sim(cfg)
inv0 = int(r.integers(-20, 20))
for s in (gpu, cpu): s.add_stream(stream); s.set_portfolio(inv0, 10**9)
og, oc = gpu.reset(starts, 0), cpu.reset(starts, 0)
A = cfg.action_dim; high = np.full(A, amax)
if cfg.market_order_clearing: high[-1] = float(r.choice([5.0, 60.0, 400.0]))
for t in range(2 * ep_steps + 8):
    act = r.uniform(0, high, size=(n, A)); m = r.random(n); act[m < 0.03] = 0.0; act[(m >= 0.03) & (m < 0.06)] = high
    og, rg, dg = gpu.step(act); oc, rc, dc = cpu.step(act)
    if dg.any():
        ids = np.nonzero(dg)[0].astype(np.int32)
        stg, stc = gpu.status(), cpu.status()
        starts[ids] += (cfg.episode_us // 1_000_000 + 2) * 1_000_000
        o1, o2 = gpu.reset(starts[ids], 0, ids), cpu.reset(starts[ids], 0, ids)
        bad = ~np.isclose(o1, o2, rtol=1e-12, atol=1e-9)
        print(f"step {t}: reset of {len(ids)} envs (status before: {sorted(set(stg[ids].tolist()))}); entries that differ: {sorted(set(np.nonzero(bad)[1].tolist()))}")
        if bad.any():
            f = int(np.nonzero(bad)[1][0])
            for k in np.nonzero(bad[:, f])[0][:12]: print(f"   env {ids[k]} status {stg[ids[k]]}/{stc[ids[k]]} entry {f}: gpu {o1[k, f]!r} oracle {o2[k, f]!r}; pre-reset obs gpu {og[ids[k], f]!r} oracle {oc[ids[k], f]!r}")
            print("   envs that agree:", int((~bad[:, f]).sum()), "of", len(ids))
            break
