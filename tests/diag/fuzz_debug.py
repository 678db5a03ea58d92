"""Debug helper for tests/diag/config_fuzz.py: replays one seed step by step and stops at the first difference of ANY kind for a
watched env (its book order by order, its portfolio, its observation row), printing both sides.
usage: python tests/diag/fuzz_debug.py <seed> [envs=96] [watch_env=-1: every env's portfolio / status / obs, books of none]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tests" / "diag"))
import config_fuzz as cf
import oracle_lib as ol
from rl4mm_amd import _lib
seed, n = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 96
watch = int(sys.argv[3]) if len(sys.argv) > 3 else -1
cfg, extra, starts, amax, ep_steps, desc, r, (first, span), chunk = cf.draw_case(seed, n)
streams, policy, masked, r2 = extra['streams'], extra['policy'], extra['masked'], extra['r2']
sids = r2.integers(0, len(streams), size=n).astype(np.int32)
print(desc)
gpu, cpu = _lib.new_sim(cfg), ol.oracle_sim(cfg)
inv0 = int(r.integers(-20, 20))
for s in (gpu, cpu):
    for st_ in streams: s.add_stream(st_)
    s.set_portfolio(inv0, 10**9)
og, oc = gpu.reset(starts, sids), cpu.reset(starts, sids)
A = cfg.action_dim; high = np.full(A, amax)
if cfg.market_order_clearing: high[-1] = float(r.choice([5.0, 60.0, 400.0]))
steps = 2 * ep_steps + 8
if chunk: steps = -(-steps // chunk)
def books(e):
    a, b = gpu.queues(e), cpu.queues(e)
    if np.array_equal(a, b): return None
    k = next((i for i in range(min(len(a), len(b))) if not np.array_equal(a[i], b[i])), min(len(a), len(b)))
    return f"books of env {e} differ from row {k} on ({len(a)} vs {len(b)} rows)\n   gpu    {a[max(0, k - 2):k + 4].tolist()}\n   oracle {b[max(0, k - 2):k + 4].tolist()}"
prev_q = cpu.queues(watch) if watch >= 0 else None
for t in range(steps):
    T = chunk or 1
    act = r.uniform(0, high, size=(T, n, A)); m = r.random((T, n)); act[m < 0.03] = 0.0; act[(m >= 0.03) & (m < 0.06)] = high
    peek = ol.oracle_peek_orders(cpu, act[0, watch], watch) if watch >= 0 and not chunk else None
    active = (r2.random(n) >= 0.1).astype(np.uint8) if masked and not chunk else None
    if chunk: (og, rg, dg, xg), (oc, rc, dc, xc) = cf._rollout(gpu, act, policy), cf._rollout(cpu, act, policy)
    else:
        og, rg, dg = gpu.step(act[0], active); oc, rc, dc = cpu.step(act[0], active)
        if masked:
            for a in (og, oc, rg, rc, dg, dc): a[active == 0] = 0
    sg, sc = gpu.status(), cpu.status()
    ig, cg, pg = gpu.portfolios(); ic, cc, pc = cpu.portfolios()
    bad = (sg != sc) | (ig != ic) | (cg != cc) | (dg != dc) | ~np.isclose(og, oc, rtol=1e-12, atol=1e-9, equal_nan=True).all(axis=1)
    msg = books(watch) if watch >= 0 else None
    if bad.any() or msg:
        e = int(np.nonzero(bad)[0][0]) if bad.any() else watch
        print(f"step {t}: env {e}: status {sg[e]}/{sc[e]} inv {ig[e]}/{ic[e]} cash {cg[e]}/{cc[e]} done {dg[e]}/{dc[e]} now {gpu.env_state(e)['now_us']}\n   action {act[-1, e].tolist()}\n   obs gpu    {og[e].tolist()}\n   obs oracle {oc[e].tolist()}")
        print(msg or books(e) or "   (books equal)")
        print("   env_state gpu   ", gpu.env_state(e)); print("   env_state oracle", cpu.env_state(e))
        if watch >= 0 and peek is not None:
            print("   the step's agent orders (oracle) {type, side, price, volume, id / rank, is_external}:\n   ", peek.tolist())
            print("   fills gpu   ", gpu.fills(e).tolist()); print("   fills oracle", ol.oracle_fills(cpu, e).tolist())
            print("   book before the step [side, price, kind, id, volume] (best 12 rows a side):")
            for side in (0, 1): print("     ", [r_.tolist() for r_ in prev_q if r_[0] == side][:12])
            print("   book after, gpu:   ", [r_.tolist() for r_ in gpu.queues(e) if r_[0] == 0][:10], [r_.tolist() for r_ in gpu.queues(e) if r_[0] == 1][:10])
            print("   book after, oracle:", [r_.tolist() for r_ in cpu.queues(e) if r_[0] == 0][:10], [r_.tolist() for r_ in cpu.queues(e) if r_[0] == 1][:10])
        break
    if watch >= 0: prev_q = cpu.queues(watch)
    if dg.any():
        ids = np.nonzero(dg)[0].astype(np.int32)
        starts[ids] = (first + r.integers(0, span, size=len(ids))) * 1_000_000
        sids[ids] = r2.integers(0, len(streams), size=len(ids))
        o1, o2 = gpu.reset(starts[ids], sids[ids], ids), cpu.reset(starts[ids], sids[ids], ids)
        badr = ~np.isclose(o1, o2, rtol=1e-12, atol=1e-9, equal_nan=True)
        s1 = gpu.status()[ids]
        badr[s1 != 0] = False
        if badr.any():
            f = int(np.nonzero(badr)[1][0])
            print(f"step {t}: reset of {len(ids)} envs; entries that differ: {sorted(set(np.nonzero(badr)[1].tolist()))}; feature {f}: {desc['feats'][f] if f < len(desc['feats']) else 'action'}")
            for k in np.nonzero(badr[:, f])[0][:8]: print(f"   env {ids[k]} start {starts[ids[k]]} entry {f}: gpu {o1[k, f]!r} oracle {o2[k, f]!r}; pre-reset obs gpu {og[ids[k], f]!r} oracle {oc[ids[k], f]!r}")
            print("   envs that agree:", int((~badr[:, f]).sum()), "of", len(ids))
            break
