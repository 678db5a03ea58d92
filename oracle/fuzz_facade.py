#!/usr/bin/env python3
"""Build container only: the package's Gym facade (rl4mm_amd.env.HistoricalOrderbookEnvironment - feature classes, reward objects,
constructor arguments -> lobsim_cfg, the step / reset / exception surface, SimpleInfoCalculator) against the upstream reference's
env, on the configurations oracle/fuzz_reference.py draws.

Run:  PYTHONPATH=oracle/refshim:/root/reference PYTHONDONTWRITEBYTECODE=1 python oracle/fuzz_facade.py [first_seed=0] [end_seed=200]

There is no GPU here and the product has no CPU path: for this comparison - test infrastructure - the facade's handle is the CPU
oracle behind a small adapter (`OracleHandle`: the oracle's entry points plus the two pinned-buffer calls the array path makes).
What is exercised is everything ABOVE the C-ABI: the mirror classes must turn the same constructor arguments into the same simulator
(feature descriptors, windows, warm-up, clocks, normalisation history sizes, reward descriptors, action / observation spaces), hand
the same observations / rewards / dones / info dicts back, and raise where the reference raises.  The C-ABI below it is compared
with the oracle on the GPU (tests/, tests/diag/config_fuzz.py), the oracle with the reference in oracle/fuzz_reference.py."""
import contextlib
import io
import sys
import time
import traceback
from datetime import timedelta
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "oracle"))
sys.path.insert(0, str(REPO / "tests"))
_args = sys.argv[1:]
sys.argv = sys.argv[:1]
import fuzz_reference as fr  # noqa: E402
import gen_golden as gg  # noqa: E402
import oracle_lib as ol  # noqa: E402
from rl4mm_amd import env as E  # noqa: E402


OracleHandle = ol.OracleHandle  # (tests/oracle_lib.py)


REWARDS = {"PnL": lambda: E.get_reward_function("PnL"), "AD": lambda: E.get_reward_function("AD"), "SD": lambda: E.get_reward_function("SD")}


def mirror_reward(spec):
    if isinstance(spec, str):
        return REWARDS[spec]()
    return E.RollingSharpe(max_window_size=spec[1], min_window_size=spec[2])


def run(seed):
    kind, feats, agent_fn, starts, episode_s, kw = fr.draw(seed)
    if kw["concentration"] is not None and kw["concentration"] < 10.0:
        return dict(skipped="concentration below max_distribution_param: both constructors assert")
    db, ps, tick, L, _, _ = fr.day(kind)
    kw_ref = dict(kw, info=True)  # (the info dict of every step too)
    with contextlib.redirect_stdout(io.StringIO()):
        try:
            cfg, out = gg.run_trajectory(f"facade{seed}", db, tick, fr.DATE, L, feats, agent_fn, 2, starts, episode_s, **kw_ref)
        except OverflowError:
            return dict(skipped="OverflowError")
    types = [str(t) for t in out.get("crash_types", [])]
    if "AttributeError" in types or "OverflowError" in types:
        return dict(skipped=types)
    # ---- the same env through the package's classes
    mirror = [getattr(E, name)(**args) for name, args in (f._spec for f in feats)]
    step, episode = timedelta(seconds=kw["step_s"]), timedelta(seconds=episode_s)
    E._lib.new_sim = OracleHandle  # (test infrastructure: see the module docstring)
    calc = E.SimpleInfoCalculator(kw["mofi"], kw["enter_spread"], concentration=kw["concentration"])
    env = E.HistoricalOrderbookEnvironment(
        catalog=E.StreamCatalog().add(ps), features=mirror, ticker=tick, step_size=step, episode_length=episode,
        initial_portfolio=E.Portfolio(inventory=kw["initial"][0], cash=kw["initial"][1]), min_quote_level=kw["minq"], max_quote_level=kw["maxq"],
        min_date=fr.DATE, max_date=fr.DATE, min_start_timedelta=timedelta(seconds=starts[0]), max_end_timedelta=timedelta(seconds=starts[0]) + episode,
        market_order_clearing=kw["moc"], inc_prev_action_in_obs=kw["inc_prev"], per_step_reward_function=mirror_reward(kw["step_reward"]),
        terminal_reward_function=mirror_reward(kw["terminal_reward"]), concentration=kw["concentration"], market_order_fraction_of_inventory=kw["mofi"],
        enter_spread=kw["enter_spread"], n_levels=L, outer_levels=kw["outer_levels"], info_calculator=calc,
    )  # fmt: skip
    assert env.action_space.shape == (cfg["action_dim"],), (env.action_space.shape, cfg["action_dim"])
    # the spaces RLlib builds its policy from (HistoricalOrderbookEnvironment.py:85-94, 125-135)
    for mine, ref, what in ((env.action_space.low, out["act_low"], "action low"), (env.action_space.high, out["act_high"], "action high"),
                            (env.observation_space.low, out["obs_low"], "observation low"), (env.observation_space.high, out["obs_high"], "observation high")):
        assert np.array_equal(np.asarray(mine), ref), f"{what}: {np.asarray(mine)} vs {ref}"
    assert str(np.dtype(env.observation_space.dtype)) == str(out["obs_dtype"]) and str(np.dtype(env.action_space.dtype)) == str(out["act_dtype"]), (env.observation_space.dtype, out["obs_dtype"])
    sharpe = any(isinstance(kw[k], tuple) for k in ("step_reward", "terminal_reward"))
    crash_by_ep = {int(e): i for i, e in enumerate(out["crash_eps"])}
    reset_crash = {int(e) for e in out["reset_crash_eps"]}
    cols = ["asset_price", "inventory", "cash", "aum", "market_spread", "agent_spread", "midprice_offset", "weighted_midprice_offset", "market_order_count", "market_order_total_volume"]
    k = 0
    for ep in range(2):
        env._venv.min_start_timedelta = timedelta(seconds=starts[ep])
        env._venv.max_end_timedelta = timedelta(seconds=starts[ep]) + episode
        if ep in reset_crash:
            try:
                env.reset()
            except Exception:
                continue
            raise AssertionError(f"episode {ep}: the reference's reset raised {out['reset_crash_types']}, the facade's did not")
        obs = env.reset()
        assert np.array_equal(obs, out["reset_obs"][ep], equal_nan=True), f"episode {ep} reset obs\n{obs}\n{out['reset_obs'][ep]}"
        while True:
            if ep in crash_by_ep and k == int(out["crash_ks"][crash_by_ep[ep]]):
                try:
                    env.step(out["crash_actions"][crash_by_ep[ep]])
                except Exception:
                    break
                raise AssertionError(f"step {k}: the reference raised {out['crash_types'][crash_by_ep[ep]]}, the facade did not")
            obs, reward, done, info = env.step(out["actions"][k])
            assert np.array_equal(obs, out["obs"][k], equal_nan=True), f"step {k} (episode {ep}) obs\n{obs}\n{out['obs'][k]}"
            if sharpe:
                assert np.isclose(reward, out["reward"][k], rtol=1e-5, atol=2e-8, equal_nan=True), f"step {k} reward {reward} vs {out['reward'][k]}"
            else:
                assert reward == out["reward"][k] or (np.isnan(reward) and np.isnan(out["reward"][k])), f"step {k} reward {reward!r} vs {out['reward'][k]!r}"
            assert bool(done) == bool(out["done"][k]), f"step {k} done"
            mine = np.array([float(np.asarray(info[c]).reshape(-1)[0]) for c in cols])
            assert np.array_equal(mine, out["info"][k], equal_nan=True), f"step {k} info\n{mine}\n{out['info'][k]}"
            k += 1
            if done:
                break
    assert k == len(out["reward"])
    env.close()
    return dict(day=kind, steps=k, crashes=types + [f"reset:{t}" for t in out.get("reset_crash_types", [])], feats=len(feats))


if __name__ == "__main__":
    first = int(_args[0]) if len(_args) > 0 else 0
    end = int(_args[1]) if len(_args) > 1 else 200
    bad, t_all = 0, time.time()
    for seed in range(first, end):
        t = time.time()
        try:
            info = run(seed)
            print(f"seed {seed:4d} OK   {time.time() - t:5.1f}s {info}", flush=True)
        except AssertionError as ex:
            bad += 1
            print(f"seed {seed:4d} DIFF {time.time() - t:5.1f}s\n    -> {str(ex)[:1500]}", flush=True)
        except Exception:
            bad += 1
            print(f"seed {seed:4d} ERROR {time.time() - t:5.1f}s\n{traceback.format_exc()[-2500:]}", flush=True)
    print(f"{end - first} cases, {bad} with a difference or an error, {time.time() - t_all:.0f} s")
    sys.exit(1 if bad else 0)
