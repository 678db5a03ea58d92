"""Replay a golden trajectory (oracle/gen_golden.py format) through any Sim (oracle or HIP) and compare."""
import numpy as np

import oracle_lib as ol


def replay_and_check(sim, g, z, stream_id, obs_rtol=0.0, obs_atol=0.0, check_fills=None, label="", reward_rtol=None):
    """Drives env 0 of `sim` with the golden actions.  Integers must match exactly; floats within
    (rtol, atol) - 0/0 means bit-exact.  `reward_rtol` (RollingSharpe only, whose exp/log are numpy's SIMD
    routines in the reference) relaxes the reward comparison alone."""
    top_k = g["top_k"]
    k = 0
    n_steps = len(z["reward"])
    crash_step = int(z["crash_step"])
    sim.set_portfolio(g["initial_inventory"], g["initial_cash"])
    n_eps = len(g["starts_us"])
    for ep in range(n_eps):
        if k >= n_steps and crash_step < 0:
            break
        obs0 = sim.reset(g["starts_us"][ep], stream_id)
        _close(obs0[0], z["reset_obs"][ep], obs_rtol, obs_atol, f"{label} reset obs ep {ep}")
        while True:
            if k == crash_step:  # the reference raised here: we must flag the env instead
                obs, reward, done = sim.step(z["crash_action"][None, :])
                st = sim.env_state(0)
                assert done[0] == 1 and st["status"] != 0, f"{label}: reference crashed ({z['crash_type']}) at step {k}, status {st['status']}"
                return
            obs, reward, done = sim.step(z["actions"][k][None, :])
            st = sim.env_state(0)
            ctx = f"{label} step {k} (episode {ep})"
            assert st["status"] == 0, f"{ctx}: status {st['status']}"
            assert st["inventory"] == z["inventory"][k], f"{ctx}: inventory {st['inventory']} != {z['inventory'][k]}"
            assert st["cash"] == z["cash"][k], f"{ctx}: cash {st['cash']} != {z['cash'][k]}"
            assert st["now_us"] == z["now_us"][k], ctx
            assert st["n_agent_orders"] == z["n_agent"][k], f"{ctx}: agent orders {st['n_agent_orders']} != {z['n_agent'][k]}"
            assert st["n_resyncs"] == z["n_resyncs"][k], f"{ctx}: resyncs {st['n_resyncs']} != {z['n_resyncs'][k]}"
            fl = z["fills"][z["fill_off"][k] : z["fill_off"][k + 1]]
            agent, ext = fl[fl[:, 0] == 0], fl[fl[:, 0] == 1]
            assert st["agent_fill_count"] == len(agent), f"{ctx}: agent fills {st['agent_fill_count']} != {len(agent)}"
            assert st["agent_buy_volume"] == agent[agent[:, 1] == 0][:, 3].sum(), ctx
            assert st["agent_sell_volume"] == agent[agent[:, 1] == 1][:, 3].sum(), ctx
            assert st["ext_buy_count"] == (ext[:, 1] == 0).sum() and st["ext_sell_count"] == (ext[:, 1] == 1).sum(), f"{ctx}: ext fill counts"
            assert st["ext_buy_volume"] == ext[ext[:, 1] == 0][:, 3].sum() and st["ext_sell_volume"] == ext[ext[:, 1] == 1][:, 3].sum(), f"{ctx}: ext fill volumes"
            if check_fills is not None:
                check_fills(k, fl)
            top = sim.top(0, top_k)
            assert np.array_equal(top, z["top"][k]), f"{ctx}: top of book differs\n{top[:, :4]}\n{z['top'][k][:, :4]}"
            _close(st["price"], z["price"][k], obs_rtol, obs_atol, f"{ctx} price")
            _close(obs[0], z["obs"][k], obs_rtol, obs_atol, f"{ctx} obs")
            if reward_rtol is not None:
                _close(reward[0], z["reward"][k], reward_rtol, 1e-12, f"{ctx} reward")
            else:
                _close(reward[0], z["reward"][k], 0.0, max(obs_atol, 1e-9) if obs_atol or obs_rtol else 0.0, f"{ctx} reward")
            assert bool(done[0]) == bool(z["done"][k]), f"{ctx}: done"
            k += 1
            if done[0]:
                break
        q = sim.queues(0)
        qe = z["queues"][z["queue_off"][ep] : z["queue_off"][ep + 1]]
        assert np.array_equal(q, qe), f"{label}: full queue dump differs at end of episode {ep}"
    assert k == n_steps


def _close(a, b, rtol, atol, what):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if rtol == 0.0 and atol == 0.0:
        assert np.array_equal(a, b, equal_nan=True), f"{what}: {a} != {b} (diff {a - b})"  # a constant normalisation history gives nan in the reference too
    else:
        assert np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True), f"{what}: {a} !~ {b} (diff {a - b})"


def run_engine_case(sim, z, name, L):
    """Returns nothing; asserts book + fills after every op equal the reference's."""
    snap = z[f"{name}/snapshot"]
    row = ol.snapshot_row_from_levels([tuple(r) for r in snap], L) if bool(z[f"{name}/has_snapshot"]) else None
    sim.debug_set_book(0, row, 0)
    sim.set_portfolio(0, 0)
    ops, dumps, doff = z[f"{name}/ops"], z[f"{name}/dumps"], z[f"{name}/dump_off"]
    fills, foff, status = z[f"{name}/fills"], z[f"{name}/fill_off"], z[f"{name}/status"]
    for i in range(len(status)):
        agg = sim.debug_process_orders(0, ops[i : i + 1])
        st = sim.env_state(0)
        if status[i]:
            assert st["status"] & int(status[i]), f"{name} op {i}: expected status {status[i]}, got {st['status']}"
            return
        assert st["status"] == 0, f"{name} op {i} {ops[i]}: status {st['status']}"
        q = sim.queues(0)
        assert np.array_equal(q, dumps[doff[i] : doff[i + 1]]), f"{name} op {i} {ops[i]}:\n{q}\n!=\n{dumps[doff[i]:doff[i + 1]]}"
        f = fills[foff[i] : foff[i + 1]]
        ag, ex = f[f[:, 0] == 0], f[f[:, 0] == 1]
        cash = int(np.sum(np.where(ag[:, 1] == 1, ag[:, 3] * ag[:, 2], -ag[:, 3] * ag[:, 2])))
        exp = [len(ag), ag[ag[:, 1] == 0][:, 3].sum(), ag[ag[:, 1] == 1][:, 3].sum(), (ex[:, 1] == 0).sum(), (ex[:, 1] == 1).sum(),
               ex[ex[:, 1] == 0][:, 3].sum(), ex[ex[:, 1] == 1][:, 3].sum(), cash]  # fmt: skip
        assert list(agg) == [int(x) for x in exp], f"{name} op {i} {ops[i]}: fills {list(agg)} != {exp}"
