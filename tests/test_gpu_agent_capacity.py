"""GPU parity where the agent's order table is pushed to its bound (`-m gpu`).

The reference's internal book is an unbounded SortedDict of deques (rl4mm/orderbook/models.py:64-69): a policy that raises its quotes
in one-share steps leaves one order per share (HistoricalOrderbookEnvironment.py:227-258 submits the DIFFERENCE to the resting
volume as a new order, and cancels from the newest order backwards).  What bounds the number of live orders is the volume a side can
quote, sum(round(100 pdf)) <= 100 + Q/2 shares (OrderDistributors.py:39-56): <= 2 x 105 orders at the default ten quote levels, plus the
<= 20 a step submits before the entries it cancelled are free again.  The handle's table (cfg.agent_orders: 0 = 256) holds that by
construction; the smaller table (128) is an opt-in that reports LOBSIM_ST_AGENT_OVERFLOW when a policy like this one outgrows it.
"""
import numpy as np
import pytest

import bench
import oracle_lib as ol
from rl4mm_amd import _capi, _lib

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-12, 1e-9
T = 200


def creeping_actions(T, n, rng):
    """All of a side's volume at the touch, moved one share at a time towards the far quote levels: Beta(a, b) from (0, 10) to
    (10, 0) over T - 20 steps, per env at its own pace, so that every level gains its volume in single shares = single orders."""
    pace = rng.uniform(0.8, 1.25, size=n)
    f = np.clip(np.arange(T)[:, None] * pace[None, :] / float(T - 20), 0.0, 1.0)
    a, b = 10.0 * f, 10.0 * (1.0 - f)
    return np.stack([a, b, a, b], axis=-1)  # [T, n, 4]


def _pair(cfg, stream, starts):
    sims = []
    for make in (_lib.new_sim, ol.oracle_sim):
        sim = make(cfg)
        sid = sim.add_stream(stream)
        sim.set_portfolio(0, 10**9)
        sims.append((sim, sim.reset(starts, sid)))
    return sims


@pytest.fixture(scope="module")
def stream():
    return bench.build_stream(60)


@pytest.mark.parametrize("features", ["default", "small"])  # default = the kernels specialised for the reference's default configuration
def test_one_share_increments_outgrow_128_entries_and_stay_exact(stream, features):
    n = 48
    F = _capi.make_feature
    feats = bench.default_features(_capi) if features == "default" else [F(_capi.F_SPREAD, 0, 100_000, 0, 1e9), F(_capi.F_INVENTORY, 0, 100_000, -1e9, 1e9)]
    # (a 0.05 s step is not the reference's default configuration: the generic kernels instead of the specialised ones)
    cfg = _capi.make_cfg(n, 10, feats, step_us=100_000 if features == "default" else 50_000, episode_us=600_000_000, inc_prev_action_in_obs=(features == "small"))
    rng = np.random.default_rng(3)
    starts = (36_000 + rng.integers(0, 30, n)) * 1_000_000
    (gpu, o_g), (cpu, o_c) = _pair(cfg, stream, starts)
    assert gpu.limits()["agent_orders"] == 256  # what a handle gets unless it asks for less
    assert np.allclose(o_g, o_c, rtol=RTOL, atol=ATOL)
    acts = creeping_actions(T, n, rng)
    live_max = np.zeros(n, dtype=np.int64)
    for t in range(T):
        og, rg, dg = gpu.step(acts[t])
        oc, rc, dc = cpu.step(acts[t])
        assert np.array_equal(dg, dc) and not dg.any(), t
        assert np.allclose(og, oc, rtol=RTOL, atol=ATOL) and np.allclose(rg, rc, rtol=0, atol=ATOL), t
        if t % 10 == 9 or t == T - 1:
            ig, cg, pg = gpu.portfolios()
            ic, cc, pc = cpu.portfolios()
            assert np.array_equal(ig, ic) and np.array_equal(cg, cc) and np.array_equal(pg, pc), t
            for e in range(0, n, 6):
                sg, sc = gpu.env_state(e), cpu.env_state(e)
                assert sg == sc, (t, e)
                live_max[e] = max(live_max[e], sg["n_agent_orders"])
    hw = ol.oracle_high_water(cpu)[:, 1]
    assert (hw > 128).all() and hw.max() > 190, hw  # every env went beyond the small table, towards the bound of 210
    assert hw.max() <= 2 * 105 + 20
    assert live_max.max() > 128
    assert not gpu.status().any()
    for e in (0, 7, 23):  # the books order by order, the agent's orders by FIFO rank
        assert np.array_equal(gpu.queues(e), cpu.queues(e)), e
    gpu.close(); cpu.close()


def test_one_share_increments_in_multi_step_launches(stream):
    """the same policy through lobsim_rollout_device (pre-drawn actions), 50 steps per launch, against the oracle's rollout"""
    import torch

    n = 64
    cfg = _capi.make_cfg(n, 10, bench.default_features(_capi), step_us=100_000, episode_us=600_000_000)
    rng = np.random.default_rng(4)
    starts = (36_000 + rng.integers(0, 30, n)) * 1_000_000
    (gpu, _), (cpu, _) = _pair(cfg, stream, starts)
    acts = creeping_actions(T, n, rng)
    dev = torch.device("cuda", 0)
    d_act = torch.from_numpy(acts).to(dev)
    obs = torch.zeros((n, gpu.obs_dim), dtype=torch.float64, device=dev)
    rew = torch.zeros(n, dtype=torch.float64, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev)
    ret = torch.zeros(n, dtype=torch.float64, device=dev); steps = torch.zeros(n, dtype=torch.int32, device=dev)
    ret_c = np.zeros(n)
    for c in range(0, T, 50):
        gpu.rollout_device(_capi.POLICY_GIVEN, [], 50, d_act[c:c + 50].contiguous().data_ptr(), 0, obs.data_ptr(), 0, rew.data_ptr(), done.data_ptr(), ret.data_ptr(), steps.data_ptr())
        gpu.sync()
        r = ol.oracle_rollout(cpu, _capi.POLICY_GIVEN, [], 50, acts[c:c + 50], want_actions=False)
        ret_c += r["ret"]
        ig, cg, pg = gpu.portfolios()
        ic, cc, pc = cpu.portfolios()
        assert np.array_equal(ig, ic) and np.array_equal(cg, cc) and np.array_equal(pg, pc), c
        assert np.array_equal(done.cpu().numpy(), r["done"]) and (steps.cpu().numpy() == 50).all()
        assert np.allclose(obs.cpu().numpy(), r["obs"], rtol=RTOL, atol=ATOL) and np.allclose(ret.cpu().numpy(), ret_c, rtol=1e-12, atol=1e-6)
    assert (ol.oracle_high_water(cpu)[:, 1] > 128).all()
    assert not gpu.status().any()
    gpu.close(); cpu.close()


def test_small_table_reports_the_overflow(stream):
    """cfg.agent_orders = 128: equal to the oracle until an env needs the 129th entry, then LOBSIM_ST_AGENT_OVERFLOW on that env,
    done = 1, never a silent difference."""
    n = 32
    F = _capi.make_feature
    cfg = _capi.make_cfg(n, 10, [F(_capi.F_SPREAD, 0, 100_000, 0, 1e9), F(_capi.F_INVENTORY, 0, 100_000, -1e9, 1e9)], step_us=100_000,
                         episode_us=600_000_000, agent_orders=128)
    rng = np.random.default_rng(3)
    starts = (36_000 + rng.integers(0, 30, n)) * 1_000_000
    (gpu, _), (cpu, _) = _pair(cfg, stream, starts)
    assert gpu.limits()["agent_orders"] == 128
    acts = creeping_actions(T, n, rng)
    alive = np.ones(n, dtype=bool)
    for t in range(T):
        og, rg, dg = gpu.step(acts[t], active=alive.astype(np.uint8))
        oc, rc, dc = cpu.step(acts[t], active=alive.astype(np.uint8))
        over = (gpu.status() & _capi.ST_AGENT_OVERFLOW) != 0
        assert np.array_equal(dg[alive & ~over], dc[alive & ~over])
        assert np.allclose(og[alive & ~over], oc[alive & ~over], rtol=RTOL, atol=ATOL), t
        assert dg[alive & over].all()
        alive &= ~over
        if not alive.any():
            break
    assert not alive.any(), "every env of this policy outgrows 128 entries"
    hw = ol.oracle_high_water(cpu)[:, 1]
    assert (hw >= 108).all(), hw  # (live orders + the entries freed in the step that overflowed + that step's new orders > 128)
    gpu.close(); cpu.close()
