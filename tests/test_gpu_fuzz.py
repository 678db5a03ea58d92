"""Steps that STOP (the reference raises out of `env.step`) and drawn configurations, the HIP path against the oracle.
Needs a real MI355X: `pytest -m gpu`.

`tests/diag/config_fuzz.py` draws env configurations (quote range, concentration, market-order clearing, enter_spread, step size,
feature subsets with random windows and clocks, rewards, outer_levels, ladder width, agent table, level budget) and synthetic days
(steady and drifting) and compares every env at every step.  Its first long run on the GPU found what the named configurations of
the suite had not: a step that raises inside `OrderbookSimulator.forward_step` - an `EmptyOrderbookError` in the middle of a sweep -
leaves `env.step` before `update_internal_state` (HistoricalOrderbookEnvironment.py:180-187, 199-204), so the fills it had produced
never reach the portfolio, and the portfolio is what survives the reset that follows (Q7).  The kernels book fills as they happen
and had kept them.  Longer runs with more dimensions found two more: with `enter_spread` on a CROSSED book (update_outer_levels puts
snapshot levels back without matching them) the highest buy quote sits at or above the best ask and executes, which the lane-parallel
path for the agent's orders did not ask about; and a feature's `current_value` outlives `reset()` (Features.py:88-96) - AmihudLambda
between two of its slow updates does not assign it, so with `normalisation_on` the first steps of a new episode z-score and record the
OLD value.  A fourth: Beta volumes that are NaN in the reference (the step raises) were quietly zero.  The cases below pin them,
and a bounded sample of the drawn configurations runs with the suite."""
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as ol
from rl4mm_amd import _capi, _lib
from rl4mm_amd.stream import SIDE_SHIFT, TYPE_SHIFT, PackedStream

sys.path.insert(0, str(Path(__file__).resolve().parent / "diag"))
import config_fuzz  # noqa: E402
import vector_env_fuzz  # noqa: E402

pytestmark = pytest.mark.gpu


def _thin_stream(bid_vol=100, n_seconds=30, crossed=False):
    """Three levels a side with `bid_vol` shares each and one small external limit order a second far from the touch.  `crossed`: the
    best bid one tick ABOVE the best ask in every snapshot (what a resync of the outer levels can leave behind)."""
    t0, mid, tick, lv = 36_000, 1_000_000, 100, 3
    rows = [((t0 + s) * 1_000_000 + 500_000, 1_000 + s, mid - 3 * tick, 10 | (1 << TYPE_SHIFT) | (0 << SIDE_SHIFT)) for s in range(n_seconds)]
    secs = np.arange(t0, t0 + n_seconds + 1)
    snaps = np.zeros((len(secs), 2 + 4 * lv), dtype=np.int64)
    snaps[:, 0], snaps[:, 1] = secs * 1_000_000, np.arange(len(secs))
    for i in range(lv):
        snaps[:, 2 + 4 * i], snaps[:, 3 + 4 * i] = mid + tick * (i + 1), 400
        snaps[:, 4 + 4 * i], snaps[:, 5 + 4 * i] = mid - tick * (i + 1) + (3 * tick if crossed and i == 0 else 0), bid_vol
    return PackedStream("THIN", "2018-02-21", lv, np.array(rows, dtype=np.int64).reshape(-1, 4), snaps)


def _moc_cfg(n):
    F = _capi.make_feature
    feats = [F(_capi.F_INVENTORY, 0, 100_000, -1e6, 1e6), F(_capi.F_SPREAD, 0, 100_000, 0, 5000), F(_capi.F_PRICE, 0, 100_000, 0, 1e8)]
    return _capi.make_cfg(n, 3, feats, step_us=100_000, episode_us=2_000_000, market_order_clearing=True, market_order_fraction_of_inventory=1.0,
                          inc_prev_action_in_obs=True, max_quote_level=3)  # fmt: skip


@pytest.mark.parametrize("launch", ["step", "rollout"])
def test_a_sweep_that_empties_the_book_leaves_the_portfolio_untouched(launch):
    """inventory 500 against 300 shares of bids, clearing threshold 0: the step's market order sells 300, then finds no bid -
    EmptyOrderbookError (Exchange.py:180-186).  The reference's portfolio still says 500 / the old cash (the exception left `step`
    before `_update_portfolio`), the env is done; after a reset it trades on from 500.  An env next to it whose order the book can
    fill is booked as always."""
    n = 4
    cfg = _moc_cfg(n)
    gpu, cpu = _lib.new_sim(cfg), ol.oracle_sim(cfg)
    inv = np.array([500, 120, -700, 0], dtype=np.int64)  # 500: more than the bids hold; 120: fits; -700: fits the asks (1200); 0: no order
    for s in (gpu, cpu):
        s.add_stream(_thin_stream())
        s.set_portfolio(inv, np.full(n, 10**9, dtype=np.int64))
    starts = np.full(n, 36_002_000_000, dtype=np.int64)
    og, oc = gpu.reset(starts, 0), cpu.reset(starts, 0)
    assert np.array_equal(og, oc)
    act = np.tile(np.array([1.0, 1.0, 1.0, 1.0, 0.0]), (n, 1))  # |inventory| > 0 -> cancel everything, market order for round(|inv| * 1.0)
    if launch == "step":
        og, rg, dg = gpu.step(act)
    else:
        import torch

        dev = torch.device("cuda", 0)
        d_act = torch.from_numpy(act[None].copy()).to(dev)
        d_obs, d_rew, d_done = torch.zeros((n, gpu.obs_dim), dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev)
        gpu.rollout_device(_capi.POLICY_GIVEN, [], 1, d_act.data_ptr(), 0, d_obs.data_ptr(), 0, d_rew.data_ptr(), d_done.data_ptr())
        gpu.sync()
        og, rg, dg = d_obs.cpu().numpy(), d_rew.cpu().numpy(), d_done.cpu().numpy()
    oc, rc, dc = cpu.step(act)
    sg, sc = gpu.status(), cpu.status()
    assert np.array_equal(sg, sc) and sg[0] == _capi.ST_EMPTY_BOOK and not sg[1:].any(), (sg, sc)
    assert np.array_equal(dg, dc) and dg[0] == 1
    ig, cg, _ = gpu.portfolios()
    ic, cc, _ = cpu.portfolios()
    assert np.array_equal(ig, ic) and np.array_equal(cg, cc), (ig, ic, cg, cc)
    assert ig[0] == 500 and cg[0] == 10**9, "the step that raised must not reach the portfolio"
    assert ig[1] == 0 and ig[2] == 0 and cg[1] > 10**9 and cg[2] < 10**9  # the envs whose orders were filled
    assert np.array_equal(og, oc) and np.array_equal(rg, rc) and rg[0] == 0.0
    # the book of the stopped env is what the sweep left (the reference's exchange is mutated up to the exception)
    assert np.array_equal(gpu.queues(0), cpu.queues(0))
    # stepped again it stays done and shows its last row; reset, it carries the 500 shares into the next episode (Q7)
    o2g, r2g, d2g = gpu.step(act)
    o2c, r2c, d2c = cpu.step(act)
    assert d2g[0] == 1 and np.array_equal(d2g, d2c) and np.array_equal(o2g, o2c) and np.array_equal(r2g, r2c)
    ids = np.array([0], dtype=np.int32)
    o3g, o3c = gpu.reset(starts[:1] + 4_000_000, 0, ids), cpu.reset(starts[:1] + 4_000_000, 0, ids)
    assert np.array_equal(o3g, o3c) and o3g[0, 0] == 500.0  # the inventory feature of the new episode
    assert not gpu.status()[0]
    gpu.close()
    cpu.close()


def test_forward_step_that_raises_books_nothing():
    """The same through OrderbookSimulator.forward_step (lobsim_forward): internal market order for more than the side holds."""
    cfg = _moc_cfg(2)
    gpu, cpu = _lib.new_sim(cfg), ol.oracle_sim(cfg)
    for s in (gpu, cpu):
        s.add_stream(_thin_stream())
        s.set_portfolio(7, 1000)
        s.reset_episode(np.full(2, 36_002_000_000, dtype=np.int64), 0)
    orders = np.zeros((2, 1, 6), dtype=np.int64)
    orders[0, 0] = (4, 1, 0, 350, 0, 0)  # {LOBSIM_MARKET, LOBSIM_SELL, price, volume, id, is_external}  # 300 shares of bids
    orders[1, 0] = (4, 1, 0, 250, 0, 0)
    until = np.full(2, 36_002_100_000, dtype=np.int64)
    gpu.forward(until, orders)
    cpu.forward(until, orders)
    assert np.array_equal(gpu.status(), cpu.status()) and gpu.status()[0] == _capi.ST_EMPTY_BOOK and gpu.status()[1] == 0
    ig, cg, _ = gpu.portfolios()
    ic, cc, _ = cpu.portfolios()
    assert np.array_equal(ig, ic) and np.array_equal(cg, cc)
    assert ig[0] == 7 and cg[0] == 1000 and ig[1] == 7 - 250
    gpu.close()
    cpu.close()


def test_enter_spread_quotes_on_a_crossed_book_execute():
    """Best bid 1 000 200 over best ask 1 000 100: the midprice 1 000 150 floors to a buy quote AT the best ask and ceils to a sell
    quote AT the best bid - both execute on submission (Exchange.py:71-83), the quotes behind them rest."""
    n = 3
    F = _capi.make_feature
    feats = [F(_capi.F_INVENTORY, 0, 100_000, -1e6, 1e6), F(_capi.F_PRICE, 0, 100_000, 0, 1e8)]
    cfg = _capi.make_cfg(n, 3, feats, step_us=100_000, episode_us=2_000_000, enter_spread=True, max_quote_level=4)
    gpu, cpu = _lib.new_sim(cfg), ol.oracle_sim(cfg)
    for s in (gpu, cpu):
        s.add_stream(_thin_stream(crossed=True))
        s.set_portfolio(0, 10**9)
    starts = np.full(n, 36_002_000_000, dtype=np.int64)
    assert np.array_equal(gpu.reset(starts, 0), cpu.reset(starts, 0))
    assert gpu.env_state(0)["best_buy"] > gpu.env_state(0)["best_sell"]
    rng = np.random.default_rng(3)
    for t in range(12):
        act = rng.uniform(0.5, 6.0, size=(n, 4))
        og, rg, dg = gpu.step(act)
        oc, rc, dc = cpu.step(act)
        ig, cg, pg = gpu.portfolios()
        ic, cc, pc = cpu.portfolios()
        assert np.array_equal(gpu.status(), cpu.status())
        assert np.array_equal(ig, ic) and np.array_equal(cg, cc) and np.array_equal(pg, pc), (t, ig, ic)
        assert np.array_equal(og, oc) and np.array_equal(rg, rc) and np.array_equal(dg, dc)
        for e in range(n):
            assert np.array_equal(gpu.queues(e), cpu.queues(e)), (t, e)
        if t == 0:
            assert np.abs(ig).max() > 0 or (cg != 10**9).any(), "the first step's quotes at the opposite best must have traded"
    gpu.close()
    cpu.close()


def test_a_feature_value_outlives_the_reset():
    """AmihudLambda with slowing_factor 2 and normalisation_on: `reset()` clears the history and the deques, not current_value; the
    first update of the new episode (an accumulating one: no assignment) clamps, records and z-scores the value the LAST episode
    ended with (Features.py:80-96, 280-290).  Every observation of three episodes, resets included, against the oracle."""
    n = 64
    F = _capi.make_feature
    feats = [F(_capi.F_AMIHUD_LAMBDA, 8, 100_000, 0, 1.0, iparam=2, norm_len=40), F(_capi.F_VOLATILITY, 10, 100_000, 0, 1.0), F(_capi.F_AMIHUD_LAMBDA, 9, 100_000, 0, 1.0, iparam=3)]
    cfg = _capi.make_cfg(n, 10, feats, step_us=100_000, episode_us=2_000_000)
    gpu, cpu = _lib.new_sim(cfg), ol.oracle_sim(cfg)
    stream = ol.golden_stream("synth_small")  # 09:54:50 .. 10:01:50
    rng = np.random.Generator(np.random.PCG64(77))
    starts = (35_720 + rng.integers(0, 200, size=n)) * 1_000_000
    for s in (gpu, cpu):
        s.add_stream(stream)
        s.set_portfolio(0, 10**9)
    assert np.array_equal(gpu.reset(starts, 0), cpu.reset(starts, 0), equal_nan=True)
    n_resets, seen_nonzero = 0, False
    for t in range(66):
        act = rng.uniform(0, 10, size=(n, 4))
        og, rg, dg = gpu.step(act)
        oc, rc, dc = cpu.step(act)
        assert np.array_equal(og, oc, equal_nan=True), (t, np.nanmax(np.abs(og - oc)))
        assert np.array_equal(dg, dc)
        if dg.any():
            ids = np.nonzero(dg)[0].astype(np.int32)
            starts[ids] += 3_000_000
            o1, o2 = gpu.reset(starts[ids], 0, ids), cpu.reset(starts[ids], 0, ids)
            assert np.array_equal(o1, o2, equal_nan=True), (t, np.nanmax(np.abs(o1 - o2)))
            n_resets += len(ids)
            seen_nonzero |= bool((np.nan_to_num(oc[ids, 0]) != 0).any())
    assert n_resets >= 2 * n and seen_nonzero  # episodes ended on a nonzero Amihud value: there was something to carry
    gpu.close()
    cpu.close()


def test_nan_volumes_raise_unless_the_step_clears_the_inventory():
    """Concentration mode with an action at or above the concentration: Beta's second shape parameter c - a + EPS is <= 0, scipy's pdf
    is NaN, np.round(nan).astype(int) the most negative integer at every level - the step raises when `_volume_diff_to_orders` looks for
    orders to cancel at a price that holds none (HistoricalOrderbookEnvironment.py:238-240), unless the inventory is being cleared:
    then the volumes are replaced by zeros before anybody looks at them (:208-209).  The same for a pdf that underflows (shape
    parameters in the hundreds, what an on-device Teradactyl asks for at fifty times its maximum inventory)."""
    n = 6
    F = _capi.make_feature
    feats = [F(_capi.F_INVENTORY, 0, 100_000, -1e6, 1e6), F(_capi.F_PRICE, 0, 100_000, 0, 1e8)]
    stream = ol.golden_stream("synth_small")
    starts = np.full(n, 35_800_000_000, dtype=np.int64)
    for conc, acts, inv, want in (
        (5.0, [[2.0, 3.0, 60.0], [5.0, 3.0, 60.0], [2.0, 7.5, 60.0], [5.0, 3.0, 10.0], [4.999, 4.999, 60.0], [0.0, 0.0, 60.0]], 30, [0, 64, 64, 0, 0, 0]),
        (None, [[2.0, 3.0, 4.0, 5.0, 60.0], [900.0, 900.0, 2.0, 2.0, 60.0], [2.0, 2.0, 1e4, 3.0, 60.0], [900.0, 900.0, 2.0, 2.0, 10.0], [200.0, 200.0, 200.0, 200.0, 60.0], [0.0, 0.0, 0.0, 0.0, 60.0]], 30, [0, 64, 0, 0, 0, 0]),  # (env 2: 0.95 ** 9999 is 1e-223, still a number)
    ):
        cfg = _capi.make_cfg(n, 10, feats, step_us=100_000, episode_us=5_000_000, concentration=conc, market_order_clearing=True, market_order_fraction_of_inventory=0.5)
        gpu, cpu = _lib.new_sim(cfg), ol.oracle_sim(cfg)
        for s_ in (gpu, cpu):
            s_.add_stream(stream)
            s_.set_portfolio(inv, 10**9)  # |inventory| 30: above the clearing threshold of env 3 only
            s_.reset(starts, 0)
        act = np.array(acts)
        og, rg, dg = gpu.step(act)
        oc, rc, dc = cpu.step(act)
        assert np.array_equal(gpu.status(), cpu.status()) and gpu.status().tolist() == want, (conc, gpu.status(), cpu.status())
        assert np.array_equal(dg, dc) and np.array_equal(rg, rc)
        ok = gpu.status() == 0
        assert np.array_equal(og[ok], oc[ok])
        for a, b in zip(gpu.portfolios(), cpu.portfolios()):
            assert np.array_equal(a, b)
        for e in range(n):
            assert np.array_equal(gpu.queues(e), cpu.queues(e)), (conc, e)
        gpu.close()
        cpu.close()


# seeds of tests/diag/config_fuzz.py.  The first eight stop envs in the middle of a sweep (they differed before the portfolio was put
# back), 612 quotes into a crossed book, 204 and 624 carry a normalised AmihudLambda over a reset; the rest is a plain sample.
# Round 6: about a hundred seeds (the drawn ones at 40 envs instead of 64: the sample trebles, the time does not), and the draw has
# three more dimensions - how the step is called (lobsim_step / lobsim_step_host_io with f32 observations / lobsim_step_device with device
# buffers and f32 observations), resets through lobsim_reset_device, a day that is dropped from the device and uploaded again mid-case.
FUZZ_SEEDS = [0, 3, 5, 13, 32, 43, 44, 52, 612, 204, 624] + list(range(1000, 1024))
MORE_FUZZ_SEEDS = list(range(2000, 2066))


@pytest.mark.parametrize("seed", FUZZ_SEEDS)
def test_drawn_configuration_against_oracle(seed):
    err, desc = config_fuzz.run_case(seed, 64)
    assert err is None, f"{err}\n{desc}"


@pytest.mark.parametrize("seed", MORE_FUZZ_SEEDS)
def test_more_drawn_configurations_against_oracle(seed):
    err, desc = config_fuzz.run_case(seed, 40)
    assert err is None, f"{err}\n{desc}"


# tests/diag/vector_env_fuzz.py: drawn batches through the Python boundary - VectorEnv(ticker=[...]) over several tickers and days (a
# weekend and a holiday in the range), a stream budget that evicts while the case runs, arrays / RLlib lists with reset_at / tensors
# with reset_done_tensors, finished envs left to sit out, envs reset in mid-episode - every env against the oracle on what it drew.
VECTOR_ENV_SEEDS = list(range(0, 16))


@pytest.mark.parametrize("seed", VECTOR_ENV_SEEDS)
def test_drawn_vector_env_against_oracle(seed):
    err, desc = vector_env_fuzz.run_case(seed, 32)
    assert err is None, f"{err}\n{desc}"
