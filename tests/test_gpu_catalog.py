"""The batched env over SEVERAL tickers and SEVERAL days on the GPU (`-m gpu`).

The reference env is one ticker and a date range (rl4mm/gym/HistoricalOrderbookEnvironment.py:58,96,136-140; its defaults train over
two weeks: rl4mm/helpers/main_helper.py:117-120,212-214); every reset draws a business day, maps it to the next NASDAQ session and reads
that day from the database.  The batch does the same per env: BASELINE configs[4] is "mixed tickers" in one batch, and the envs of one
batch sit on different days.  Checked here through the Python boundary (HistoricalOrderbookVectorEnv) against the raw C-ABI and the oracle.
"""
import multiprocessing as mp
from datetime import datetime, timedelta

import numpy as np
import pytest

import bench
import oracle_lib as ol
from rl4mm_amd import _capi, _lib, env as E
from rl4mm_amd.synth import synthetic_stream

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-12, 1e-9


def _deep_stream(i):
    return synthetic_stream(20180221 + i, 50, bench.DEEP50_RATES[i], 9.5 * 3600 + 1490, 9.5 * 3600 + 1800 + 60 + 40, mid_price=2_700_000 - 350_000 * i, ticker=f"SY{i}", book_snapshot_freq="100ms")


def test_four_ticker_8192_env_vector_env_equals_the_raw_handle():
    """configs[4]'s shard through the Python boundary: HistoricalOrderbookVectorEnv(ticker=[four tickers]) - env i on ticker i mod 4,
    50-level books, rate-ordered launch (the handle holds more envs than the chip and streams of different message rates) - against a
    raw `_capi.Sim` driven with the same stream ids, start times and actions: every observation, reward and done flag of 300 steps
    bit for bit, portfolios at the end; the tensor path on a twin."""
    import torch

    n, steps = 8192, 300
    with mp.get_context("spawn").Pool(4) as pool:
        streams = pool.map(_deep_stream, range(4))
    cat = E.StreamCatalog()
    for s in streams:
        cat.add(s)
    day = datetime(2018, 2, 21)
    kw = dict(ticker=[s.ticker for s in streams], n_levels=50, min_date=day, max_date=day, min_start_timedelta=timedelta(hours=10), max_end_timedelta=timedelta(hours=10, minutes=31),
              episode_length=timedelta(minutes=30), initial_portfolio=E.Portfolio(inventory=0, cash=10**12), per_step_reward_function=E.PnL(), terminal_reward_function=E.PnL())  # fmt: skip
    v = E.HistoricalOrderbookVectorEnv(n, cat, **kw)
    np.random.seed(21)
    obs0 = v.vector_reset_arrays()
    ids, keys, starts = v.last_draw
    assert [k[0] for k in keys[:8]] == ["SY0", "SY1", "SY2", "SY3"] * 2 and len(set(keys)) == 4
    raw = _lib.new_sim(bench.workload_cfg(_capi, "deep50", n))
    sid_raw = {s.ticker: raw.add_stream(s) for s in streams}
    raw.set_portfolio(0, 10**12)
    o_raw = raw.reset(starts, np.array([sid_raw[k[0]] for k in keys], dtype=np.int32))
    assert np.array_equal(obs0, o_raw)
    twin = E.HistoricalOrderbookVectorEnv(n, cat, obs_dtype=np.float32, **kw)
    np.random.seed(21)
    ot = twin.vector_reset_tensors()
    assert np.array_equal(ot.cpu().numpy(), obs0.astype(np.float32))
    rng = np.random.Generator(np.random.PCG64(8))
    dev = torch.device("cuda", 0)
    for t in range(steps):
        a = rng.uniform(0, 10, size=(n, 4))
        o, r, d, _ = v.vector_step_arrays(a, want_infos=False, copy=False)
        o2, r2, d2 = raw.step(a)
        assert np.array_equal(o, o2) and np.array_equal(r, r2) and np.array_equal(d, d2.astype(bool)) and not d.any(), t
        o3, r3, d3 = twin.vector_step_tensors(torch.from_numpy(a).to(dev))
        if t % 50 == 49:
            torch.cuda.synchronize()
            assert np.array_equal(o3.cpu().numpy(), o.astype(np.float32)) and np.array_equal(r3.cpu().numpy(), r)
    for x, y in zip(v.portfolios(), raw.portfolios()):
        assert np.array_equal(x, y)
    for x, y in zip(v.portfolios(), twin.portfolios()):
        assert np.array_equal(x, y)
    assert not v.status.any()
    v.close(); twin.close(); raw.close()


def _day_stream(args):
    ticker, date, seed, lam = args
    return synthetic_stream(seed, 10, lam, 9.5 * 3600 + 1700, 9.5 * 3600 + 1800 + 100, mid_price=2_700_000 - 1000 * (seed % 7), ticker=ticker, trading_date=date)


def _catalog(dates, tickers=("SYN",)):
    jobs = [(tk, d, 100 + 17 * i + 1000 * j, 150.0 + 25 * j) for j, tk in enumerate(tickers) for i, d in enumerate(dates)]
    with mp.get_context("spawn").Pool(4) as pool:
        streams = pool.map(_day_stream, jobs)
    cat = E.StreamCatalog()
    for s in streams:
        cat.add(s)
    return cat, {(s.ticker, s.trading_date): s for s in streams}


def _features(episode):
    return [E.Spread(), E.PriceMove(name="pm", update_frequency=timedelta(seconds=0.1), lookback_periods=5), E.Volatility(name="v", update_frequency=timedelta(seconds=0.1), lookback_periods=30),
            E.Inventory(), E.EpisodeProportion(episode_length=episode), E.TradeDirectionImbalance(update_frequency=timedelta(seconds=0.1), lookback_periods=30)]  # fmt: skip


def _mirror(v, cfg_env, streams, n):
    """an oracle simulator holding every stream of the catalog, and the map (ticker, day) -> its stream id"""
    cpu = ol.oracle_sim(cfg_env)
    sid = {k: cpu.add_stream(s) for k, s in streams.items()}
    cpu.set_portfolio(0, 10**9)
    return cpu, sid


def _run_against_oracle(v, cpu, sid, n, rounds, steps_per_round, rng):
    """vector_reset + steps, then rounds of: step until the episode ends, reset the finished envs (new days drawn) - every env against
    the oracle on the (ticker, day, start) the env drew"""
    np.random.seed(5)
    obs = v.vector_reset_arrays()
    ids, keys, starts = v.last_draw
    oc = cpu.reset(starts, np.array([sid[k] for k in keys], dtype=np.int32))
    assert np.allclose(obs, oc, rtol=RTOL, atol=ATOL)
    seen = set(keys)
    for rnd in range(rounds):
        for t in range(steps_per_round):
            a = rng.uniform(0, 10, size=(n, 4))
            o, r, d, _ = v.vector_step_arrays(a, want_infos=False)
            o2, r2, d2 = cpu.step(a)
            assert np.array_equal(d, d2.astype(bool)), (rnd, t)
            assert np.allclose(o, o2, rtol=RTOL, atol=ATOL) and np.allclose(r, r2, rtol=0, atol=ATOL), (rnd, t)
            if d.any():
                assert d.all()  # (every env of the batch has the same episode length)
                for x, y in zip(v.portfolios(), cpu.portfolios()):
                    assert np.array_equal(x, y)
                fin = np.flatnonzero(d).astype(np.int32)
                o_new = v.reset_many(fin)
                ids, keys, starts = v.last_draw
                assert np.array_equal(ids, fin)
                oc = cpu.reset(starts, np.array([sid[k] for k in keys], dtype=np.int32), env_ids=fin)
                assert np.allclose(o_new, oc, rtol=RTOL, atol=ATOL)
                seen |= set(keys)
                break
    assert not v.status.any()
    return seen


def test_three_day_catalog_with_a_weekend_envs_of_one_batch_on_different_days():
    """min_date .. max_date = Friday 2018-02-16 .. Tuesday 2018-02-20, and Monday the 19th is Presidents' Day: the business days drawn
    are Fri / Mon / Tue, Monday maps to Tuesday's session (rl4mm/utils/utils.py:54-57, rl4mm_amd/trading_calendar.py).  256 envs of
    one batch start on different days; two tickers; several episodes with re-draws - every env against the oracle."""
    n = 256
    episode = timedelta(seconds=3)
    cat, streams = _catalog(["2018-02-16", "2018-02-20"], tickers=("SYA", "SYB"))
    v = E.HistoricalOrderbookVectorEnv(
        n, cat, features=_features(episode), ticker=["SYA", "SYB"], n_levels=10, min_date=datetime(2018, 2, 16), max_date=datetime(2018, 2, 20),
        min_start_timedelta=timedelta(hours=9, minutes=58, seconds=30), max_end_timedelta=timedelta(hours=10, minutes=1), episode_length=episode,
        initial_portfolio=E.Portfolio(inventory=0, cash=10**9), per_step_reward_function=E.PnL(), terminal_reward_function=E.PnL())  # fmt: skip
    assert v._day_lookup() == ["2018-02-16", "2018-02-20", "2018-02-20"]  # Friday, the holiday Monday -> Tuesday, Tuesday
    cpu, sid = _mirror(v, v.sim.cfg, streams, n)
    seen = _run_against_oracle(v, cpu, sid, n, rounds=4, steps_per_round=40, rng=np.random.Generator(np.random.PCG64(3)))
    assert seen == set(streams)  # both days of both tickers were drawn
    ids, keys, _ = v.last_draw
    assert len({k[1] for k in keys}) == 2 and {k[0] for k in keys} == {"SYA", "SYB"}
    assert all(k[0] == ("SYA", "SYB")[i % 2] for i, k in zip(ids, keys))
    v.close(); cpu.close()


def test_ten_day_catalog_under_a_three_day_stream_budget():
    """stream_budget_mb: a day goes to the device when a start draw first selects it and the least recently drawn days no env replays
    leave it again (lobsim_drop_stream; their ids are reused).  Ten days, room for about three: 96 envs x 8 episodes against the oracle,
    which holds all ten; streams were dropped and uploaded again along the way, never one an env was replaying."""
    n = 96
    episode = timedelta(seconds=2)
    dates = ["2018-02-05", "2018-02-06", "2018-02-07", "2018-02-08", "2018-02-09", "2018-02-12", "2018-02-13", "2018-02-14", "2018-02-15", "2018-02-16"]
    cat, streams = _catalog(dates)
    one = max(s.msgs.nbytes + s.snaps.nbytes for s in streams.values())
    v = E.HistoricalOrderbookVectorEnv(
        n, cat, features=_features(episode), ticker="SYN", n_levels=10, min_date=datetime(2018, 2, 5), max_date=datetime(2018, 2, 16),
        min_start_timedelta=timedelta(hours=9, minutes=58, seconds=30), max_end_timedelta=timedelta(hours=10, minutes=1), episode_length=episode,
        initial_portfolio=E.Portfolio(inventory=0, cash=10**9), per_step_reward_function=E.PnL(), terminal_reward_function=E.PnL(),
        stream_budget_mb=3.2 * one / 2**20)  # fmt: skip
    cpu, sid = _mirror(v, v.sim.cfg, streams, n)
    res = v._residency
    assert len(res.sid) == 0  # nothing is uploaded before a draw names it

    seen = _run_against_oracle(v, cpu, sid, n, rounds=8, steps_per_round=30, rng=np.random.Generator(np.random.PCG64(4)))
    assert len(seen) == 10
    # 96 envs over ten days hold all ten days at once: nothing can be dropped while they do - the budget is exceeded rather than a
    # stream pulled from under an env
    assert res.drops == 0 and len(res.sid) == 10
    # a batch that sits on few days: reset every env onto three days (a narrower date range on the same handle), the other seven leave
    v.min_date, v.max_date, v._day_table = datetime(2018, 2, 7), datetime(2018, 2, 9), None
    np.random.seed(6)
    obs = v.vector_reset_arrays()
    ids, keys, starts = v.last_draw
    assert {k[1] for k in keys} == {"2018-02-07", "2018-02-08", "2018-02-09"}
    assert sorted(k[1] for k in res.sid) == ["2018-02-07", "2018-02-08", "2018-02-09"] and res.drops == 7 and res.resident_bytes <= 3.2 * one
    oc = cpu.reset(starts, np.array([sid[k] for k in keys], dtype=np.int32))
    assert np.allclose(obs, oc, rtol=RTOL, atol=ATOL)
    # ... and on to three other days: uploads again, into the slots the dropped streams gave back
    uploads = res.uploads
    v.min_date, v.max_date, v._day_table = datetime(2018, 2, 13), datetime(2018, 2, 15), None
    obs = v.vector_reset_arrays()
    ids, keys, starts = v.last_draw
    assert res.uploads == uploads + 3 and sorted(k[1] for k in res.sid) == ["2018-02-13", "2018-02-14", "2018-02-15"] and max(res.sid.values()) <= 5
    oc = cpu.reset(starts, np.array([sid[k] for k in keys], dtype=np.int32))
    assert np.allclose(obs, oc, rtol=RTOL, atol=ATOL)
    rng = np.random.Generator(np.random.PCG64(9))
    for t in range(15):
        a = rng.uniform(0, 10, size=(n, 4))
        o, r, d, _ = v.vector_step_arrays(a, want_infos=False)
        o2, r2, d2 = cpu.step(a)
        assert np.array_equal(d, d2.astype(bool)) and np.allclose(o, o2, rtol=RTOL, atol=ATOL) and np.allclose(r, r2, rtol=0, atol=ATOL), t
    with pytest.raises(_capi.LobsimError):
        v.sim.drop_stream(res.sid[keys[0]])  # an env is replaying it
    assert not v.status.any()
    v.close(); cpu.close()


def test_env_creator_forwards_device_and_sizing_keys():
    """register_env's entry point (rl4mm/gym/utils.py:42-87): the env_config keys of the reference plus the ones that place and size the
    envs on the GPU reach the handle; the device defaults from the rollout worker's index."""
    s = ol.golden_stream("synth_small")
    cfg = dict(ticker=s.ticker, min_date=s.trading_date, max_date=s.trading_date, step_size=0.1, episode_length=1 / 30, n_levels=s.n_levels, features="agent_state",
               max_inventory=1000, normalisation_on=False, initial_cash=10**9, initial_inventory=0, per_step_reward_function="PnL", terminal_reward_function="PnL",
               market_order_clearing=False, market_order_fraction_of_inventory=0.0, min_quote_level=0, max_quote_level=10, enter_spread=False, concentration=None,
               info_calculator=None, catalog=E.StreamCatalog().add(s), num_envs=32, obs_dtype="float32", max_levels=128, agent_orders=128, stream_budget_mb=64)  # fmt: skip

    class Ctx(dict):  # RLlib's EnvContext: a dict with the worker's index
        worker_index = 3

    v = E.env_creator(Ctx(cfg))
    assert isinstance(v, E.HistoricalOrderbookVectorEnv) and v.num_envs == 32
    assert v.obs_dtype == np.float32 and v.limits["level_slabs"] == 256 and v.limits["agent_orders"] == 128
    import torch

    assert v.sim.cfg.device_id == (3 - 1) % torch.cuda.device_count()
    assert v._residency.budget == 64 * 2**20
    v.min_start_timedelta, v.max_end_timedelta = timedelta(hours=9, minutes=56), timedelta(hours=10, minutes=1)
    obs = v.vector_reset_arrays()
    assert obs.dtype == np.float32 and obs.shape == (32, 1)
    v.close()
    single = E.env_creator(dict(cfg, num_envs=1, device_id=0))
    assert isinstance(single, E.HistoricalOrderbookEnvironment) and single._venv.sim.cfg.device_id == 0
    single.close()


def test_default_device_id_with_the_real_device_count():
    """`default_device_id` against the device count of THIS box (the CPU test monkeypatches it): every rollout worker index, the local
    worker and a torchrun rank map to a device that exists - on a one-GPU box all of them to device 0 - and a handle opens there.  The info
    calculator is a required env_config key, as in the reference (rl4mm/gym/utils.py:81)."""
    import os

    import torch

    n_dev = torch.cuda.device_count()
    assert n_dev >= 1
    for w in range(0, 17):
        d = E.default_device_id({"worker_index": w})
        assert 0 <= d < n_dev and (n_dev > 1 or d == 0), (w, d)
        assert d == (0 if w == 0 else (w - 1) % n_dev)
    old = os.environ.get("LOCAL_RANK")
    try:
        for r in range(0, 9):
            os.environ["LOCAL_RANK"] = str(r)
            assert E.default_device_id({}) == r % n_dev
    finally:
        if old is None:
            os.environ.pop("LOCAL_RANK", None)
        else:
            os.environ["LOCAL_RANK"] = old
    s = ol.golden_stream("synth_small")
    cfg = dict(ticker=s.ticker, min_date=s.trading_date, max_date=s.trading_date, step_size=0.1, episode_length=1 / 30, n_levels=s.n_levels, features="agent_state",
               max_inventory=1000, normalisation_on=False, initial_cash=10**9, initial_inventory=0, per_step_reward_function="PnL", terminal_reward_function="PnL",
               market_order_clearing=False, market_order_fraction_of_inventory=0.0, min_quote_level=0, max_quote_level=10, enter_spread=False, concentration=None,
               info_calculator=None, catalog=E.StreamCatalog().add(s), num_envs=8, worker_index=16)  # fmt: skip
    v = E.env_creator(cfg)
    assert v.sim.cfg.device_id == 15 % n_dev
    v.close()
    with pytest.raises(KeyError):
        E.env_creator({k: x for k, x in cfg.items() if k != "info_calculator"})
