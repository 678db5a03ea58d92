"""CPU-only tests of the host side: packer vs the reference's DB ordering (golden), synthetic generator
determinism, start-time draw, agents, env sharding + the world_size-2 gloo gather."""
import os
import socket
from datetime import datetime, timedelta

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import oracle_lib as ol
from rl4mm_amd import agents, dist as rdist
from rl4mm_amd.stream import PackedStream, pack_lobster, unpack_meta
from rl4mm_amd.synth import generate_lobster, synthetic_stream


def test_packer_reproduces_db_order_quirks():
    """Q1-Q3 on the reference's MSFT slice: floor-to-us timestamps, lexicographic row-index tie-break, hidden
    executions dropped, executions name the aggressor."""
    ps = ol.golden_stream("msft")
    assert ps.n_msgs == 995 and ps.n_snaps == 1000  # 5 hidden executions dropped; one snapshot per row (freq None)
    t = ps.msgs[:, 0]
    assert np.all(np.diff(t) >= 0)
    raw = np.loadtxt(ol.GOLD / "msft_message_50.csv", delimiter=",")
    # rows 9 and 10..29 share a microsecond in the slice: DB order puts row 9 after row 29 ("10" < ... < "29" < "9")
    vol, typ, side = unpack_meta(ps.msgs[:, 3])
    assert set(np.unique(typ)) <= {1, 2, 3, 4}
    ex = raw[raw[:, 1] == 4]
    packed_ex = ps.msgs[typ == 4]
    assert len(ex) == len(packed_ex)
    # a LOBSTER execution of a resting SELL (-1) is a BUY aggressor (side 0)
    for row in ex[:20]:
        m = packed_ex[(packed_ex[:, 1] == int(row[2])) & ((packed_ex[:, 3] & ((1 << 48) - 1)) == int(row[3]))]
        assert len(m) >= 1 and all(((mm[3] >> 52) & 1) == (0 if row[5] == -1 else 1) for mm in m)
    # S-frequency snapshots: exactly the reference's 4 rows
    ps_s = pack_lobster(ol.GOLD / "msft_message_50.csv", ol.GOLD / "msft_orderbook_50.csv.gz", "MSFT", "2012-06-21", 50, "S", 1000, 1000)
    assert ps_s.n_snaps == 4 and np.array_equal(ps_s.msgs, ps.msgs)
    assert ps_s.last_snapshot_at(36_000_000_000) == 0 and ps_s.last_snapshot_at(35_999_000_000) == -1


def test_synthetic_generator_is_deterministic_and_matches_golden_stream():
    a = synthetic_stream(20180221, 10, 200.0, 9.5 * 3600 + 1490, 9.5 * 3600 + 1490 + 420)
    g = ol.golden_stream("synth_small")  # generated when the goldens were made, via CSV files the reference ingested
    assert np.array_equal(a.msgs, g.msgs) and np.array_equal(a.snaps, g.snaps)
    raw = generate_lobster(3, 10, 100.0, 34200, 34260)
    assert np.all(np.diff(raw["time_ns"]) >= 0) and set(np.unique(raw["type"])) <= {1, 2, 3, 4}
    assert np.all(raw["books"][:, 0] > raw["books"][:, 2])  # ask above bid on every row


def test_stream_roundtrip(tmp_path):
    g = ol.golden_stream("synth_small")
    g.save(tmp_path / "s.npz")
    h = PackedStream.load(tmp_path / "s.npz")
    assert np.array_equal(g.msgs, h.msgs) and h.ticker == g.ticker and h.n_levels == 10
    assert g.first_msg_after(int(g.msgs[10, 0])) > 10


def test_teradactyl_matches_reference_actions():
    """G4 pins the agent: the actions the reference's Teradactyl took on the observations it saw."""
    for name, idx, kw in (("msft_teradactyl_ad", 0, dict(max_inventory=100, default_kappa=7.0, max_kappa=9.0)), ("synth_teradactyl_ad_spread", 3, dict(max_inventory=300))):
        g, z = ol.load_traj(name)
        agent = agents.Teradactyl(inventory_index=idx, **kw)
        obs = np.concatenate([z["reset_obs"][:1], z["obs"][:-1]])
        assert np.allclose(agent.get_action(obs), z["actions"], rtol=1e-13, atol=0)
        assert np.allclose(agent.get_action(obs[3]), z["actions"][3], rtol=1e-13, atol=0)


def test_start_time_draw_follows_reference_rng_calls():
    from rl4mm_amd.env import HistoricalOrderbookVectorEnv as V

    env = V.__new__(V)
    env.min_date = env.max_date = datetime(2018, 2, 21)
    env.min_start_timedelta, env.max_end_timedelta = timedelta(hours=10), timedelta(hours=15, minutes=30)
    env.episode_length, env.step_size = timedelta(minutes=30), timedelta(seconds=0.1)
    np.random.seed(7)
    t = env._get_random_start_time()
    np.random.seed(7)
    np.random.choice(np.arange(1))  # trading day draw
    steps = np.random.randint(low=0, high=int((env.max_end_timedelta - env.episode_length - env.min_start_timedelta) / env.step_size))
    exp = timedelta(hours=10) + steps * env.step_size
    exp -= timedelta(microseconds=exp.microseconds)
    assert t == datetime(2018, 2, 21) + exp and t.microsecond == 0


def test_shard_range_partitions():
    for n, w in ((32768, 8), (4096, 1), (10, 3), (7, 8)):
        spans = [rdist.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def _gather_worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = rdist.shard_range(n_total, rank, world)
    local = torch.arange(lo, hi, dtype=torch.float64) * 0.5  # stand-in for per-env episode returns
    out = rdist.gather_episode_returns(local)
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_episode_return_gather_world_size_2_gloo():
    """The N>1 path on CPU: two processes, contiguous env shards (unequal sizes), the one collective of the design."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_total = 11
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        assert np.array_equal(res[r], np.arange(n_total) * 0.5)


@pytest.mark.parametrize("world", [2, 8])
def test_bench_gpus_flag_launches_ranks_gloo_stub(world):
    """`python bench.py --gpus N` (no torchrun around it) must itself start N ranks: the launcher path of bench.py with
    the CPU stand-in step (gloo) through the SAME harness as the GPU run - the stream generated once by rank 0 and handed to
    the others, five timed regions each inside its own barrier pair, the region time the MAX over ranks, the reported time
    the median region.  Rank 0 prints one JSON line with n_gpus = N and the gathered return vector of all ranks in env order."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", str(world), "--stub", "--steps", "3", "--envs", "37", "--repeats", "5"], env=env,
                       capture_output=True, text=True, timeout=600)  # fmt: skip
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["gathered_returns"] == 37 * world and out["gather_in_env_order"] and len(out["ms_per_step_per_rank"]) == world
    assert out["repeats"] == 5 and len(out["ms_per_step_regions"]) == 5 and out["stream_msgs"] > 0
    assert out["ms_per_step_min"] <= out["ms_per_step"] <= out["ms_per_step_max"]
    assert out["ms_per_step"] == pytest.approx(sorted(out["ms_per_step_regions"])[2], abs=1e-4)


def test_nasdaq_trading_calendar():
    """rl4mm/utils/utils.py:48-62 without pandas_market_calendars: the exchange's holiday rules restated.  Checked against
    known closures, against pandas' federal-holiday rules where the two coincide, and through the env's day draw."""
    from datetime import date

    import pandas as pd
    from pandas.tseries.holiday import USFederalHolidayCalendar

    from rl4mm_amd import trading_calendar as tc

    closed = [date(2012, 1, 2), date(2012, 1, 16), date(2012, 2, 20), date(2012, 4, 6), date(2012, 5, 28), date(2012, 7, 4), date(2012, 9, 3), date(2012, 10, 29),
              date(2012, 10, 30), date(2012, 11, 22), date(2012, 12, 25), date(2018, 1, 1), date(2018, 1, 15), date(2018, 2, 19), date(2018, 3, 30), date(2018, 5, 28),
              date(2018, 7, 4), date(2018, 9, 3), date(2018, 11, 22), date(2018, 12, 5), date(2018, 12, 25), date(2022, 6, 20), date(2021, 12, 24), date(2023, 1, 2),
              date(2015, 7, 3), date(2019, 4, 19), date(2024, 3, 29)]  # fmt: skip
    for d in closed:
        assert not tc.is_trading_day(d), d
    for d in (date(2012, 6, 21), date(2018, 2, 21), date(2021, 12, 31), date(2010, 12, 31), date(2021, 6, 18), date(2018, 11, 23), date(2012, 10, 31), date(2011, 11, 11)):
        assert tc.is_trading_day(d), d  # (New Year's Day on a Saturday is not observed; Veterans / Columbus Day are sessions)
    # rules shared with the federal calendar: same dates apart from Columbus Day, Veterans Day and the Saturday rule of 1 January
    fed = USFederalHolidayCalendar().holidays(start="2005-01-01", end="2030-12-31")
    for ts in fed:
        d = ts.date()
        if d.month in (10,) or (d.month == 11 and d.day in (10, 11, 12)) or (d.month == 12 and d.day == 31) or (d.month == 6 and d.year < 2022):
            continue
        assert not tc.is_trading_day(d), d
    for y in range(2005, 2031):
        n = sum(tc.is_trading_day(ts.date()) for ts in pd.bdate_range(f"{y}-01-01", f"{y}-12-31"))
        assert 249 <= n <= 253, (y, n)
    assert tc.get_next_trading_dt(datetime(2018, 3, 30)) == datetime(2018, 4, 2, 9, 30)
    assert tc.get_last_trading_dt(datetime(2018, 4, 1)) == datetime(2018, 3, 29, 16)
    assert [d.day for d in tc.get_trading_datetimes(datetime(2018, 12, 24), datetime(2018, 12, 27))] == [24, 26, 27]


def test_ingest_archives_and_multi_day_catalog(tmp_path):
    """rl4mm_amd/ingest.py against run_populate_database_from_zipped.py:50-109 / populate_database.py:38-95: archives named like
    LOBSTER's monthly downloads are taken in date order (not in download-counter order), their trading days (NASDAQ
    sessions only) are packed, the extracted CSVs are deleted again, a second run re-adds nothing, the cache serves repeats."""
    import zipfile

    from rl4mm_amd import ingest
    from rl4mm_amd.env import StreamCatalog
    from rl4mm_amd.synth import generate_lobster, write_lobster_csv

    assert ingest.parse_archive_name("_data_dwn_50_385__KO_2018-04-01_2018-04-30_200.7z") == ("KO", datetime(2018, 4, 1), datetime(2018, 4, 30), 200)
    days = {"_data_dwn_9_385__SYN_2018-03-29_2018-04-02_10.zip": ["2018-03-29", "2018-04-02"], "_data_dwn_9_390__SYN_2018-02-20_2018-02-21_10.zip": ["2018-02-20", "2018-02-21"]}
    packed = {}
    for k, (arc, dates) in enumerate(days.items()):
        with zipfile.ZipFile(tmp_path / arc, "w") as z:
            for j, d in enumerate(dates):
                g = generate_lobster(100 * k + j, 10, 150.0, 9.5 * 3600, 9.5 * 3600 + 20)
                sub = tmp_path / f"raw{k}{j}"
                mp_, bp_ = write_lobster_csv(g, sub, "SYN", d, 10)
                z.write(mp_, mp_.name); z.write(bp_, bp_.name)
                packed[d] = pack_lobster(mp_, bp_, "SYN", d, 10, "S")
    cat = ingest.populate_catalog_from_archives(StreamCatalog(), tmp_path, "S", cache_dir=tmp_path / "cache")
    keys = [k for k, _ in cat.items()]
    assert keys == [("SYN", "2018-02-20"), ("SYN", "2018-02-21"), ("SYN", "2018-03-29"), ("SYN", "2018-04-02")]  # date order; 30 March 2018 (Good Friday) and the weekend are no sessions
    for (_, d), ps in cat.items():
        assert np.array_equal(ps.msgs, packed[d].msgs) and np.array_equal(ps.snaps, packed[d].snaps)
    assert not list(tmp_path.glob("*.csv")) and len(list((tmp_path / "cache").glob("*.npz"))) == 4
    n = len(cat)
    ingest.populate_catalog_from_archives(cat, tmp_path, "S", cache_dir=tmp_path / "cache")
    assert len(cat) == n
    cat2 = ingest.populate_catalog_from_archives(StreamCatalog(), tmp_path, "S", cache_dir=tmp_path / "cache")  # served by the cache
    assert [k for k, _ in cat2.items()] == keys


def test_id_suffix_order_equals_string_sort():
    """Q2 (ORDER BY timestamp, id with a string id): the integer restatement against an actual string sort."""
    from rl4mm_amd.stream import _id_suffix_order

    rng = np.random.default_rng(3)
    for n, batch in ((5000, 1000), (30000, 1_000_000)):
        t = np.sort(rng.integers(0, n // 4, size=n))
        ids = np.arange(n) + (np.arange(n) // batch) * batch
        ref = np.lexsort((np.array([str(int(i)) for i in ids]), t))
        assert np.array_equal(_id_suffix_order(t, ids), ref)


def _bare_vector_env(min_date, max_date, episode_minutes, step_seconds, n_envs=512, tickers=("SYN",)):
    """A HistoricalOrderbookVectorEnv without a device handle: enough of it for the start-time draw."""
    import pandas as pd

    from rl4mm_amd.env import HistoricalOrderbookVectorEnv as V
    from rl4mm_amd.trading_calendar import get_next_trading_dt

    class AnyDayResidency:  # every (ticker, day) exists; ids in order of first use
        def __init__(self):
            self.sid = {}

        def has(self, key):
            return True

        def ensure(self, keys):
            for k in keys:
                self.sid.setdefault(k, len(self.sid))
            return self.sid

        def trim(self, in_use):
            pass

    env = V.__new__(V)
    env.ticker, env.tickers, env._day_table = "SYN", list(tickers), None
    env.min_date, env.max_date = min_date, max_date
    env.min_start_timedelta, env.max_end_timedelta = timedelta(hours=10), timedelta(hours=15, minutes=30)
    env.episode_length, env.step_size = timedelta(minutes=episode_minutes), timedelta(seconds=step_seconds)
    env._residency, env._env_key, env._env_ticker = AnyDayResidency(), [None] * n_envs, np.arange(n_envs) % len(tickers)
    return env


@pytest.mark.parametrize("case", [(datetime(2018, 2, 21), datetime(2018, 2, 21), 30, 0.1), (datetime(2018, 2, 1), datetime(2018, 3, 15), 30, 0.1),
                                  (datetime(2018, 2, 1), datetime(2018, 2, 2), 330, 0.1), (datetime(2018, 2, 1), datetime(2018, 2, 9), 329, 60.0),
                                  (datetime(2018, 3, 26), datetime(2018, 4, 3), 30, 1.0)])  # fmt: skip
def test_batched_start_draw_equals_per_env_reference_calls(case):
    """vector_reset draws every env's (day, start second) in one pass; values AND the global np.random state afterwards must be
    what n calls of the reference's _get_random_start_time (HistoricalOrderbookEnvironment.py:196,333-351) leave behind - one
    day (no word drawn), many days (rejection sampling on both draws), an episode that fills the window (randint raises ->
    0), a window of exactly one step, a range with a holiday (Good Friday 2018 maps to the Monday after)."""
    env = _bare_vector_env(*case)
    for seed in (1, 7, 123):
        np.random.seed(seed)
        slow = env._draw_slow(257)
        after_slow = np.random.randint(0, 1 << 30)
        np.random.seed(seed)
        fast = env._draw_days(257)
        assert np.array_equal(slow[0], fast[0]) and np.array_equal(slow[1], fast[1])
        assert np.random.randint(0, 1 << 30) == after_slow
    assert (fast[1] % 1_000_000 == 0).all()


def test_env_classes_are_gym_env_and_rllib_vector_env_when_those_import(monkeypatch):
    """RLlib's env check accepts gym.Env / VectorEnv instances only (the reference's class is a gym.Env,
    HistoricalOrderbookEnvironment.py:51; its env_creator sets env.spec, rl4mm/gym/utils.py:83-86).  gym and ray are not in
    this image: stub modules with the surface the package touches are injected, the package re-imported."""
    import importlib
    import sys
    import types

    gym = types.ModuleType("gym")
    gym.Env = type("Env", (), {"spec": None, "metadata": {}})
    spaces = types.ModuleType("gym.spaces")

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            shape = np.shape(low) if shape is None else shape
            self.low, self.high = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy(), np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()
            self.shape, self.dtype = tuple(shape), np.dtype(dtype)

    spaces.Box = Box
    gym.spaces = spaces
    envs, reg = types.ModuleType("gym.envs"), types.ModuleType("gym.envs.registration")

    class EnvSpec:
        def __init__(self, id, max_episode_steps=None):  # noqa: A002
            self.id, self.max_episode_steps = id, max_episode_steps

    reg.EnvSpec = EnvSpec
    envs.registration = reg
    gym.envs = envs
    ray, rllib, renv, vec = (types.ModuleType(n) for n in ("ray", "ray.rllib", "ray.rllib.env", "ray.rllib.env.vector_env"))

    class VectorEnv:
        def __init__(self, observation_space, action_space, num_envs):
            self.observation_space, self.action_space, self.num_envs = observation_space, action_space, num_envs

    vec.VectorEnv = VectorEnv
    for name, mod in (("gym", gym), ("gym.spaces", spaces), ("gym.envs", envs), ("gym.envs.registration", reg), ("ray", ray), ("ray.rllib", rllib),
                      ("ray.rllib.env", renv), ("ray.rllib.env.vector_env", vec)):  # fmt: skip
        monkeypatch.setitem(sys.modules, name, mod)
    import rl4mm_amd.env as E

    E = importlib.reload(E)
    try:
        assert issubclass(E.HistoricalOrderbookEnvironment, gym.Env) and issubclass(E.HistoricalOrderbookVectorEnv, VectorEnv)
        assert E.Box is Box
        made = {}

        def fake_sim(cfg):  # no GPU here: the handle is not what this test is about
            made["cfg"] = cfg
            return types.SimpleNamespace(add_stream=lambda s: 0, set_portfolio=lambda *a: None, close=lambda: None, n_envs=cfg.n_envs,
                                         limits=lambda: dict(level_slabs=1024, ladder_levels=64, far_levels=2048, queue_slots=1024, agent_orders=128, fill_log_rows=1024, bytes_per_env=9 << 20))

        monkeypatch.setattr(E._lib, "new_sim", fake_sim)
        cat = E.StreamCatalog().add(synthetic_stream(3, 10, 50.0, 9.5 * 3600, 9.5 * 3600 + 5, ticker="SYN", trading_date="2018-02-21"))
        env_config = dict(ticker="SYN", min_date="2018-02-21", max_date="2018-02-21", step_size=0.1, episode_length=30, n_levels=10, features="full_state", max_inventory=1000,
                          normalisation_on=False, initial_cash=10**6, initial_inventory=0, per_step_reward_function="PnL", terminal_reward_function="PnL", market_order_clearing=False,
                          market_order_fraction_of_inventory=0.0, min_quote_level=0, max_quote_level=10, enter_spread=False, concentration=None, info_calculator=None)  # fmt: skip
        env = E.env_creator(env_config, catalog=cat)
        assert isinstance(env, gym.Env) and isinstance(env.spec, EnvSpec)
        assert env.spec.id == "HistoricalOrderbookEnvironment-v0" and env.spec.max_episode_steps == 18000
        assert isinstance(env.observation_space, Box) and env.observation_space.shape == (10,) and env.action_space.shape == (4,)
        venv = E.env_creator(env_config, catalog=cat, num_envs=8)
        assert isinstance(venv, VectorEnv) and venv.num_envs == 8 and venv.spec.max_episode_steps == 18000 and made["cfg"].n_envs == 8
    finally:
        monkeypatch.undo()
        importlib.reload(E)
    assert not issubclass(E.HistoricalOrderbookEnvironment, gym.Env)


def test_info_dicts_feed_the_reference_metrics_callback():
    """rl4mm/utils/rllib/custom_metrics_callback.py:14-47 copies every key of `episode._agent_to_last_info["agent0"]` into
    `episode.custom_metrics`, which RLlib then averages per key: the info dict of a step must be a flat dict whose keys are the
    reference calculator's (InfoCalculators.py:31-89) and whose values survive np.mean.  The callback's loop is restated on
    a stub episode (ray is not installed); the dict comes from SimpleInfoCalculator.calculate_dicts on a stub handle."""
    import types

    from rl4mm_amd.env import SimpleInfoCalculator

    n = 3
    sim = types.SimpleNamespace(
        portfolios=lambda: (np.array([5, -7, 0]), np.array([10**6, 2 * 10**6, 1000]), np.array([2_700_050.0, 2_700_010.5, 2_699_990.0])),
        best_prices=lambda: (np.array([2_700_000] * n), np.array([2_700_100] * n)), status=lambda: np.zeros(n, dtype=np.int32),
    )
    actions = np.array([[1.0, 2.0, 1.0, 2.0, 6.0], [0.5, 0.5, 3.0, 3.0, 6.0], [2.0, 1.0, 2.0, 1.0, 0.0]])
    calc = SimpleInfoCalculator(market_order_fraction_of_inventory=0.5)
    infos = calc.calculate_dicts(sim, actions)
    reference_keys = ["asset_price", "inventory", "cash", "aum", "market_spread", "agent_spread", "agent_weighted_spread", "midprice_offset", "weighted_midprice_offset",
                      "bid_action", "ask_action", "market_order_action", "market_order_count", "market_order_total_volume"]  # fmt: skip
    for info in infos:
        assert list(info) == reference_keys
        episode = types.SimpleNamespace(length=1, custom_metrics={}, _agent_to_last_info={"agent0": info})
        assert episode.length > 0
        for key in episode._agent_to_last_info["agent0"]:  # on_episode_step / on_episode_end, :26-27 and :46-47
            episode.custom_metrics[key] = episode._agent_to_last_info["agent0"][key]
        assert set(episode.custom_metrics) == set(reference_keys)
        for key, value in episode.custom_metrics.items():  # what RLlib's metrics summary does with a custom metric
            assert np.isfinite(np.mean(value)), key
    assert [i["market_order_count"] for i in infos] == [0, 1, 0] and infos[1]["market_order_total_volume"] == 4.0  # |inventory| 7 > 6: round(3.5) = 4
    assert infos[0]["aum"] == 10**6 + 2_700_050.0 * 5 and infos[0]["market_spread"] == 100


def test_7z_archives_go_through_the_7z_executable(tmp_path, monkeypatch):
    """run_populate_database_from_zipped.py:100 shells out to `7z x`; so does ingest._extract for `.7z` archives.  There is no
    7z in this image: a stand-in executable on PATH (it unpacks a zip that was given the .7z name) shows the branch passes the
    right command line, extracts next to the archive, and the day ends up in the catalog."""
    import stat
    import zipfile

    from rl4mm_amd import ingest
    from rl4mm_amd.env import StreamCatalog
    from rl4mm_amd.synth import generate_lobster, write_lobster_csv

    data = tmp_path / "data"
    data.mkdir()
    g = generate_lobster(5, 10, 150.0, 9.5 * 3600, 9.5 * 3600 + 20)
    mp_, bp_ = write_lobster_csv(g, tmp_path / "raw", "SYN", "2018-02-21", 10)
    arc = data / "_data_dwn_9_391__SYN_2018-02-21_2018-02-21_10.7z"
    with zipfile.ZipFile(arc, "w") as z:
        z.write(mp_, mp_.name); z.write(bp_, bp_.name)
    bindir = tmp_path / "bin"
    bindir.mkdir()
    fake = bindir / "7z"
    fake.write_text('#!/usr/bin/env python3\nimport sys, zipfile\nopen(sys.argv[0] + ".log", "a").write(" ".join(sys.argv[1:]) + "\\n")\n'
                    'assert sys.argv[1] == "x" and sys.argv[2] == "-y", sys.argv\nout = [a[2:] for a in sys.argv if a.startswith("-o")][0]\n'
                    'zipfile.ZipFile(sys.argv[3]).extractall(out)\n')
    fake.chmod(fake.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv("PATH", f"{bindir}{os.pathsep}{os.environ['PATH']}")
    cat = ingest.populate_catalog_from_archives(StreamCatalog(), data, "S")
    assert [k for k, _ in cat.items()] == [("SYN", "2018-02-21")]
    ref = pack_lobster(mp_, bp_, "SYN", "2018-02-21", 10, "S")
    assert np.array_equal(cat.get("SYN", "2018-02-21").msgs, ref.msgs) and np.array_equal(cat.get("SYN", "2018-02-21").snaps, ref.snaps)
    assert (bindir / "7z.log").read_text().split()[:3] == ["x", "-y", str(arc)] and not list(data.glob("*.csv"))
    monkeypatch.setenv("PATH", str(tmp_path / "nowhere"))
    with pytest.raises(RuntimeError, match="no 7z executable"):
        ingest._extract(arc, data)


def test_orderbook_simulator_class_host_logic_on_the_reference_simulator_tests():
    """rl4mm_amd.simulator.OrderbookSimulator's own logic (datetime <-> stream time, internal ids = the device's order tags, Orderbook /
    FilledOrders decoding) through the reference's simulator tests (testOrderbookSimulator.py:46-108), with the class bound to
    the CPU restatement (same C-ABI).  The HIP library runs the same checks in tests/test_gpu_simulator.py."""
    import sim_facade_checks as fc

    fc.check_agreement_with_lobster(ol.oracle_sim)
    fc.check_update_outer_levels_with_internal_orders(ol.oracle_sim)
    fc.check_internal_orders_by_id(ol.oracle_sim)
    fc.check_cancellations_the_host_could_not_rank(ol.oracle_sim)
    fc.check_reset_episode_with_a_start_book(ol.oracle_sim)


def test_reset_at_resets_all_finished_envs_in_one_launch_and_keeps_the_rng_sequence():
    """RLlib calls reset_at(i) once per finished env, in index order, and all envs of a batch finish together: the first call
    resets every finished env in ONE reset launch (one launch replays the warm-up of 4096 envs as fast as of one), the others
    return cached rows.  Start draws and the state np.random is left in must equal one-env-at-a-time resets."""

    class StubSim:
        def __init__(self):
            self.launches = []

        def reset(self, starts, sids, ids):
            self.launches.append(np.array(ids))
            return np.stack([np.asarray(starts, dtype=np.float64), np.asarray(ids, dtype=np.float64)], axis=1)

        def status(self):
            return np.zeros(16, dtype=np.int32)

    def fresh():
        env = _bare_vector_env(datetime(2018, 2, 1), datetime(2018, 3, 15), 30, 0.1, n_envs=16)
        env.num_envs, env.sim, env.obs_dtype = 16, StubSim(), np.dtype(np.float64)
        env._dones, env._any_done, env._reset_cache, env.status = np.zeros(16, dtype=bool), False, {}, np.zeros(16, dtype=np.int32)
        return env

    finished = [1, 2, 5, 6, 7, 11, 15]
    np.random.seed(5)
    one_by_one = fresh()
    rows_a = [one_by_one.reset_many([i])[0] for i in finished]  # round 3's reset_at: a launch per env
    after_a = np.random.randint(0, 1 << 30)
    assert len(one_by_one.sim.launches) == len(finished)

    np.random.seed(5)
    batched = fresh()
    batched._dones[finished] = True
    batched._any_done = True
    rows_b = [batched.reset_at(i) for i in finished]
    after_b = np.random.randint(0, 1 << 30)
    assert len(batched.sim.launches) == 1 and list(batched.sim.launches[0]) == finished
    assert np.array_equal(np.array(rows_a), np.array(rows_b)) and after_a == after_b
    assert not batched._dones.any() and not batched._any_done and batched._reset_cache == {}
    # an env that is not finished (or a second reset_at of one that was) is a reset of its own
    batched.reset_at(3)
    batched.reset_at(5)
    assert [list(x) for x in batched.sim.launches[1:]] == [[3], [5]]


def test_mixed_ticker_draw_and_stream_residency_lru():
    """ticker=[...]: env i replays tickers[i mod T], every env draws its own day (np.random consumed in env order exactly as for one
    ticker), and the day is looked up under the env's ticker.  Streams go to the device on first use; under a byte budget the least
    recently drawn days that no env replays are dropped after the reset, never one an env still holds."""
    from rl4mm_amd.env import _StreamResidency

    tickers = ("AAA", "BBB", "CCC")
    env = _bare_vector_env(datetime(2018, 2, 1), datetime(2018, 2, 14), 30, 0.1, n_envs=12, tickers=tickers)
    np.random.seed(3)
    day_ref, start_ref = env._draw_days(12)
    np.random.seed(3)
    sids, starts = env._draw(12)
    assert np.array_equal(starts, start_ref)
    dates = env._day_lookup()
    ids, keys, st = env.last_draw
    assert keys == [(tickers[i % 3], dates[int(d)]) for i, d in enumerate(day_ref)]
    assert env._env_key == [None] * 12  # (the bookkeeping moves once the device HAS reset the envs: a reset that raises leaves it alone)
    env._after_reset()
    assert env._env_key == keys
    assert [env._residency.sid[k] for k in keys] == sids.tolist()
    # a partial reset: only those envs' keys move, np.random advances by their draws only
    np.random.seed(9)
    d2, _ = env._draw_days(2)
    np.random.seed(9)
    env._draw(2, env_ids=[4, 7])
    assert env._env_key[4] == keys[4]
    env._after_reset()
    assert env._env_key[4] == ("BBB", dates[int(d2[0])]) and env._env_key[7] == ("BBB", dates[int(d2[1])]) and env._env_key[5] == keys[5]

    class Stream:  # what _StreamResidency touches of a PackedStream
        def __init__(self, mb):
            self.msgs, self.snaps = np.zeros(mb * 2**20 // 8, dtype=np.int64), np.zeros(0, dtype=np.int64)

    class Catalog:
        def get(self, ticker, date):
            if ticker == "ZZZ":
                raise KeyError(ticker)
            return Stream(10)

    class Sim:
        def __init__(self):
            self.live, self.log = {}, []

        def add_stream(self, stream):
            sid = min(set(range(len(self.live) + 1)) - set(self.live))  # the lowest free slot, like lobsim_add_stream
            self.live[sid] = stream
            self.log.append(("add", sid))
            return sid

        def drop_stream(self, sid):
            del self.live[sid]
            self.log.append(("drop", sid))

    sim = Sim()
    res = _StreamResidency(sim, Catalog(), budget_bytes=35 * 2**20)  # room for three 10 MB days
    assert res.has(("AAA", "2018-02-01")) and not res.has(("ZZZ", "2018-02-01"))
    a, b, c, d, e = [("AAA", f"2018-02-0{i}") for i in range(1, 6)]
    assert res.ensure([a, b, c]) == {a: 0, b: 1, c: 2}
    res.trim({a, b, c})
    assert len(sim.live) == 3 and res.drops == 0
    res.ensure([d, a])  # a fourth day: over budget until the trim
    assert len(sim.live) == 4
    res.trim({a, d, c})  # b is the only one nobody holds
    assert sorted(res.sid) == sorted([a, c, d]) and ("drop", 1) in sim.log and res.resident_bytes == 30 * 2**20
    assert res.ensure([e])[e] == 1  # the freed slot is handed out again
    res.trim({a, c, d, e})  # all in use: the budget is exceeded rather than a stream pulled from under an env
    assert len(sim.live) == 4 and res.drops == 1
    res.trim({e})  # least recently used first: c (drawn at clock 1), then a and d (clock 2) until it fits
    assert e in res.sid and len(res.sid) == 3 and c not in res.sid


def test_default_device_id_spreads_rollout_workers_over_gpus(monkeypatch):
    """env_creator's device: RLlib's EnvContext.worker_index (1 .. num_workers for remote workers) modulo the visible GPUs, else
    LOCAL_RANK, else 0 (rl4mm/helpers/main_helper.py:262-263 sets num_workers; the reference's envs are CPU objects and need none)."""
    import torch

    from rl4mm_amd.env import default_device_id

    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("LOCAL_RANK", raising=False)

    class Ctx(dict):
        worker_index = 0

    assert default_device_id({}) == 0 and default_device_id(None) == 0
    ctx = Ctx()
    assert default_device_id(ctx) == 0
    for w, dev in ((1, 0), (2, 1), (8, 7), (9, 0), (12, 3)):
        ctx.worker_index = w
        assert default_device_id(ctx) == dev
    assert default_device_id({"worker_index": 3}) == 2
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert default_device_id({}) == 5
    ctx.worker_index = 2
    assert default_device_id(ctx) == 1  # the worker index wins
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert default_device_id(ctx) == 0 and default_device_id({}) == 0


def test_array_path_output_pool_never_reuses_memory_the_caller_still_holds():
    """the output pool of vector_step_arrays(copy=True) decides by weak references to the arrays it handed out, not by
    reference counts: while the caller holds an array - or any view, slice or reshape of one - its memory is not handed out again."""
    from rl4mm_amd.env import HistoricalOrderbookVectorEnv as V

    class S:
        obs_dim = 3

    env = V.__new__(V)
    env.num_envs, env.sim, env._out_pool = 4, S(), []
    o1, r1, d1 = env._out_buffers(np.dtype(np.float64))
    o1[:] = 1.0
    keep = o1[2]  # a view only
    del o1, r1, d1
    o2, r2, d2 = env._out_buffers(np.dtype(np.float64))
    o2[:] = 2.0
    assert keep.tolist() == [1.0, 1.0, 1.0] and len(env._out_pool) == 2  # the first buffer is still referenced through the view
    del keep, o2, r2, d2
    o3, _, _ = env._out_buffers(np.dtype(np.float64))
    assert len(env._out_pool) == 2 and o3.shape == (4, 3)  # recycled
    flat = o3.reshape(-1)
    del o3
    o4, _, _ = env._out_buffers(np.dtype(np.float64))
    o4[:] = 4.0
    flat[:] = 3.0
    assert (o4 == 4.0).all()  # (a reshape kept o3's memory out of the pool)
    o5, _, _ = env._out_buffers(np.dtype(np.float32))
    assert o5.dtype == np.float32 and o5.shape == (4, 3)


def test_gym_facade_on_the_oracle_replays_the_reference_through_an_exception(monkeypatch):
    """Everything ABOVE the C-ABI without a GPU: rl4mm_amd.env.HistoricalOrderbookEnvironment with the CPU oracle as its handle
    (oracle_lib.OracleHandle - test infrastructure, the product has no CPU path) against the reference's trajectory thin_sweep_crash:
    observations and rewards step by step, EmptyOrderbookError out of `step` where the reference raised it, the portfolio the failed
    step did not touch, the next episode from it.  (oracle/fuzz_facade.py draws thousands of such cases against the reference itself
    in the build container; tests/test_gpu_parity.py runs this one on the HIP path.)"""
    from datetime import datetime, timedelta

    from rl4mm_amd import env as E

    monkeypatch.setattr(E._lib, "new_sim", ol.OracleHandle)
    g, z = ol.load_traj("thin_sweep_crash")
    ms = timedelta(seconds=0.1)
    start = timedelta(microseconds=g["starts_us"][0])
    e = E.HistoricalOrderbookEnvironment(
        catalog=E.StreamCatalog().add(ol.golden_stream("thin")), features=[E.Inventory(), E.Spread(), E.Price(update_frequency=ms)], ticker="THN", step_size=ms,
        episode_length=timedelta(seconds=2), min_date=datetime(2018, 2, 21), max_date=datetime(2018, 2, 21), min_start_timedelta=start, max_end_timedelta=start + timedelta(seconds=2),
        n_levels=3, max_quote_level=3, market_order_clearing=True, market_order_fraction_of_inventory=1.0, initial_portfolio=E.Portfolio(inventory=500, cash=10**9),
        per_step_reward_function=E.get_reward_function("PnL"), terminal_reward_function=E.get_reward_function("PnL"),
    )  # fmt: skip
    assert np.array_equal(e.reset(), z["reset_obs"][0])
    k_crash = int(z["crash_ks"][0])
    for k in range(k_crash):
        obs, reward, done, _ = e.step(z["actions"][k])
        assert np.array_equal(obs, z["obs"][k]) and reward == z["reward"][k] and not done
    with pytest.raises(E.EmptyOrderbookError):
        e.step(z["crash_actions"][0])
    inv, cash, _ = e._venv.portfolios()
    assert inv[0] == int(z["crash_inventory"][0]) == 500 and cash[0] == int(z["crash_cash"][0])
    e._venv.min_start_timedelta = timedelta(microseconds=g["starts_us"][1])
    e._venv.max_end_timedelta = e._venv.min_start_timedelta + timedelta(seconds=2)
    assert np.array_equal(e.reset(), z["reset_obs"][1])
    for k in range(k_crash, len(z["reward"])):
        obs, reward, done, _ = e.step(z["actions"][k])
        assert np.array_equal(obs, z["obs"][k]) and reward == z["reward"][k] and done == bool(z["done"][k])
    assert done
    e.close()


def test_bench_refuses_pmc_counters_of_another_kernel(tmp_path, monkeypatch):
    """`roofline.traffic` of a bench line comes from committed rocprofv3 counter files (counters cannot be collected inside the timed
    run).  A file taken on another kernel - another agent table or ladder, the generic variant, an older round - or at another batch
    size must give `traffic: null`, not a stale number."""
    import bench

    d = tmp_path / "profiles" / "rX"
    d.mkdir(parents=True)
    body = "pass_,counter,mean_per_dispatch,dispatches\nfetch,FETCH_SIZE,1000.0,85\nsq1,SQ_WAVES,4112.0,85\nwrite,WRITE_SIZE,500.0,85\n"
    (d / "pmc_step_kernel.csv").write_text("# kernel: lob_step_kernel<false, 64, 256, true>\n" + body)
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    monkeypatch.setattr(bench, "PROFILE_ROUND", "rX")
    monkeypatch.delenv("LOBSIM_GENERIC_KERNELS", raising=False)
    t, src = bench.pmc_traffic("random", 4096, "lob_step_kernel<false, 64, 256, true>")
    assert t == (2 * 1000.0 + 500.0) * 1024 and src.endswith("pmc_step_kernel.csv")
    assert bench.pmc_traffic("random", 4096, "lob_step_kernel<false, 64, 128, true>") == (None, None)  # another agent table
    assert bench.pmc_traffic("random", 8192, "lob_step_kernel<false, 64, 256, true>") == (None, None)  # another batch size
    (d / "pmc_step_kernel.csv").write_text(body)  # a file from before the header existed
    assert bench.pmc_traffic("random", 4096, "lob_step_kernel<false, 64, 256, true>") == (None, None)
    assert bench.pmc_traffic("deep50", 8192, "lob_step_kernel<false, 64, 256, true>") == (None, None)  # no file at all


def test_vector_env_fuzz_draws_cases_the_constructor_accepts():
    """tests/diag/vector_env_fuzz.py's draw (the GPU sample of it runs in tests/test_gpu_fuzz.py): every case has a warm-up of whole
    seconds set by its last feature (the reference asserts it, OrderbookSimulator.py:61), a stream for every (ticker, session) the date
    range can draw - weekend days and the holiday of 2018-02-19 mapped to the next session - covering warm-up, every start and the
    episode, and the constructor's own assertions hold (HistoricalOrderbookEnvironment.py:85-100,372-384)."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent / "diag"))
    import vector_env_fuzz as vf

    paths, multi_session, budgets = set(), 0, 0
    for seed in range(120):
        kw, streams, desc, x = vf.draw_case(seed, 32)
        feats = kw["features"]
        warm = max(f.window_size for f in feats)
        assert warm == feats[-1].window_size and warm.total_seconds() == int(warm.total_seconds()) >= 1
        assert kw["episode_length"] % kw["step_size"] == timedelta(0) and kw["min_start_timedelta"] + kw["episode_length"] <= kw["max_end_timedelta"]
        assert (kw["concentration"] is None) or kw["concentration"] >= kw["max_distribution_param"]
        assert kw["market_order_clearing"] == (kw["market_order_fraction_of_inventory"] > 0)
        tickers = [kw["ticker"]] if isinstance(kw["ticker"], str) else kw["ticker"]
        assert set(streams) == {(t, d) for t in tickers for d in desc["sessions"]} and "2018-02-19" not in desc["sessions"]
        for a in streams.values():
            assert a["start_s"] <= kw["min_start_timedelta"].total_seconds() - warm.total_seconds() - 2
            assert a["end_s"] >= kw["max_end_timedelta"].total_seconds() + 2
        paths.add(x["path"])
        multi_session += len(desc["sessions"]) > 1
        budgets += x["budget_streams"] > 0
    assert paths == {"arrays", "lists", "tensors"} and multi_session > 30 and budgets > 20
