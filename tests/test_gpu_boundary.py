"""The Python boundary on the GPU: the tensor path (torch tensors in and out on torch's stream), the host-mapped array path
(pinned buffers the kernel reads and writes itself), per-env error reporting in the batched env, RLlib's reset_at protocol
without a launch per env, and what the handle reports about its capacities.  Needs a real MI355X: `pytest -m gpu`."""
from datetime import datetime, timedelta

import numpy as np
import pytest

import drift_stream as ds
import oracle_lib as ol
from rl4mm_amd import _capi, _lib, env as E

pytestmark = pytest.mark.gpu


def _venv(n, stream=None, episode=timedelta(seconds=4), obs_dtype=np.float64, start=timedelta(hours=9, minutes=56), end=timedelta(hours=10, minutes=1), **kw):
    stream = stream or ol.golden_stream("synth_small")  # 09:54:50 .. 10:01:50
    day = datetime.strptime(stream.trading_date, "%Y-%m-%d")
    feats = [E.Spread(), E.PriceMove(name="pm", update_frequency=timedelta(seconds=0.1), lookback_periods=3), E.Volatility(name="v", update_frequency=timedelta(seconds=0.1), lookback_periods=20),
             E.Inventory(), E.EpisodeProportion(episode_length=episode), E.TradeVolumeImbalance(update_frequency=timedelta(seconds=0.1), lookback_periods=20)]  # fmt: skip
    return E.HistoricalOrderbookVectorEnv(
        n, E.StreamCatalog().add(stream), features=feats, ticker=stream.ticker, n_levels=stream.n_levels, min_date=day, max_date=day, min_start_timedelta=start,
        max_end_timedelta=end, episode_length=episode, initial_portfolio=E.Portfolio(inventory=0, cash=10**9), per_step_reward_function=E.PnL(),
        terminal_reward_function=E.PnL(), obs_dtype=obs_dtype, **kw)  # fmt: skip


@pytest.mark.parametrize("obs_dtype", [np.float64, np.float32])
def test_tensor_path_equals_array_path_bit_for_bit(obs_dtype):
    """vector_step_tensors (device tensors, torch's stream, done envs skipped on the device) against vector_step_arrays (pinned
    host buffers) on twin envs: observations, rewards and done flags of every env at every step, through an episode end and
    the reset of the finished envs, on the null stream and on a side stream."""
    import torch

    n, steps = 256, 100
    a, b = _venv(n, obs_dtype=obs_dtype), _venv(n, obs_dtype=obs_dtype)
    np.random.seed(3)
    oa = a.vector_reset_arrays()
    np.random.seed(3)
    ob = b.vector_reset_tensors()
    assert oa.dtype == np.dtype(obs_dtype) and np.array_equal(oa, ob.cpu().numpy())
    rng = np.random.Generator(np.random.PCG64(4))
    side = torch.cuda.Stream()
    n_done = 0
    for t in range(steps):
        act = rng.uniform(0, 10, size=(n, 4))
        idle = a._dones.copy()  # finished and not yet reset: skipped by both paths (the tensor path leaves their reward row as it was)
        oa, ra, da, _ = a.vector_step_arrays(act)
        d_act = torch.from_numpy(act).cuda()
        if t % 2:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ob, rb, db = b.vector_step_tensors(d_act)
            torch.cuda.current_stream().wait_stream(side)
        else:
            ob, rb, db = b.vector_step_tensors(d_act)
        torch.cuda.synchronize()
        assert np.array_equal(oa, ob.cpu().numpy()), f"step {t}: obs"
        assert np.array_equal(ra[~idle], rb.cpu().numpy()[~idle]) and np.array_equal(da, db.cpu().numpy().astype(bool)), f"step {t}"
        if da.any() and t % 3 == 1:  # (not at once: for a step or two the finished envs are skipped on both paths)
            ids = np.nonzero(a._dones)[0]
            n_done += len(ids)
            np.random.seed(100 + t)
            ra_obs = a.reset_many(ids)
            np.random.seed(100 + t)
            ids_b = b.reset_done_tensors()
            assert np.array_equal(ids, ids_b)
            assert np.array_equal(ra_obs, b._t["obs"].cpu().numpy()[ids])
    assert n_done >= n
    ia, ca, _ = a.portfolios()
    ib, cb, _ = b.portfolios()
    assert np.array_equal(ia, ib) and np.array_equal(ca, cb) and np.abs(ia).max() > 0
    a.close()
    b.close()


def test_host_mapped_step_equals_the_host_pointer_step_and_the_device_pointer_step():
    """lobsim_step_host_io (the kernel reads / writes pinned host memory), lobsim_step (host pointers, copies around it) and
    lobsim_step_device (device pointers) on three handles: identical outputs, f32 observations = the f64 ones rounded."""
    import torch

    n = 128
    g, _ = ol.load_traj("synth_default_random")
    cfg = ol.cfg_from_golden(g, n_envs=n)
    sims = [_lib.new_sim(cfg) for _ in range(3)]
    starts = (36_000 + np.arange(n) % 40) * 1_000_000
    for s in sims:
        s.add_stream(ol.golden_stream(g["stream_file"]))
        s.set_portfolio(0, 10**9)
        s.reset(starts, 0)
    io = sims[0].host_io()
    D = sims[0].obs_dim
    d_obs, d_o32 = torch.zeros((n, D), dtype=torch.float64).cuda(), torch.zeros((n, D), dtype=torch.float32).cuda()
    d_rew, d_done = torch.zeros(n, dtype=torch.float64).cuda(), torch.zeros(n, dtype=torch.uint8).cuda()
    rng = np.random.Generator(np.random.PCG64(8))
    for t in range(40):
        act = rng.uniform(0, 10, size=(n, 4))
        active = (rng.random(n) < 0.9).astype(np.uint8) if t % 5 == 4 else None
        io["actions"][:] = act
        if active is not None:
            io["active"][:] = active
        sims[0].step_host_io(active is not None, 3)
        o1, r1, d1 = sims[1].step(act, active=active)
        d_act = torch.from_numpy(act).cuda()
        d_actv = None if active is None else torch.from_numpy(active).cuda()
        sims[2].step_device(d_act.data_ptr(), 0 if d_actv is None else d_actv.data_ptr(), d_obs.data_ptr(), d_o32.data_ptr(), d_rew.data_ptr(), d_done.data_ptr())
        sims[2].sync()
        m = slice(None) if active is None else active.astype(bool)
        assert np.array_equal(io["obs"][m], o1[m]) and np.array_equal(io["reward"][m], r1[m]) and np.array_equal(io["done"][m], d1[m]), t
        assert np.array_equal(io["obs"][m], d_obs.cpu().numpy()[m]) and np.array_equal(io["reward"][m], d_rew.cpu().numpy()[m]), t
        assert np.array_equal(io["obs_f32"][m], io["obs"][m].astype(np.float32)) and np.array_equal(io["obs_f32"][m], d_o32.cpu().numpy()[m]), t
    for s in sims:
        s.close()


def test_reset_at_protocol_of_rllib_is_one_launch_per_episode_boundary():
    """All envs of a batch finish together; RLlib then calls reset_at(i) for i = 0 .. N-1.  The observations and the np.random
    state must be those of one-env-at-a-time resets (the reference's sequence), in a fraction of their time."""
    import time

    n = 512
    a, b = _venv(n, episode=timedelta(seconds=2)), _venv(n, episode=timedelta(seconds=2))
    for v in (a, b):
        np.random.seed(9)
        v.vector_reset()
        rng = np.random.Generator(np.random.PCG64(10))
        for t in range(20):
            obs, rew, done, infos = v.vector_step(rng.uniform(0, 10, size=(n, 4)))
        assert all(done) and infos[0] == {}
    np.random.seed(21)
    t0 = time.perf_counter()
    rows_a = [a.reset_at(i) for i in range(n)]
    dt_a = time.perf_counter() - t0
    after_a = np.random.randint(0, 1 << 30)
    np.random.seed(21)
    t0 = time.perf_counter()
    rows_b = [b.reset_many([i])[0] for i in range(n)]  # a launch per env: what reset_at did before
    dt_b = time.perf_counter() - t0
    after_b = np.random.randint(0, 1 << 30)
    assert np.array_equal(np.array(rows_a), np.array(rows_b)) and after_a == after_b
    assert dt_a < dt_b / 5, (dt_a, dt_b)
    rng = np.random.Generator(np.random.PCG64(12))
    for t in range(5):
        act = rng.uniform(0, 10, size=(n, 4))
        oa, ob = a.vector_step_arrays(act), b.vector_step_arrays(act)
        assert np.array_equal(oa[0], ob[0]) and np.array_equal(oa[1], ob[1]) and not oa[2].any()
    a.close()
    b.close()


def test_batched_env_reports_why_an_env_stopped():
    """Where the reference raises (EmptyOrderbookError: a clearing market order larger than the whole ask side) or a capacity
    is exceeded (a drifting day with a small level budget), the batched env says so: done=True with infos[i]["error"], the
    status array on every path, the reference's exception type from the single-env facade."""
    # (1) LEVEL_OVERFLOW: price drifting one tick per 2 s, the smallest level budget (max_levels=64 -> 256 slabs): the episode cannot hold its stale levels
    ps = ds.drift_stream(11, ds.ramp_path(2.0), duration_s=1_000.0)
    day = datetime.strptime(ps.trading_date, "%Y-%m-%d")
    kw = dict(features=[E.Spread(), E.Inventory()], ticker=ps.ticker, n_levels=ps.n_levels, min_date=day, max_date=day, min_start_timedelta=timedelta(hours=9, minutes=31),
              max_end_timedelta=timedelta(hours=9, minutes=46), episode_length=timedelta(minutes=15), per_step_reward_function=E.PnL(), terminal_reward_function=E.PnL())  # fmt: skip
    v = E.HistoricalOrderbookVectorEnv(8, E.StreamCatalog().add(ps), max_levels=64, **kw)  # (the smallest budget: raised to 128 levels = 256 slabs)
    assert v.limits["level_slabs"] == 256 and "256 price levels" in repr(v)
    v.vector_reset()
    act = np.tile([1.0, 1.0, 1.0, 1.0], (8, 1))
    hit = None
    for t in range(9000):
        obs, rew, done, infos = v.vector_step_arrays(act)
        if done.any():
            hit = (t, done.copy(), infos)
            break
    assert hit is not None and hit[0] < 8990, "the level budget must run out before the episode ends"
    t, done, infos = hit
    assert done.all() and (v.status == _capi.ST_LEVEL_OVERFLOW).all()
    assert all("LOBSIM_ST_LEVEL_OVERFLOW" in d["error"] and d["status"] == _capi.ST_LEVEL_OVERFLOW for d in infos)
    obs, rew, done, infos = v.vector_step(act)  # finished envs keep saying done, nothing new to report
    assert all(done) and infos[0] == {}
    v.reset_at(0)  # (resets every finished env, see reset_at)
    assert not v.status.any() and not v._dones.any()
    v.close()
    # (2) EmptyOrderbookError: market-order clearing of an inventory larger than the ask side
    stream = ol.golden_stream("synth_small")
    day = datetime.strptime(stream.trading_date, "%Y-%m-%d")
    kw = dict(features=[E.Spread(), E.Inventory()], ticker=stream.ticker, n_levels=10, min_date=day, max_date=day, min_start_timedelta=timedelta(hours=10),
              max_end_timedelta=timedelta(hours=10, minutes=1), episode_length=timedelta(seconds=30), per_step_reward_function=E.PnL(), terminal_reward_function=E.PnL(),
              market_order_clearing=True, market_order_fraction_of_inventory=1.0, initial_portfolio=E.Portfolio(inventory=-10**7, cash=10**12))  # fmt: skip
    v = E.HistoricalOrderbookVectorEnv(4, E.StreamCatalog().add(stream), **kw)
    v.vector_reset()
    obs, rew, done, infos = v.vector_step(np.tile([1.0, 1.0, 1.0, 1.0, 5.0], (4, 1)))
    assert all(done) and all("EmptyOrderbookError" in d["error"] for d in infos) and (v.status & _capi.ST_EMPTY_BOOK).all()
    v.close()
    e = E.HistoricalOrderbookEnvironment(catalog=E.StreamCatalog().add(stream), **kw)
    e.reset()
    with pytest.raises(E.EmptyOrderbookError):
        e.step(np.array([1.0, 1.0, 1.0, 1.0, 5.0]))
    e.close()


def test_limits_and_default_level_budget():
    """The default level budget is a fixed 512 per side (1024 levels per env) whatever else holds device memory, an explicit
    one is honoured, and lobsim_get_limits says what the handle has."""
    feats = [_capi.make_feature(_capi.F_INVENTORY, min_value=-1e6, max_value=1e6)]
    for max_levels, slabs in ((0, 1024), (64, 256), (200, 448), (2048, 4096)):
        s = _lib.new_sim(_capi.make_cfg(4, 10, feats, max_levels=max_levels))
        lim = s.limits()
        assert lim["level_slabs"] == slabs and lim["ladder_levels"] == 64 and lim["queue_slots"] == 4096 and lim["agent_orders"] in (128, 256) and lim["far_levels"] == 2048
        assert lim["queue_pages"] >= max(slabs, 1024 + 256)  # at least a page per level of the budget
        assert lim["bytes_per_env"] > lim["queue_pages"] * 1024  # a level's queue: 1, 4 or 16 KiB by the orders it holds
        if max_levels == 0:
            assert lim["queue_pages"] == 256 + 2048 and lim["bytes_per_env"] < 2.8 * 2**20  # (16.5 MiB until round 5: every level owned 16 KiB)
        s.close()
    s = _lib.new_sim(_capi.make_cfg(4, 10, feats, queue_pages=4000))  # an explicit pool: 256 + a multiple of 16 pages
    assert s.limits()["queue_pages"] == 256 + 16 * 234 and s.limits()["level_slabs"] == 1024
    s.close()
    with pytest.raises(_capi.LobsimError):
        _lib.new_sim(_capi.make_cfg(4, 10, feats, queue_pages=100))
    assert _lib.binding().fn["abi_version"]() == _capi.ABI_VERSION


def test_array_path_outputs_stay_valid_while_referenced():
    """vector_step_arrays(copy=True) recycles its output arrays only once the caller has dropped them: arrays (or views of them)
    that are kept hold their values over later steps; copy=False hands out the pinned buffers themselves, which the next step
    overwrites."""
    n = 64
    v = _venv(n)
    np.random.seed(1)
    v.vector_reset_arrays()
    rng = np.random.Generator(np.random.PCG64(2))
    kept = []
    for t in range(12):
        obs, rew, done, _ = v.vector_step_arrays(rng.uniform(0, 10, size=(n, 4)))
        kept.append((obs, obs.copy(), rew[3:5], rew[3:5].copy()))  # the array itself and a view of the reward array
    for obs, obs_then, rv, rv_then in kept:
        assert np.array_equal(obs, obs_then) and np.array_equal(rv, rv_then)
    assert len({k[0].__array_interface__["data"][0] for k in kept}) == 12  # twelve different pieces of memory
    ids = set()
    for t in range(12):  # dropped every step: the same memory comes back (under a new array object each time)
        obs, rew, done, _ = v.vector_step_arrays(rng.uniform(0, 10, size=(n, 4)))
        ids.add(obs.__array_interface__["data"][0])
        del obs, rew, done
    assert len(ids) <= 2
    a = v.vector_step_arrays(rng.uniform(0, 10, size=(n, 4)), copy=False)[0]
    a_then = a.copy()
    b = v.vector_step_arrays(rng.uniform(0, 10, size=(n, 4)), copy=False)[0]
    assert a is b or np.shares_memory(a, b)
    assert not np.array_equal(a_then, b)
    v.close()


def test_tensor_path_can_be_captured_in_a_hip_graph():
    """A closed loop on the device - policy network, env step - is launch-bound on the host; the tensor path is capturable: one
    `vector_step_tensors` call (bound to the capture stream beforehand) recorded in a torch.cuda.CUDAGraph and replayed with the
    actions updated in place gives, step for step, what eager calls give on a twin env."""
    import torch

    n, steps = 256, 30
    a, b = _venv(n, obs_dtype=np.float32), _venv(n, obs_dtype=np.float32)
    np.random.seed(5)
    a.vector_reset_tensors()
    np.random.seed(5)
    b.vector_reset_tensors()
    rng = np.random.Generator(np.random.PCG64(6))
    acts = torch.from_numpy(rng.uniform(0, 10, size=(steps + 1, n, 4))).cuda()
    s = torch.cuda.Stream()
    static_act = acts[0].clone()
    with torch.cuda.stream(s):
        b.vector_step_tensors(static_act)  # binds the handle to the stream (lobsim_set_stream synchronises: not allowed during capture)
        a.vector_step_tensors(acts[0])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        ob, rb, db = b.vector_step_tensors(static_act)
    for t in range(1, steps + 1):
        static_act.copy_(acts[t])
        g.replay()
        with torch.cuda.stream(s):
            oa, ra, da = a.vector_step_tensors(acts[t])
        torch.cuda.synchronize()
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db), t
    assert float(ob.abs().max()) > 0
    a.close()
    b.close()
