"""BASELINE.json configs 2, 3 and 4 at their per-GPU shard size on one MI355X (the 8-GPU forms shard these env ranges with
no data-path collective, rl4mm_amd/dist.py).  Configs 2 and 3 are checked against the CPU oracle on EVERY env of the batch (the oracle runs ~4 M env-steps/s on the
box's host cores), config 4 on 256 envs across its four tickers over a full episode: bit-exact for inventory / cash / price /
fill counters / book contents, rtol 1e-12 / atol 1e-9 for observations and rewards.

Real LOBSTER SPY 2018-02-21 data (configs[2]) is not available offline: the SPY-shaped synthetic stream stands in; the
bit-exact fill check against the reference itself is on the MSFT slice (tests/test_gpu_parity.py, trajectories)."""
import multiprocessing as mp

import numpy as np
import pytest

import bench
import oracle_lib as ol
from rl4mm_amd import _capi, _lib

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-12, 1e-9


def _fill_state(sim, e):
    st = sim.env_state(int(e))
    return [st[k] for k in ("agent_fill_count", "agent_buy_volume", "agent_sell_volume", "ext_buy_count", "ext_sell_count", "ext_buy_volume", "ext_sell_volume", "status", "n_resyncs")]


def _assert_portfolios(gpu, cpu, what):
    for a, b, name in zip(gpu.portfolios(), cpu.portfolios(), ("inventory", "cash", "price")):
        assert np.array_equal(a, b), f"{what}: {name} differs in envs {np.nonzero(a != b)[0][:8]}"


def _assert_fill_states(gpu, cpu, n):
    for e in range(n):
        a, b = _fill_state(gpu, e), _fill_state(cpu, e)
        assert a == b and a[7] == 0, (e, a, b)


def test_config2_shape_4096_envs_teradactyl_pnl():
    """configs[2] shape: 4096 envs x 1 GPU, replay + the Teradactyl rule-based agent evaluated on the device, PnL reward,
    default full_state features.  240 steps in one launch (lobsim_rollout_device), then 60 more one launch per step.  EVERY
    env against the oracle: steps taken, done, inventory, cash, price (bit-exact), observation and episode return after the
    launch and after each of the single steps; the fill counters of every env and sampled per-order books at the end."""
    import torch

    n, T, singles = 4096, 240, 60
    stream = bench.build_streams("teradactyl", 40)[0]
    cfg = bench.workload_cfg(_capi, "teradactyl", n)
    gpu, cpu = _lib.new_sim(cfg), ol.oracle_sim(cfg)
    starts, sids = bench.workload_starts("teradactyl", n, 1, 7)
    for sim in (gpu, cpu):
        sim.add_stream(stream)
        sim.set_portfolio(0, 10**12)
        sim.reset(starts, 0)
    D = gpu.obs_dim
    obs, reward, done = torch.zeros((n, D), dtype=torch.float64).cuda(), torch.zeros(n, dtype=torch.float64).cuda(), torch.zeros(n, dtype=torch.uint8).cuda()
    ret, steps = torch.zeros(n, dtype=torch.float64).cuda(), torch.zeros(n, dtype=torch.int32).cuda()
    params = [300, 10.0, 0.5, 10.0, 1.0, 5, 0, 300]  # max_inventory 300 so that the policy's skew is exercised
    gpu.rollout_device(_capi.POLICY_TERADACTYL, params, T, 0, 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), ret.data_ptr(), steps.data_ptr())
    gpu.sync()
    out = ol.oracle_rollout(cpu, _capi.POLICY_TERADACTYL, params, T, want_actions=False)
    assert np.array_equal(steps.cpu().numpy(), np.full(n, T)) and np.array_equal(out["steps"], np.full(n, T)) and not done.cpu().numpy().any()
    _assert_portfolios(gpu, cpu, "after the 240-step launch")
    assert np.allclose(obs.cpu().numpy(), out["obs"], rtol=RTOL, atol=ATOL)
    total = out["ret"].copy()
    for t in range(singles):  # the same policy, one launch per env.step()
        gpu.rollout_device(_capi.POLICY_TERADACTYL, params, 1, 0, 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), ret.data_ptr(), 0)
        gpu.sync()
        out = ol.oracle_rollout(cpu, _capi.POLICY_TERADACTYL, params, 1, want_actions=False)
        total += out["ret"]
        _assert_portfolios(gpu, cpu, f"single step {t}")
        assert np.array_equal(done.cpu().numpy(), out["done"])
        assert np.allclose(obs.cpu().numpy(), out["obs"], rtol=RTOL, atol=ATOL) and np.allclose(reward.cpu().numpy(), out["reward"], rtol=0, atol=ATOL), t
    assert np.abs(gpu.portfolios()[0]).max() > 0, "the agent must have traded"
    assert np.allclose(ret.cpu().numpy(), total, rtol=0, atol=1e-6)  # sums of 300 rewards of size ~1e3
    _assert_fill_states(gpu, cpu, n)
    for e in (0, 1666, n - 1):
        assert np.array_equal(gpu.queues(e), cpu.queues(e))
    gpu.close()


def test_config3_shard_4096_envs_ad_reward():
    """configs[3] shard (4096 of the 32768 envs): random Beta-policy actions, the asymmetrically dampened per-step reward
    (get_reward_function("AD")), default full_state features, one launch per env.step() through device pointers as an RLlib
    rollout worker would drive it, 300 steps.  EVERY env against the oracle at every step: observation and reward
    (rtol 1e-12 / atol 1e-9), done; inventory / cash / price of every env every 25 steps and at the end (bit-exact), the
    fill counters of every env at the end.  Twin envs (same start, same actions) give identical results wherever they
    sit in the batch."""
    import torch

    n, steps = 4096, 300
    stream = bench.build_streams("ad", 40)[0]
    cfg = bench.workload_cfg(_capi, "ad", n)
    assert cfg.per_step_reward.kind == _capi.R_INVENTORY_ADJUSTED and cfg.per_step_reward.asymmetrically_dampened == 1
    gpu, cpu = _lib.new_sim(cfg), ol.oracle_sim(cfg)
    s_half, _ = bench.workload_starts("ad", n // 2, 1, 11)
    starts = np.concatenate([s_half, s_half])
    for sim in (gpu, cpu):
        sim.add_stream(stream)
        sim.set_portfolio(0, 10**12)
        sim.reset(starts, 0)
    rng = np.random.Generator(np.random.PCG64(5))
    half = rng.uniform(0, 10, size=(steps, n // 2, 4))
    full = np.concatenate([half, half], axis=1)
    acts = torch.from_numpy(full).cuda()
    D = gpu.obs_dim
    obs, reward, done = torch.zeros((n, D), dtype=torch.float64).cuda(), torch.zeros(n, dtype=torch.float64).cuda(), torch.zeros(n, dtype=torch.uint8).cuda()
    traded = 0.0
    for t in range(steps):
        gpu.step_device(acts[t].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
        gpu.sync()
        o, r, d = obs.cpu().numpy(), reward.cpu().numpy(), done.cpu().numpy()
        oc, rc, dc = cpu.step(full[t])
        assert np.array_equal(o[: n // 2], o[n // 2 :]) and np.array_equal(r[: n // 2], r[n // 2 :])
        assert np.array_equal(d, dc) and not d.any()
        assert np.allclose(o, oc, rtol=RTOL, atol=ATOL) and np.allclose(r, rc, rtol=0, atol=ATOL), t
        traded += (r != 0).mean()
        if t % 25 == 24 or t == steps - 1:
            _assert_portfolios(gpu, cpu, f"step {t}")
    assert traded / steps > 0.1
    _assert_fill_states(gpu, cpu, n)
    gpu.close()


def _deep_stream(i):
    lam = bench.DEEP50_RATES[i]
    from rl4mm_amd.synth import synthetic_stream

    return synthetic_stream(20180221 + i, 50, lam, 9.5 * 3600 + 1490, 9.5 * 3600 + 1800 + 60 + 1800 + 6, mid_price=2_700_000 - 350_000 * i, ticker=f"SY{i}", book_snapshot_freq="100ms")


def test_config4_shard_8192_envs_50_levels_full_episode():
    """configs[4] shard (8192 of the 65536 envs): 50-level books, four tickers at 50 / 150 / 300 / 600 messages per second in
    one batch, 100 ms snapshot frequency, and a FULL 30-minute episode (18000 steps after the 5-minute warm-up) - the run that
    shows whether the per-env capacities (price levels, queue slots, agent orders) hold on deep books: no env may raise a
    status bit, every env must report done exactly at the last step, and 256 envs (64 per ticker) must agree with the oracle
    over the whole episode: inventory / cash / price (bit-exact) and the observation after every 300 steps, the fill
    counters and resync counts at the end, the whole book of sixteen of them order by order.  The first 2100 steps are 2100
    launches of the step kernel (lobsim_step_device, what `bench.py --workload deep50` times), the rest 300-step launches of
    the rollout kernel."""
    import torch

    n, T, chunk = 8192, 18000, 300
    per_step_chunks = 7
    with mp.get_context("spawn").Pool(4) as pool:
        streams = pool.map(_deep_stream, range(4))
    cfg = bench.workload_cfg(_capi, "deep50", n)
    assert cfg.n_levels == 50
    sample = np.concatenate([np.arange(0, 128), np.arange(4096, 4224)])  # 256 envs, 64 per ticker (env i replays ticker i mod 4)
    gpu, cpu = _lib.new_sim(cfg), ol.oracle_sim(bench.workload_cfg(_capi, "deep50", len(sample)))
    starts, sids = bench.workload_starts("deep50", n, 4, 3)
    for sim, st, sd in ((gpu, starts, sids), (cpu, starts[sample], sids[sample])):
        for s in streams:
            sim.add_stream(s)
        sim.set_portfolio(0, 10**12)
        og = sim.reset(st, sd)
    D = gpu.obs_dim
    obs, reward, done = torch.zeros((n, D), dtype=torch.float64).cuda(), torch.zeros(n, dtype=torch.float64).cuda(), torch.zeros(n, dtype=torch.uint8).cuda()
    steps = torch.zeros(n, dtype=torch.int32).cuda()
    gen = torch.Generator(device="cuda").manual_seed(77)
    for c in range(T // chunk):
        acts = torch.rand((chunk, n, 4), dtype=torch.float64, device="cuda", generator=gen) * 10.0
        torch.cuda.synchronize()  # the handle launches on its own stream: the actions (torch's stream) must be complete
        if c < per_step_chunks:  # one launch of the deep-book STEP kernel per env.step(): 2100 consecutive steps of the whole shard
            for t in range(chunk):
                gpu.step_device(acts[t].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr())
            gpu.sync()
        else:
            gpu.rollout_device(_capi.POLICY_GIVEN, [], chunk, acts.data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0, steps.data_ptr())
            gpu.sync()
            assert int(steps.min()) == chunk, f"chunk {c}: an env stopped early (status bit or early done)"
        last = c == T // chunk - 1
        assert int(done.sum()) == (n if last else 0), f"chunk {c}: {int(done.sum())} envs done"
        out = ol.oracle_rollout(cpu, _capi.POLICY_GIVEN, [], chunk, acts[:, torch.from_numpy(sample).cuda()].cpu().numpy(), want_actions=False)
        ig, cg, pg = gpu.portfolios()
        ic, cc, pc = cpu.portfolios()
        assert np.array_equal(ig[sample], ic) and np.array_equal(cg[sample], cc) and np.array_equal(pg[sample], pc), f"chunk {c}"
        assert np.allclose(obs.cpu().numpy()[sample], out["obs"], rtol=RTOL, atol=ATOL), f"chunk {c}"
    worst = {"levels": 0, "agent": 0}
    for e in list(range(0, n, 64)) + list(sample):
        st = gpu.env_state(int(e))
        assert st["status"] == 0, (e, st["status"])
        worst["levels"] = max(worst["levels"], st["n_buy_levels"], st["n_sell_levels"])
        worst["agent"] = max(worst["agent"], st["n_agent_orders"])
    print("config 4 shard: deepest ladder side", worst["levels"], "levels; most live agent orders", worst["agent"])
    assert worst["agent"] <= 128
    for k, e in enumerate(sample):
        assert _fill_state(gpu, e) == _fill_state(cpu, k), e
    for k in range(0, len(sample), 16):
        assert np.array_equal(gpu.queues(int(sample[k])), cpu.queues(k)), f"env {sample[k]}: book contents differ after a full episode"
    gpu.close()
