"""Diagnostic (not part of the suite): RANDOM env configurations on RANDOM synthetic days, the HIP library against the CPU oracle,
every env compared at every step.  The suite's parity tests walk a handful of named configurations; this draws them - quote range
(`min_quote_level`, `max_quote_level`), concentration mode, market-order clearing, `enter_spread`, previous action in the
observation, step size, feature subset with random look-backs and clocks, reward kind, `outer_levels`, on-chip ladder width,
agent-order table, level budget, z-score normalisation, RollingSharpe rewards, per-step or multi-step launches, a few malformed
messages in the day - so that combinations nobody thought of get their chance (mostly on the generic kernels: only the
default configuration runs the specialised ones).  A case that differs is printed with its seed: `python tests/diag/config_fuzz.py
<seed> <seed+1>` replays it alone.
usage: python tests/diag/config_fuzz.py [first_seed=0] [end_seed=40] [envs=96]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as ol
from rl4mm_amd import _capi, _lib
from rl4mm_amd.synth import synthetic_stream

RTOL, ATOL = 1e-12, 1e-9
STATE_KEYS = ("now_us", "n_agent_orders", "n_resyncs", "min_buy_price", "max_sell_price", "status", "agent_fill_count", "agent_buy_volume", "agent_sell_volume",
              "ext_buy_count", "ext_sell_count", "ext_buy_volume", "ext_sell_volume", "n_buy_levels", "n_sell_levels", "best_buy", "best_sell")  # fmt: skip


def draw_case(seed: int, n: int):
    r = np.random.Generator(np.random.PCG64(90_000 + seed))
    F = _capi.make_feature
    step_us = int(r.choice([50_000, 100_000, 100_000, 100_000, 200_000, 250_000, 500_000, 1_000_000]))
    ep_steps = int(r.integers(15, 60))
    episode_us = ep_steps * step_us
    n_levels = int(r.choice([5, 10, 10, 20, 50]))
    rate = float(r.choice([40, 120, 200, 400, 900]))
    clock = lambda: int(r.choice([step_us, step_us, 2 * step_us, 1_000_000 if 1_000_000 % step_us == 0 else step_us]))
    # every feature's window must end inside the day's first minute and a half (the stream starts 100 s before the first episode)
    lb = lambda u, hi: int(r.integers(1, max(2, min(hi, 60_000_000 // u))))
    pool = [
        lambda: F(_capi.F_SPREAD, 0, clock(), 0, float(r.choice([300, 5000]))),
        lambda: (lambda u: F(_capi.F_PRICE_MOVE, lb(u, 30), u, -1e4, 1e4))(clock()),
        lambda: (lambda u: F(_capi.F_VOLATILITY, lb(u, 80), u, 0, 1.0))(clock()),
        lambda: F(_capi.F_INVENTORY, 0, clock(), -float(r.choice([50, 1e6])), float(r.choice([50, 1e6]))),
        lambda: F(_capi.F_EPISODE_PROPORTION, 0, step_us, 0, 1.0, dparam=step_us / episode_us),
        lambda: F(_capi.F_TIME_OF_DAY, 0, int(r.choice([step_us, 60_000_000])), 0, 9, iparam=int(r.choice([1_980_000_000, 7_000_000, 600_000_000]))),
        lambda: (lambda u: F(_capi.F_TRADE_DIRECTION_IMBALANCE, lb(u, 60), u, -1, 1, iparam=int(r.integers(0, 2))))(clock()),
        lambda: (lambda u: F(_capi.F_TRADE_VOLUME_IMBALANCE, lb(u, 60), u, -1, 1, iparam=int(r.integers(0, 2))))(clock()),
        lambda: F(_capi.F_BOOK_IMBALANCE, 0, clock(), -1, 1),
        lambda: (lambda u: F(_capi.F_PRICE_RANGE, lb(u, 12), u, 0, 1e4))(clock()),
        lambda: F(_capi.F_PRICE, 0, clock(), 0, 1e8),
        lambda: (lambda u, s: F(_capi.F_AMIHUD_LAMBDA, s * int(r.integers(2, 6)), u, 0, 1.0, iparam=s))(step_us, int(r.integers(1, 4))),
    ]
    k = int(r.integers(2, 13))
    feats = [pool[i]() for i in sorted(r.choice(len(pool), size=min(k, len(pool)), replace=False))]
    if r.random() < 0.3:
        feats += [pool[int(r.integers(1, 3))]() for _ in range(int(r.integers(1, 4)))]  # second and third windows of the same kind
    feats = feats[: _capi.MAX_FEATURES]
    extras = r.random()  # (drawn for every case so that a seed's other draws do not depend on the mode)
    norm_lens = r.choice([2, 7, 40, 600], size=len(feats))
    norm_on = r.random(len(feats)) < 0.5
    if 0.70 <= extras < 0.85:  # normalisation_on (Features.py:60-74): some features z-scored against a history that wraps or only grows
        for f, ln_, on in zip(feats, norm_lens, norm_on):
            if on: f.norm_len = int(ln_)
    # the longest window is the warm-up, and an episode's replay starts `warm-up` before a whole second: the reference asserts that
    # this is a whole second too (OrderbookSimulator.py:61), so the longest window is stretched to one
    w = max(range(len(feats)), key=lambda i: feats[i].lookback * feats[i].update_us)
    while (feats[w].lookback * feats[w].update_us) % 1_000_000 != 0 or any(f.lookback * f.update_us > feats[w].lookback * feats[w].update_us for f in feats):
        feats[w].lookback += int(feats[w].iparam) if feats[w].kind == _capi.F_AMIHUD_LAMBDA else 1
    minq = int(r.choice([0, 0, 1, 2]))
    q = int(r.choice([2, 3, 5, 8, 10, 10, 12, 16, 18]))
    conc = float(r.choice([5.0, 11.0, 30.0])) if r.random() < 0.3 else None
    moc = bool(r.random() < 0.35)
    rewards = [(_capi.R_PNL, 0, 0.0), (_capi.R_INVENTORY_ADJUSTED, 1, 0.1), (_capi.R_INVENTORY_ADJUSTED, 0, float(r.choice([0.01, 0.5])))]
    rs = (_capi.R_ROLLING_SHARPE, 0, 0.0, int(r.choice([8, 32, 120])), int(r.choice([2, 5, 8])))  # (kind, -, -, max_window, min_window)
    sharpe = extras >= 0.85
    table = int(r.choice([0, 0, 128, 256])) if q <= 10 else int(r.choice([0, 256]))
    kw = dict(step_us=step_us, episode_us=episode_us, outer_levels=int(r.choice([3, 10, 20, 40])), min_quote_level=minq, max_quote_level=minq + q, concentration=conc,
              market_order_clearing=moc, market_order_fraction_of_inventory=float(r.choice([0.25, 0.5, 1.0])) if moc else 0.0, enter_spread=bool(r.random() < 0.4),
              inc_prev_action_in_obs=bool(r.random() < 0.4), per_step_reward=rewards[int(r.integers(0, 3))], terminal_reward=rewards[int(r.integers(0, 3))],
              ladder_levels=int(r.choice([0, 0, 64, 128])) if n_levels <= 64 else int(r.choice([0, 128])), max_levels=int(r.choice([0, 0, 64, 256])), agent_orders=table)  # fmt: skip
    if sharpe:  # RollingSharpe per step, at the end, or both (RewardFunctions.py:38-94)
        which = int(r.integers(0, 3))
        if which != 1: kw["per_step_reward"] = rs
        if which != 0: kw["terminal_reward"] = rs
    cfg = _capi.make_cfg(n, n_levels, feats, **kw)
    warm_s = -(-cfg.warmup_us // 1_000_000)
    t0 = 35_600  # 09:53:20
    total_s = warm_s + 3 * (episode_us // 1_000_000 + 2) + 40  # (an env that stops early - a clearing order of volume 0 is the reference's `assert order.volume > 0` - restarts somewhere in here)
    if r.random() < 0.25:  # a day whose price walks away: far levels, outer-level resyncs (tests/drift_stream.py)
        from drift_stream import drift_stream, ramp_path, triangle_path

        if r.random() < 0.5: drift = ("ramp", float(r.choice([0.2, 0.5, 2.0])), int(r.choice([-1, 1])))
        else: drift = ("triangle", int(r.choice([15, 40, 120])), float(r.choice([0.1, 0.3, 1.0])))
        path = ramp_path(drift[1], drift[2]) if drift[0] == "ramp" else triangle_path(drift[1], drift[2])
        drift += (float(r.choice([12.0, 60.0])), int((total_s + 5) / 0.1) + 200, t0, total_s + 5)  # noise per second, max excursion, start, duration
        stream = drift_stream(7_000 + seed, path, n_levels=n_levels, start_s=t0, duration_s=drift[6], noise_per_s=drift[3], max_excursion=drift[4])
        rate = -1.0
    else:
        stream = synthetic_stream(7_000 + seed, n_levels=n_levels, msgs_per_s=rate, start_s=t0, end_s=t0 + total_s + 5, mid_price=int(r.choice([60_000, 2_700_000, 2_700_000, 9_990_000])))
    drift = locals().get("drift")
    bad_msgs = 0
    if r.random() < 0.12 and stream.n_msgs > 2000:  # a cross trade / trading halt / zero-volume message somewhere in the day: the reference
        from rl4mm_amd.stream import TYPE_SHIFT  # asserts (HistoricalOrderGenerator.py:54-56, Exchange.py:59-60), every env that replays it stops

        for k in r.integers(stream.n_msgs // 3, stream.n_msgs, size=int(r.integers(1, 3))):
            kind = int(r.integers(0, 3))
            if kind == 2: stream.msgs[k, 3] &= ~((1 << TYPE_SHIFT) - 1)  # volume 0
            else: stream.msgs[k, 3] = (stream.msgs[k, 3] & ~(0xF << TYPE_SHIFT)) | ((6 + kind) << TYPE_SHIFT)
            bad_msgs += 1
    first = t0 + warm_s + 2
    starts = (first + r.integers(0, 30, size=n)) * 1_000_000
    amax = float(r.choice([10.0, 10.0, 3.0, 50.0]))  # (max_distribution_param of the Box: the reference default is 10)
    chunk = int(r.integers(2, 8)) if r.random() < 0.35 else 0  # steps per multi-step launch (lobsim_rollout_device); 0 = one launch per step
    # dimensions added later draw from a generator of their own: the cases of the seeds that found something stay what they were
    r2 = np.random.Generator(np.random.PCG64(190_000 + seed))
    streams = [stream]
    for k in range(int(r2.choice([0, 0, 0, 1, 2]))):  # more days on the handle: every env replays one of them, redrawn at its resets
        streams.append(synthetic_stream(17_000 + 10 * seed + k, n_levels=n_levels, msgs_per_s=float(r2.choice([40, 200, 600])), start_s=t0, end_s=t0 + total_s + 5,
                                        mid_price=int(r2.choice([60_000, 2_700_000])), book_snapshot_freq=str(r2.choice(["S", "S", "100ms"]))))
    policy = (_capi.POLICY_GIVEN, [])
    if chunk and r2.random() < 0.4:  # the policy of a multi-step launch computed on the device (baseline_agents.py:21-115)
        if r2.random() < 0.5: policy = (_capi.POLICY_FIXED, r2.uniform(0.2, amax, size=cfg.action_dim).tolist())
        else:
            inv_idx = next((i for i, f in enumerate(feats) if f.kind == _capi.F_INVENTORY), 0)
            mi = float(r2.choice([20, 150, 1000]))
            policy = (_capi.POLICY_TERADACTYL, [mi, float(r2.uniform(3, 9)), float(r2.choice([0.3, 0.5, 0.8])), float(r2.uniform(9, 12)), float(r2.choice([1.0, 2.0])), inv_idx, float(moc), mi])
    masked = (not chunk) and r2.random() < 0.2  # lobsim_step's `active` mask: a tenth of the envs sits a step out
    # SimpleInfoCalculator's arrays written by the step kernel's epilogue (lobsim_set_info_buffer) against the package's host mirror of
    # the calculator working on the ORACLE's state (InfoCalculators.py:17-89); the kernel takes the env's own quote range, so level 0 up
    with_info = (not chunk) and (not masked) and minq == 0 and r2.random() < 0.35
    # round 6 dimensions (a generator of their own again): how the step is CALLED - the host-pointer entry (lobsim_step), the pinned
    # host-io entry with f32 observations (lobsim_step_host_io), or the device entry a torch learner uses (lobsim_step_device: device
    # buffers, f32 observations) - resets through lobsim_reset_device (first observations into the caller's device buffers, done flags
    # cleared on the device), and a day that leaves the device and comes back while no env replays it (lobsim_drop_stream: the stream
    # budget of the Python env evicts like that)
    r3 = np.random.Generator(np.random.PCG64(290_000 + seed))
    call = "host" if (chunk or masked or with_info) else str(r3.choice(["host", "host_io_f32", "device_f32"]))
    reset_dev = (not chunk) and bool(r3.random() < 0.4)
    evict = len(streams) > 1 and bool(r3.random() < 0.5)
    desc = dict(kw, call=call, reset_dev=reset_dev, evict=evict, n_levels=n_levels, rate=rate, feats=[(f.kind, f.lookback, f.update_us, f.norm_len) for f in feats], amax=amax, ep_steps=ep_steps, chunk=chunk, bad_msgs=bad_msgs,
                sharpe=sharpe, normalised=int(sum(f.norm_len > 0 for f in feats)), days=len(streams), policy=policy[0], masked=masked, drift=drift, with_info=with_info)
    extra = dict(streams=streams, policy=policy, masked=masked, r2=r2, with_info=with_info, call=call, reset_dev=reset_dev, evict=evict, r3=r3)
    return cfg, extra, starts, amax, ep_steps, desc, r, (first, total_s - warm_s - episode_us // 1_000_000 - 6), chunk


def _rollout(sim, acts, policy=(_capi.POLICY_GIVEN, [])):
    """acts [T, n, A] through one multi-step launch -> (obs, reward, done, (steps taken, return))."""
    T, n = acts.shape[0], acts.shape[1]
    if sim.b.prefix == "oracle_":
        out = ol.oracle_rollout(sim, policy[0], policy[1], T, acts if policy[0] == _capi.POLICY_GIVEN else None, want_actions=False)
        return out["obs"], out["reward"], out["done"], (out["steps"], out["ret"])
    import torch

    dev = torch.device("cuda", 0)
    d_act = torch.from_numpy(np.ascontiguousarray(acts)).to(dev)
    obs, rew, done = torch.zeros((n, sim.obs_dim), dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev)
    ret, steps = torch.zeros(n, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)
    sim.rollout_device(policy[0], policy[1], T, d_act.data_ptr() if policy[0] == _capi.POLICY_GIVEN else 0, 0, obs.data_ptr(), 0, rew.data_ptr(), done.data_ptr(), ret.data_ptr(), steps.data_ptr())
    sim.sync()
    return obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), (steps.cpu().numpy(), ret.cpu().numpy())


class _DeviceIO:
    """The step / reset entries a caller with device or pinned memory uses, for run_case: `device_f32` = lobsim_step_device with torch tensors
    (f32 observations), `host_io_f32` = lobsim_step_host_io on the handle's pinned buffers (f32 observations), `host` = lobsim_step (only
    the resets go through lobsim_reset_device then)."""

    def __init__(self, sim, n, A, call):
        import torch

        self.torch, self.call, self.n, self.A = torch, call, n, A
        self.f32 = call != "host"
        dev = torch.device("cuda", 0)
        D = sim.obs_dim
        self.act = torch.zeros((n, A), dtype=torch.float64, device=dev)
        self.obs = torch.zeros((n, D), dtype=torch.float64, device=dev)
        self.obs32 = torch.zeros((n, D), dtype=torch.float32, device=dev)
        self.rew = torch.zeros(n, dtype=torch.float64, device=dev)
        self.done = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.io = sim.host_io() if call == "host_io_f32" else None

    def step(self, sim, act, active=None):
        if self.call == "host":
            return sim.step(act, active)
        if self.call == "host_io_f32":
            self.io["actions"][:, : self.A] = act
            sim.step_host_io(False, 2)
            return self.io["obs_f32"].astype(np.float64), self.io["reward"].copy(), self.io["done"].copy()
        self.act.copy_(self.torch.from_numpy(np.ascontiguousarray(act)))
        self.torch.cuda.synchronize()
        sim.step_device(self.act.data_ptr(), 0, 0, self.obs32.data_ptr(), self.rew.data_ptr(), self.done.data_ptr(), 0)
        sim.sync()
        return self.obs32.cpu().numpy().astype(np.float64), self.rew.cpu().numpy(), self.done.cpu().numpy()

    def reset(self, sim, starts, sids, ids):
        """lobsim_reset_device: the observations land in the rows of the reset envs, their done flags are cleared on the device."""
        self.done.fill_(1)
        self.torch.cuda.synchronize()
        sim.reset_device(starts, sids, ids, 0 if self.f32 else self.obs.data_ptr(), self.obs32.data_ptr() if self.f32 else 0, self.done.data_ptr())
        sim.sync()
        d = self.done.cpu().numpy()
        assert not d[ids].any() and d[np.setdiff1d(np.arange(self.n), ids)].all(), "reset_device must clear the done flags of exactly the envs it reset"
        o = (self.obs32 if self.f32 else self.obs).cpu().numpy().astype(np.float64)
        return o[ids]


def run_case(seed: int, n: int, new_gpu=None):
    """`new_gpu(cfg)`: the handle under test (default: the product library; tests/test_gpu_bounds.py passes the bounds build)."""
    cfg, extra, starts, amax, ep_steps, desc, r, (first, span), chunk = draw_case(seed, n)
    streams, policy, masked, r2 = extra["streams"], extra["policy"], extra["masked"], extra["r2"]
    sids = r2.integers(0, len(streams), size=n).astype(np.int32)
    # RollingSharpe is mean / std of a window's log returns.  With a portfolio of 1e9 in cash and a few shares the AUM moves by O(1) a
    # step: log returns of 1e-9, which f64 resolves to 1e-7 relative - the last bits in which ocml's exp / log differ from libm's (and
    # both from the reference's numpy: 4.4e-8 on the recorded trajectory, tests/test_gpu_parity.py) show at 1e-6 in the ratio.  That is
    # the formula's conditioning, not a difference of the paths: every other reward is compared to 1e-9 absolute
    r_rtol, r_atol = (1e-5, 1e-6) if desc["sharpe"] else (0.0, ATOL)
    gpu, cpu = (ol.oracle_sim(cfg) if os.environ.get("FUZZ_SELF") else (new_gpu or _lib.new_sim)(cfg)), ol.oracle_sim(cfg)  # (FUZZ_SELF=1: the harness against itself, no GPU)
    try:
        inv0 = int(r.integers(-20, 20))
        for s in (gpu, cpu):
            for st_ in streams: s.add_stream(st_)
            s.set_portfolio(inv0, 10**9)
        og, oc = gpu.reset(starts, sids), cpu.reset(starts, sids)
        sg, sc = gpu.status(), cpu.status()
        lost = (sg & 28) != 0  # (a warm-up that outgrew a capacity of the HIP layout: see below)
        reset_failed = sc != 0
        if not np.array_equal(sg[~lost], sc[~lost]):
            return f"first reset: status differs: {sg.tolist()} vs {sc.tolist()}", desc
        og[lost | reset_failed], oc[lost | reset_failed] = 0.0, 0.0
        if not np.allclose(og, oc, rtol=RTOL, atol=ATOL, equal_nan=True):
            return f"reset obs differ by {np.nanmax(np.abs(og - oc))}", desc
        calc_dev = calc_host = None
        if extra["with_info"] and not os.environ.get("FUZZ_SELF"):
            from rl4mm_amd.env import SimpleInfoCalculator

            ikw = dict(concentration=cfg.concentration if cfg.use_concentration else None, quote_levels=cfg.max_quote_level - cfg.min_quote_level)
            calc_dev = SimpleInfoCalculator(cfg.market_order_fraction_of_inventory, bool(cfg.enter_spread), **ikw).attach(gpu)
            calc_host = SimpleInfoCalculator(cfg.market_order_fraction_of_inventory, bool(cfg.enter_spread), **ikw)
            if calc_dev._device_rows is None:
                return "the info calculator did not attach to the device", desc
        A = cfg.action_dim
        high = np.full(A, amax)
        if cfg.market_order_clearing:
            high[-1] = float(r.choice([5.0, 60.0, 400.0]))
        steps = 2 * ep_steps + 8
        if chunk: steps = -(-steps // chunk)
        resets = flagged_steps = evictions = 0
        dev_io = None
        if not os.environ.get("FUZZ_SELF") and (extra["call"] != "host" or extra["reset_dev"]):
            dev_io = _DeviceIO(gpu, n, cfg.action_dim, extra["call"])
        capacity_stops = int(lost.sum())
        for t in range(steps):
            T = chunk or 1
            act = r.uniform(0, high, size=(T, n, A))
            m = r.random((T, n))
            act[m < 0.03] = 0.0  # (the Box's corners: the distributor adds EPS)
            act[(m >= 0.03) & (m < 0.06)] = high
            active = None
            if chunk:
                (og, rg, dg, xg), (oc, rc, dc, xc) = _rollout(gpu, act, policy), _rollout(cpu, act, policy)
            else:
                if masked: active = (r2.random(n) >= 0.1).astype(np.uint8)
                oc, rc, dc = cpu.step(act[0], active)
                if dev_io is None: og, rg, dg = gpu.step(act[0], active)
                else:
                    og, rg, dg = dev_io.step(gpu, act[0], active)
                    if dev_io.f32: oc = oc.astype(np.float32).astype(np.float64)  # (the kernel writes the observation_space's dtype: the f64 value rounded once)
                if masked:  # (an env that sits the step out has no row in it)
                    for a in (og, oc, rg, rc, dg, dc): a[active == 0] = 0
            sg, sc = gpu.status(), cpu.status()
            # a capacity of the HIP layout (level budget, queue pages, agent table: bits 4 / 8 / 16) is not a difference: the oracle
            # has none (include/lobsim.h).  Such an env leaves the comparison until its next reset
            capacity_stops += int((((sg & 28) != 0) & ~lost).sum())
            lost |= (sg & 28) != 0
            if not np.array_equal(sg[~lost], sc[~lost]):
                e = int(np.nonzero((sg != sc) & ~lost)[0][0])
                return f"step {t}: status differs in env {e}: {sg[e]} vs {sc[e]}", desc
            skip = lost | reset_failed  # (an env whose RESET failed has no observation at all: include/lobsim.h)
            if chunk and (not np.array_equal(xg[0][~lost], xc[0][~lost]) or not np.allclose(xg[1][~lost], xc[1][~lost], rtol=max(r_rtol, 1e-12), atol=max(r_atol, 1e-6), equal_nan=True)):
                return f"launch {t}: steps taken / returns of the multi-step launch differ: envs {np.nonzero((xg[0] != xc[0]) & ~lost)[0][:6].tolist()}, {np.nanmax(np.abs(xg[1] - xc[1])[~lost])}", desc
            for a in (og, oc, rg, rc): a[skip] = 0.0
            dg[lost], dc[lost] = 0, 0
            flagged_steps += int((sg != 0).sum())
            ig, cg, pg = gpu.portfolios()
            ic, cc, pc = cpu.portfolios()
            for what, a, b in (("inventory", ig, ic), ("cash", cg, cc), ("done", dg, dc), ("microprice", pg, pc)):
                a, b = np.where(lost, 0, a), np.where(lost, 0, b)
                if not np.array_equal(a, b):
                    e = int(np.nonzero(a != b)[0][0])
                    return f"step {t}: {what} differs in env {e}: {a[e]} vs {b[e]} (status gpu {gpu.env_state(e)['status']} / oracle {cpu.env_state(e)['status']})", desc
            if not np.allclose(og, oc, rtol=RTOL, atol=ATOL, equal_nan=True):  # (NaN: the z-score of a constant history, both sides)
                e, f = np.unravel_index(np.nanargmax(np.abs(og - oc)), og.shape)
                return f"step {t}: obs differ by {np.abs(og - oc).max()} (env {e}, entry {f}; status gpu {gpu.status()[e]} / oracle {cpu.status()[e]})\n       gpu    {og[e].tolist()}\n       oracle {oc[e].tolist()}", desc
            if calc_dev is not None:
                a_dev, a_host = calc_dev.calculate_arrays(gpu, act[0]), calc_host.calculate_arrays(cpu, act[0])
                live = (sg == 0) & ~skip
                for col in ("asset_price", "inventory", "cash", "aum", "market_spread", "agent_spread", "agent_weighted_spread_value", "midprice_offset", "weighted_midprice_offset"):
                    x, y = np.asarray(a_dev[col], dtype=np.float64)[live], np.asarray(a_host[col], dtype=np.float64)[live]
                    if not np.allclose(x, y, rtol=1e-12, atol=1e-9, equal_nan=True):
                        e = int(np.nonzero(live)[0][np.nanargmax(np.abs(x - y))])
                        return f"step {t}: info column {col} differs in env {e}: device {a_dev[col][e]!r} vs host mirror on the oracle {a_host[col][e]!r}", desc
            if not np.allclose(rg, rc, rtol=r_rtol, atol=r_atol, equal_nan=True):  # (NaN: a RollingSharpe window of equal values, both sides)
                return f"step {t}: reward differs by {np.abs(rg - rc).max()}", desc
            if dg.any() or lost.any():
                ids = np.nonzero((dg != 0) | lost)[0].astype(np.int32)
                resets += len(ids)
                st_before = sg
                starts[ids] = (first + r.integers(0, span, size=len(ids))) * 1_000_000
                if lost.any():  # (the portfolio outlives the episode, Q7: an env that left the comparison re-enters it with the oracle's)
                    li = np.nonzero(lost)[0].astype(np.int32)
                    gpu.set_portfolio(ic[li], cc[li], li)
                sids[ids] = r2.integers(0, len(streams), size=len(ids))
                o2 = cpu.reset(starts[ids], sids[ids], ids)
                if dev_io is not None and extra["reset_dev"]: o1 = dev_io.reset(gpu, starts[ids], sids[ids], ids)
                else: o1 = gpu.reset(starts[ids], sids[ids], ids)
                if extra["evict"] and not os.environ.get("FUZZ_SELF"):  # a day no env replays any more (after this reset) leaves the device and comes back, into the slot it had: the lowest free one
                    for k in range(len(streams)):
                        if not (sids == k).any():
                            gpu.drop_stream(k)
                            back = gpu.add_stream(streams[k])
                            if back != k: return f"step {t}: stream {k} came back as {back}", desc
                            evictions += 1
                            break
                if dev_io is not None and extra["reset_dev"] and dev_io.f32: o2 = o2.astype(np.float32).astype(np.float64)
                s1, s2 = gpu.status()[ids], cpu.status()[ids]
                cap = (s1 & 28) != 0  # (a warm-up that outgrew a capacity)
                capacity_stops += int(cap.sum())
                if not np.array_equal(s1[~cap], s2[~cap]):
                    return f"step {t}: status after the reset differs: {s1.tolist()} vs {s2.tolist()}", desc
                o1[(s1 != 0) | (s2 != 0)], o2[(s1 != 0) | (s2 != 0)] = 0.0, 0.0
                lost[ids] = cap | (lost[ids] & desc["sharpe"])  # (the RollingSharpe window outlives the episode - env.reset never resets the reward function - so an env that left the comparison cannot re-enter it)
                reset_failed[ids] = s2 != 0
                if not np.allclose(o1, o2, rtol=RTOL, atol=ATOL, equal_nan=True):
                    k, f = np.unravel_index(np.nanargmax(np.abs(o1 - o2)), o1.shape)
                    return f"step {t}: obs after reset differ by {np.abs(o1 - o2).max()} (env {ids[k]}, entry {f}; status before the reset {st_before[ids[k]]})\n       gpu    {o1[k].tolist()}\n       oracle {o2[k].tolist()}", desc
        for e in np.nonzero(~lost)[0]:
            sg, sc = gpu.env_state(int(e)), cpu.env_state(int(e))
            for k in STATE_KEYS:
                if sg[k] != sc[k]:
                    return f"end: env {e} {k} {sg[k]} vs {sc[k]}", desc
        for e in range(0, n, 7):
            if not lost[e] and not np.array_equal(gpu.queues(e), cpu.queues(e)):
                return f"end: env {e}: per-order book contents differ", desc
        st = np.array([gpu.env_state(e)["status"] for e in range(n)])
        if gpu.b.prefix == "lobsim_":  # (the diagnostic build's sentinel zones; the product library has none: 0 of 0)
            zones, damaged = gpu.check_redzones()
            if damaged:
                return f"end: {damaged} of {zones} red zones damaged: {gpu.b.fn['last_error'](gpu.h).decode()}", desc
            desc = dict(desc, red_zones=zones)
        desc = dict(desc, evictions=evictions, resets=resets, flagged=flagged_steps, capacity_stops=capacity_stops, max_agent_orders=max(gpu.env_state(e)["n_agent_orders"] for e in range(n)), inv_max=int(np.abs(ig).max()))
        return None, desc
    finally:
        gpu.close()
        cpu.close()


BOUNDS_SIM = None
if os.environ.get("FUZZ_BOUNDS"):  # FUZZ_BOUNDS=1: the handle under test is the diagnostic build (liblobsim_bounds.so: bounds checks + red zones)
    _bb = _lib.bounds_binding()
    BOUNDS_SIM = lambda cfg: _capi.Sim(_bb, cfg)

if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    end = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 96
    bad = 0
    for seed in range(first, end):
        t = time.time()
        try:
            err, desc = run_case(seed, n, new_gpu=BOUNDS_SIM)
        except Exception as ex:  # a configuration one side refuses: both must refuse it (reported, not counted as a difference)
            err, desc = f"EXCEPTION {type(ex).__name__}: {ex}", {}
        short = {k: desc[k] for k in ("step_us", "n_levels", "rate", "min_quote_level", "max_quote_level", "concentration", "market_order_clearing", "enter_spread", "ladder_levels",
                                        "agent_orders", "max_levels", "outer_levels", "chunk", "bad_msgs", "sharpe", "normalised", "days", "policy", "masked", "with_info", "call", "reset_dev", "evictions", "resets", "flagged", "capacity_stops", "max_agent_orders", "inv_max") if k in desc}  # fmt: skip
        print(f"seed {seed:4d} {'OK  ' if err is None else 'DIFF'} {time.time() - t:5.1f}s {short}" + (f"\n    -> {err}\n    {desc}" if err else ""), flush=True)
        bad += err is not None
    print(f"{end - first} cases, {bad} with a difference or an exception")
    sys.exit(1 if bad else 0)
