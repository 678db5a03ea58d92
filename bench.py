#!/usr/bin/env python3
"""bench.py - env-steps/s of the batched LOB simulation step on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: every env of the rank converts its action to orders, replays
its 100 ms message window, updates features and reward (one `lobsim_step_device` call = one env.step() per env).
Workload at N=1 = BASELINE.json configs[1]: 4096 vectorised envs on one MI355X, synthetic 10-level SPY-shaped stream
with 1 s snapshots, random Beta-policy actions (uniform in Box[0,10]^4, PCG64(42)), default `full_state` features,
PnL reward.  The other configs run as `--workload` shards (they are parity-test cases first, tests/test_gpu_parity.py):
  teradactyl  configs[2] shape: 4096 envs, on-device Teradactyl policy, PnL (SPY-shaped synthetic; real SPY absent)
  ad          configs[3] shard: 4096 envs, random actions, asymmetrically dampened per-step reward
  deep50      configs[4] shard: 8192 envs, 50-level books, 4 tickers (50/150/300/600 msg/s), 100 ms snapshots

`--gpus N` (the reference's num_workers knob, rl4mm/helpers/main_helper.py:262-263): when not already running under
torch.distributed.run, this process starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
child BEFORE touching the GPU and relays its output; every rank owns `--envs` envs of its own (weak scaling; envs
are independent, the stream is replicated), the only collective is the episode-return gather after the timed loop.

Inputs (actions, stream, env state) are resident in HBM when the timed region starts.  Prints ONE JSON line.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md
ENVS_PER_GPU = 4096
STEP_US = 100_000
EPISODE_US = 1_800_000_000  # 30 min, the reference default
PROFILE_ROUND = "r6"
REPEATS = 5  # timed regions of --steps steps per invocation; `value` is the median region

WORKLOADS = {
    "random": "4096 vectorised envs per GPU, synthetic 10-level SPY-shaped LOBSTER stream @1s snapshots, random Beta-policy actions (BASELINE.json configs[1])",
    "teradactyl": "4096 envs per GPU, synthetic 10-level SPY-shaped stream, on-device Teradactyl rule-based agent, PnL reward (BASELINE.json configs[2] shape; real SPY 2018-02-21 data absent)",
    "ad": "4096 envs per GPU, synthetic 10-level stream, random actions, AD (asymmetrically dampened) per-step reward (BASELINE.json configs[3] shard)",
    "deep50": "8192 envs per GPU, 50-level books, 4 tickers at 50/150/300/600 msg/s, 100 ms snapshots (BASELINE.json configs[4] shard)",
    "rs": "4096 envs per GPU, synthetic 10-level stream, random actions, RollingSharpe per-step and terminal reward (env_config -psr RS: the kernel variants that carry RollingSharpe / normalisation)",
    "norm": "4096 envs per GPU, synthetic 10-level stream, random actions, default features with normalisation_on (env_config -n True: every feature update z-scores its value against the feature's whole history, Features.py:60-74 - the one HBM-bound configuration of the path)",
}
# max_norm_len of the reference's default features (rl4mm/features/Features.py: Spread 10 000, the others 100 000), in the order of default_features()
DEFAULT_MAX_NORM_LEN = (10_000, 100_000, 100_000, 100_000, 100_000, 100_000, 100_000, 100_000, 100_000, 100_000)
WARMUP_US = 300_000_000  # the longest window of the default features (5 min Volatility)
DEEP50_RATES = (50.0, 150.0, 300.0, 600.0)
TERADACTYL_PARAMS = [1000, 10.0, 0.5, 10.0, 1.0, 5, 0, 1000]  # max_inventory, kappa, omega, max_kappa, exponent, inventory feature index, no MOC, denom


def default_features(capi, normalisation_on=False):
    """HistoricalOrderbookEnvironment.get_default_features (HistoricalOrderbookEnvironment.py:395-440).  normalisation_on: every feature
    carries its z-score history, sized like rl4mm_amd/env.py:Feature.desc_for sizes it - what warm-up + episode can append."""
    F = capi.make_feature
    ep = STEP_US / EPISODE_US
    feats = _default_features(F, capi, ep)
    if normalisation_on:
        for f, mx in zip(feats, DEFAULT_MAX_NORM_LEN):
            f.norm_len = min(mx, (WARMUP_US + EPISODE_US) // f.update_us + 4)
    return feats


def _default_features(F, capi, ep):
    return [
        F(capi.F_SPREAD, 0, STEP_US, 0, 5000),
        F(capi.F_PRICE_MOVE, 1, 100_000, -1e4, 1e4),
        F(capi.F_PRICE_MOVE, 10, 1_000_000, -1e4, 1e4),
        F(capi.F_VOLATILITY, 600, 100_000, 0, 1.0),
        F(capi.F_VOLATILITY, 300, 1_000_000, 0, 1.0),
        F(capi.F_INVENTORY, 0, STEP_US, -1e6, 1e6),
        F(capi.F_EPISODE_PROPORTION, 0, STEP_US, 0, 1.0, dparam=ep),
        F(capi.F_TIME_OF_DAY, 0, 60_000_000, 0, 9, iparam=1_980_000_000),
        F(capi.F_TRADE_DIRECTION_IMBALANCE, 600, 100_000, -1, 1),
        F(capi.F_TRADE_VOLUME_IMBALANCE, 600, 100_000, -1, 1),
    ]


def build_streams(workload, sim_seconds):
    """Synthetic LOBSTER streams of the workload: 09:54:50 (a 300 s feature warm-up before 10:00:00) up to the last
    start (10:00:59) + sim_seconds."""
    from rl4mm_amd.synth import synthetic_stream

    t0, t1 = 9.5 * 3600 + 1490, 9.5 * 3600 + 1800 + 60 + sim_seconds + 5
    if workload == "deep50":
        return [synthetic_stream(20180221 + i, 50, lam, t0, t1, mid_price=2_700_000 - 350_000 * i, ticker=f"SY{i}", book_snapshot_freq="100ms")
                for i, lam in enumerate(DEEP50_RATES)]  # fmt: skip
    return [synthetic_stream(20180221, 10, 200.0, t0, t1)]


def build_stream(sim_seconds):
    """The configs[1] stream alone (tools/)."""
    return build_streams("random", sim_seconds)[0]


def shared_streams(workload, sim_seconds, rank, dist):
    """The streams of a multi-rank job (one node): rank 0 generates them once and parks them as .npz, the others load them
    (eight generators next to eight ranks on a box that grants 16 CPUs made every rank's set-up - and its first timed
    steps - jittery).  One rank: just build."""
    if dist is None:
        return build_streams(workload, sim_seconds)
    import tempfile

    from rl4mm_amd.stream import PackedStream

    n = len(DEEP50_RATES) if workload == "deep50" else 1
    d = Path(tempfile.gettempdir()) / f"lobsim_bench_{os.environ.get('MASTER_PORT', '0')}_{workload}_{sim_seconds}"
    # (the hand-over goes through a node-local directory: one node, as the contract says - `--nnodes=1`)
    assert int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))) == int(os.environ.get("WORLD_SIZE", "1")), "bench.py runs the ranks of ONE node"
    if rank == 0:
        d.mkdir(parents=True, exist_ok=True)
        for i, s in enumerate(build_streams(workload, sim_seconds)):
            s.save(d / f"stream{i}.npz")
    dist.barrier()
    streams = [PackedStream.load(d / f"stream{i}.npz") for i in range(n)]
    dist.barrier()
    if rank == 0:
        import shutil

        shutil.rmtree(d, ignore_errors=True)
    return streams


def timed_regions(step, K, repeats, first_step, dist, sync, make_events=None):
    """`repeats` regions of exactly K steps, each bracketed by a barrier + device synchronise on both sides.  Per region the
    job's time is the MAX over ranks of the rank's own clock around its K steps (all ranks released together by the opening
    barrier; the closing barrier's own 50-100 us of RCCL launch is outside every rank's clock).  Returns (per-region job
    times [R] in seconds, this rank's times [R], per-region event times in ms or None)."""
    import torch

    local, ev_ms = [], []
    # everything the loop needs is made BEFORE it (event objects, the clock function): between two regions the host does the
    # bracket's barrier + synchronise and nothing else, so a region does not start on a GPU that idled through Python bookkeeping
    all_evs = [make_events() for _ in range(repeats)] if make_events else None
    clock = time.perf_counter
    gc.collect()
    gc.disable()  # (a cyclic collection inside a 0.7 ms region would be a third of it)
    for r in range(repeats):
        evs = all_evs[r] if all_evs else None
        if dist:
            dist.barrier()
        sync()
        if evs:
            evs[0].record()  # on the idle stream, just ahead of the clock: instrumentation, not steps
        t0 = clock()
        for t in range(first_step + r * K, first_step + (r + 1) * K):
            step(t)
        if evs:
            evs[1].record()
        sync()
        local.append(clock() - t0)
        if dist:
            dist.barrier()  # the closing side of the bracket: every rank is done before anything else happens
        sync()
    gc.enable()
    if all_evs:
        ev_ms = [e[0].elapsed_time(e[1]) for e in all_evs]
    mine = torch.tensor(local, dtype=torch.float64)
    job = mine.clone()
    if dist:
        dev = None if dist.get_backend() == "gloo" else torch.device("cuda", torch.cuda.current_device())
        job = job.to(dev) if dev else job
        dist.all_reduce(job, op=dist.ReduceOp.MAX)
        job = job.cpu()
    return job.tolist(), local, (ev_ms if make_events else None)


def region_summary(job_times, K):
    """median / min / max region of a run, as ms per step"""
    srt = sorted(job_times)
    med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    return med, {"repeats": len(srt), "ms_per_step_min": srt[0] / K * 1e3, "ms_per_step_max": srt[-1] / K * 1e3,
                 "ms_per_step_regions": [round(x / K * 1e3, 5) for x in job_times]}  # fmt: skip


def workload_cfg(capi, workload, n_envs, device_id=0, ladder_levels=0, agent_orders=0):
    kw = dict(step_us=STEP_US, episode_us=EPISODE_US, device_id=device_id, ladder_levels=ladder_levels, agent_orders=agent_orders)
    if workload == "ad":  # get_reward_function("AD"), rl4mm/gym/utils.py:29-39
        kw.update(per_step_reward=(capi.R_INVENTORY_ADJUSTED, 1, 0.1), terminal_reward=(capi.R_INVENTORY_ADJUSTED, 1, 0.1))
    if workload == "rs":  # get_reward_function("RS"): RollingSharpe(max_window_size=120, min_window_size=60)
        kw.update(per_step_reward=(capi.R_ROLLING_SHARPE, 0, 0.0, 120, 60), terminal_reward=(capi.R_ROLLING_SHARPE, 0, 0.0, 120, 60))
    return capi.make_cfg(n_envs, 50 if workload == "deep50" else 10, default_features(capi, normalisation_on=workload == "norm"), **kw)


def norm_history_bytes(cfg, start_us, first_step, n_steps):
    """Algorithmic bytes per env-step of the z-score histories over steps [first_step, first_step + n_steps) after the reset: every update
    of a normalised feature reads its history twice (the mean, then the squared deviations), 8 B per element.  The history holds one
    value per update since the reset's warm-up began (the first append counts twice: Features.py:67-72), up to norm_len."""
    total = 0
    t0 = start_us - cfg.warmup_us  # the clock at the top of the warm-up
    for i in range(cfg.n_features):
        f = cfg.features[i]
        if f.norm_len <= 0:
            continue
        first_usage = start_us - f.lookback * f.update_us
        t = t0 + cfg.step_us * np.arange(1, cfg.warmup_us // cfg.step_us + first_step + n_steps + 1)  # the end of every step since the reset
        upd = (t >= first_usage) & (t % f.update_us == 0)
        n_hist = np.minimum(np.cumsum(upd) + 1, f.norm_len)  # history length when the update's sums run
        timed = slice(cfg.warmup_us // cfg.step_us + first_step, None)
        total += int(np.sum(n_hist[timed][upd[timed]])) * 16
    return total / n_steps


def workload_starts(workload, n_envs, n_streams, seed):
    """Episode starts 10:00:00 .. 10:00:59 and the stream of every env (round-robin over the tickers)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    starts = (36_000 + rng.integers(0, 60, size=n_envs)) * 1_000_000
    sids = (np.arange(n_envs) % n_streams).astype(np.int32)
    return starts, sids


def pmc_traffic(workload, n_envs, kernel=None):
    """HBM bytes per launch of the step kernel from the committed rocprofv3 --pmc passes of this same command
    (tools/pmc_passes.sh -> profiles/<round>/pmc_step_kernel.csv; counters cannot be collected from inside the timed
    run).  FETCH_SIZE / WRITE_SIZE are in KB per dispatch; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes
    for wide reads, so it is doubled (MI355X_MICROARCH.md, HBM section): an upper bound for this kernel's mix of
    8- and 16-byte-per-lane loads.  None when the file is absent or was taken at another batch size / workload."""
    if os.environ.get("LOBSIM_GENERIC_KERNELS"):
        workload = "generic" if workload == "random" else workload + "_generic"
    f = ROOT / "profiles" / PROFILE_ROUND / ("pmc_step_kernel.csv" if workload == "random" else f"pmc_step_kernel_{workload}.csv")
    try:
        lines = f.read_text().splitlines()
        # the file names the kernel its counters were taken on (tools/pmc_csv.py): a csv of another kernel - another agent table, another
        # ladder, the generic instead of the specialised variant, a stale round - is refused, `traffic` is then null rather than wrong
        named = [ln_[len("# kernel:"):].strip() for ln_ in lines if ln_.startswith("# kernel:")]
        if kernel is not None and (not named or named[0] != kernel):
            return None, None
        rows = {r.split(",")[1]: float(r.split(",")[2]) for r in lines if r.count(",") >= 3 and not r.startswith(("#", "pass_"))}
        if int(rows["SQ_WAVES"]) not in (n_envs, n_envs + 16):  # (+ 16: the sorter waves at the head of a step launch, lpt_sort)
            return None, None
        return (2.0 * rows["FETCH_SIZE"] + rows["WRITE_SIZE"]) * 1024.0, str(f.relative_to(ROOT))
    except (OSError, KeyError, ValueError):
        return None, None


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline leg (child process: no torch, its own OpenMP runtime with the threads pinned)
# ---------------------------------------------------------------------------------------------------------------------
def effective_cpus():
    """CPUs this process can actually keep busy: the affinity mask, capped by the cgroup's CPU quota (a container that sees
    256 CPUs but is granted 16 CPUs' worth of time runs 256 threads at a sixteenth of their speed)."""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    eff = n if quota is None else max(1, min(n, int(quota + 0.5)))
    return eff, n, quota


def cpu_baseline_child(workload, budget_s):
    """The CPU oracle (a port: the reference itself is Python and cannot travel) on the host cores: same stream
    generator, same features, same action distribution.  The oracle is driven through oracle_rollout - T steps per
    env inside ONE OpenMP parallel region, 64 envs per thread - first on all cores, then on one thread."""
    import ctypes as C

    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as ol

    from rl4mm_amd import _capi

    cores, visible, quota = effective_cpus()
    T, chunks_max, per_thread = 500, 24, 64
    streams = build_streams(workload, T * chunks_max * STEP_US // 1_000_000 + 1)
    lib = ol.oracle_binding().lib
    lib.oracle_set_threads.restype = None
    lib.oracle_set_threads.argtypes = [C.c_int]

    def run(threads, n_envs, budget):
        lib.oracle_set_threads(threads)
        cfg = workload_cfg(_capi, workload, n_envs)
        cpu = ol.oracle_sim(cfg)
        for s in streams:
            cpu.add_stream(s)
        cpu.set_portfolio(0, 10**12)
        starts, sids = workload_starts(workload, n_envs, len(streams), 1234)
        t0 = time.perf_counter()
        cpu.reset(starts, sids)
        reset_s = time.perf_counter() - t0
        rng = np.random.Generator(np.random.PCG64(42))
        steps, dt = 0, 0.0
        for _ in range(chunks_max):  # bounded sample: whole chunks until the budget is spent
            if workload == "teradactyl":
                policy, params, acts = _capi.POLICY_TERADACTYL, TERADACTYL_PARAMS, None
            else:
                policy, params, acts = _capi.POLICY_GIVEN, [], rng.uniform(0, 10, size=(T, n_envs, 4))
            t0 = time.perf_counter()
            out = ol.oracle_rollout(cpu, policy, params, T, acts, want_actions=False)
            dt += time.perf_counter() - t0
            steps += int(out["steps"].sum())
            if dt > budget:
                break
        cpu.close()
        return steps / dt, steps, dt, reset_s

    all_v, all_steps, all_dt, all_reset = run(cores, per_thread * cores, budget_s)
    one_v, one_steps, one_dt, _ = run(1, per_thread, budget_s / 3)
    return {
        "value": all_v, "unit": "env-steps/s", "cores": cores, "threads": cores, "per_core": one_v, "kind": "port",
        "visible_cpus": visible, "cgroup_cpu_quota": quota,
        "sample": f"all cores: {per_thread * cores} envs, {all_steps} env-steps in {all_dt:.1f} s (+{all_reset:.1f} s reset incl. 3000 warm-up steps per env); "
                  f"1 thread: {per_thread} envs, {one_steps} env-steps in {one_dt:.1f} s; oracle/lobsim_oracle.c through oracle_rollout "
                  f"({T} steps per env per OpenMP region, static schedule, OMP_PROC_BIND=close), workload '{workload}'; cores = CPUs usable: "
                  f"{visible} visible, cgroup quota {quota}",
    }  # fmt: skip


def cpu_baseline(workload, budget_s=10.0):
    env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores", OMP_NUM_THREADS=str(effective_cpus()[0]))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--cpu-baseline-only", "--workload", workload, "--cpu-budget", str(budget_s)],
                       env=env, capture_output=True, text=True, timeout=600)  # fmt: skip
    if p.returncode != 0:
        return {"error": (p.stderr or p.stdout)[-400:]}
    return json.loads(p.stdout.strip().splitlines()[-1])


# ---------------------------------------------------------------------------------------------------------------------
# the Python boundary (reported beside `value`, never as it)
# ---------------------------------------------------------------------------------------------------------------------
def facade_leg(sim, stream, n_envs, device_id, steps=60):
    """env-steps/s through the API the reference's callers hold (reported beside `value`, never as it):
      vector_step_tensors  torch CUDA tensors in and out on torch's current stream, no PCIe, no host sync per step - the path of a
                           learner that lives on the GPU (rl4mm/helpers/main_helper.py:260-261: the reference's consumer is a torch PPO)
      vector_step_arrays   numpy in, numpy out: the kernel reads the actions from / writes its outputs to pinned host memory itself
      lobsim_step          the C-ABI's host-pointer entry underneath (caller's pageable arrays: two memcpys around the same launch)
      vector_step          RLlib's list-shaped VectorEnv call
    plus vector_reset (start draws on the host + 3000 warm-up steps per env on the device) and RLlib's episode boundary: N
    consecutive reset_at(i) calls after all envs finished (one launch for all of them since round 4)."""
    from datetime import datetime, timedelta

    import torch

    from rl4mm_amd import env as E

    rng = np.random.Generator(np.random.PCG64(7))
    acts = rng.uniform(0, 10, size=(steps + 5, n_envs, 4))
    out = {"unit": "env-steps/s", "steps": steps, "envs": n_envs,
           "note": "vector_step_tensors: device tensors, asynchronous; the others: host numpy arrays in and out every step (PCIe inclusive); `value` is the device-pointer entry"}

    def rate(fn, sync=None):  # fn(t): one step with the t-th action batch
        for t in range(5):
            fn(t)
        if sync:
            sync()
        t0 = time.perf_counter()
        for t in range(5, 5 + steps):
            fn(t)
        if sync:
            sync()
        return n_envs * steps / (time.perf_counter() - t0)

    out["lobsim_step_host_pointers"] = rate(lambda t: sim.step(acts[t]))
    day = datetime.strptime(stream.trading_date, "%Y-%m-%d")

    def make(obs_dtype, **kw):
        return E.HistoricalOrderbookVectorEnv(
            n_envs, E.StreamCatalog().add(stream), ticker=stream.ticker, n_levels=stream.n_levels, min_date=day, max_date=day,
            min_start_timedelta=timedelta(hours=10), max_end_timedelta=timedelta(hours=10, minutes=31), episode_length=timedelta(minutes=30),
            initial_portfolio=E.Portfolio(inventory=0, cash=10**12), per_step_reward_function=E.PnL(), terminal_reward_function=E.PnL(), device_id=device_id,
            obs_dtype=obs_dtype, **kw)  # fmt: skip

    venv = make(np.float64)
    np.random.seed(11)
    venv.vector_reset_arrays()
    t0 = time.perf_counter()
    venv.vector_reset_arrays()
    out["vector_reset_ms"] = (time.perf_counter() - t0) * 1e3
    out["vector_step_arrays"] = rate(lambda t: venv.vector_step_arrays(acts[t]))
    out["vector_step_arrays_views"] = rate(lambda t: venv.vector_step_arrays(acts[t], copy=False))
    out["vector_step_lists"] = rate(lambda t: venv.vector_step(acts[t]))
    venv.close()
    venv = make(np.float32)
    venv.vector_reset_arrays()
    out["vector_step_arrays_f32_obs"] = rate(lambda t: venv.vector_step_arrays(acts[t]))
    # tensor path on a torch side stream (the null stream costs two stream waits per step)
    d_acts = torch.from_numpy(acts).to(torch.device("cuda", device_id))
    venv.vector_reset_tensors()
    with torch.cuda.stream(torch.cuda.Stream(device=torch.device("cuda", device_id))):
        out["vector_step_tensors_f32_obs"] = rate(lambda t: venv.vector_step_tensors(d_acts[t]), sync=torch.cuda.synchronize)
    try:
        out["closed_loop"] = closed_loop_leg(venv, n_envs, device_id)
    except Exception as e:  # noqa: BLE001  (a stretch entry: never the reason a bench line is lost)
        out["closed_loop"] = {"error": repr(e)[:300]}
    venv.close()
    # RLlib's episode boundary: every env finishes on the same step, then reset_at(i) for each
    short = E.HistoricalOrderbookVectorEnv(
        n_envs, E.StreamCatalog().add(stream), ticker=stream.ticker, n_levels=stream.n_levels, min_date=day, max_date=day, min_start_timedelta=timedelta(hours=10),
        max_end_timedelta=timedelta(hours=10, seconds=30), episode_length=timedelta(seconds=0.5), initial_portfolio=E.Portfolio(inventory=0, cash=10**12),
        per_step_reward_function=E.PnL(), terminal_reward_function=E.PnL(), device_id=device_id)  # fmt: skip
    short.vector_reset_arrays()
    for t in range(5):
        _, _, done, _ = short.vector_step_arrays(acts[t])
    assert done.all()
    t0 = time.perf_counter()
    for i in range(n_envs):
        short.reset_at(i)
    out["reset_at_all_ms"] = (time.perf_counter() - t0) * 1e3
    short.close()
    return out


def closed_loop_leg(venv, n_envs, device_id, steps=200):
    """The rate a learner on the GPU sees (reported beside `value`, never as it; outside SURVEY 8's rows): a 2 x 256 tanh MLP policy
    (the reference's PPO default, rl4mm/helpers/main_helper.py:253-300: fcnet_hiddens [256, 256]) between vector_step_tensors calls -
    observation f32 [N, D] -> Beta parameters f64 [N, 4] in (0, 10) -> env step - once as eager torch calls on a side stream, once
    with policy + step captured in ONE hipGraph and replayed."""
    import torch

    dev = torch.device("cuda", device_id)
    D = venv.sim.obs_dim
    torch.manual_seed(0)
    policy = torch.nn.Sequential(torch.nn.Linear(D, 256), torch.nn.Tanh(), torch.nn.Linear(256, 256), torch.nn.Tanh(), torch.nn.Linear(256, 4)).to(dev)
    lo = torch.as_tensor(venv.observation_space.low, device=dev)
    span = torch.clamp(torch.as_tensor(venv.observation_space.high, device=dev) - lo, min=1e-6)
    act = torch.full((n_envs, 4), 1.0, dtype=torch.float64, device=dev)

    def think(obs):  # -> writes the next action batch in place
        with torch.no_grad():
            x = (obs.float() - lo) / span
            act.copy_(torch.sigmoid(policy(x)).double() * 10.0)

    out = {}
    s = torch.cuda.Stream(device=dev)
    venv.vector_reset_tensors()
    with torch.cuda.stream(s):
        obs, _, _ = venv.vector_step_tensors(act)  # binds the handle to the stream
        for _ in range(5):
            think(obs)
            obs, _, _ = venv.vector_step_tensors(act)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            think(obs)
            obs, _, _ = venv.vector_step_tensors(act)
        torch.cuda.synchronize()
        out["eager"] = n_envs * steps / (time.perf_counter() - t0)
    # (torch.cuda.graph ends the capture in its __exit__ also when the body raises: a failed capture cannot leave the stream capturing for
    # the legs that follow; the replays stay inside the try of the caller)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        think(obs)
        venv.vector_step_tensors(act)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    out["hip_graph"] = n_envs * steps / (time.perf_counter() - t0)
    out["hip_graph_note"] = "a replayed graph reads the launch order it was captured with: the longest-job-first order of the step launch (lpt_sort) is stale under replay - valid, not current"
    out["note"] = "2 x 256 tanh MLP (f32) -> sigmoid x 10 -> f64 actions -> vector_step_tensors (f32 observations), per env-step of the closed loop; done envs are skipped on the device, none finishes inside the leg"
    return out


def facade_tensor_leg(streams, n_envs, device_id, n_levels, steps=60):
    """The batched env over SEVERAL tickers through the Python boundary (configs[4]: mixed tickers in one batch):
    HistoricalOrderbookVectorEnv(ticker=[...]).vector_step_tensors, env i on ticker i mod T, the rate-ordered launch picked by the library."""
    from datetime import datetime, timedelta

    import torch

    from rl4mm_amd import env as E

    cat = E.StreamCatalog()
    for s in streams:
        cat.add(s)
    day = datetime.strptime(streams[0].trading_date, "%Y-%m-%d")
    venv = E.HistoricalOrderbookVectorEnv(
        n_envs, cat, ticker=[s.ticker for s in streams] if len(streams) > 1 else streams[0].ticker, n_levels=n_levels, min_date=day, max_date=day,
        min_start_timedelta=timedelta(hours=10), max_end_timedelta=timedelta(hours=10, minutes=31), episode_length=timedelta(minutes=30),
        initial_portfolio=E.Portfolio(inventory=0, cash=10**12), per_step_reward_function=E.PnL(), terminal_reward_function=E.PnL(), device_id=device_id, obs_dtype=np.float32)  # fmt: skip
    np.random.seed(11)
    dev = torch.device("cuda", device_id)
    d_acts = torch.from_numpy(np.random.Generator(np.random.PCG64(7)).uniform(0, 10, size=(steps + 5, n_envs, 4))).to(dev)
    t0 = time.perf_counter()
    venv.vector_reset_tensors()
    out = {"unit": "env-steps/s", "envs": n_envs, "tickers": len(streams), "vector_reset_ms": (time.perf_counter() - t0) * 1e3, "limits": venv.limits}
    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
        for t in range(5):
            venv.vector_step_tensors(d_acts[t])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(5, 5 + steps):
            venv.vector_step_tensors(d_acts[t])
        torch.cuda.synchronize()
        out["vector_step_tensors_f32_obs"] = n_envs * steps / (time.perf_counter() - t0)
    venv.close()
    return out


# ---------------------------------------------------------------------------------------------------------------------
# multi-GPU launcher
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """`bench.py --gpus N` outside torchrun: start N ranks (one per GPU) as a child job and relay its output.  Nothing in
    this process has touched the GPU (torch is not even imported), and the child is a new process, not an exec."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), *argv]  # fmt: skip
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))


def stub_rank(args, rank, world):
    """CPU stand-in of one rank (tests/test_host_logic.py): same launcher, same timed_regions harness (barriers, max over
    ranks per region, median region), same stream hand-over and episode-return gather - over gloo, with a step that only
    fills the return vector."""
    import torch
    import torch.distributed as dist

    from rl4mm_amd.dist import gather_episode_returns, shard_range

    dist.init_process_group("gloo")
    streams = shared_streams("random", 1, rank, dist)
    lo, hi = shard_range(world * args.envs, rank, world)
    returns = torch.zeros(hi - lo, dtype=torch.float64)

    def step(t):
        returns.add_(torch.arange(lo, hi, dtype=torch.float64))

    job, local, _ = timed_regions(step, args.steps, args.repeats, 0, dist, lambda: None)
    med, summary = region_summary(job, args.steps)
    mine = torch.tensor([sorted(local)[len(local) // 2] / args.steps * 1e3], dtype=torch.float64)
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(per_rank, mine)
    allr = gather_episode_returns(returns)
    if rank == 0:
        ok = bool(torch.equal(allr, args.steps * args.repeats * torch.arange(world * args.envs, dtype=torch.float64)))
        print(json.dumps({"metric": "stub", "n_gpus": world, "steps": args.steps, "ms_per_step": med / args.steps * 1e3, **summary,
                          "gathered_returns": int(allr.numel()), "gather_in_env_order": ok, "stream_msgs": int(streams[0].n_msgs),
                          "ms_per_step_per_rank": [float(x.item()) for x in per_rank]}))  # fmt: skip
    dist.barrier()
    dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default 4096; 8192 for deep50)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="random")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-budget", type=float, default=10.0, help="seconds of all-core CPU work in the cpu_baseline sample")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)  # CPU test of the launcher (gloo, no lobsim)
    ap.add_argument("--repeats", type=int, default=REPEATS, help="timed regions of --steps steps; `value` is the median region")
    ap.add_argument("--ladder-levels", type=int, default=0, help="price levels per side kept on chip (0 = the library's choice: 64 for books of <= 56 visible levels, else 128)")
    ap.add_argument("--agent-orders", type=int, default=0, help="the env's table of live agent orders: 0 = the library's default (256: cannot overflow), 128 = the smaller, faster table (cfg.agent_orders)")
    ap.add_argument("--generic-kernels", action="store_true", help="run the default configuration through the GENERIC kernels (what any non-default env_config runs) instead of the ones specialised for it")
    ap.add_argument("--no-facade", action="store_true", help="skip the Python-boundary leg (VectorEnv.vector_step_arrays, host-pointer lobsim_step)")
    # multi-rank readiness on a ONE-GPU box (tests/test_gpu_multirank.py): every rank uses device 0; RCCL refuses two ranks on one
    # device, so the collectives of that mode run over gloo on host copies - the rank body, sharding, seeding, timing harness and
    # gather are the ones of the real multi-GPU run
    ap.add_argument("--ranks-share-gpu", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--seed-rank", type=int, default=None, help=argparse.SUPPRESS)  # a single-rank run with the seeds of this rank of a larger job
    ap.add_argument("--dump-returns", default=None, help=argparse.SUPPRESS)  # .npy of the gathered per-env episode returns
    args = ap.parse_args()
    if args.envs is None:
        args.envs = 8192 if args.workload == "deep50" else ENVS_PER_GPU
    if args.generic_kernels:
        os.environ["LOBSIM_GENERIC_KERNELS"] = "1"  # (read by lobsim_create)

    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_child(args.workload, args.cpu_budget)))
        return 0
    under_torchrun = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if args.gpus > 1 and not under_torchrun:
        return launch_ranks(args.gpus, sys.argv[1:])

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.stub:
        stub_rank(args, rank, world)
        return 0

    import torch

    from rl4mm_amd import _capi, _lib

    dist = None
    if args.ranks_share_gpu:
        local_rank = 0
    if under_torchrun:  # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist

        if args.ranks_share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    seed_rank = rank if args.seed_rank is None else args.seed_rank
    K, W, N, wl, R = args.steps, args.warmup, args.envs, args.workload, max(1, args.repeats)
    if os.environ.get("LOBSIM_LIB") and rank == 0:
        print(f"bench.py: LOBSIM_LIB override active, running {os.environ['LOBSIM_LIB']} (no staleness check)", file=sys.stderr)

    sim_seconds = ((R + 1) * K + W) * STEP_US // 1_000_000 + 8  # the timed regions, the open-loop rollout leg's K more, the facade leg
    streams = shared_streams(wl, sim_seconds, rank, dist)
    cfg = workload_cfg(_capi, wl, N, device_id=local_rank, ladder_levels=args.ladder_levels, agent_orders=args.agent_orders)
    sim = _lib.new_sim(cfg)
    # kernels run on a torch-owned (non-default) stream so the torch.cuda.Event pairs below time exactly them
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    sim.set_stream(torch.cuda.current_stream().cuda_stream)
    for s in streams:
        sim.add_stream(s)
    sim.set_portfolio(0, 10**12)  # initial_cash 1e12 (main_helper.py:78)
    starts, sids = workload_starts(wl, N, len(streams), 1234 + seed_rank)
    t0 = time.perf_counter()
    sim.reset(starts, sids)  # snapshot + 3000 warm-up steps per env (5 min Volatility window)
    reset_s = time.perf_counter() - t0

    obs = torch.zeros((N, sim.obs_dim), dtype=torch.float64, device=dev)
    obs32 = torch.zeros((N, sim.obs_dim), dtype=torch.float32, device=dev)
    reward = torch.zeros(N, dtype=torch.float64, device=dev)
    done = torch.zeros(N, dtype=torch.uint8, device=dev)
    returns = torch.zeros(N, dtype=torch.float64, device=dev)
    if wl == "teradactyl":
        acts = None

        out_ptrs = (obs.data_ptr(), obs32.data_ptr(), reward.data_ptr(), done.data_ptr(), returns.data_ptr())

        def step(t):  # one env.step() per env per launch, the rule-based agent's action computed in the kernel prologue
            sim.rollout_device(_capi.POLICY_TERADACTYL, TERADACTYL_PARAMS, 1, 0, 0, *out_ptrs, 0)
    else:
        acts = torch.from_numpy(np.random.Generator(np.random.PCG64(42 + seed_rank)).uniform(0, 10, size=(R * K + W, N, 4))).to(dev)
        # device addresses worked out ahead of the timed region (a short region - the driver times 20 steps - would otherwise
        # charge the tensor indexing of its first step as GPU idle time)
        act_ptrs = [acts[t].data_ptr() for t in range(R * K + W)]
        out_ptrs = (obs.data_ptr(), obs32.data_ptr(), reward.data_ptr(), done.data_ptr(), returns.data_ptr())

        def step(t):  # one launch of lob_step_kernel over the rank's envs; the episode return is accumulated in its epilogue
            sim.step_device(act_ptrs[t], 0, *out_ptrs)

    for t in range(W):
        step(t)
    returns.zero_()
    # HIP events on the launch stream: one pair around each timed region (back-to-back launches of the one kernel)
    warm_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    for ev in warm_ev:
        ev.record()  # a dry run: the very first timed event record of a process takes ~50 us on the host (lazy set-up of the
                     # profiling signals) - as much as 1.5 steps of a 20-step region - and none afterwards
    torch.cuda.synchronize()
    job_times, local_times, ev_ms = timed_regions(step, K, R, W, dist, torch.cuda.synchronize,
                                                  lambda: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
    # `value` = the MEDIAN region (each region is K steps inside its own barrier + synchronise bracket, its time the MAX over
    # ranks): one host hiccup inside a 0.7 ms region - the driver's 20-step form - no longer decides the number
    elapsed, region_stats = region_summary(job_times, K)
    local_elapsed = sorted(local_times)[len(local_times) // 2]
    kernel_ms = sorted(ev_ms)[len(ev_ms) // 2] / K  # average launch duration over the median region (back-to-back launches)

    # extra (not `value`): K more steps as ONE launch (lobsim_rollout_device) - the mode the reference's rule-based
    # baseline sweeps can use; no per-step launch tail, state stays on chip
    rollout = None
    if wl in ("random", "teradactyl"):
        roll_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        steps_taken = torch.zeros(N, dtype=torch.int32, device=dev)
        acts2 = None if acts is None else torch.from_numpy(np.random.Generator(np.random.PCG64(4242 + rank)).uniform(0, 10, size=(K, N, 4))).to(dev)
        roll_ev[0].record()
        if wl == "teradactyl":
            sim.rollout_device(_capi.POLICY_TERADACTYL, TERADACTYL_PARAMS, K, 0, 0, obs.data_ptr(), obs32.data_ptr(), reward.data_ptr(), done.data_ptr(), 0, steps_taken.data_ptr())
        else:
            sim.rollout_device(_capi.POLICY_GIVEN, [], K, acts2.data_ptr(), 0, obs.data_ptr(), obs32.data_ptr(), reward.data_ptr(), done.data_ptr(), 0, steps_taken.data_ptr())
        roll_ev[1].record()
        torch.cuda.synchronize()
        rollout = (roll_ev[0].elapsed_time(roll_ev[1]), int(steps_taken.sum().item()))

    # the only collective: gather per-env episode returns (RCCL over xGMI; 32 KiB per GPU)
    per_rank_ms = [local_elapsed / K * 1e3]
    if dist:
        from rl4mm_amd.dist import gather_episode_returns

        cdev = torch.device("cpu") if dist.get_backend() == "gloo" else dev  # (gloo: ranks sharing one GPU, host copies)
        mine = torch.tensor([local_elapsed / K * 1e3], dtype=torch.float64, device=cdev)
        allms = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allms, mine)
        per_rank_ms = [float(x.item()) for x in allms]
        all_returns = gather_episode_returns(returns.to(cdev))
    else:
        all_returns = returns
    n_bad = int((torch.isnan(all_returns)).sum().item())

    if rank == 0 and args.dump_returns:
        np.save(args.dump_returns, all_returns.cpu().numpy())
    if rank == 0:
        # algorithmic bytes per env-step: 32 B per replayed message + 105 B of step I/O
        t_lo = starts + W * STEP_US
        t_hi = t_lo + K * STEP_US
        n_msgs = 0
        for i, s in enumerate(streams):
            m = sids == i
            n_msgs += int(np.sum(np.searchsorted(s.msgs[:, 0], t_hi[m], side="right") - np.searchsorted(s.msgs[:, 0], t_lo[m], side="right")))
        m_bar = n_msgs / (N * K)
        b_step = 32.0 * m_bar + 105.0
        norm_bytes = 0.0
        if wl == "norm":  # + 2 passes x 8 B x history length for every feature update of the timed steps (mean, then squared deviations)
            norm_bytes = norm_history_bytes(cfg, int(starts[0]), W, K * R)
            b_step += norm_bytes
        achieved = N * b_step / (kernel_ms * 1e-3)
        sample = range(0, N, max(1, N // 64))
        st_bad = sum(sim.env_state(int(e))["status"] != 0 for e in sample)
        lim = sim.limits()
        extras = wl in ("rs", "norm")
        kernel = "lob_step_kernel<%s, %d, %d, %s>" % ("true" if extras else "false", lim["ladder_levels"], lim["agent_orders"],
                                                     "false" if os.environ.get("LOBSIM_GENERIC_KERNELS") else "true")  # <EXTRAS, ladder, agent table, SPEC>
        traffic, traffic_src = pmc_traffic(wl, N, kernel)
        # (Teradactyl: lobsim_rollout_device with T = 1 runs the step kernel, the policy in its prologue)
        out = {
            "metric": "env-steps/sec (batched LOB sims)", "value": world * N * K / elapsed, "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, **region_stats, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int64+f64", "data": "synthetic",
            "config": {
                "workload": WORKLOADS[wl], "envs_per_gpu": N, "global_envs": world * N, "features": "full_state (10 default features, 5 min warm-up)",
                "agent_orders": lim["agent_orders"], "device_bytes_per_env": lim["bytes_per_env"], "step_size_s": 0.1, "reward": "AD" if wl == "ad" else "RS" if wl == "rs" else "PnL", **({"normalisation_on": True, "normalisation_history_bytes_per_env_step": round(norm_bytes)} if wl == "norm" else {}), "messages_per_env_step": round(m_bar, 2), "messages_per_s": round(world * N * K / elapsed * m_bar), "parallelism": f"env-sharded x{world}",
                "reset_seconds": round(reset_s, 3), "envs_in_error_state_sampled": int(st_bad), "nan_returns": n_bad,
                "ranks": world, "ms_per_step_per_rank": [round(x, 5) for x in per_rank_ms], "gathered_returns": int(all_returns.numel()),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
                "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src, "kernel": kernel, "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_env_step": round(b_step, 1),
                **({"traffic_note": "normalisation histories grow with the episode: the PMC passes ran steps 5 - 85 after the reset (tools/pmc_passes.sh), whose algorithmic bytes are `algorithmic_bytes_at_pmc_steps` per launch, the timed regions steps %d - %d" % (W, W + K * R),
                    "algorithmic_bytes_at_pmc_steps": round(N * (norm_history_bytes(cfg, int(starts[0]), 5, 80) + 32.0 * m_bar + 105.0))} if wl == "norm" else {}),
            },
        }  # fmt: skip
        if rollout:
            out["open_loop_rollout"] = {
                "value": rollout[1] / (rollout[0] * 1e-3), "unit": "env-steps/s (rank 0)", "steps_per_launch": K, "launch_ms": rollout[0],
                "note": "lob_rollout_kernel: K steps per env in one launch; reported beside, never as, `value`",
            }
        if os.environ.get("LOBSIM_LIB"):
            out["config"]["lobsim_lib_override"] = os.environ["LOBSIM_LIB"]
        if args.ranks_share_gpu:
            out["config"]["ranks_share_one_gpu"] = True
            out["config"]["collective_backend"] = "gloo on host copies (RCCL refuses two ranks on one device); not a scaling measurement"
        if dist:
            out["config"]["collective_backend"] = out["config"].get("collective_backend", f"{dist.get_backend()} (RCCL)" if dist.get_backend() == "nccl" else dist.get_backend())
        if world == 1 and wl == "random" and not args.no_facade:
            out["facade"] = facade_leg(sim, streams[0], N, local_rank)
        if world == 1 and wl == "deep50" and not args.no_facade:
            sim.close()  # (two 8192-env handles of 50-level books at once would be 2 x 22 GB: fine, but the first one is done)
            out["facade"] = facade_tensor_leg(streams, N, local_rank, 50)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_budget)
        print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    sim.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
