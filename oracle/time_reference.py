#!/usr/bin/env python3
"""Time the upstream reference's env.step() (build container only) and the CPU oracle on the SAME inputs.

Run:  PYTHONPATH=oracle/refshim:/root/reference PYTHONDONTWRITEBYTECODE=1 python oracle/time_reference.py [--procs 8] [--steps 2000]

SURVEY.md 8(d), BASELINE.md section 3: N_proc processes x 1 env of the reference (rl4mm/gym/HistoricalOrderbookEnvironment.py
:163-178 driven like rl4mm/gym/utils.py:100-117) on the synthetic SPY-shaped 10-level stream (the generator's LOBSTER CSV
pair -> populate_database -> sqlite in memory, exactly like the reference's own tests), default `full_state` features
(5 min warm-up), 0.1 s steps, random Beta-policy actions, PnL reward; then the C restatement (oracle/lobsim_oracle.c) on
the same packed stream, same starts, same actions, 1 thread and all cores.  Writes profiles/r2/reference_timing.json;
the ratio oracle / reference translates the GPU box's cpu_baseline into reference-Python terms.
Nothing of the reference travels: only the numbers.
"""
import argparse
import json
import multiprocessing as mp
import os
import shutil
import sys
import tempfile
import time
import warnings
from datetime import datetime, timedelta
from pathlib import Path

import numpy as np

warnings.simplefilter("ignore")
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))

T0_S, T1_S = 9.5 * 3600 + 1490, 9.5 * 3600 + 1800 + 60 + 260  # 09:54:50 .. 10:05:20
SDATE = datetime(2018, 2, 21)


def make_csv(tmp):
    from rl4mm_amd.stream import pack_lobster
    from rl4mm_amd.synth import generate_lobster, write_lobster_csv

    g = generate_lobster(20180221, 10, 200.0, T0_S, T1_S)
    mpath, bpath = write_lobster_csv(g, tmp, "SYN", "2018-02-21", 10)
    return mpath, bpath, pack_lobster(mpath, bpath, "SYN", "2018-02-21", 10, "S")


def ref_worker(rank, tmp, n_steps, q):
    """One process = one reference env (the reference is single-threaded, one env per Ray actor slot)."""
    import contextlib
    import io

    from sqlalchemy import create_engine

    from rl4mm.database.HistoricalDatabase import HistoricalDatabase
    from rl4mm.database.populate_database import populate_database
    from rl4mm.features.Features import Portfolio
    from rl4mm.gym.HistoricalOrderbookEnvironment import HistoricalOrderbookEnvironment
    from rl4mm.gym.utils import get_reward_function
    from rl4mm.simulation.HistoricalOrderGenerator import HistoricalOrderGenerator
    from rl4mm.simulation.OrderbookSimulator import OrderbookSimulator

    t0 = time.perf_counter()
    db = HistoricalDatabase(engine=create_engine("sqlite:///:memory:"))
    with contextlib.redirect_stdout(io.StringIO()):
        populate_database(("SYN",), (SDATE,), n_levels=10, database=db, path_to_lobster_data=str(tmp), book_snapshot_freq="S", max_rows=99999999999, batch_size=1000000)
    ingest_s = time.perf_counter() - t0
    step, episode = timedelta(seconds=0.1), timedelta(seconds=n_steps * 0.1)
    features = HistoricalOrderbookEnvironment.get_default_features(step, episode)
    warm = max(f.window_size for f in features)
    gen = HistoricalOrderGenerator("SYN", db, True)
    sim = OrderbookSimulator(ticker="SYN", order_generators=[gen], n_levels=10, database=db, preload_orders=True, episode_length=episode, warm_up=warm, outer_levels=20)
    start = 36000 + 7 * rank
    env = HistoricalOrderbookEnvironment(
        features=features, ticker="SYN", step_size=step, episode_length=episode, initial_portfolio=Portfolio(inventory=0, cash=10**12),
        min_quote_level=0, max_quote_level=10, min_date=SDATE, max_date=SDATE, simulator=sim, per_step_reward_function=get_reward_function("PnL"),
        terminal_reward_function=get_reward_function("PnL"), n_levels=10, min_start_timedelta=timedelta(seconds=start), max_end_timedelta=timedelta(seconds=start) + episode,
    )  # fmt: skip
    rng = np.random.Generator(np.random.PCG64(42 + rank))
    acts = rng.uniform(0, 10, size=(n_steps, 4))
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        env.reset()
        reset_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        n = 0
        for a in acts:
            _, _, done, _ = env.step(a)
            n += 1
            if done:
                break
        step_s = time.perf_counter() - t0
    q.put(dict(rank=rank, steps=n, step_s=step_s, reset_s=reset_s, ingest_s=ingest_s, inventory=int(env.state.portfolio.inventory), cash=int(env.state.portfolio.cash), start=start))


def oracle_run(ps, starts, n_steps, threads, seeds):
    import ctypes as C

    import bench
    import oracle_lib as ol
    from rl4mm_amd import _capi

    lib = ol.oracle_binding().lib
    lib.oracle_set_threads.restype = None
    lib.oracle_set_threads.argtypes = [C.c_int]
    lib.oracle_set_threads(threads)
    n = len(starts)
    F = _capi.make_feature
    feats = bench.default_features(_capi)
    feats[6] = F(_capi.F_EPISODE_PROPORTION, 0, 100_000, 0, 1.0, dparam=100_000 / (n_steps * 100_000))
    cfg = _capi.make_cfg(n, 10, feats, step_us=100_000, episode_us=n_steps * 100_000)
    cpu = ol.oracle_sim(cfg)
    cpu.add_stream(ps)
    cpu.set_portfolio(0, 10**12)
    t0 = time.perf_counter()
    cpu.reset(np.asarray(starts) * 1_000_000, 0)
    reset_s = time.perf_counter() - t0
    acts = np.stack([np.random.Generator(np.random.PCG64(s)).uniform(0, 10, size=(n_steps, 4)) for s in seeds], axis=1)
    t0 = time.perf_counter()
    out = ol.oracle_rollout(cpu, _capi.POLICY_GIVEN, [], n_steps, acts, want_actions=False)
    dt = time.perf_counter() - t0
    inv, cash, _ = cpu.portfolios()
    return dict(threads=threads, envs=n, steps=int(out["steps"].sum()), step_s=dt, reset_s=reset_s, inventory=inv.tolist(), cash=cash.tolist())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--steps", type=int, default=2000)
    args = ap.parse_args()
    tmp = Path(tempfile.mkdtemp(prefix="ref_timing_"))
    try:
        _, _, ps = make_csv(tmp)
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=ref_worker, args=(r, tmp, args.steps, q)) for r in range(args.procs)]
        t0 = time.perf_counter()
        for p in procs:
            p.start()
        ref = sorted((q.get(timeout=7200) for _ in procs), key=lambda d: d["rank"])
        for p in procs:
            p.join()
        wall = time.perf_counter() - t0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    starts = [d["start"] for d in ref]
    seeds = [42 + d["rank"] for d in ref]
    one = oracle_run(ps, starts[:1], args.steps, 1, seeds[:1])
    many = oracle_run(ps, starts * 16, args.steps, args.procs, seeds * 16)  # 16 envs per thread, all cores
    same = oracle_run(ps, starts, args.steps, args.procs, seeds)
    parity = all(same["inventory"][i] == ref[i]["inventory"] and same["cash"][i] == ref[i]["cash"] for i in range(len(ref)))
    ref_rate_core = float(np.mean([d["steps"] / d["step_s"] for d in ref]))
    ref_rate_all = float(sum(d["steps"] for d in ref) / max(d["step_s"] for d in ref))
    out = dict(
        host=dict(cpus=len(os.sched_getaffinity(0)), python=sys.version.split()[0]),
        workload="synthetic 10-level SPY-shaped stream @1s snapshots, default full_state features (5 min warm-up), 0.1 s steps, random Beta-policy actions, PnL",
        reference=dict(processes=args.procs, steps_per_process=[d["steps"] for d in ref], env_steps_per_s_per_core=ref_rate_core, env_steps_per_s_all=ref_rate_all,
                       reset_s_mean=float(np.mean([d["reset_s"] for d in ref])), ingest_s_mean=float(np.mean([d["ingest_s"] for d in ref])), wall_s=wall),
        oracle=dict(one_thread_env_steps_per_s=one["steps"] / one["step_s"], all_cores_env_steps_per_s=many["steps"] / many["step_s"], threads=many["threads"], envs=many["envs"],
                    reset_s_one_env=one["reset_s"]),
        ratio_oracle_over_reference_per_core=(one["steps"] / one["step_s"]) / ref_rate_core,
        final_portfolios_equal=bool(parity),
    )  # fmt: skip
    (REPO / "profiles" / "r2").mkdir(parents=True, exist_ok=True)
    (REPO / "profiles" / "r2" / "reference_timing.json").write_text(json.dumps(out, indent=1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
