#!/usr/bin/env python3
"""bench.py - env-steps/s of the batched LOB simulation step on MI355X (BASELINE.json metric).

A "step" is one launch of the hot path (lob_step_kernel) over one batch: every env of the rank converts its action
to orders, replays its 100 ms message window, updates features and reward.  Workload at N=1 = BASELINE.json
configs[1]: 4096 vectorised envs on one MI355X, synthetic 10-level SPY-shaped stream with 1 s snapshots, random
Beta-policy actions (uniform in Box[0,10]^4, PCG64(42)), default `full_state` features, PnL reward.  With --gpus N
every rank owns 4096 envs of its own (weak scaling; envs are independent, the stream is replicated), the only
collective is the episode-return gather after the timed loop.

Inputs (actions, stream, env state) are resident in HBM when the timed region starts.  Prints ONE JSON line.
"""
import argparse
import json
import os
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


def default_features(capi):
    """HistoricalOrderbookEnvironment.get_default_features (HistoricalOrderbookEnvironment.py:395-440)."""
    F = capi.make_feature
    ep = STEP_US / EPISODE_US
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


def build_stream(sim_seconds):
    from rl4mm_amd.synth import synthetic_stream

    # 09:54:50 -> enough for a 300 s feature warm-up before 10:00:00 plus the stepped window
    return synthetic_stream(20180221, 10, 200.0, 9.5 * 3600 + 1490, 9.5 * 3600 + 1800 + 60 + max(sim_seconds, 105) + 5)


def pmc_traffic(n_envs):
    """HBM bytes per launch of the step kernel from the committed rocprofv3 --pmc passes of this same command
    (tools/pmc_passes.sh -> profiles/r1/h_pmc_step_kernel.csv; counters cannot be collected from inside the timed
    run).  FETCH_SIZE / WRITE_SIZE are in KB per dispatch; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes
    for wide reads, so it is doubled (MI355X_MICROARCH.md, HBM section): an upper bound for this kernel's mix of
    8- and 16-byte-per-lane loads.  None when the file is absent or was taken at another batch size."""
    f = ROOT / "profiles" / "r1" / "h_pmc_step_kernel.csv"
    try:
        rows = {r.split(",")[1]: float(r.split(",")[2]) for r in f.read_text().splitlines()[1:] if r.count(",") >= 3}
        if int(rows["SQ_WAVES"]) != n_envs:
            return None, None
        return (2.0 * rows["FETCH_SIZE"] + rows["WRITE_SIZE"]) * 1024.0, str(f.relative_to(ROOT))
    except (OSError, KeyError, ValueError):
        return None, None


def cpu_baseline(capi, stream, n_envs, n_steps):
    """The CPU oracle (a port: the reference itself is Python and cannot travel) on the host cores, same stream,
    same features, same action distribution."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as ol

    cores = len(os.sched_getaffinity(0))
    cfg = capi.make_cfg(n_envs, 10, default_features(capi), step_us=STEP_US, episode_us=EPISODE_US)
    cpu = ol.oracle_sim(cfg)
    cpu.add_stream(stream)
    cpu.set_portfolio(0, 10**12)
    rng = np.random.Generator(np.random.PCG64(1234))
    starts = (36_000 + rng.integers(0, 60, size=n_envs)) * 1_000_000
    cpu.reset(starts, 0)
    acts = np.random.Generator(np.random.PCG64(42)).uniform(0, 10, size=(n_steps, n_envs, 4))
    t0 = time.perf_counter()
    done_steps = 0
    for t in range(n_steps):  # bounded sample: stop after ~12 s of CPU work
        cpu.step(acts[t])
        done_steps += 1
        if time.perf_counter() - t0 > 12.0:
            break
    dt = time.perf_counter() - t0
    return {
        "value": n_envs * done_steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": f"{n_envs} envs x {done_steps} steps of the same workload (oracle/lobsim_oracle.c, OpenMP over envs), {dt:.1f} s",
    }  # fmt: skip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", type=int, default=ENVS_PER_GPU, help="envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    from rl4mm_amd import _capi, _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if "RANK" in os.environ and "MASTER_ADDR" in os.environ:  # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    K, W, N = args.steps, args.warmup, args.envs

    sim_seconds = (K + W) * STEP_US // 1_000_000 + 1
    stream = build_stream(sim_seconds)
    cfg = _capi.make_cfg(N, 10, default_features(_capi), step_us=STEP_US, episode_us=EPISODE_US, device_id=local_rank)
    sim = _lib.new_sim(cfg)
    # kernels run on a torch-owned (non-default) stream so the torch.cuda.Event pairs below time exactly them
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    sim.set_stream(torch.cuda.current_stream().cuda_stream)
    sim.add_stream(stream)
    sim.set_portfolio(0, 10**12)  # initial_cash 1e12 (main_helper.py:78)
    rng = np.random.Generator(np.random.PCG64(1234 + rank))
    starts = (36_000 + rng.integers(0, 60, size=N)) * 1_000_000  # episode starts 10:00:00 .. 10:00:59
    t0 = time.perf_counter()
    sim.reset(starts, 0)  # snapshot + 3000 warm-up steps per env (5 min Volatility window)
    reset_s = time.perf_counter() - t0

    acts = torch.from_numpy(np.random.Generator(np.random.PCG64(42 + rank)).uniform(0, 10, size=(K + W, N, 4))).to(dev)
    obs = torch.zeros((N, sim.obs_dim), dtype=torch.float64, device=dev)
    obs32 = torch.zeros((N, sim.obs_dim), dtype=torch.float32, device=dev)
    reward = torch.zeros(N, dtype=torch.float64, device=dev)
    done = torch.zeros(N, dtype=torch.uint8, device=dev)
    returns = torch.zeros(N, dtype=torch.float64, device=dev)

    def step(t):  # one launch of lob_step_kernel over the rank's envs; the episode return is accumulated in its epilogue
        sim.step_device(acts[t].data_ptr(), 0, obs.data_ptr(), obs32.data_ptr(), reward.data_ptr(), done.data_ptr(), returns.data_ptr())

    for t in range(W):
        step(t)
    returns.zero_()
    # HIP events on the launch stream: one pair around the whole timed region, plus a pair around every 8th launch
    ev_all = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev = {t: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for t in range(0, K, 8)}
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev_all[0].record()
    for t in range(K):
        if t in ev:
            ev[t][0].record()
        step(W + t)
        if t in ev:
            ev[t][1].record()
    ev_all[1].record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms_sampled = float(np.mean([a.elapsed_time(b) for a, b in ev.values()]))  # event pair around single launches
    kernel_ms = ev_all[0].elapsed_time(ev_all[1]) / K  # average launch duration over the timed region (back-to-back launches)

    # extra (not `value`): the same K steps as ONE launch with the pre-drawn actions (lobsim_rollout_device) - the
    # mode the reference's rule-based baseline sweeps can use; no per-step launch tail, state stays on chip
    roll_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    steps_taken = torch.zeros(N, dtype=torch.int32, device=dev)
    roll_ev[0].record()
    sim.rollout_device(_capi.POLICY_GIVEN, [], K, acts[W:].data_ptr(), 0, obs.data_ptr(), obs32.data_ptr(), reward.data_ptr(), done.data_ptr(), 0, steps_taken.data_ptr())
    roll_ev[1].record()
    torch.cuda.synchronize()
    rollout_ms = roll_ev[0].elapsed_time(roll_ev[1])
    rollout_steps = int(steps_taken.sum().item())

    # the only collective: gather per-env episode returns (RCCL over xGMI; 32 KiB per GPU)
    if dist:
        el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
        gathered = [torch.zeros_like(returns) for _ in range(world)]
        dist.all_gather(gathered, returns)
        all_returns = torch.cat(gathered)
    else:
        all_returns = returns
    n_bad = int((torch.isnan(all_returns)).sum().item())

    if rank == 0:
        # algorithmic bytes per env-step: 32 B per replayed message + 105 B of step I/O
        t_lo = starts + W * STEP_US
        t_hi = t_lo + K * STEP_US
        n_msgs = int(np.sum(np.searchsorted(stream.msgs[:, 0], t_hi, side="right") - np.searchsorted(stream.msgs[:, 0], t_lo, side="right")))
        m_bar = n_msgs / (N * K)
        b_step = 32.0 * m_bar + 105.0
        achieved = N * b_step / (kernel_ms * 1e-3)
        inv, cash, _ = sim.portfolios()
        st_bad = sum(sim.env_state(int(e))["status"] != 0 for e in range(0, N, max(1, N // 64)))
        traffic, traffic_src = pmc_traffic(N)
        out = {
            "metric": "env-steps/sec (batched LOB sims)", "value": world * N * K / elapsed, "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int64+f64", "data": "synthetic",
            "config": {
                "workload": "4096 vectorised envs per GPU, synthetic 10-level SPY-shaped LOBSTER stream @1s snapshots, random Beta-policy actions (BASELINE.json configs[1])",
                "envs_per_gpu": N, "global_envs": world * N, "features": "full_state (10 default features, 5 min warm-up)",
                "step_size_s": 0.1, "reward": "PnL", "messages_per_env_step": round(m_bar, 2), "parallelism": f"env-sharded x{world}",
                "reset_seconds": round(reset_s, 3), "envs_in_error_state_sampled": int(st_bad), "nan_returns": n_bad,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
                "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src, "kernel": "lob_step_kernel", "kernel_ms": kernel_ms, "kernel_ms_sampled_pairs": kernel_ms_sampled, "algorithmic_bytes_per_env_step": round(b_step, 1),
            },
        }  # fmt: skip
        out["open_loop_rollout"] = {
            "value": rollout_steps / (rollout_ms * 1e-3), "unit": "env-steps/s (rank 0)", "steps_per_launch": K, "launch_ms": rollout_ms,
            "note": "lob_rollout_kernel: K steps per env in one launch with pre-drawn actions; reported beside, never as, `value`",
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(_capi, stream, max(256, 4 * len(os.sched_getaffinity(0))), 3000)
        print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    sim.close()


if __name__ == "__main__":
    main()
