"""Batched trajectories and episode summaries of the reference's evaluation scripts (rl4mm/gym/utils.py:100-257).

``generate_trajectory(agent, env)`` there steps ONE env through an episode on the host, one Python call per step, and
``get_episode_summary_dict`` repeats that ``n_iterations`` times (thread pool).  Here the iterations are the envs of one
batch and, for the reference's rule-based agents (FixedActionAgent, Teradactyl) or pre-drawn RandomAgent actions, the whole
episode is ONE ``lobsim_rollout_device`` launch: the policy runs on the device and the per-step series the summary needs
(reward, aum, inventory, asset price, market spread, weighted mid-price offset, action) are written by the kernel into a
trace buffer (``lobsim_set_trace_buffer``).  The values are the reference's: the trace columns are the entries of
``SimpleInfoCalculator.calculate`` (rl4mm/gym/order_tracking/InfoCalculators.py:31-89)."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import _capi
from .agents import FixedActionAgent, RandomAgent, Teradactyl


def policy_of(agent, action_dim: int, n_envs: int, n_steps: int):
    """-> (policy, params, actions or None) for lobsim_rollout_device."""
    if isinstance(agent, FixedActionAgent):
        return _capi.POLICY_FIXED, list(np.asarray(agent.fixed_action, dtype=np.float64)), None
    if isinstance(agent, Teradactyl):
        if agent.max_inventory is None:
            raise ValueError("Teradactyl on the device needs max_inventory (get_kappa divides by it)")
        return _capi.POLICY_TERADACTYL, [agent.max_inventory, agent.default_kappa, agent.default_omega, agent.max_kappa, agent.exponent, agent.inventory_index,
                                         1.0 if agent.market_clearing is True else 0.0, agent.denom], None  # fmt: skip
    if isinstance(agent, RandomAgent):  # pre-drawn samples of the action space, one row per (step, env)
        acts = np.stack([np.stack([agent.get_action(None) for _ in range(n_envs)]) for _ in range(n_steps)]).astype(np.float64)
        return _capi.POLICY_GIVEN, [], acts
    raise TypeError(f"no device policy for {type(agent).__name__}: step the env from the host (HistoricalOrderbookVectorEnv.vector_step)")


def generate_trajectories(agent, venv, n_steps: Optional[int] = None) -> Dict[str, np.ndarray]:
    """One episode of every env of ``venv`` (HistoricalOrderbookVectorEnv) under ``agent``, in one launch.
    -> dict(steps [N], trace [T, N, TRACE_DIM], final observation / reward / done).  Trace rows beyond an env's steps are NaN; so
    is the row of a step that ended with a status bit (counted in `steps`, where the reference would have raised)."""
    import torch

    sim, n = venv.sim, venv.num_envs
    T = int(n_steps or round(venv.episode_length / venv.step_size))
    venv.vector_reset()
    policy, params, acts = policy_of(agent, sim.action_dim, n, T)
    dev = torch.device("cuda", sim.cfg.device_id)
    trace = torch.full((T, n, _capi.TRACE_DIM), float("nan"), dtype=torch.float64, device=dev)
    obs = torch.zeros((n, sim.obs_dim), dtype=torch.float64, device=dev)
    reward, done = torch.zeros(n, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev)
    steps = torch.zeros(n, dtype=torch.int32, device=dev)
    d_acts = None if acts is None else torch.from_numpy(acts).to(dev)
    torch.cuda.synchronize(dev)  # the handle launches on its own stream
    sim.set_trace_buffer(trace.data_ptr(), T)
    try:
        sim.rollout_device(policy, params, T, 0 if d_acts is None else d_acts.data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0, steps.data_ptr())
        sim.sync()
    finally:
        sim.set_trace_buffer(0, 0)
    venv._dones[:] = done.cpu().numpy().astype(bool)
    venv._any_done = bool(venv._dones.any())
    return dict(steps=steps.cpu().numpy(), trace=trace.cpu().numpy(), observations=obs.cpu().numpy(), rewards=reward.cpu().numpy(), dones=done.cpu().numpy().astype(bool))


def init_episode_summary_dict() -> dict:  # rl4mm/gym/utils.py:139-153
    return {"equity_curves": [], "reward_series": [], "rewards": [], "actions": [], "spread": [], "inventory": [], "inventories": [], "asset_prices": [],
            "agent_midprice_offsets": []}  # fmt: skip


def get_episode_summary_dict(agent, venv, n_steps: Optional[int] = None) -> dict:
    """rl4mm/gym/utils.py:120-257: one entry per iteration = per env of the batch, the same keys and reductions
    (append_to_episode_summary_dict, :156-198 - including its mean action without the last component)."""
    out = generate_trajectories(agent, venv, n_steps)
    col = {name: i for i, name in enumerate(_capi.TRACE_COLS)}
    A = venv.sim.action_dim
    esd = init_episode_summary_dict()
    for e in range(venv.num_envs):
        k = int(out["steps"][e])
        tr = out["trace"][:k, e]
        esd["equity_curves"].append(tr[:, col["aum"]].copy())
        esd["reward_series"].append(list(tr[:, col["reward"]]))
        esd["rewards"].append(np.mean(tr[:, col["reward"]]))
        esd["actions"].append(np.mean(tr[:, col["a0"] : col["a0"] + A], axis=0)[:-1])
        esd["spread"].append(np.mean(tr[:, col["market_spread"]]))
        esd["inventory"].append(np.mean(tr[:, col["inventory"]]))
        esd["inventories"].append(tr[:, col["inventory"]].copy())
        esd["asset_prices"].append(tr[:, col["asset_price"]].copy())
        esd["agent_midprice_offsets"].append(tr[:, col["weighted_midprice_offset"]].copy())
    return esd
