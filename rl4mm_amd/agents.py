"""Rule-based baseline agents of the reference (rl4mm/agents/baseline_agents.py), restated for one observation or
a batch of observations ([N, obs_dim]); pure host arithmetic on the observation's inventory entry."""
from __future__ import annotations

import numpy as np


class RandomAgent:  # baseline_agents.py:9-18
    def __init__(self, env, seed: int = None):
        self.action_space = env.action_space
        self.action_space.seed(seed)

    def get_action(self, state: np.ndarray) -> np.ndarray:
        return self.action_space.sample()

    def get_name(self):
        return "RandomAgent"


class FixedActionAgent:  # :21-30
    def __init__(self, fixed_action: np.ndarray):
        self.fixed_action = fixed_action

    def get_action(self, state: np.ndarray) -> np.ndarray:
        state = np.asarray(state)
        if state.ndim == 2:
            return np.broadcast_to(np.asarray(self.fixed_action, dtype=np.float64), (state.shape[0], len(self.fixed_action))).copy()
        return self.fixed_action

    def get_name(self):
        return "FixedAction_" + "_".join(map(str, self.fixed_action))


class Teradactyl:  # :33-115
    def __init__(self, max_inventory=None, default_kappa=10.0, default_omega=0.5, max_kappa=10.0, exponent=1.0, market_clearing=False, inventory_index=3):
        self.max_inventory, self.default_kappa, self.default_omega = max_inventory, default_kappa, default_omega
        self.max_kappa, self.exponent, self.market_clearing, self.inventory_index = max_kappa, exponent, market_clearing, inventory_index
        self.eps = 0.00001
        self.denom = 100 if max_inventory is None else max_inventory

    def clamp_to_unit(self, x):  # :103-107, strict containment
        return np.maximum(np.minimum(x, 1 - self.eps), -1 + self.eps)

    def get_omega_bid_and_ask(self, inventory):  # :58-69
        inventory = np.asarray(inventory, dtype=np.float64)
        u = self.clamp_to_unit(inventory / self.denom)
        w, e = self.default_omega, self.exponent
        pos = inventory >= 0
        bid = np.where(pos, w * (1 + (1 / w - 1) * np.abs(u) ** e), w * (1 - np.abs(u) ** e))
        ask = np.where(pos, w * (1 - np.abs(u) ** e), w * (1 + (1 / w - 1) * np.abs(u) ** e))
        return bid, ask

    def get_kappa(self, inventory):  # :71-74
        return (self.max_kappa - self.default_kappa) * np.abs(np.asarray(inventory, dtype=np.float64) / self.max_inventory) ** self.exponent + self.default_kappa

    def get_action(self, state: np.ndarray) -> np.ndarray:  # :76-87
        state = np.asarray(state, dtype=np.float64)
        inventory = state[..., self.inventory_index]
        omega_bid, omega_ask = self.get_omega_bid_and_ask(inventory)
        kappa = self.get_kappa(inventory)
        cols = [omega_bid * (kappa - 2) + 1, (1 - omega_bid) * (kappa - 2) + 1, omega_ask * (kappa - 2) + 1, (1 - omega_ask) * (kappa - 2) + 1]
        if self.market_clearing is True:
            cols.append(np.full_like(kappa, self.max_inventory * 2))
        return np.stack(cols, axis=-1)

    def get_name(self):
        return (f"Teradactyl_def_omega_{self.default_omega}_def_kappa_{self.default_kappa}_"
                + f"max_inv_{self.max_inventory}_max_kappa_{self.max_kappa}_exponent_{self.exponent}")
