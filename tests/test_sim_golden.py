"""The simulator class against episodes recorded from the reference's OrderbookSimulator with internal-order lists of every shape and with a
caller's start book (tests/golden/sim_internal_orders.npz, oracle/gen_golden_simulator.py) - here on the CPU oracle; the same replay runs on
the HIP library in tests/test_gpu_simulator.py."""
import pytest

import oracle_lib as ol
import sim_golden_replay as sg


@pytest.mark.parametrize("k", range(sg.n_cases()))
def test_recorded_internal_order_episodes_on_the_oracle(k):
    stats = sg.replay_case(k, ol.oracle_sim)
    assert stats["steps"] >= 2


def test_the_fixture_holds_what_it_is_for():
    """cancellations behind own orders of the same call, episodes from a start book, calls that raise: all present"""
    import numpy as np

    behind = start_books = raises = 0
    with np.load(sg.GOLD) as g:
        for k in range(int(g["n_cases"])):
            rows = g[f"c{k}_orders"]
            start_books += len(g[f"c{k}_book_edit"]) > 0
            raises += bool(len(rows)) and rows[-1, 1] == -1
            for step in np.unique(rows[:, 0]):
                call = rows[rows[:, 0] == step]
                seen_own = False
                for r in call:
                    behind += seen_own and r[1] in (sg.T_CANCEL, sg.T_DELETE)
                    seen_own |= r[1] in (sg.T_LIMIT, sg.T_MARKET)
    assert behind >= 30 and start_books == 2 and raises >= 3
