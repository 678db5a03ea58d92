"""The reference's own simulator tests (rl4mm/simulation/tests/testOrderbookSimulator.py:46-108), written against
rl4mm_amd.simulator.OrderbookSimulator the way the reference writes them against its class.  `new_sim` binds the library
under test (the HIP one on the GPU box; the oracle for the CPU test of the class's host logic)."""
from datetime import datetime, timedelta


import oracle_lib as ol
from rl4mm_amd.simulator import Cancellation, LimitOrder, MarketOrder, OrderbookSimulator, convert_to_lobster_format, create_order

TICKER, N_LEVELS = "MSFT", 50


def _simulator(new_sim):
    return OrderbookSimulator(TICKER, None, None, N_LEVELS, ol.golden_stream("msft"), preload_orders=False, outer_levels=48, _new_sim=new_sim)


def _lobster_row_dict(ps, t_us):
    """HistoricalDatabase.get_last_snapshot(t).to_dict() restricted to the book columns (rl4mm/orderbook/helpers.py:52-55)."""
    row = ps.snaps[ps.last_snapshot_at(t_us), 2:]
    d = {}
    for i in range(N_LEVELS):
        d[f"sell_price_{i}"], d[f"sell_volume_{i}"], d[f"buy_price_{i}"], d[f"buy_volume_{i}"] = (float(x) for x in row[4 * i : 4 * i + 4])
    return d


def _episode_start(ps):
    """start_of_episode = the first snapshot after 09:30 rounded up to the next second (testOrderbookSimulator.py:47-49)."""
    t = int(ps.snaps[0, 0])
    t += 10**6 - t % 10**6
    return datetime(2012, 6, 21) + timedelta(microseconds=t), t


def check_agreement_with_lobster(new_sim):
    """test_agreement_with_lobster (:46-75)."""
    ps = ol.golden_stream("msft")
    sim = _simulator(new_sim)
    start, start_us = _episode_start(ps)
    sim.reset_episode(start)
    assert _lobster_row_dict(ps, start_us) == convert_to_lobster_format(sim.exchange.central_orderbook, N_LEVELS)
    for secs, drop in ((1, set()), (2, {"buy_price_45", "buy_volume_45"})):
        sim.forward_step(start + timedelta(seconds=secs))
        expected = _lobster_row_dict(ps, start_us + secs * 10**6)
        actual = convert_to_lobster_format(sim.exchange.central_orderbook, N_LEVELS)
        for key in (expected.keys() - actual.keys()) | drop:
            expected.pop(key, None)
            actual.pop(key, None)
        assert expected == actual, secs
    assert sim.now_is == start + timedelta(seconds=2)
    sim.close()


def check_update_outer_levels_with_internal_orders(new_sim):
    """test_update_outer_levels_with_internal_orders (:77-108)."""
    ps = ol.golden_stream("msft")
    sim = _simulator(new_sim)
    start, _ = _episode_start(ps)
    sim.reset_episode(start)
    min_buy_price_0 = sim.min_buy_price
    orderbook = sim.exchange.central_orderbook
    worst_buy_price, worst_sell_price = min(orderbook.buy.keys()), max(orderbook.sell.keys())
    worst_buy_order_dict = dict(orderbook.buy[worst_buy_price][0].__dict__)
    worst_sell_order_dict = dict(orderbook.sell[worst_sell_price][0].__dict__)
    worst_buy_order_dict["volume"], worst_sell_order_dict["volume"] = 100, 200
    worst_buy_order_dict["is_external"] = worst_sell_order_dict["is_external"] = False
    worst_buy_order_dict["price"] -= 100
    internal_buy, internal_sell = create_order("limit", worst_buy_order_dict), create_order("limit", worst_sell_order_dict)
    sim.forward_step(start + timedelta(seconds=1), internal_orders=[internal_buy, internal_sell])
    assert sim.min_buy_price < min_buy_price_0
    internal_level = sim.exchange.internal_orderbook.buy[worst_buy_order_dict["price"]]
    external_level = sim.exchange.central_orderbook.buy[worst_buy_order_dict["price"]]
    assert internal_level[0].internal_id is not None and len(internal_level) == 1
    # it joins the back of the queue in the central book, behind the level the resync put there
    assert external_level[0].is_external and not external_level[1].is_external
    assert external_level[0].volume == 200 and external_level[1].volume == 100
    sim.close()


def check_internal_orders_by_id(new_sim):
    """The agent's orders through the class: booked limits get internal ids, cancellations name them, a marketable agent order
    returns its fills in FilledOrders.internal and the resting side in .external (Exchange.py:85-120)."""
    ps = ol.golden_stream("msft")
    sim = _simulator(new_sim)
    start, _ = _episode_start(ps)
    book = sim.reset_episode(start)
    bb, bs = book.best_buy_price, book.best_sell_price
    mk = lambda cls, direction, **kw: cls(timestamp=start, direction=direction, ticker=TICKER, internal_id=None, external_id=None, is_external=False, **kw)  # noqa: E731
    a, b, c = mk(LimitOrder, "buy", price=bb, volume=50), mk(LimitOrder, "buy", price=bb, volume=70), mk(LimitOrder, "sell", price=bs + 100, volume=30)
    t = start + timedelta(milliseconds=1)
    filled = sim.forward_step(t, internal_orders=[a, b, c])
    assert a.internal_id and b.internal_id and c.internal_id and len({a.internal_id, b.internal_id, c.internal_id}) == 3
    ib = sim.exchange.internal_orderbook
    assert [o.internal_id for o in ib.buy[bb]] == [a.internal_id, b.internal_id] and [o.volume for o in ib.buy[bb]] == [50, 70]
    assert [o.volume for o in ib.sell[bs + 100]] == [30]
    # cancel 20 of b, then all of a (named by internal id): b stays, with 50
    t += timedelta(milliseconds=1)
    cancel = lambda o, v: Cancellation(timestamp=t, direction="buy", ticker=TICKER, internal_id=o.internal_id, external_id=None, is_external=False, price=bb, volume=v)  # noqa: E731
    sim.forward_step(t, internal_orders=[cancel(b, 20), cancel(a, 50)])
    ib = sim.exchange.internal_orderbook
    assert [(o.internal_id, o.volume) for o in ib.buy[bb]] == [(b.internal_id, 50)]
    # a marketable sell of the agent hits the best bid's queue from the front: its fill comes back in FilledOrders.internal (a
    # sell at the bid's price), the resting external order in .external (Exchange.py:104-115)
    head = sim.exchange.central_orderbook.buy[bb][0]
    assert head.is_external and sim.exchange.best_buy_price == bb
    v = min(10, head.volume)
    t += timedelta(milliseconds=1)
    filled = sim.forward_step(t, internal_orders=[mk(MarketOrder, "sell", volume=v)])
    assert [(o.direction, o.price, o.volume) for o in filled.internal][:1] == [("sell", bb, v)]
    assert [(o.direction, o.price, o.volume) for o in filled.external][:1] == [("buy", bb, v)]
    sim.close()
