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


def check_cancellations_the_host_could_not_rank(new_sim):
    """The calls rl4mm_amd.simulator refused until round 6 (it turned internal ids into FIFO ranks on the host): a cancellation BEHIND a
    marketable own limit order at the same price, and one behind an own order that sweeps the agent's quotes, in ONE forward_step -
    the list is processed in order, as Exchange.process_order does (Exchange.py:58-69), and the target is found by its id on the device."""
    ps = ol.golden_stream("msft")
    sim = _simulator(new_sim)
    start, _ = _episode_start(ps)
    book = sim.reset_episode(start)
    bb, bs = book.best_buy_price, book.best_sell_price
    mk = lambda cls, direction, **kw: cls(timestamp=start, direction=direction, ticker=TICKER, internal_id=None, external_id=None, is_external=False, **kw)  # noqa: E731
    cancel = lambda o, v, t: Cancellation(timestamp=t, direction=o.direction, ticker=TICKER, internal_id=o.internal_id, external_id=None, is_external=False, price=o.price, volume=v)  # noqa: E731
    t = start + timedelta(milliseconds=1)
    a = mk(LimitOrder, "sell", price=bs, volume=40)  # rests at the best ask, behind the external orders there
    sim.forward_step(t, internal_orders=[a])
    ask_vol = sum(o.volume for o in sim.exchange.central_orderbook.sell[bs] if o.is_external)
    # (1) a buy limit AT the best ask for more than the externals hold - it executes, deletes the agent's own resting sell on its way
    # (Exchange.py:91-94) and its remainder rests at bs as the new best bid - and, in the same call, a cancellation of that remainder's
    # price level naming `a`: a is gone, the level's head is now the agent's own buy order (not a snapshot order): nothing happens
    t += timedelta(milliseconds=1)
    m = mk(LimitOrder, "buy", price=bs, volume=ask_vol + 25)
    filled = sim.forward_step(t, internal_orders=[m, cancel(a, 40, t)])
    assert sum(o.volume for o in filled.internal) == ask_vol  # the own sell was deleted, not filled
    ib = sim.exchange.internal_orderbook
    assert bs not in ib.sell and [(o.internal_id, o.volume) for o in ib.buy[bs]] == [(m.internal_id, 25)]
    # (2) behind a marketable own order at the SAME price as its target: 10 of the remainder are cancelled by id
    t += timedelta(milliseconds=1)
    book = sim.exchange.central_orderbook
    bs2 = book.best_sell_price
    m2 = mk(LimitOrder, "buy", price=bs, volume=5)  # joins the agent's own level behind m
    sim.forward_step(t, internal_orders=[m2, cancel(m, 10, t)])
    ib = sim.exchange.internal_orderbook
    assert [(o.internal_id, o.volume) for o in ib.buy[bs]] == [(m.internal_id, 15), (m2.internal_id, 5)] and bs2 == sim.exchange.central_orderbook.best_sell_price
    # (3) an own sell that sweeps the agent's whole bid level (it deletes its own resting orders, Exchange.py:91-94) and, behind it in the
    # same call, a cancellation of one of them: the order is gone, the cancellation finds nothing
    t += timedelta(milliseconds=1)
    bids = sum(o.volume for p, q in sim.exchange.central_orderbook.buy.items() if p >= bs - 100 for o in q if o.is_external)
    sweep = mk(MarketOrder, "sell", volume=bids + 1 if bids else 1)
    sim.forward_step(t, internal_orders=[sweep, cancel(m, 15, t)])
    assert bs not in sim.exchange.internal_orderbook.buy
    sim.close()


def check_reset_episode_with_a_start_book(new_sim):
    """OrderbookSimulator.reset_episode(start_date, start_book) (OrderbookSimulator.py:55-68): the episode starts from the caller's book -
    here the historical one with its best bid thickened by two more orders and its worst ask taken out - the price ranges are that
    book's, and the messages of the stream are replayed against it."""
    ps = ol.golden_stream("msft")
    sim = _simulator(new_sim)
    start, _ = _episode_start(ps)
    hist = sim.reset_episode(start)
    bb, worst_ask = hist.best_buy_price, max(hist.sell)
    extra = [LimitOrder(timestamp=start, direction="buy", ticker=TICKER, internal_id=7, external_id=9001 + k, is_external=True, price=bb, volume=30 + k) for k in range(2)]
    hist.buy[bb] = list(hist.buy[bb]) + extra
    del hist.sell[worst_ask]
    book = sim.reset_episode(start, start_book=hist)
    assert [o.volume for o in book.buy[bb]][-2:] == [30, 31] and book.buy[bb][0].internal_id == -1 and worst_ask not in book.sell
    assert sim.max_sell_price == max(book.sell) and sim.min_buy_price == min(book.buy)
    assert sim.initial_buy_price_range == bb - min(book.buy) and sim.initial_sell_price_range == max(book.sell) - book.best_sell_price
    sim.forward_step(start + timedelta(seconds=1))
    assert sim.now_is == start + timedelta(seconds=1)
    sim.close()
