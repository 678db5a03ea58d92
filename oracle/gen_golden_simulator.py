#!/usr/bin/env python3
"""Build container only (test infrastructure: imports the upstream reference): records tests/golden/sim_internal_orders.npz - episodes of
`OrderbookSimulator.reset_episode` / `forward_step(until, internal_orders)` driven with internal-order lists of every shape, from the
REFERENCE (rl4mm/simulation/OrderbookSimulator.py:55-88, rl4mm/orderbook/Exchange.py:58-247): cancellations and deletions of the agent's
own orders ANYWHERE in a call's list - also behind own limit and market orders of the same call that cross the book or the agent's own
quotes, and at the price of a marketable own limit order - and one episode that starts from a caller's `start_book`.  Those are the calls
the simulator class refused until round 6 (the device now finds a cancellation's target by its internal id); the fixture pins them where
the reference does not travel: tests/test_sim_golden.py replays it through rl4mm_amd.simulator on the oracle (CPU suite) and on the HIP
library (`-m gpu`).

Run:  PYTHONPATH=oracle/refshim:/root/reference PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_simulator.py

Per case: the packed day, the start, the outer-level budget; per step `until` and the call's order rows (type, side, price, volume,
target) - a LimitOrder gets the next SUBMISSION NUMBER of the episode, a Cancellation / Deletion names its target by that number (the
reference's and the class's internal ids are each their own counters); after every step the FilledOrders (internal and external, in
order), the central book order by order, the internal book with the submission numbers, the price-range bookkeeping.  While recording,
the class on the oracle runs beside the reference and must agree at every step (as in oracle/fuzz_simulator.py): a fixture the
restatement cannot reproduce is never written."""
import contextlib
import copy
import io
import sys
from datetime import timedelta
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "oracle"))
sys.path.insert(0, str(REPO / "tests"))
sys.argv = sys.argv[:1]
import fuzz_reference as fr  # noqa: E402
import fuzz_simulator as fs  # noqa: E402
import gen_golden as gg  # noqa: E402
import oracle_lib as ol  # noqa: E402
from rl4mm_amd import simulator as S  # noqa: E402

from rl4mm.orderbook import models as RM  # noqa: E402

T_LIMIT, T_CANCEL, T_DELETE, T_MARKET = 1, 2, 3, 4
BIG = 2**62  # (an infinite price bound)


def _bound(v):
    return int(v) if np.isfinite(v) else (BIG if v > 0 else -BIG)


def record(seed, kind, with_start_book=False, n_steps=32):
    r = np.random.Generator(np.random.PCG64(88_000 + seed))
    db, ps, tick, L, t0, t1 = fr.day(kind)
    outer = int(r.choice([5, 20, 40]))
    ref = gg.OrderbookSimulator(ticker=tick, order_generators=[gg.HistoricalOrderGenerator(tick, db, True)], n_levels=L, database=db, preload_orders=True,
                                episode_length=timedelta(seconds=t1 - t0), warm_up=timedelta(0), outer_levels=outer)  # fmt: skip
    mine = S.OrderbookSimulator(ticker=tick, n_levels=L, database=ps, episode_length=timedelta(seconds=t1 - t0), outer_levels=outer, _new_sim=ol.oracle_sim)
    start_s = int(r.integers(t0 + 1, max(t0 + 2, t1 - 30)))
    start = fr.DATE + timedelta(seconds=start_s)
    rb, mb = ref.reset_episode(start_date=start), mine.reset_episode(start)
    book_edit = np.zeros((0, 4), dtype=np.int64)
    if with_start_book:  # the historical book with its best bid thickened by two external orders and its worst ask taken out (OrderbookSimulator.py:55-58)
        bb, worst_ask = int(mb.best_buy_price), int(max(mb.sell))
        kw = lambda k: dict(timestamp=start, direction="buy", ticker=tick, internal_id=7, external_id=9001 + k, is_external=True, price=bb, volume=30 + k)  # noqa: E731
        rb2 = copy.deepcopy(rb)
        for k in range(2):
            rb2.buy[bb].append(RM.LimitOrder(**kw(k)))
        del rb2.sell[worst_ask]
        mb.buy[bb] = list(mb.buy[bb]) + [S.LimitOrder(**kw(k)) for k in range(2)]
        del mb.sell[worst_ask]
        rb, mb = ref.reset_episode(start_date=start, start_book=rb2), mine.reset_episode(start, start_book=mb)
        book_edit = np.array([[0, bb, 30, 9001], [0, bb, 31, 9002], [1, worst_ask, -1, 0]], dtype=np.int64)  # (side, price, volume of an added order | -1: level removed, external id)
    assert fs.book_rows(rb) == fs.book_rows(mb), "start books differ"
    now = start
    sub_r, sub_m = {}, {}  # internal id -> submission number (reference / class)
    n_sub = 0
    alive = []  # [(submission number, direction, price, volume)] of the own orders in the book, plus a few that are gone
    until_us, rows, fills, central, internal, ranges = [], [], [], [], [], []
    day0 = fr.DATE
    behind = 0
    for step in range(n_steps):
        dt = float(r.choice([0.05, 0.1, 0.1, 0.25, 0.5, 1.0, 2.0]))
        until = now + timedelta(seconds=dt)
        if until > fr.DATE + timedelta(seconds=t1 - 2):
            break
        if r.random() < 0.3:
            until = until.replace(microsecond=0) + timedelta(seconds=1)
        call = []  # (type, side name, price, volume or None, target submission number)
        bb, bs = ref.exchange.best_buy_price, ref.exchange.best_sell_price
        for _ in range(int(r.choice([0, 1, 1, 2, 3, 4, 5]))):
            what = r.random()
            if what < 0.4 and alive:
                tgt, direction, price, v0 = alive[int(r.integers(0, len(alive)))]
                whole = r.random() < 0.5
                vol = None if whole and r.random() < 0.5 else int(v0 if whole else max(1, int(r.integers(1, v0 + 1))))
                at = int(r.integers(0, len(call) + 1))  # anywhere: in front of, between and behind the call's other orders
                behind += any(c[0] in (T_LIMIT, T_MARKET) for c in call[:at])
                call.insert(at, (T_DELETE if vol is None else T_CANCEL, direction, price, vol, tgt))
            elif what < 0.85:
                side = "buy" if r.random() < 0.5 else "sell"
                ref_p = bb if side == "buy" else bs
                if not np.isfinite(ref_p) or ref_p <= 0: continue
                off = int(r.choice([-3, -2, -1, 0, 0, 1, 2, 4, 8]))  # > 0: away from the touch; <= 0: at and through it (marketable)
                price = int(ref_p - off * 100) if side == "buy" else int(ref_p + off * 100)
                if price <= 0: continue
                call.append((T_LIMIT, side, price, int(r.choice([1, 5, 10, 50, 200])), -1))
            else:
                call.append((T_MARKET, "buy" if r.random() < 0.5 else "sell", 0, int(r.choice([1, 5, 10, 100])), -1))
        ro, mo = [], []
        inv_r, inv_m = {v: k for k, v in sub_r.items()}, {v: k for k, v in sub_m.items()}
        for typ, side, price, vol, tgt in call:
            base = dict(timestamp=now, direction=side, ticker=tick, external_id=None, is_external=False)
            if typ == T_LIMIT:
                ro.append(RM.LimitOrder(internal_id=None, price=price, volume=vol, **base)); mo.append(S.LimitOrder(internal_id=None, price=price, volume=vol, **base))
            elif typ == T_MARKET:
                ro.append(RM.MarketOrder(internal_id=None, volume=vol, **base)); mo.append(S.MarketOrder(internal_id=None, volume=vol, **base))
            else:
                cr, cm = (RM.Deletion, S.Deletion) if typ == T_DELETE else (RM.Cancellation, S.Cancellation)
                ro.append(cr(internal_id=inv_r[tgt], price=price, volume=vol, **base)); mo.append(cm(internal_id=inv_m[tgt], price=price, volume=vol, **base))
        lim_m = [o for o in mo if isinstance(o, S.LimitOrder)]
        err_r = err_m = None
        try:
            fm = mine.forward_step(until, mo)
        except Exception as ex:  # noqa: BLE001
            err_m = type(ex).__name__
        with contextlib.redirect_stdout(io.StringIO()):
            try:
                fr_ = ref.forward_step(until=until, internal_orders=list(ro))
            except Exception as ex:  # noqa: BLE001
                err_r = type(ex).__name__
        assert (err_r is None) == (err_m is None), f"step {step}: reference {err_r}, class {err_m}"
        until_us.append(int((until - day0) / timedelta(microseconds=1)))
        for typ, side, price, vol, tgt in call:
            sub = -1
            if typ == T_LIMIT:
                sub, n_sub = n_sub, n_sub + 1
            rows.append((step, typ, 0 if side == "buy" else 1, price, -1 if vol is None else vol, tgt, sub))
        if err_r is not None:
            rows.append((step, -1, 0, 0, 0, -1, -1))  # the call raised: the episode ends here
            break
        # the class stamps every LimitOrder it submits with its internal id; the reference files a COPY of the order under a new id when it rests
        # (OrderIDConvertor.py:12-18) and gives none to an order that executes at once: its ids are learnt from the internal book below, order by
        # order against the class's (equal) book
        k0 = n_sub - len(lim_m)
        for k, o_m in enumerate(lim_m):
            assert o_m.internal_id is not None
            sub_m[o_m.internal_id] = k0 + k
        for which, name in ((0, "internal"), (1, "external")):
            a = [(o.direction, int(o.price), int(o.volume)) for o in getattr(fr_, name)]
            b = [(o.direction, int(o.price), int(o.volume)) for o in getattr(fm, name)]
            assert a == b, f"step {step}: {name} fills differ"
            fills += [(step, which, 0 if d == "buy" else 1, p, v) for d, p, v in a]
        a, b = fs.book_rows(ref.exchange.central_orderbook), fs.book_rows(mine.exchange.central_orderbook)
        assert a == b, f"step {step}: central books differ"
        central += [(step, s_, p, v, int(x)) for s_, p, v, x in a]
        ib_r, ib_m = ref.exchange.internal_orderbook, mine.exchange.internal_orderbook
        now_alive = []
        for s_, side_name in ((0, "buy"), (1, "sell")):
            dr, dm = getattr(ib_r, side_name), getattr(ib_m, side_name)
            assert list(dr.keys()) == list(dm.keys()), f"step {step}: internal books differ"
            prices = list(dr.keys())
            for p in (reversed(prices) if s_ == 0 else prices):
                assert [int(o.volume) for o in dr[p]] == [int(o.volume) for o in dm[p]], f"step {step}: internal books differ"
                for o_r, o_m in zip(dr[p], dm[p]):
                    if o_m.internal_id not in sub_m:  # an own order the outer-level resync took out and put back (OrderbookSimulator.py:105-135): a new order with a
                        sub_m[o_m.internal_id], n_sub = n_sub, n_sub + 1  # new id on both sides - it gets the next submission number, in the book's order
                    assert sub_r.setdefault(o_r.internal_id, sub_m[o_m.internal_id]) == sub_m[o_m.internal_id], f"step {step}: internal books differ in who is who"
                for o in dr[p]:
                    internal.append((step, s_, int(p), int(o.volume), sub_r[o.internal_id]))
                    now_alive.append((sub_r[o.internal_id], side_name, int(p), int(o.volume)))
        assert (ref.min_buy_price, ref.max_sell_price) == (mine.min_buy_price, mine.max_sell_price)
        ranges.append((_bound(ref.min_buy_price), _bound(ref.max_sell_price)))
        gone = [e for e in alive if e[0] not in {x[0] for x in now_alive}]
        alive = now_alive + gone[-2:]  # (a cancellation of an order that is gone already: the reference warns and goes on)
        now = until
    mine.close()
    A = lambda x, w: np.array(x, dtype=np.int64).reshape(-1, w)  # noqa: E731
    case = dict(msgs=ps.msgs, snaps=ps.snaps, meta=np.array([L, outer, int((start - day0) / timedelta(microseconds=1)), t1 - t0], dtype=np.int64), ticker=np.array(tick), date=np.array(ps.trading_date),
                until_us=np.array(until_us, dtype=np.int64), orders=A(rows, 7), fills=A(fills, 5), central=A(central, 5), internal=A(internal, 5), ranges=A(ranges, 2), book_edit=book_edit)  # fmt: skip
    info = dict(kind=kind, steps=len(until_us), orders=len(rows), cancels_behind_own_orders=behind, fills=len(fills), raised=err_r)
    return case, info


if __name__ == "__main__":
    out, n = {}, 0
    # (the days of oracle/fuzz_reference.py: two small synthetic ones, the drifting one whose outer levels resync, the three-level thin
    # books - one of them crossed - and the 10-level day; the last case starts from a caller's book)
    # (`want`: the first seed from the given one on whose episode runs that many steps - a market order for more than a thin side holds, or a
    # cancellation that asks for more than is left, raises in the reference and ends the episode: two such endings are kept on purpose)
    for seed, kind, sb, want in ((1, "syn5", False, 32), (2, "syn5", False, 32), (3, "drift", False, 6), (4, "thin", False, 0), (5, "thinx", False, 0), (6, "syn10", False, 30), (7, "syn5", True, 25), (8, "thin", True, 8)):
        for attempt in range(40):
            case, info = record(seed, kind, sb)
            if info["steps"] >= want:
                break
            print(f"   (seed {seed}: {info['steps']} steps, {info['raised']})", flush=True)
            seed += 100
        else:
            raise SystemExit(f"no seed of {kind} reaches {want} steps")
        print(f"case {n}: seed {seed} {info}", flush=True)
        for k, v in case.items():
            out[f"c{n}_{k}"] = v
        n += 1
    out["n_cases"] = np.array(n)
    path = REPO / "tests" / "golden" / "sim_internal_orders.npz"
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({path.stat().st_size / 1024:.0f} KiB)")
