"""NASDAQ trading-day calendar (host side of env.reset(): rl4mm/utils/utils.py:48-57, called from
rl4mm/gym/HistoricalOrderbookEnvironment.py:346-349).

The reference asks ``pandas_market_calendars.get_calendar("NASDAQ")`` for the next session at or after a drawn business
day.  That package is not a dependency here; the exchange's holiday rules are few and fixed, so they are restated:
New Year's Day, Martin Luther King Jr. Day (from 1998), Washington's Birthday, Good Friday, Memorial Day, Juneteenth
(from 2022), Independence Day, Labor Day, Thanksgiving, Christmas - with the exchange's observance rule (a holiday on a
Sunday is observed on Monday; on a Saturday, on the Friday before, except New Year's Day, which is then not observed) -
plus the unscheduled full-day closures since 2001.  Early closes (13:00) are sessions and do not matter for day selection.
"""
from __future__ import annotations

from datetime import date, datetime, timedelta
from functools import lru_cache

# unscheduled full-day closures (national days of mourning, 9/11, Hurricane Sandy)
_SPECIAL_CLOSURES = {
    date(2001, 9, 11), date(2001, 9, 12), date(2001, 9, 13), date(2001, 9, 14),
    date(2004, 6, 11),   # Reagan
    date(2007, 1, 2),    # Ford
    date(2012, 10, 29), date(2012, 10, 30),  # Sandy
    date(2018, 12, 5),   # G. H. W. Bush
    date(2025, 1, 9),    # Carter
}  # fmt: skip


def _nth_weekday(year: int, month: int, weekday: int, n: int) -> date:
    d = date(year, month, 1)
    d += timedelta(days=(weekday - d.weekday()) % 7)
    return d + timedelta(weeks=n - 1)


def _last_weekday(year: int, month: int, weekday: int) -> date:
    d = date(year + (month == 12), month % 12 + 1, 1) - timedelta(days=1)
    return d - timedelta(days=(d.weekday() - weekday) % 7)


def _easter(year: int) -> date:
    """Gregorian Easter Sunday (anonymous algorithm)."""
    a, b, c = year % 19, year // 100, year % 100
    d, e = b // 4, b % 4
    f = (b + 8) // 25
    g = (b - f + 1) // 3
    h = (19 * a + b - d - g + 15) % 30
    i, k = c // 4, c % 4
    l = (32 + 2 * e + 2 * i - h - k) % 7
    m = (a + 11 * h + 22 * l) // 451
    month, day = divmod(h + l - 7 * m + 114, 31)
    return date(year, month, day + 1)


def _observed(d: date, saturday_to_friday: bool = True):
    if d.weekday() == 6:
        return d + timedelta(days=1)
    if d.weekday() == 5:
        return d - timedelta(days=1) if saturday_to_friday else None
    return d


@lru_cache(maxsize=None)
def holidays(year: int) -> frozenset:
    out = {
        _observed(date(year, 1, 1), saturday_to_friday=False),
        _nth_weekday(year, 2, 0, 3),              # Washington's Birthday: third Monday of February
        _easter(year) - timedelta(days=2),        # Good Friday
        _last_weekday(year, 5, 0),                # Memorial Day
        _observed(date(year, 7, 4)),
        _nth_weekday(year, 9, 0, 1),              # Labor Day
        _nth_weekday(year, 11, 3, 4),             # Thanksgiving: fourth Thursday of November
        _observed(date(year, 12, 25)),
    }  # fmt: skip
    if year >= 1998:
        out.add(_nth_weekday(year, 1, 0, 3))      # Martin Luther King Jr. Day
    if year >= 2022:
        out.add(_observed(date(year, 6, 19)))     # Juneteenth
    out.discard(None)
    return frozenset(out | {d for d in _SPECIAL_CLOSURES if d.year == year})


def is_trading_day(d) -> bool:
    d = d.date() if isinstance(d, datetime) else d
    return d.weekday() < 5 and d not in holidays(d.year)


def get_next_trading_dt(timestamp: datetime) -> datetime:
    """rl4mm/utils/utils.py:54-57: 09:30 of the first session on or after `timestamp`'s date."""
    d = timestamp.date()
    while not is_trading_day(d):
        d += timedelta(days=1)
    return datetime.combine(d, datetime.min.time()) + timedelta(hours=9, minutes=30)


def get_last_trading_dt(timestamp: datetime) -> datetime:
    """rl4mm/utils/utils.py:48-51: 16:00 of the last session on or before `timestamp`'s date."""
    d = timestamp.date()
    while not is_trading_day(d):
        d -= timedelta(days=1)
    return datetime.combine(d, datetime.min.time()) + timedelta(hours=16)


def get_trading_datetimes(start_date: datetime, end_date: datetime):
    """rl4mm/utils/utils.py:60-62: the sessions between two dates (inclusive)."""
    d, out = start_date.date(), []
    while d <= end_date.date():
        if is_trading_day(d):
            out.append(datetime.combine(d, datetime.min.time()))
        d += timedelta(days=1)
    return out
