"""Business-day NASDAQ calendar stand-in: schedule() rows = weekdays, first column = 09:30 open."""
import pandas as pd


class _Cal:
    def schedule(self, start_date, end_date):
        days = pd.bdate_range(pd.Timestamp(start_date).normalize(), pd.Timestamp(end_date).normalize())
        return pd.DataFrame(
            {"market_open": days + pd.Timedelta(hours=9, minutes=30), "market_close": days + pd.Timedelta(hours=16)},
            index=days,
        )


def get_calendar(name):
    return _Cal()
