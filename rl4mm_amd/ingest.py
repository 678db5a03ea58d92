"""LOBSTER files and archives -> StreamCatalog of packed streams, skipping the SQL database.

Replaces, as a *format* step, the reference's ingest drivers: ``populate_database`` over a set of tickers and trading
days (rl4mm/database/populate_database.py:38-95) and its archive driver (run_populate_database_from_zipped.py:50-109:
monthly LOBSTER ``.7z`` downloads named ``_data_dwn_<..>__<TICKER>_<start>_<end>_<levels>.7z``, processed in date order,
extracted, ingested day by day, CSVs deleted).  Day selection uses the NASDAQ session calendar like the reference
(rl4mm/utils/utils.py:48-62, here rl4mm_amd/trading_calendar.py).  A packed day can be cached as ``.npz`` next to the
data, which is what makes re-runs cheap (the reference's equivalent is "already in database", populate_database.py:61-63).
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import tarfile
import zipfile
from datetime import datetime
from pathlib import Path
from typing import Iterable, Optional, Sequence, Tuple, Union

from .stream import PackedStream, pack_lobster
from .trading_calendar import get_trading_datetimes


def get_book_and_message_paths(data_path: Union[str, Path], ticker: str, trading_date: str, n_levels: int) -> Tuple[Path, Path]:
    """rl4mm/database/database_population_helpers.py:107-113 (also accepts gzip-compressed CSVs)."""
    for ext in ("csv", "csv.gz"):
        books = glob.glob(f"{data_path}/{ticker}_{trading_date}_*_orderbook_{n_levels}.{ext}")
        msgs = glob.glob(f"{data_path}/{ticker}_{trading_date}_*_message_{n_levels}.{ext}")
        if books and msgs:
            return Path(books[0]), Path(msgs[0])
    raise FileNotFoundError(f"Level {n_levels} data for ticker {ticker} on {trading_date} not found in {data_path}")


def pack_day(data_path, ticker: str, trading_date: str, n_levels: int, book_snapshot_freq: Optional[str] = "S", max_rows: Optional[int] = None,
             batch_size: int = 1_000_000, cache_dir: Union[str, Path, None] = None) -> PackedStream:  # fmt: skip
    """One (ticker, day): the packed stream, from the cache when it is there."""
    cache = None
    if cache_dir is not None:
        Path(cache_dir).mkdir(parents=True, exist_ok=True)
        cache = Path(cache_dir) / f"{ticker}_{trading_date}_L{n_levels}_{book_snapshot_freq or 'all'}_{max_rows or 'all'}_{batch_size}.npz"
        if cache.exists():
            return PackedStream.load(cache)
    book_path, message_path = get_book_and_message_paths(data_path, ticker, trading_date, n_levels)
    ps = pack_lobster(message_path, book_path, ticker, trading_date, n_levels, book_snapshot_freq, max_rows, batch_size)
    if cache is not None:
        ps.save(cache)
    return ps


def populate_catalog(catalog, tickers: Sequence[str], trading_datetimes: Iterable[datetime], n_levels: int = 50, path_to_lobster_data: str = "",
                     book_snapshot_freq: Optional[str] = None, max_rows: Optional[int] = None, batch_size: int = 1_000_000, cache_dir=None, skip_missing: bool = False):  # fmt: skip
    """populate_database (populate_database.py:38-95) with a StreamCatalog in the place of the database: every (ticker,
    trading day) pair is packed and added; a pair that is already in the catalog is not re-added (:61-63)."""
    for ticker in tickers:
        for dt in trading_datetimes:
            trading_date = datetime.strftime(dt, "%Y-%m-%d")
            if (ticker, trading_date) in dict(catalog.items()):
                continue
            try:
                catalog.add(pack_day(path_to_lobster_data, ticker, trading_date, n_levels, book_snapshot_freq, max_rows, batch_size, cache_dir))
            except FileNotFoundError:
                if not skip_missing:
                    raise
    return catalog


def _extract(archive: Path, out_dir: Path) -> None:
    name = archive.name.lower()
    if name.endswith(".zip"):
        with zipfile.ZipFile(archive) as z:
            z.extractall(out_dir)
    elif name.endswith((".tar", ".tar.gz", ".tgz", ".tar.xz")):
        with tarfile.open(archive) as t:
            t.extractall(out_dir)
    elif name.endswith(".7z"):
        exe = next((e for e in ("7z", "7za", "7zr") if shutil.which(e)), None)
        if exe is None:
            raise RuntimeError(f"{archive}: no 7z executable on PATH (the reference shells out to `7z x` as well, run_populate_database_from_zipped.py:100)")
        subprocess.run([exe, "x", "-y", str(archive), "-o" + str(out_dir)], check=True, stdout=subprocess.DEVNULL)
    else:
        raise ValueError(f"unknown archive type: {archive}")


def parse_archive_name(filename: str):
    """``_data_dwn_50_385__KO_2018-04-01_2018-04-30_200.7z`` -> (ticker, start, end, n_levels)
    (run_populate_database_from_zipped.py:88-92)."""
    stem = os.path.basename(filename)
    for ext in (".tar.gz", ".tar.xz", ".tgz", ".tar", ".zip", ".7z"):
        if stem.lower().endswith(ext):
            stem = stem[: -len(ext)]
            break
    parts = stem.split("_")
    return parts[-4], datetime.strptime(parts[-3], "%Y-%m-%d"), datetime.strptime(parts[-2], "%Y-%m-%d"), int(parts[-1])


def populate_catalog_from_archives(catalog, path_to_lobster_data: Union[str, Path], book_snapshot_freq: Optional[str] = "S", batch_size: int = 1_000_000,
                                   cache_dir=None, keep_csvs: bool = False):  # fmt: skip
    """run_populate_database_from_zipped.py:50-109: every archive of the directory in chronological order (sorted on what
    follows the double underscore, i.e. ticker and dates, not on the download counter in front), extracted next to it,
    its trading days packed into the catalog, the extracted CSVs removed again."""
    root = Path(path_to_lobster_data)
    archives = [p for p in root.iterdir() if p.name.lower().endswith((".7z", ".zip", ".tar", ".tar.gz", ".tgz", ".tar.xz"))]
    archives.sort(key=lambda p: p.name.split("__")[1] if "__" in p.name else p.name)
    for archive in archives:
        ticker, start, end, n_levels = parse_archive_name(archive.name)
        days = get_trading_datetimes(start, end)
        have = dict(catalog.items())
        if all((ticker, d.strftime("%Y-%m-%d")) in have for d in days):
            continue  # "already in database so not re-added" (:96-99)
        before = set(glob.glob(str(root / "*.csv")))
        _extract(archive, root)
        try:
            populate_catalog(catalog, (ticker,), days, n_levels, str(root), book_snapshot_freq, None, batch_size, cache_dir, skip_missing=True)
        finally:
            if not keep_csvs:
                for f in set(glob.glob(str(root / "*.csv"))) - before:  # delete_csvs (:44-47), only what this archive brought
                    os.remove(f)
    return catalog
