"""Synthetic LOBSTER-format streams (host tooling around csrc/synth_lob.c).

``generate_lobster`` returns raw LOBSTER columns + book rows; ``synthetic_stream`` packs them exactly
as a real LOBSTER file pair would be packed (:func:`rl4mm_amd.stream.pack_arrays`), and
``write_lobster_csv`` writes the pair to disk so the reference itself can ingest it (oracle/gen_golden.py).
"""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path
from typing import Optional

import numpy as np

from .stream import PackedStream, pack_arrays

_CSRC = Path(__file__).resolve().parent / "csrc"
_LIB_PATH = _CSRC / "libsynthlob.so"
_lib = None


def build_synth_lib(force: bool = False) -> Path:
    src = _CSRC / "synth_lob.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", str(_LIB_PATH), str(src), "-lm"])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build_synth_lib()
        _lib = ctypes.CDLL(str(_LIB_PATH))
        p = ctypes.POINTER(ctypes.c_int64)
        _lib.synth_lob_generate.restype = ctypes.c_int64
        _lib.synth_lob_generate.argtypes = [
            ctypes.c_uint64, ctypes.c_int, ctypes.c_double, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int64, p, p, p, p, p, p, p,
        ]  # fmt: skip
    return _lib


def generate_lobster(
    seed: int,
    n_levels: int = 10,
    msgs_per_s: float = 200.0,
    start_s: float = 9.5 * 3600,
    end_s: float = 16 * 3600,
    mid_price: int = 2_700_000,
    max_msgs: Optional[int] = None,
):
    """-> dict(time_s, type, id, size, price, direction, books[n, 4 * n_levels]) in LOBSTER conventions."""
    lib = _load()
    if max_msgs is None:
        max_msgs = int((end_s - start_s) * msgs_per_s * 1.3) + 1024
    cols = [np.zeros(max_msgs, dtype=np.int64) for _ in range(6)]
    books = np.zeros((max_msgs, 4 * n_levels), dtype=np.int64)
    p = ctypes.POINTER(ctypes.c_int64)
    n = lib.synth_lob_generate(
        seed, n_levels, msgs_per_s, mid_price, int(round(start_s * 1e9)), int(round(end_s * 1e9)), max_msgs,
        *[c.ctypes.data_as(p) for c in cols], books.ctypes.data_as(p),
    )  # fmt: skip
    t_ns, typ, oid, size, price, direction = [c[:n] for c in cols]
    return dict(
        time_s=t_ns / 1e9, time_ns=t_ns, type=typ, id=oid, size=size, price=price, direction=direction, books=books[:n]
    )


def synthetic_stream(
    seed: int,
    n_levels: int = 10,
    msgs_per_s: float = 200.0,
    start_s: float = 9.5 * 3600,
    end_s: float = 16 * 3600,
    mid_price: int = 2_700_000,
    ticker: str = "SYN",
    trading_date: str = "2018-02-21",
    book_snapshot_freq: Optional[str] = "S",
    max_msgs: Optional[int] = None,
) -> PackedStream:
    g = generate_lobster(seed, n_levels, msgs_per_s, start_s, end_s, mid_price, max_msgs)
    return pack_arrays(
        g["time_s"], g["type"], g["id"], g["size"], g["price"], g["direction"], g["books"],
        ticker, trading_date, n_levels, book_snapshot_freq,
    )  # fmt: skip


def write_lobster_csv(g: dict, directory, ticker: str, trading_date: str, n_levels: int):
    """Write ``{ticker}_{date}_{start}_{end}_message_{L}.csv`` / ``..._orderbook_{L}.csv`` like LOBSTER does."""
    directory = Path(directory)
    directory.mkdir(parents=True, exist_ok=True)
    stem = f"{ticker}_{trading_date}_34200000_57600000"
    mpath = directory / f"{stem}_message_{n_levels}.csv"
    bpath = directory / f"{stem}_orderbook_{n_levels}.csv"
    with open(mpath, "w") as f:
        for i in range(len(g["type"])):
            ns = int(g["time_ns"][i])
            f.write(
                f"{ns // 10**9}.{ns % 10**9:09d},{g['type'][i]},{g['id'][i]},{g['size'][i]},{g['price'][i]},{g['direction'][i]}\n"
            )
    np.savetxt(bpath, g["books"], fmt="%d", delimiter=",")
    return mpath, bpath
