"""srsRAN unit-test vector formats, so the estimator can be checked against real
`port_channel_estimator_test_*.dat` files when someone supplies them (they are git-ignored upstream and absent
here; the reference's own scripts keep working unchanged through `compat/`).

Formats, as the reference's harness reads them:
* resource-grid entries (`..._input_rg*.dat`, `..._output_ch_est*.dat`): 12-byte little-endian records
  `<HHff` = `(symbol << 8 | port, subcarrier, re, im)` (`scripts/validation/validate_all.py:28-49`,
  `validate_case0.py:15-36`);
* pilots (`..._pilots*.dat`): a raw complex64 stream whose axis order varies by case
  (`validate_all.py:306-344`; case 0 is `[symbol, re, layer]`, `validate_case0.py:156-159`).
"""
from __future__ import annotations

from pathlib import Path
from typing import Iterable, Tuple

import numpy as np

ENTRY_DTYPE = np.dtype([("sym_port", "<u2"), ("sc", "<u2"), ("re", "<f4"), ("im", "<f4")])


def read_entries(path) -> np.ndarray:
    """Structured array with fields symbol, port, sc, value (complex64)."""
    raw = Path(path).read_bytes()
    if len(raw) % ENTRY_DTYPE.itemsize:
        raise ValueError(f"{path} size {len(raw)} not multiple of {ENTRY_DTYPE.itemsize} bytes.")
    rec = np.frombuffer(raw, dtype=ENTRY_DTYPE)
    out = np.empty(rec.size, dtype=[("symbol", "i4"), ("port", "i4"), ("sc", "i4"), ("value", "c8")])
    out["symbol"] = rec["sym_port"] >> 8
    out["port"] = rec["sym_port"] & 0xFF
    out["sc"] = rec["sc"]
    out["value"] = rec["re"] + 1j * rec["im"]
    return out


def write_entries(path, symbols: Iterable[int], ports: Iterable[int], scs: Iterable[int], values: Iterable[complex]) -> None:
    symbols, ports, scs = np.asarray(symbols), np.asarray(ports), np.asarray(scs)
    values = np.asarray(values, np.complex64)
    rec = np.empty(values.size, dtype=ENTRY_DTYPE)
    rec["sym_port"] = (symbols.astype(np.uint16) << 8) | ports.astype(np.uint16)
    rec["sc"] = scs
    rec["re"], rec["im"] = values.real, values.imag
    Path(path).write_bytes(rec.tobytes())


def entries_to_grid(entries: np.ndarray, n_sc: int, n_sym: int = 14) -> np.ndarray:
    """Dense `(n_sc, n_sym, n_ports)` complex64 grid, zeros where the file has no entry."""
    n_ports = int(entries["port"].max()) + 1 if entries.size else 1
    grid = np.zeros((n_sc, n_sym, n_ports), np.complex64)
    grid[entries["sc"], entries["symbol"], entries["port"]] = entries["value"]
    return grid


def grid_to_entries(grid: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Non-zero REs of a `(n_sc, n_sym, n_ports)` grid as (symbols, ports, subcarriers, values)."""
    sc, sym, port = np.nonzero(grid)
    return sym, port, sc, grid[sc, sym, port]


def read_pilots(path, n_dmrs_symbols: int, n_re: int, n_layers: int, order: str = "sym-re-layer") -> np.ndarray:
    """Pilots as `[re, symbol, layer]` (the estimator's axis order, T:760) from a raw complex64 stream stored
    in `order` (any permutation of "sym", "re", "layer")."""
    axes = order.split("-")
    sizes = {"sym": n_dmrs_symbols, "re": n_re, "layer": n_layers}
    data = np.fromfile(path, dtype=np.complex64)
    if data.size != n_dmrs_symbols * n_re * n_layers or sorted(axes) != ["layer", "re", "sym"]:
        raise ValueError("pilot file size / axis order mismatch")
    arr = data.reshape([sizes[a] for a in axes])
    return np.ascontiguousarray(arr.transpose(axes.index("re"), axes.index("sym"), axes.index("layer")))


def compare_at_entries(estimate: np.ndarray, entries: np.ndarray) -> Tuple[float, float]:
    """(max |diff|, RMS |diff|) of `estimate[sc, symbol, port]` over the coordinates the file lists only --
    the harness's comparison protocol (`validate_all.py:547-561`)."""
    d = estimate[entries["sc"], entries["symbol"], entries["port"]].astype(np.complex128) - entries["value"].astype(np.complex128)
    return float(np.abs(d).max()), float(np.sqrt(np.mean(np.abs(d) ** 2)))
