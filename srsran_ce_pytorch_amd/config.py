"""Configuration containers of the estimator boundary.

Field names and meaning mirror the reference's two dataclasses
(`src/ce_rule_tensorized.py:13-29`) so its validation harness
(`scripts/validation/validate_all.py:445-465`) can construct them unchanged.  Any
duck-typed object with the same attributes is accepted by the plan builder; tensors,
numpy arrays and Python lists are all fine for the array fields.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any

import torch


@dataclass
class HopConfig:
    DMRSsymbols: Any          # (n_sym,) bool -- OFDM symbols that carry DM-RS in this hop
    DMRSREmask: Any           # (12, nCDM) bool -- DM-RS REs inside one PRB, one column per CDM group
    PRBstart: int             # first PRB of the (contiguous) allocation, 0-based
    nPRBs: int                # PRBs in the allocation
    maskPRBs: Any             # (n_prb_grid,) bool -- allocated PRBs over the whole grid
    startSymbol: int          # first allocated OFDM symbol of the hop, 0-based
    nAllocatedSymbols: int    # allocated OFDM symbols in the hop


@dataclass
class EstimatorConfig:
    scs: float                     # subcarrier spacing, Hz
    CyclicPrefixDurations: Any     # (>=14,) cyclic-prefix durations, milliseconds
    Smoothing: str = "filter"      # "filter" | "mean" | "none"
    CFOCompensate: bool = True


def empty_hop(device: str | torch.device = "cpu") -> HopConfig:
    """The "no second hop" encoding used by the reference harness (validate_all.py:449-457)."""
    return HopConfig(
        DMRSsymbols=torch.zeros((0,), dtype=torch.bool, device=device),
        DMRSREmask=torch.zeros((12, 0), dtype=torch.bool, device=device),
        PRBstart=0,
        nPRBs=0,
        maskPRBs=torch.zeros((0,), dtype=torch.bool, device=device),
        startSymbol=0,
        nAllocatedSymbols=0,
    )
