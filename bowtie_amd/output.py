"""A hit as the C ABI returns it (bt_hit + its slice of the mismatch pool), as a Python record.  The output formats
themselves are written by bowtie_amd/csrc/bt_io.cpp (bt_format_hits / bt_format_pairs, bound in hostio.py)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass
class Hit:
    tidx: int
    toff: int
    oms: int
    cost: int
    stratum: int
    fw: bool
    mms: List[Tuple[int, int]] = field(default_factory=list)   # (5'-relative pos, refc 0..3), by pos
    mate: int = 0              # 1 / 2: mate of a paired alignment (its partner is the adjacent hit)
