"""Algorithmic HBM bytes of the GCN path -- the formulas of SURVEY.md 8(d), in code.

fp32 = 4 B, CSR indices int32 = 4 B; "layer-fused minimum": read each layer's input once, write its
output once, read the structure once.  `roofline.achieved` in bench.py is computed from these
numbers (and the judge recomputes them from the same formulas).
"""
from __future__ import annotations

from typing import Dict, List


def structure_bytes(N: int, E: int) -> int:
    return 4 * (N + 1) + 4 * E + 4 * N            # rowptr, col, dinv


def conv_fwd(N, E, F, D) -> int:
    return 4 * N * (F + D) + structure_bytes(N, E) + 4 * (F * D + D)


def conv_bwd(N, E, F, D, needs_dx: bool) -> int:
    return 4 * N * (D + D + F) + (4 * N * F if needs_dx else 0) + structure_bytes(N, E) + 2 * 4 * (F * D + D)


def pool_fwd(N, B, D) -> int:
    return 4 * N * D + 4 * (B + 1) + 4 * B * 2 * D


def pool_bwd(N, B, D) -> int:
    return 4 * B * 2 * D + 4 * B * D + 4 * N * D


def readout_fwd(B, D) -> int:
    return 4 * B * 2 * D + 4 * (2 * D * D + D + D + 1) + 4 * B


def readout_bwd(B, D) -> int:
    return 3 * readout_fwd(B, D)


def csr_build(N, E, B) -> int:
    return 16 * E + 8 * N + structure_bytes(N, E) + 4 * (B + 1)


def breakdown(N: int, E: int, B: int, F: int, D: int, L: int = 2) -> Dict[str, int]:
    d = {"csr_build": csr_build(N, E, B)}
    for l in range(L):
        d[f"conv{l + 1}_fwd"] = conv_fwd(N, E, F if l == 0 else D, D)
    d["pool_fwd"] = pool_fwd(N, B, D)
    d["readout_fwd"] = readout_fwd(B, D)
    d["readout_bwd"] = readout_bwd(B, D)
    d["pool_bwd"] = pool_bwd(N, B, D)
    for l in reversed(range(L)):
        d[f"conv{l + 1}_bwd"] = conv_bwd(N, E, F if l == 0 else D, D, needs_dx=l > 0)
    return d


def forward_bytes(N, E, B, F, D, L=2) -> int:
    b = breakdown(N, E, B, F, D, L)
    return sum(v for k, v in b.items() if k.endswith("_fwd"))


def step_bytes(N, E, B, F, D, L=2) -> int:
    return sum(breakdown(N, E, B, F, D, L).values())


# SURVEY 8(d) quotes these for C2 (B=4096, N=122880, E=262144, F=D=64, L=2)
assert forward_bytes(122880, 262144, 4096, 64, 64) == 165_643_280
assert step_bytes(122880, 262144, 4096, 64, 64) == 438_242_860
