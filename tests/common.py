"""Shared problem set-ups for the parity tests: the same problem built with the oracle and with the product."""
from __future__ import annotations

import numpy as np

from oracle import penguin_oracle as po
from oracle.geometry import Ball, MultiBall


def oracle_capacity_from_product(pcap, omesh, body=None) -> po.Capacity:
    """Wrap capacities computed by the HIP kernels into the oracle's Capacity (isolates the solve path)."""
    N = omesh.N
    Cg = pcap.C_γ
    return po.Capacity(tuple(pcap.A), tuple(pcap.B), pcap.V, tuple(pcap.W), pcap.C_ω, Cg, pcap.Γ, pcap.cell_types, omesh, body)


def rel_l2(a, b):
    d = np.linalg.norm(np.asarray(a) - np.asarray(b))
    n = np.linalg.norm(np.asarray(b))
    return d / n if n > 0 else d
