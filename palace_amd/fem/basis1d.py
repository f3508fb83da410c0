"""1-D building blocks of the tensor-product hex elements.

Palace fixes the 1-D bases of its Nedelec / H1 collections to closed Gauss-Lobatto (`b1`) and
open Gauss-Legendre (`b2`) nodal bases (reference: palace/fem/multigrid.hpp:35,49) and the
quadrature to a Gauss-Legendre tensor rule of order 2p (palace/fem/integrator.cpp:14-22,
palace/utils/configfile.hpp:1134-1138 => p+1 points per direction on hexes).

MFEM, which evaluates those bases for Palace, is not vendored in the reference tree; the
definitions below restate the public definitions (nodal Lagrange bases on the Gauss-Legendre /
Gauss-Lobatto points of [0,1]).
"""
from __future__ import annotations

import numpy as np


def gauss_legendre(n: int):
    """n-point Gauss-Legendre rule on [0,1]: (points, weights)."""
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def gauss_lobatto(n: int) -> np.ndarray:
    """n Gauss-Lobatto points on [0,1] (n >= 2): endpoints + roots of P'_{n-1}."""
    if n < 2:
        raise ValueError("Gauss-Lobatto needs >= 2 points")
    if n == 2:
        return np.array([0.0, 1.0])
    N = n - 1
    # Chebyshev-Gauss-Lobatto initial guess, Newton on (1-x^2) P'_N(x).
    x = -np.cos(np.pi * np.arange(n) / N)
    for _ in range(100):
        P = np.zeros((n, N + 1))
        P[:, 0] = 1.0
        P[:, 1] = x
        for k in range(2, N + 1):
            P[:, k] = ((2 * k - 1) * x * P[:, k - 1] - (k - 1) * P[:, k - 2]) / k
        dx = (x * P[:, N] - P[:, N - 1]) / ((N + 1) * P[:, N])
        x = x - dx
        if np.max(np.abs(dx)) < 1e-16:
            break
    x[0], x[-1] = -1.0, 1.0
    x = 0.5 * (x - x[::-1])  # symmetrise
    return 0.5 * (x + 1.0)


def lagrange_eval(nodes: np.ndarray, x: np.ndarray):
    """Values and derivatives of the Lagrange basis on `nodes` at points `x`.

    Returns (B, G) with B[q, i] = l_i(x_q), G[q, i] = l_i'(x_q).
    """
    nodes = np.asarray(nodes, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    n = nodes.size
    B = np.ones((x.size, n))
    G = np.zeros((x.size, n))
    for i in range(n):
        denom = 1.0
        for m in range(n):
            if m != i:
                denom *= nodes[i] - nodes[m]
        for q in range(x.size):
            val = 1.0
            for m in range(n):
                if m != i:
                    val *= x[q] - nodes[m]
            B[q, i] = val / denom
            d = 0.0
            for m in range(n):
                if m == i:
                    continue
                t = 1.0
                for l in range(n):
                    if l != i and l != m:
                        t *= x[q] - nodes[l]
                d += t
            G[q, i] = d / denom
    return B, G


class Tables1D:
    """The 1-D tables a tensor-product hex element of order p needs at Q1d quadrature points.

    Bc/Gc: closed (Gauss-Lobatto, p+1 nodes) basis values / derivatives, shape [Q1d, p+1].
    Bo:    open (Gauss-Legendre, p nodes) basis values, shape [Q1d, p].
    """

    def __init__(self, p: int, q1d: int):
        self.p = p
        self.q1d = q1d
        self.qx, self.qw = gauss_legendre(q1d)
        self.cp = gauss_lobatto(p + 1)
        self.op, _ = gauss_legendre(p)
        self.Bc, self.Gc = lagrange_eval(self.cp, self.qx)
        self.Bo, self.Go = lagrange_eval(self.op, self.qx)
