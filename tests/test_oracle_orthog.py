"""The reference's own unit tests of OrthogonalizeColumn{MGS,CGS} (test/unit/test-orthog.cpp), restated on the oracle
(MFEM-free arithmetic): the cases "Real Empty", "Real 1", "Real 2" (with its known-answer
coefficients), "Complex 1", "Weighted - Real 1", "Weighted - Complex 1", each for MGS, CGS and CGS with refinement,
at communicator size 1."""
import numpy as np
import pytest

from oracle import palace_oracle as po

KINDS = ["MGS", "CGS", "CGS2"]
W3 = np.array([[2.0, 2.0, 0.0], [2.0, 1.0, 0.0], [0.0, 0.0, 2.0]])  # test-orthog.cpp:281-287


@pytest.mark.parametrize("kind", KINDS)
def test_real_empty(kind):  # test-orthog.cpp:98-121
    w = np.arange(4.0)
    H, w2 = po.orthogonalize_column(kind, [], w, 0)
    assert np.array_equal(w2, w) and H.size == 0


@pytest.mark.parametrize("kind", KINDS)
def test_real_1(kind):  # :123-158, mpi_size = 1
    V = [np.array([1.0, 0.0])]
    w = np.random.default_rng(0).uniform(-1, 1, 2)
    H, w2 = po.orthogonalize_column(kind, V, w, 1)
    assert abs(w2[0]) < 1e-12 and abs(np.dot(w2, V[0])) < 1e-12
    assert H[0] == pytest.approx(w[0])


@pytest.mark.parametrize("kind", KINDS)
def test_real_2(kind):  # :162-229, mpi_rank = 0, mpi_size = 1
    V = [np.array([1.0, 0, 0, 0]), np.array([0.0, 1, 0, 0])]
    H1, v1 = po.orthogonalize_column(kind, V, V[1], 1)
    assert np.array_equal(v1, [0.0, 1.0, 0.0, 0.0])  # exact: multiply by zero
    w = np.arange(4.0)
    H, w2 = po.orthogonalize_column(kind, V, w, 2)
    assert abs(np.dot(w2, V[0])) < 1e-12 and abs(np.dot(w2, V[1])) < 1e-12
    assert w2[2] == 2.0 and w2[3] == 3.0
    # H[0] = size (size - 1) / (2 |v0|) = 0, H[1] = size (size + 1) / (2 |v1|) = 1
    assert H[0] == pytest.approx(0.0, abs=1e-15) and H[1] == pytest.approx(1.0)


@pytest.mark.parametrize("kind", KINDS)
def test_complex_1(kind):  # :231-268
    V = [np.array([1.0 + 0.0j, 0.0])]
    rng = np.random.default_rng(1)
    w = rng.uniform(-1, 1, 2) + 1j * rng.uniform(-1, 1, 2)
    H, w2 = po.orthogonalize_column(kind, V, w, 1)
    assert abs(w2[0]) < 1e-12 and abs(np.vdot(V[0], w2)) < 1e-12


@pytest.mark.parametrize("kind", KINDS)
def test_weighted_real_1(kind):  # :270-317
    V = [np.array([1.0 / np.sqrt(2), 0, 0]), np.array([0, 0, 1.0 / np.sqrt(2)])]
    assert V[0] @ W3 @ V[0] == pytest.approx(1.0) and V[1] @ W3 @ V[1] == pytest.approx(1.0)
    w = np.random.default_rng(314159).uniform(-1, 1, 3)
    H, w2 = po.orthogonalize_column(kind, V, w, 2, weight=W3)
    assert abs(w2 @ (W3 @ V[0])) < 1e-12 and abs(w2 @ (W3 @ V[1])) < 1e-12


@pytest.mark.parametrize("kind", KINDS)
def test_weighted_complex_1(kind):  # :319-374
    V = [np.array([1.0 / np.sqrt(2), 0, 0], dtype=complex), np.array([0, 0, 1j / np.sqrt(2)])]
    rng = np.random.default_rng(314160)
    w = rng.uniform(-1, 1, 3) + 1j * rng.uniform(-1, 1, 3)
    H, w2 = po.orthogonalize_column(kind, V, w, 2, weight=W3)
    for v in V:
        assert abs(np.vdot(W3 @ v, w2)) < 1e-12  # Dot(w, W v) = (W v)^H w


def test_cgs2_is_more_orthogonal_than_cgs():
    """The point of the refinement pass (orthog.hpp:75-87): an ill-conditioned but normalised basis."""
    rng = np.random.default_rng(5)
    n, m = 200, 12
    Q, _ = np.linalg.qr(rng.normal(size=(n, m)))
    V = [Q[:, j] for j in range(m)]
    w = Q @ rng.normal(size=m) * 1e8 + rng.normal(size=n)
    _, w1 = po.orthogonalize_column("CGS", V, w, m)
    _, w2 = po.orthogonalize_column("CGS2", V, w, m)
    e1 = max(abs(np.dot(w1, v)) for v in V)
    e2 = max(abs(np.dot(w2, v)) for v in V)
    assert e2 < e1 and e2 < 1e-9 * np.linalg.norm(w2) + 1e-9
