"""The device cases of the reference's vector unit test (test/unit/test-vector.cpp: "Vector Sum - Real" :16-40,
"Vector Sum - Complex" :75-110, "Sqrt function" :315-348) at communicator size 1, plus sizes that exercise the
16-byte-lane / tail / unaligned paths."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


@pytest.fixture(scope="module")
def ctx():
    from palace_amd import linalg

    return linalg.Context()


def test_vector_sum_real_and_complex(ctx):
    rank, size = 0, 1
    v = _dev([rank * 3 + i + 1 for i in range(3)])
    assert ctx.sum(v) == pytest.approx(9.0 * size * (size - 1) / 2.0 + 6.0 * size, rel=1e-15)
    re, im = _dev([rank, rank]), _dev([rank + 0, rank + 1])
    s = complex(ctx.sum(re), ctx.sum(im))  # Sum(ComplexVector) = (Sum(Real), Sum(Imag)), vector.cpp:696-699
    assert s.real == pytest.approx(0.0, abs=1e-15) and s.imag == pytest.approx(1.0, rel=1e-15)


@pytest.mark.parametrize("offset", [0, 1])
@pytest.mark.parametrize("n", [0, 1, 2, 5, 256, 257, 100003])
def test_vector_sum_any_size(ctx, n, offset):
    rng = np.random.default_rng(n + offset)
    a = rng.uniform(-1, 1, n + 2)
    x = _dev(a)[offset:offset + n]
    ref = float(np.sum(a[offset:offset + n]))
    assert abs(ctx.sum(x) - ref) <= 1e-13 * max(1.0, np.abs(a).sum())


def test_sqrt_function(ctx):
    v = _dev([4.0, 9.0, 16.0, 25.0])
    ctx.sqrt(v)
    assert np.array_equal(v.cpu().numpy(), [2.0, 3.0, 4.0, 5.0])
    v = _dev([1.0, 4.0, 9.0])
    ctx.sqrt(v, 4.0)  # sqrt(4 x)
    assert np.array_equal(v.cpu().numpy(), [2.0, 4.0, 6.0])
    rng = np.random.default_rng(0)
    for n, off in ((1, 0), (7, 1), (1000, 0), (1001, 1)):
        a = rng.uniform(0.0, 3.0, n + 2)
        b = _dev(a)
        ctx.sqrt(b[off:off + n], 0.5)
        ref = a.copy()
        ref[off:off + n] = np.sqrt(0.5 * a[off:off + n])
        assert np.abs(b.cpu().numpy() - ref).max() <= 2e-16 * 3.0
