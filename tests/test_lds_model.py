"""The LDS bank-conflict model behind the element stride of the streaming kernels (scripts/lds_conflict_model.py):
with the banking rules of MI355X_MICROARCH.md the contraction passes of the p = 3 kernel are conflict-free exactly when the stride
between the elements of a wave is 16 (mod 32) doubles -- what pa_nd_hex_stream.hip: stream_lds_elem() produces -- and were two- to
four-way conflicted at the 300 doubles of rounds 1-3.  Pins the numbers the design text quotes."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_element_stride_of_the_streaming_kernel_is_conflict_free_for_the_contraction_passes():
    from lds_conflict_model import batch_extra_cycles

    n_r, old_r, n_w, old_w = batch_extra_cycles(300, True)   # rounds 1-3: contraction buffers + side buffers, parity flip
    assert (n_r, n_w) == (102, 102) and old_r == 236 and old_w == 64
    _, new_r, _, new_w = batch_extra_cycles(304, False)       # round 4
    assert new_r == 32 and new_w == 64                        # what is left: the tensor-order staging rows (both layouts)
    # any stride of 16 (mod 32) doubles does, with or without room to spare; 0 (mod 32) needs the flip and keeps 54
    for stride in (272, 336, 432):
        assert batch_extra_cycles(stride, False, inplace=stride != 432)[1] == 32
    assert batch_extra_cycles(320, True)[1] == 54


def test_stream_lds_elem_rounds_to_sixteen_mod_thirty_two():
    """The formula in the kernel source (a constexpr the host and the device share) against the strides the design text names."""
    src = open(os.path.join(ROOT, "palace_amd", "csrc", "pa_nd_hex_stream.hip")).read()
    m = re.search(r"constexpr int stream_lds_elem\(const int raw\) \{ return \(raw \+ 15\) / 32 \* 32 \+ 16; \}", src)
    assert m, "stream_lds_elem changed: update scripts/lds_conflict_model.py and DESIGN.md 3.1"
    f = lambda raw: (raw + 15) // 32 * 32 + 16  # noqa: E731
    assert [f(r) for r in (300, 295, 210, 428)] == [304, 304, 240, 432]
    assert all(f(r) % 32 == 16 and f(r) >= r for r in range(1, 2000))
