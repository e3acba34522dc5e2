"""Offline weight preparation (host only): quantiser, packer, container."""
import numpy as np
import pytest

from composable_kernel_amd import weights


@pytest.mark.parametrize("bits,sdt", [(4, "f16"), (4, "bf16"), (8, "f16"), (8, "bf16")])
def test_quantize_pack_roundtrip(tmp_path, port, bits, sdt):
    rng = np.random.default_rng(bits)
    N, K = 24, 64
    w = rng.normal(0, 0.05, (N, K)).astype(np.float32)
    w[3] = 0                                   # an all-zero channel keeps a finite scale
    q, s = weights.quantize_per_channel(w, bits, sdt)
    qmax = 2 ** (bits - 1) - 1
    assert q.min() >= -qmax - 1 and q.max() <= qmax and np.abs(q).max() == qmax
    pw = weights.pack(q, s, bits, sdt)
    if bits == 4:
        # the packer agrees with the checker's format and unpack_int4 inverts it
        np.testing.assert_array_equal(pw.weight, port.pack_int4(q, K, N))
        np.testing.assert_array_equal(weights.unpack_int4(pw.weight, K), q)
    # quantisation error: half a step, plus the rounding of q * scale to the 16-bit type (bf16 truncates: 2^-7
    # relative, and its truncated scale can push the largest element one step past the grid)
    deq = weights.dequantize(pw)
    step = weights._from_bits(s, sdt)[:, None]
    steps, rel = (0.51, 2.0 ** -10) if sdt == "f16" else (1.51, 2.0 ** -6)
    assert (np.abs(deq - w) <= steps * step + np.abs(w) * rel).all()
    path = str(tmp_path / "w.npz")
    weights.save_packed(path, pw)
    back = weights.load_packed(path)
    np.testing.assert_array_equal(back.weight, pw.weight)
    np.testing.assert_array_equal(back.scale, pw.scale)
    assert (back.bits, back.scale_dtype, back.N, back.K) == (bits, sdt, N, K)


def test_container_rejects_foreign_files(tmp_path):
    path = str(tmp_path / "x.npz")
    np.savez(path, format=np.array("something-else"), weight=np.zeros(1))
    with pytest.raises(ValueError):
        weights.load_packed(path)
