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


@pytest.mark.parametrize("bits,group_k", [(4, 64), (4, 128), (8, 64)])
def test_group_wise_quantize_pack_roundtrip(tmp_path, bits, group_k):
    rng = np.random.default_rng(group_k + bits)
    N, K = 24, 256
    # channels whose magnitude changes along k: the case group-wise scales exist for
    w = (rng.normal(0, 0.05, (N, K)) * np.linspace(0.2, 3.0, K)[None, :]).astype(np.float32)
    q, s = weights.quantize_per_group(w, group_k, bits, "f16")
    assert s.shape == (K // group_k, N) and q.shape == (N, K)
    pw = weights.pack(q, s, bits, "f16", group_k)
    deq = weights.dequantize(pw)
    step = np.repeat(weights._from_bits(s, "f16").T, group_k, axis=1)
    assert (np.abs(deq - w) <= 0.51 * step + np.abs(w) * 2.0 ** -10).all()
    # finer than one scale per channel where the magnitudes differ
    qc, sc = weights.quantize_per_channel(w, bits, "f16")
    err_c = np.abs(weights.dequantize(weights.pack(qc, sc, bits, "f16")) - w)[:, : K // 4].mean()
    assert np.abs(deq - w)[:, : K // 4].mean() < 0.7 * err_c
    path = str(tmp_path / "g.npz")
    weights.save_packed(path, pw)
    back = weights.load_packed(path)
    np.testing.assert_array_equal(back.weight, pw.weight)
    np.testing.assert_array_equal(back.scale, pw.scale)
    assert back.group_k == group_k
    with pytest.raises(ValueError):
        weights.pack(q, s[:-1], bits, "f16", group_k)
    with pytest.raises(ValueError):
        weights.quantize_per_group(w, 96, bits)


def test_version_1_container_still_loads(tmp_path):
    q = np.zeros((4, 16), np.int8)
    path = str(tmp_path / "v1.npz")
    np.savez(path, format=np.array("wqgemm-weights-1"), bits=np.array(8), scale_dtype=np.array("f16"), N=np.array(4),
             K=np.array(16), weight=q, scale=np.full(4, 0x3C00, np.uint16))
    pw = weights.load_packed(path)
    assert pw.group_k == 0 and pw.scale.shape == (4,)


def test_preshuffled_layout_roundtrip(tmp_path, port):
    """preshuffle -> container -> load -> unshuffle returns the packed bytes; the layout tag survives; the numpy
    inverse agrees with the checker's statement of the format; version-2 files (no layout entry) load as packed."""
    from composable_kernel_amd import weights
    rng = np.random.default_rng(5)
    N, K = 96, 256
    w = rng.normal(0, 0.1, (N, K)).astype(np.float32)
    q, s = weights.quantize_per_group(w, 64, 4, "f16")
    pk = weights.pack(q, s, 4, "f16", 64)
    pre = weights.preshuffle(pk)
    np.testing.assert_array_equal(pre.weight.reshape(-1), port.preshuffle_int4(pk.weight, K, N))
    path = str(tmp_path / "pre.npz")
    weights.save_packed(path, pre)
    back = weights.load_packed(path)
    assert back.layout == "preshuffled" and back.group_k == 64 and back.b_dtype == weights.capi.I4_PRESHUFFLED
    np.testing.assert_array_equal(weights.unshuffle(back).weight, pk.weight)
    np.testing.assert_array_equal(weights.dequantize(back), weights.dequantize(pk))
    # 8-bit weights: their own unit order
    q8, s8 = weights.quantize_per_channel(w, 8, "bf16")
    pk8 = weights.pack(q8, s8, 8, "bf16")
    pre8 = weights.preshuffle(pk8)
    np.testing.assert_array_equal(pre8.weight.reshape(-1), port.preshuffle_int8(pk8.weight, K, N))
    weights.save_packed(str(tmp_path / "pre8.npz"), pre8)
    back8 = weights.load_packed(str(tmp_path / "pre8.npz"))
    assert back8.layout == "preshuffled" and back8.b_dtype == weights.capi.I8_PRESHUFFLED
    np.testing.assert_array_equal(weights.unshuffle(back8).weight, pk8.weight)
    # sizes that are not whole blocks have no such layout
    for bad in (weights.pack(q[:40], s[:, :40], 4, "f16", 64),):
        try:
            weights.preshuffle(bad)
            raise AssertionError("accepted")
        except ValueError:
            pass
    # a version-2 file
    np.savez(str(tmp_path / "v2.npz"), format=np.array("wqgemm-weights-2"), bits=np.array(4), scale_dtype=np.array("f16"),
             N=np.array(N), K=np.array(K), group_k=np.array(64), weight=pk.weight, scale=pk.scale)
    assert weights.load_packed(str(tmp_path / "v2.npz")).layout == "packed"
