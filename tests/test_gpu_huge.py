"""Maximum-size test: > 2^31 elements (8.6 GB per tensor) with a ragged tail -- 64-bit index arithmetic in the
streaming kernels, block counts beyond 2^20, partial counts beyond 10^6.  The element-wise reference is torch on
the same device in chunks (IEEE division, floor, multiply: the same three fp32 ops as the oracle, which is checked
on slices at the start, across the 2^31 boundary and at the end); reductions are accumulated in float64."""
import numpy as np
import pytest
import torch

from oracle import lq_oracle as O

pytestmark = pytest.mark.gpu


def test_more_than_2_to_31_elements():
    import learned_quantization_amd as lq
    dev = torch.device("cuda:0")
    if torch.cuda.get_device_properties(0).total_memory < 60 * 2 ** 30:
        pytest.skip("needs ~40 GB of device memory")
    n = 2 ** 31 + 4096 + 3
    g = torch.Generator(device=dev).manual_seed(1)
    P = torch.empty(n, device=dev)
    dy = torch.empty(n, device=dev)
    step = 1 << 28
    for a in range(0, n, step):
        b = min(n, a + step)
        P[a:b] = torch.rand(b - a, device=dev, generator=g) * 200.0 - 100.0
        dy[a:b] = torch.randn(b - a, device=dev, generator=g) * 1e-3
    s = torch.tensor([0.37], device=dev)
    lam = 2e-5
    out, q = lq.fq_forward(P, s, q_dtype=torch.int32)
    # oracle on slices (start, across 2^31, ragged end)
    for a, b in ((0, 5000), (2 ** 31 - 3000, 2 ** 31 + 3000), (n - 5000, n)):
        q_o, out_o = O.fq_forward(P[a:b].cpu().numpy(), s.cpu().numpy())
        np.testing.assert_array_equal(out[a:b].cpu().numpy(), out_o)
        np.testing.assert_array_equal(q[a:b].cpu().numpy(), q_o.astype(np.int32))
    # whole tensor against the same three fp32 ops in torch, chunked
    maxq, below, total = 0.0, 0, 0.0
    for a in range(0, n, step):
        b = min(n, a + step)
        qr = torch.floor(P[a:b] / s)
        outr = qr * s
        assert torch.equal(out[a:b], outr), f"chunk {a}"
        assert torch.equal(q[a:b], qr.to(torch.int32))
        maxq = max(maxq, float(qr.abs().max()))
        nz = torch.where(outr == 0, torch.full_like(outr, O.EPS_F32), outr)
        ratio = dy[a:b].abs() / nz.abs()
        m = ~(ratio >= lam)
        below += int(m.sum())
        total += float((-torch.tanh(lam - ratio[m]).abs()).double().sum())
        del qr, outr, nz, ratio, m
    ds, parts = lq.fq_scale_grad(P, s, dy, lam, return_parts=True)
    parts = parts.cpu().numpy()
    assert parts[0, 0] == maxq
    assert int(parts[2, 0]) == below or abs(parts[2, 0] - below) <= 128      # float32 readout of an exact count
    mean = total / n
    np.testing.assert_allclose(parts[1, 0], mean, rtol=1e-5)
    np.testing.assert_allclose(float(ds), mean * maxq, rtol=1e-5)
    out2, ds2 = lq.fq_fwd_bwd_fused(P, s, dy, lam, out=out)
    # K4 and K2 may use different unit sizes on streaming-size tensors: max|q| and the count are exact, the vote sum
    # differs by fp32 summation order only
    np.testing.assert_allclose(ds2.cpu().numpy(), ds.cpu().numpy(), rtol=1e-6)
    for a, b in ((2 ** 31 - 3000, 2 ** 31 + 3000), (n - 5000, n)):
        _, out_o = O.fq_forward(P[a:b].cpu().numpy(), s.cpu().numpy())
        np.testing.assert_array_equal(out2[a:b].cpu().numpy(), out_o)


def test_rows_longer_than_2_to_30_with_per_row_scales():
    import learned_quantization_amd as lq
    dev = torch.device("cuda:0")
    if torch.cuda.get_device_properties(0).total_memory < 60 * 2 ** 30:
        pytest.skip("needs ~30 GB of device memory")
    L = 2 ** 30 + 2048 + 4
    P = torch.empty(2, L, device=dev)
    g = torch.Generator(device=dev).manual_seed(2)
    for r in range(2):
        for a in range(0, L, 1 << 28):
            b = min(L, a + (1 << 28))
            P[r, a:b] = torch.randn(b - a, device=dev, generator=g) * 0.05
    s = torch.tensor([[0.011], [0.0037]], device=dev)
    out = lq.fq_forward(P, s)
    for r in range(2):
        for a in range(0, L, 1 << 28):
            b = min(L, a + (1 << 28))
            assert torch.equal(out[r, a:b], torch.floor(P[r, a:b] / s[r]) * s[r])
    _, out_o = O.fq_forward(P[:, L - 3000:].cpu().numpy(), s.cpu().numpy())
    np.testing.assert_array_equal(out[:, L - 3000:].cpu().numpy(), out_o)


@pytest.mark.parametrize("L", [1028, 1029, 1001])
def test_flat_forward_beyond_2_to_32_elements(L):
    """More than 2^32 elements in rows that are not whole 128-byte lines: the line-aligned flat forward with 64-bit group
    arithmetic (k_flat_fwd group mode 2 of csrc/lq_stream2.hpp for L = 1028; L = 1029 keeps the row stream with its scalar head /
    tail elements; L = 1001 < 1024: group mode 7, float4s straddle row ends) -- and, for L = 1001, the scale gradient through the
    row-window kernel (k_row_win) whose element offsets pass 2^32."""
    import learned_quantization_amd as lq
    dev = torch.device("cuda:0")
    if torch.cuda.get_device_properties(0).total_memory < 100 * 2 ** 30:
        pytest.skip("needs ~40 GB of device memory")
    R = 4300000
    assert R * L > 2 ** 32
    g = torch.Generator(device=dev).manual_seed(3)
    P = torch.empty(R, L, device=dev)
    rows = 1 << 18
    for a in range(0, R, rows):
        b = min(R, a + rows)
        P[a:b] = torch.randn(b - a, L, device=dev, generator=g) * 0.05
    s = (torch.rand(R, 1, device=dev, generator=g) * 9e-3 + 1e-3)
    out = lq.fq_forward(P, s)
    for a in range(0, R, rows):
        b = min(R, a + rows)
        assert torch.equal(out[a:b], torch.floor(P[a:b] / s[a:b]) * s[a:b]), f"rows {a}..{b}"
    for a in (0, (2 ** 32) // L - 2, R - 3):                                  # oracle across the 2^32 boundary and at both ends
        _, out_o = O.fq_forward(P[a:a + 3].cpu().numpy(), s[a:a + 3].cpu().numpy())
        np.testing.assert_array_equal(out[a:a + 3].cpu().numpy(), out_o)
    if L != 1001:
        return
    lam = 2e-5
    dy = torch.empty(R, L, device=dev)
    for a in range(0, R, rows):
        b = min(R, a + rows)
        dy[a:b] = torch.randn(b - a, L, device=dev, generator=g) * 1e-3
    ds, parts = lq.fq_scale_grad(P, s, dy, lam, return_parts=True)
    out2, ds2 = lq.fq_fwd_bwd_fused(P, s, dy, lam, out=out)
    for a in range(0, R, rows):
        b = min(R, a + rows)
        qr = torch.floor(P[a:b] / s[a:b])
        outr = qr * s[a:b]
        assert torch.equal(out2[a:b], outr), f"fused rows {a}..{b}"
        maxq = qr.abs().amax(1)
        assert torch.equal(parts[0, a:b].reshape(-1), maxq), f"max|q| rows {a}..{b}"
        nz = torch.where(outr == 0, torch.full_like(outr, O.EPS_F32), outr)
        ratio = dy[a:b].abs() / nz.abs()
        m = ~(ratio >= lam)
        vote = torch.where(m, -torch.tanh(lam - ratio).abs(), torch.zeros_like(ratio)).double().sum(1)
        cnt = m.sum(1)
        mean = torch.where(cnt == 0, torch.full_like(vote, -float(np.abs(np.tanh(np.float32(lam))))), vote / L)
        ref = (mean * maxq.double()).float()
        for got, name in ((ds, "split"), (ds2, "fused")):
            torch.testing.assert_close(got[a:b].reshape(-1), ref, rtol=2e-5, atol=1e-30, msg=lambda m_: f"{name} ds rows {a}..{b}: {m_}")
        del qr, outr, nz, ratio, m, vote


@pytest.mark.parametrize("C", [4, 68, 65])
def test_column_forward_and_statistics_beyond_2_to_32_elements(C):
    """More than 2^32 elements in column mode (`inner == 1`, per-column scales: NHWC per-channel activations): the flat forward
    with 64-bit group arithmetic -- C = 4 / 68: the four scales of a float4 are one aligned float4 of the scale vector (k_flat_fwd
    group mode 5); C = 65: the same, dword-aligned, wrapping around the row end (mode 9) -- and the 64-bit index forms of the
    tracking statistics (custom_callbacks.py:85-99: unique integers and max|q| over an axis).  Reference: the same three fp32
    operations in torch on the device, chunked; the oracle on rows across the 2^32 boundary."""
    import learned_quantization_amd as lq
    dev = torch.device("cuda:0")
    if torch.cuda.get_device_properties(0).total_memory < 100 * 2 ** 30:
        pytest.skip("needs ~40 GB of device memory")
    R = (2 ** 32) // C + 1001
    assert R * C > 2 ** 32
    g = torch.Generator(device=dev).manual_seed(4 + C)
    P = torch.empty(R, C, device=dev)
    rows = (1 << 28) // C
    for a in range(0, R, rows):
        b = min(R, a + rows)
        P[a:b] = torch.randn(b - a, C, device=dev, generator=g) * 0.05
    s = torch.rand(1, C, device=dev, generator=g) * 9e-3 + 1e-3
    out = lq.fq_forward(P, s)
    qmax = torch.zeros(C, device=dev)
    qlo, qhi = 0.0, 0.0
    for a in range(0, R, rows):
        b = min(R, a + rows)
        qr = torch.floor(P[a:b] / s)
        assert torch.equal(out[a:b], qr * s), f"rows {a}..{b}"
        qmax = torch.maximum(qmax, qr.abs().amax(0))
        qlo, qhi = min(qlo, float(qr.min())), max(qhi, float(qr.max()))
        del qr
    a = (2 ** 32) // C - 2
    _, out_o = O.fq_forward(P[a:a + 4].cpu().numpy(), s.cpu().numpy())
    np.testing.assert_array_equal(out[a:a + 4].cpu().numpy(), out_o)
    del out
    if C != 4:
        return
    # statistics with 64-bit element indices: max|q| over axis 0 (per column), and the unique integers with their counts
    got = lq.q_absmax_over_axis(P, s, 0)
    assert torch.equal(got.reshape(-1), qmax)
    values, counts = lq.q_unique(P, s)
    assert int(counts.sum()) == R * C
    assert float(values.min()) == qlo and float(values.max()) == qhi
    # the count of one integer against torch, chunked
    v0 = int(values[len(values) // 2])
    n0 = 0
    for a in range(0, R, rows):
        b = min(R, a + rows)
        n0 += int((torch.floor(P[a:b] / s) == v0).sum())
    assert int(counts[len(values) // 2]) == n0
