"""GPU parity tests proper: the HIP path (through the C ABI of include/lq_hip.h) against the CPU oracle.

Bars (north_star): integer q and the dequantized product out = q*s bit-exact; max|q| bit-exact;
tanh/mean-derived quantities (ds, penalties) within rtol 1e-5.
"""
import hashlib

import numpy as np

from _bounds import stable_seed
import pytest
import torch

from oracle import lq_oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import learned_quantization_amd._hip as _hip
    _hip.load()   # fail loudly if the extension is missing
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _check_case(P, s, dy, lam, dev, name=""):
    import learned_quantization_amd as lq
    q_o, out_o = O.fq_forward(P, s)
    _, ds_o, im = O.nq_backward(P, s, lam, dy, return_intermediates=True)
    Pt, st, dyt = _t(P, dev), _t(s, dev), _t(dy, dev)
    out, q = lq.fq_forward(Pt, st, q_dtype=torch.float32)
    np.testing.assert_array_equal(q.cpu().numpy(), q_o, err_msg=f"{name}: q")
    np.testing.assert_array_equal(out.cpu().numpy(), out_o, err_msg=f"{name}: out")
    ds, parts = lq.fq_scale_grad(Pt, st, dyt, lam, return_parts=True)
    parts = parts.cpu().numpy()
    np.testing.assert_array_equal(parts[0], np.asarray(im["maxvalue"], np.float32).reshape(-1), err_msg=f"{name}: max|q|")
    np.testing.assert_allclose(parts[1], np.asarray(im["mean"], np.float32).reshape(-1), rtol=RTOL, atol=1e-30,
                               equal_nan=True, err_msg=f"{name}: mean")
    assert ds.shape == st.shape
    np.testing.assert_allclose(ds.cpu().numpy(), ds_o, rtol=RTOL, atol=1e-30, equal_nan=True, err_msg=f"{name}: ds")
    # fused single-pass kernel gives the same out (bit-exact) and ds
    out2, ds2 = lq.fq_fwd_bwd_fused(Pt, st, dyt, lam)
    np.testing.assert_array_equal(out2.cpu().numpy(), out_o, err_msg=f"{name}: fused out")
    # the fused and the split traversal may cut a group into different partials (rows per block by operation): same terms,
    # another summation order -- both are held to the oracle, and to each other at float32 resolution
    np.testing.assert_allclose(ds2.cpu().numpy(), ds_o, rtol=RTOL, atol=1e-30, equal_nan=True, err_msg=f"{name}: fused ds")
    np.testing.assert_allclose(ds2.cpu().numpy(), ds.cpu().numpy(), rtol=2e-6, atol=1e-30, equal_nan=True, err_msg=f"{name}: fused ds vs unfused ds")


def test_kat_on_gpu(kat, dev):
    P, s, dy = (np.array(kat[k], np.float32) for k in ("P", "s", "dy"))
    import learned_quantization_amd as lq
    out, q = lq.fq_forward(_t(P, dev), _t(s, dev), q_dtype=torch.float32)
    np.testing.assert_array_equal(q.cpu().numpy(), np.array(kat["q"], np.float32))
    np.testing.assert_array_equal(out.cpu().numpy(), np.array(kat["out"], np.float32))
    for lam in ("0.5", "0.05"):
        ds = lq.fq_scale_grad(_t(P, dev), _t(s, dev), _t(dy, dev), float(lam))
        np.testing.assert_allclose(ds.cpu().numpy(), np.array(kat[f"lambda_{lam}"]["ds"], np.float32), rtol=1e-6)


def test_golden_cases_on_gpu(golden_cases, dev):
    for name, c in golden_cases.items():
        _check_case(c["P"], c["s"], c["dy"], float(c["lam"]), dev, name)
        import learned_quantization_amd as lq
        out, q = lq.fq_forward(_t(c["P"], dev), _t(c["s"], dev), q_dtype=torch.float32)
        np.testing.assert_array_equal(q.cpu().numpy(), c["q"], err_msg=name)       # committed fixture, not live oracle
        np.testing.assert_array_equal(out.cpu().numpy(), c["out"], err_msg=name)


# every traversal mode of the kernels: row-big (vector + scalar), row-small (all team widths), column
SHAPES = [
    ((784, 128), "rowwise"), ((784, 128), "columnwise"), ((784, 128), "channelwise"), ((784, 128), "scalar"),
    ((128, 10), "rowwise"), ((128, 10), "columnwise"), ((10,), "scalar"), ((128,), "scalar"),
    ((3, 3, 3, 32), "rowwise"), ((3, 3, 3, 32), "columnwise"), ((3, 3, 3, 32), "channelwise"), ((3, 3, 3, 32), "scalar"),
    ((3, 3, 64, 128), "rowwise"), ((3, 3, 64, 128), "columnwise"), ((3, 3, 64, 128), "channelwise"),
    ((7, 7, 3, 64), "channelwise"), ((1, 1, 64, 128), "channelwise"), ((1, 1, 64, 128), "rowwise"),
    ((5, 1031), "rowwise"),            # L >= 1024 and L % 4 != 0 -> scalar row-big path
    ((3, 4100), "rowwise"),            # L % 4 == 0 -> vector path with a partial chunk
    ((2, 3, 9000), "columnwise"),      # several chunks per row
    ((100003,), "scalar"),             # flat, ragged tail on the vector path
    ((33, 5, 3), "columnwise"),        # inner = 3 < 16, outer > 1 -> column mode
    ((1000, 7, 2), "columnwise"),      # column mode with row slices
    ((50, 9), "rowwise"),              # tiny rows (lanes-per-row = 4)
    ((6, 1), "rowwise"),               # L = 1
    ((4, 3, 16, 16), "columnwise"),    # NCHW per-channel, small
    ((1398103, 3), "columnwise"),      # column mode C = 3 at streaming size (>= 4 M): float4 grid-stride variant, numel % 4 == 1
    ((140000, 4, 8), "columnwise"),    # C = 32 (G = 4, inner = 8)
    ((66601, 7, 9), "columnwise"),     # C = 63, numel % 4 == 3
    ((840000, 5), "columnwise"),       # C = 5
]


@pytest.mark.parametrize("shape,orient", SHAPES)
def test_random_parity_all_modes(shape, orient, dev):
    rng = np.random.default_rng(stable_seed(shape, orient))
    for lam in (0.0, 1e-10, 3e-2):
        P = rng.normal(0, 0.05, size=shape).astype(np.float32)
        dy = rng.normal(0, 1e-3, size=shape).astype(np.float32)
        s = rng.uniform(1e-3, 3e-2, size=O.scale_shape(shape, orient)).astype(np.float32)
        _check_case(P, s, dy, lam, dev, f"{shape} {orient} lam={lam}")


# streaming-size (>= 4 M elements) forms of csrc/lq_stream2.hpp: flat K1, pipelined column tile, pipelined periodic
# columns, tiny-row passes -- every template branch (group arithmetic, team width, ragged edges)
STREAM2_SHAPES = [
    ((20000, 256), "columnwise"),      # column tile, one column block, contiguous tiles
    ((2100, 2100), "columnwise"),      # column tile, 9 column blocks, the last one partial (2100 = 8*256 + 52)
    ((4099, 1028), "columnwise"),      # rows not a multiple of the row block
    ((600, 1024, 8), "columnwise"),    # inner = 8: one group per float4 (flat K1 group mode 0)
    ((1100, 1024, 4), "columnwise"),   # inner = 4
    ((700, 1000, 6), "columnwise"),    # inner = 6: float4s straddle groups (C = 6000)
    ((140000, 32), "rowwise"),         # tiny rows, 8 lanes per row, outer == 1: direct emit
    ((3, 50000, 32), "columnwise"),    # tiny rows with outer > 1: group-major partials + finalize
    ((200000, 24), "rowwise"),         # 6 of 8 lanes active
    ((300000, 16), "rowwise"),         # 4 lanes per row
    ((600000, 8), "rowwise"),          # 2 lanes per row
    ((100000, 48), "rowwise"),         # 12 of 16 lanes
    ((70001, 64), "rowwise"),          # 16 lanes per row, ragged last wave
    ((9000, 512), "rowwise"),          # rows of 512: round-1 row-small traversal, flat forward
    ((5, 1000000), "columnwise"),      # G = 1 M groups of one element each (inner = 1, outer = 5): column tile with C = 1 M
    ((2100001, 2), "columnwise"),      # C = 2: one-shot flat column kernel (per-column wave reductions), numel % 4 == 2
    ((1100000, 2, 2), "columnwise"),   # C = 4 with inner = 2: two groups inside every float4
    ((1050001, 4), "columnwise"),      # C = 4, inner = 1: scale-float4 flat forward + flat column kernel
    ((1398101, 3), "columnwise"),      # C = 3 (NHWC RGB), numel % 4 == 3
    ((600001, 8), "columnwise"),       # C = 8: one-shot flat column kernel, shuffle tree over 5 lane bits
    ((300000, 16), "columnwise"),      # C = 16
    ((150001, 32), "columnwise"),      # C = 32
    ((70001, 64), "columnwise"),       # C = 64: shuffle tree over 2 lane bits
    ((33001, 16, 4), "columnwise"),    # C = 64 with inner = 4 (16 groups)
    ((1100, 4100), "rowwise"),         # row stream, 4-element tail folded into the last chunk (loads hoisted: TAIL instantiation)
    ((1100, 4099), "rowwise"),         # row stream, chunks off the 16-byte grid: scalar head / tail elements
    ((3, 1500001), "rowwise"),         # long ragged rows, many chunks
    ((4200, 1001), "rowwise"),         # row-small rows off the 16-byte grid: flat K1 whose float4s may straddle ONE row end (mode 6)
    ((3, 1400, 1001), "columnwise"),   # the same with outer > 1: group = (i / inner) % G wraps around
    ((250001, 17), "rowwise"),         # 17-element rows: every 4th/5th float4 straddles, numel % 4 == 1
    ((840001, 5), "rowwise"),          # 5-element rows: EVERY float4 straddles a row end
    ((1030, 4100), "columnwise"),      # inner = 1, G = 4100: scale-float4 forward with the invariant-divisor modulo
    # rows off the 16-byte grid, scale-gradient ops: aligned float4 windows (k_row_win), team sizes 2..64 lanes, 1..5 float4 per lane
    ((43, 2048, 49), "columnwise"),    # 7x7 activation planes, outer > 1: 16 lanes per row, group-major partials
    ((90001, 49), "rowwise"),          # the same rows with outer == 1 (direct emit), numel % 4 == 1: the tensor ends inside a float4
    ((17000, 253), "rowwise"),         # widest window 64 float4: 64 lanes x 1, two rows per wave
    ((16500, 257), "rowwise"),         # 65 float4: 64 lanes x 2
    ((8500, 515), "rowwise"),          # 64 lanes x 3
    ((6000, 777), "rowwise"),          # 64 lanes x 4
    ((4300, 1023), "rowwise"),         # 257 float4: 64 lanes x 5
    ((140001, 30), "rowwise"),         # L % 4 == 2: rows start on two of the four phases only
    # (rows of 5..64 off-grid elements run k_row_seg: one flat window per block, segmented reduction through LDS)
    ((600001, 7), "rowwise"),          # 292 rows per block: the row walk loops
    ((70001, 63), "rowwise"),          # 32 rows per block
    ((3, 30000, 61), "columnwise"),    # outer > 1: the next row's group wraps around at G
    # rows of 1153..1533 elements (two poorly filled 1024-chunks in the row stream): one wave per row, 5 or 6 float4 per lane
    ((3500, 1225), "rowwise"),         # 35 x 35 planes: off the 16-byte grid, flat straddling forward
    ((3300, 1300), "rowwise"),         # aligned, 325 float4: 6 per lane
    ((2900, 1500), "rowwise"),         # 375 float4
    ((3, 1200, 1229), "columnwise"),   # outer > 1: group-major partials replace the row-stream layout
    ((2800, 1537), "rowwise"),         # off the grid up to 1945 elements: 7 or 8 float4 per lane
    ((2300, 1901), "rowwise"),
    # short rows in many layers (outer >= 32, G * inner a whole number of 128-byte lines): column mode with inner >= 16
    ((48, 2048, 49), "columnwise"),    # 7 x 7 planes: 4.8 M elements, 48 layers per block
    ((40, 1024, 132), "columnwise"),   # rows of 132: 128 layers per block (one partial per column)
    ((64, 4096, 20), "columnwise"),    # aligned short rows, one group per float4 in the forward
    # column tiles whose rows are off the 16-byte grid (C % 4 != 0): dword-aligned float4 access, the last lane re-reads columns
    ((70000, 67), "columnwise"),       # C = 67: one column block, the last lane repeats one column
    ((33000, 130), "columnwise"),      # C % 4 == 2
    ((4200, 1001), "columnwise"),      # 4 column blocks, the last one 233 wide
    ((1100, 4099), "columnwise"),      # 17 column blocks, the last one 3 columns wide (its only lane repeats one column)
    ((2100, 682, 3), "columnwise"),    # inner = 3: C = 2046
    # 64 < C <= 256: the periodic form (flat stream, LDS combine with H = 256 / C helpers per column)
    ((42000, 100), "columnwise"),      # C = 100: 2 helpers per column
    ((17000, 250), "columnwise"),      # C = 250: one helper per column, numel % 4 == 0
    ((42001, 33, 3), "columnwise"),    # C = 99 with inner = 3, numel % 4 == 3
    ((21000, 200), "columnwise"),      # C = 200 (C % 32 != 0)
    ((21900, 192), "columnwise"),      # C = 192 keeps the column tile (3 full lines per row)
    # (at >= 64 MiB) 256 < C <= 512 with a mostly empty second column block or rows off the 128-byte grid: periodic form, 512-thread blocks
    ((56000, 320), "columnwise"),      # 17.9 M elements: nontemporal size
    ((38001, 450), "columnwise"),      # C % 4 == 2, numel % 4 == 2
]


@pytest.mark.parametrize("shape,orient", STREAM2_SHAPES)
def test_streaming_forms_parity(shape, orient, dev):
    rng = np.random.default_rng(stable_seed(shape, orient))
    for lam in (1e-10, 3e-2):
        P = rng.normal(0, 0.05, size=shape).astype(np.float32)
        dy = (rng.normal(0, 1, size=shape) * 10.0 ** rng.uniform(-9, -2, size=shape)).astype(np.float32)
        s = rng.uniform(1e-3, 3e-2, size=O.scale_shape(shape, orient)).astype(np.float32)
        _check_case(P, s, dy, lam, dev, f"{shape} {orient} lam={lam}")
    # integer view through the same traversals
    import learned_quantization_amd as lq
    q = lq.quantized_integers(_t(P, dev), _t(s, dev), torch.int32).cpu().numpy()
    np.testing.assert_array_equal(q, O.quantized_integers(P, s).astype(np.int32))


RESNET18_KERNELS = [(7, 7, 3, 64), (3, 3, 64, 64), (3, 3, 64, 128), (3, 3, 128, 128), (1, 1, 64, 128), (3, 3, 128, 256),
                    (3, 3, 256, 256), (1, 1, 128, 256), (3, 3, 256, 512), (3, 3, 512, 512), (1, 1, 256, 512)]


@pytest.mark.parametrize("orient", ["rowwise", "columnwise", "channelwise", "scalar"])
def test_resnet18_kernel_shapes_full_size(orient, dev):
    """BASELINE configs[2] at full size: every distinct conv kernel of the ResNet-18-like topology (HWIO,
    IMAGENETTE/nested_quantization_layer/experiment.py:627-760), each orientation, trained-like and initial scales."""
    rng = np.random.default_rng(1234)
    for shape in RESNET18_KERNELS:
        P = rng.normal(0, 0.05, size=shape).astype(np.float32)
        dy = rng.normal(0, 1e-3, size=shape).astype(np.float32)
        sshape = O.scale_shape(shape, orient)
        for s, lam in ((rng.uniform(1e-3, 3e-2, size=sshape).astype(np.float32), 3e-2),
                       (np.full(sshape, O.SCALE_MIN, np.float32), 1e-11)):
            _check_case(P, s, dy, lam, dev, f"{shape} {orient} lam={lam}")


# BASELINE.json configs[4] (ResNet-50, "mixed 4/8-bit"): every distinct quantised tensor of the bottleneck topology
# (learned_quantization_amd/models.py::ResNet50Like -- an extension, the reference has no ResNet-50; the op is shape-generic)
RESNET50_TENSORS = [(7, 7, 3, 64), (1, 1, 64, 64), (1, 1, 64, 256), (1, 1, 256, 64), (3, 3, 64, 64), (1, 1, 256, 128),
                    (3, 3, 128, 128), (1, 1, 128, 512), (1, 1, 256, 512), (1, 1, 512, 128), (1, 1, 512, 256), (3, 3, 256, 256),
                    (1, 1, 256, 1024), (1, 1, 512, 1024), (1, 1, 1024, 256), (1, 1, 1024, 512), (3, 3, 512, 512),
                    (1, 1, 512, 2048), (1, 1, 1024, 2048), (1, 1, 2048, 512), (2048, 10), (2048, 1000), (2048,), (1000,)]


def test_resnet50_tensor_list_matches_the_model():
    import learned_quantization_amd as lq
    from learned_quantization_amd.models import ResNet50Like
    lq.reset_layer_names()
    m = ResNet50Like()
    shapes = {tuple(p.shape) for l in lq.custom_layers_of(m) for p in l._regularized()}
    assert {sh for sh in shapes if len(sh) > 1} <= set(RESNET50_TENSORS)
    assert len(lq.custom_layers_of(m)) == 54 and sum(len(l._regularized()) for l in lq.custom_layers_of(m)) == 108


@pytest.mark.parametrize("orient", ["rowwise", "columnwise", "channelwise", "scalar"])
def test_resnet50_kernel_shapes_full_size(orient, dev):
    """BASELINE configs[4] at full size: every bottleneck shape (1x1x64x256 ... 1x1x1024x2048, the 3x3 mids, the 7x7 stem, the
    2048x10 / 2048x1000 classifiers and the biases), each orientation, "mixed" thresholds per layer: the coarse one for 3x3
    kernels, the fine one elsewhere, once with trained-like scales and once at the reference's initial scale."""
    rng = np.random.default_rng(5050)
    for shape in RESNET50_TENSORS:
        o = orient if len(shape) > 1 else "scalar"                 # biases always have a scalar scale (NQ-L:225-227)
        if o == "channelwise" and len(shape) < 3:
            o = "rowwise"
        P = rng.normal(0, 0.05, size=shape).astype(np.float32)
        dy = (rng.normal(0, 1, size=shape) * 10.0 ** rng.uniform(-9, -2, size=shape)).astype(np.float32)
        sshape = O.scale_shape(shape, o)
        lam_trained, lam_init = (3e-2, 1e-10) if (len(shape) == 4 and shape[0] == 3) else (1e-6, 1e-11)
        for s, lam in ((rng.uniform(1e-3, 3e-2, size=sshape).astype(np.float32), lam_trained),
                       (np.full(sshape, O.SCALE_MIN, np.float32), lam_init)):
            _check_case(P, s, dy, lam, dev, f"{shape} {o} lam={lam}")


def test_init_scale_large_integers(dev):
    """s = 100*eps: |q| ~ 2e4, the regime where reciprocal-multiply differs (SURVEY section 7 hard part 1)."""
    rng = np.random.default_rng(1)
    P = rng.normal(0, 0.05, size=(1 << 22,)).astype(np.float32)
    s = np.array([O.SCALE_INIT], np.float32)
    import learned_quantization_amd as lq
    q = lq.quantized_integers(_t(P, dev), _t(s, dev), torch.int32).cpu().numpy()
    q_o = O.quantized_integers(P, s).astype(np.int32)
    np.testing.assert_array_equal(q, q_o)
    assert np.count_nonzero(q_o != np.floor(P * (np.float32(1) / s[0]))) > 0   # the trap is real on this data
    for sv in (1e-4, 3.3e-3, 0.77):
        s = np.array([sv], np.float32)
        np.testing.assert_array_equal(lq.quantized_integers(_t(P, dev), _t(s, dev), torch.int32).cpu().numpy(),
                                      O.quantized_integers(P, s).astype(np.int32))


def test_mnist_real_weights_on_gpu(mnist_weights, mnist_expected, dev):
    import learned_quantization_amd as lq
    meta, exp = mnist_expected
    for name in ("W1", "b1", "W2", "b2"):
        P = mnist_weights[name]
        q = lq.quantized_integers(_t(P, dev), _t(np.array([O.SCALE_INIT], np.float32), dev), torch.int32).cpu().numpy()
        np.testing.assert_array_equal(q, exp[f"{name}_q_init"].astype(np.int32))
        for key, m in meta.items():
            if not key.startswith(name + "@"):
                continue
            tag = key.split("@")[1]
            if tag.endswith("absmax12"):
                orient = tag.split("_")[0]
                sshape = O.scale_shape(P.shape, orient)
                s = (np.abs(P).max(axis=1 if orient == "rowwise" else 0).reshape(sshape) / 12.0 + 1e-6).astype(np.float32)
            else:
                s = np.array([np.float32(float(tag))], np.float32)
            q = lq.quantized_integers(_t(P, dev), _t(s, dev), torch.int32).cpu().numpy()
            assert hashlib.sha256(np.ascontiguousarray(q).tobytes()).hexdigest() == m["sha256"], key


def test_q_dtypes_and_int8_wrap(dev):
    import learned_quantization_amd as lq
    P = np.array([0.0, 1.0, 127.0, 128.0, 255.0, 256.0, -1.0, -128.0, -129.0, 300.7, -0.3], np.float32)
    s = np.array([1.0], np.float32)
    q8 = lq.quantized_integers(_t(P, dev), _t(s, dev), torch.int8).cpu().numpy()
    np.testing.assert_array_equal(q8, O.export_int8(P, s))
    q32 = lq.quantized_integers(_t(P, dev), _t(s, dev), torch.int32).cpu().numpy()
    np.testing.assert_array_equal(q32, np.floor(P).astype(np.int32))


def test_edge_empty_and_bad_arguments(dev):
    import learned_quantization_amd as lq
    with pytest.raises(ValueError):
        lq.fq_forward(torch.empty(0, device=dev), torch.ones(1, device=dev))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lq.fq_forward(torch.ones(4), torch.ones(1))
    with pytest.raises(TypeError):
        lq.fq_forward(torch.ones(4, device=dev, dtype=torch.float64), torch.ones(1, device=dev))
    with pytest.raises(ValueError):
        lq.fq_forward(torch.ones(4, 4, device=dev), torch.ones(2, 2, device=dev))
    # non-contiguous inputs are accepted (made contiguous)
    P = torch.randn(16, 8, device=dev).t()
    s = torch.full((8, 1), 0.1, device=dev)
    out = lq.fq_forward(P, s)
    q_o, out_o = O.fq_forward(P.cpu().numpy(), s.cpu().numpy())
    np.testing.assert_array_equal(out.cpu().numpy(), out_o)
    # misaligned base pointer (offset by one float) still correct: falls to the scalar kernels
    base = torch.randn(4097 * 3 + 1, device=dev)
    Pm = base[1:].view(3, 4097)
    sm = torch.tensor([[0.01], [0.02], [0.03]], device=dev)
    _, out_o = O.fq_forward(Pm.cpu().numpy(), sm.cpu().numpy())
    np.testing.assert_array_equal(lq.fq_forward(Pm, sm).cpu().numpy(), out_o)


def test_nan_and_inf_inputs(dev):
    import learned_quantization_amd as lq
    P = np.array([[0.1, np.nan, 0.3, -0.2], [np.inf, 0.2, -0.1, 0.05]], np.float32)
    dy = np.array([[1e-3, 1e-3, np.nan, 1e-4], [1e-3, 2e-3, 1e-3, 1e-3]], np.float32)
    s = np.array([[0.01], [0.02]], np.float32)
    q_o, out_o = O.fq_forward(P, s)
    out, q = lq.fq_forward(_t(P, dev), _t(s, dev), q_dtype=torch.float32)
    np.testing.assert_array_equal(q.cpu().numpy(), q_o)
    np.testing.assert_array_equal(out.cpu().numpy(), out_o)
    _, ds_o = O.nq_backward(P, s, 1e-2, dy)
    ds = lq.fq_scale_grad(_t(P, dev), _t(s, dev), _t(dy, dev), 1e-2).cpu().numpy()
    assert np.array_equal(np.isnan(ds), np.isnan(ds_o))


def test_determinism_bitwise(dev):
    import learned_quantization_amd as lq
    g = torch.Generator(device="cpu").manual_seed(0)
    P = torch.randn(64, 3, 56, 56, generator=g).mul_(50).to(dev)
    dy = torch.randn(64, 3, 56, 56, generator=g).mul_(1e-3).to(dev)
    s = torch.tensor([0.5, 1.0, 2.0], device=dev).view(1, 3, 1, 1)
    ref = lq.fq_scale_grad(P, s, dy, 1e-3)
    for _ in range(5):
        assert torch.equal(lq.fq_scale_grad(P, s, dy, 1e-3), ref)


def test_bench_shape_full_size_properties(dev):
    """BASELINE workload 256x3x224x224 (154 MB): too big for the NumPy oracle to be quick on every
    element-wise check, so: (1) exact identities on the device, (2) oracle on a strided sample of
    whole (n, c) planes, (3) ds against the oracle's per-plane pieces merged on the host."""
    import learned_quantization_amd as lq
    torch.manual_seed(42)
    x = torch.rand(256, 3, 224, 224, device=dev) * 255.0
    dy = torch.randn(256, 3, 224, 224, device=dev) * 1e-3
    s = torch.tensor([0.5, 1.0, 2.0], device=dev).view(1, 3, 1, 1)
    out, q = lq.fq_forward(x, s, q_dtype=torch.float32)
    assert torch.equal(q, torch.floor(q))                              # integers
    assert torch.equal(out, q * s)                                     # out == q*s exactly
    assert bool(((x - out) >= 0).all()) and bool(((x - out) < s).all())   # floor: 0 <= x - out < s
    assert torch.equal(lq.fq_forward(out, s), out)                     # idempotent on its own output (s = 2^k)
    # oracle on sampled planes
    xs, qs, outs = x[::37].cpu().numpy(), q[::37].cpu().numpy(), out[::37].cpu().numpy()
    q_o, out_o = O.fq_forward(xs, s.cpu().numpy())
    np.testing.assert_array_equal(qs, q_o)
    np.testing.assert_array_equal(outs, out_o)
    for lam in (0.0, 1e-11, 1e-3):
        ds, parts = lq.fq_scale_grad(x, s, dy, lam, return_parts=True)
        out2, ds2 = lq.fq_fwd_bwd_fused(x, s, dy, lam)
        assert torch.equal(out2, out)
        np.testing.assert_allclose(ds2.cpu().numpy(), ds.cpu().numpy(), rtol=1e-6)     # unit sizes may differ: summation order only
        # full oracle in float32 NumPy, channel by channel (a few seconds)
        xn, dyn = x.cpu().numpy(), dy.cpu().numpy()
        _, ds_o, im = O.nq_backward(xn, s.cpu().numpy(), lam, dyn, return_intermediates=True)
        np.testing.assert_array_equal(parts[0].cpu().numpy(), im["maxvalue"].reshape(-1))
        np.testing.assert_allclose(ds.cpu().numpy(), ds_o, rtol=RTOL, atol=1e-30)
    # per-tensor scale variant
    s1 = torch.tensor([1.0], device=dev)
    ds = lq.fq_scale_grad(x, s1, dy, 1e-3)
    _, ds_o = O.nq_backward(x.cpu().numpy(), s1.cpu().numpy(), 1e-3, dy.cpu().numpy())
    np.testing.assert_allclose(ds.cpu().numpy(), ds_o, rtol=RTOL)
    # linearity-type property: scaling dy and lambda by the same factor leaves the below-count unchanged
    _, p1 = lq.fq_scale_grad(x, s, dy, 1e-3, return_parts=True)
    _, p2 = lq.fq_scale_grad(x, s, dy * 4.0, 4e-3, return_parts=True)
    assert torch.equal(p1[2], p2[2]) and torch.equal(p1[0], p2[0])


def test_c_abi_without_python(dev, tmp_path):
    """Builds tests/tools/abi_smoke.cpp with hipcc and runs it: the boundary is usable from plain C/C++."""
    import os
    import shutil
    import subprocess
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(root, "learned_quantization_amd", "csrc")
    subprocess.check_call([hipcc, "-O2", "--offload-arch=gfx950", "-I", os.path.join(root, "include"), "-o", exe,
                           os.path.join(root, "tests", "tools", "abi_smoke.cpp"), "-L", libdir, "-llq_hip",
                           f"-Wl,-rpath,{libdir}"])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "0 mismatches" in res.stdout


def test_window_division_selftest(dev):
    """2^33 random in-window (a, b) pairs: the packed rcp+fma ratio division == IEEE '/' bit for bit on this device."""
    from learned_quantization_amd import _hip
    lib = _hip.load()
    bad = torch.zeros(1, dtype=torch.int64, device=dev)
    for seed in (1, 0xDEADBEEF, 20240229, 7):
        _hip.check(lib.lq_selftest_ratio_division(seed, 8192, 1024, bad.data_ptr(), None), "selftest")
    torch.cuda.synchronize()
    assert int(bad.item()) == 0
    bad.zero_()
    for seed in (3, 0xC0FFEE, 99):
        _hip.check(lib.lq_selftest_uniform_division(seed, 8192, 1024, bad.data_ptr(), None), "selftest")
    torch.cuda.synchronize()
    assert int(bad.item()) == 0


@pytest.mark.parametrize("orient", ["rowwise", "columnwise", "scalar"])
def test_mnist_real_weights_backward(mnist_weights, dev, orient):
    """Scale gradient on the shipped MNIST baseline weights (realistic distribution, |w| <= 0.23) at the thresholds the
    thesis sweeps, with gradient magnitudes of a trained net (1e-6 .. 1e-2)."""
    import learned_quantization_amd as lq
    rng = np.random.default_rng(12)
    for name in ("W1", "W2"):
        P = mnist_weights[name]
        sshape = O.scale_shape(P.shape, orient)
        s = (np.abs(P).max() / rng.uniform(8, 40, size=sshape)).astype(np.float32)
        dy = (rng.normal(0, 1, size=P.shape) * 10.0 ** rng.uniform(-6, -2, size=P.shape)).astype(np.float32)
        for lam in (1e-11, 1e-10, 1e-6, 1e-3):
            _check_case(P, s, dy, lam, dev, f"{name} {orient} lam={lam}")


def test_profile_events_stamp_the_kernel_itself(dev):
    """lq_profile_events(start, stop): the row-stream traversal is launched through hipExtLaunchKernelGGL and the two events
    carry the kernel's own begin / end -- what bench.py's roofline leg relies on.  The interval must be the kernel (tens of
    microseconds on the 154 MB tensor), not the call (which also holds the finalize launch) and not zero; switching the hook
    off again must leave later launches unstamped."""
    from learned_quantization_amd import _hip
    lib = _hip.load()
    x = torch.rand(256, 3, 224, 224, device=dev) * 255.0
    dy = torch.randn(256, 3, 224, 224, device=dev) * 1e-3
    out = torch.empty_like(x)
    s = torch.tensor([0.5, 1.0, 2.0], device=dev)
    ds = torch.empty(3, device=dev)
    ws = torch.empty(lib.lq_workspace_bytes(256, 3, 50176), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)

    def live():
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        return e
    times = {}
    for name in ("fwd", "bwd"):
        vals = []
        for _ in range(6):
            a, b = live(), live()
            torch.cuda.synchronize()
            lib.lq_profile_events(a.cuda_event, b.cuda_event)
            if name == "fwd":
                rc = lib.lq_fq_forward(x.data_ptr(), s.data_ptr(), out.data_ptr(), None, 0, 256, 3, 50176, None)
            else:
                rc = lib.lq_fq_scale_grad(x.data_ptr(), s.data_ptr(), dy.data_ptr(), 1e-11, ds.data_ptr(), None, ws.data_ptr(), ws.numel(),
                                          256, 3, 50176, None)
            lib.lq_profile_events(None, None)
            assert rc == 0
            torch.cuda.synchronize()
            vals.append(a.elapsed_time(b) * 1e3)
        times[name] = sorted(vals)[len(vals) // 2]
    # 308 MB at 4-7 TB/s
    assert 40.0 < times["fwd"] < 80.0 and 40.0 < times["bwd"] < 80.0, times
    a, b = live(), live()
    torch.cuda.synchronize()
    assert lib.lq_fq_forward(x.data_ptr(), s.data_ptr(), out.data_ptr(), None, 0, 256, 3, 50176, None) == 0     # hook off
    torch.cuda.synchronize()
    assert abs(a.elapsed_time(b)) < 0.03          # the two records were back to back: nothing stamped them again


MISALIGNED_STREAMING = [
    ((48, 2048, 49), "columnwise"),    # column mode with inner = 49 (short rows in many layers): scalar column kernel, any inner
    ((4200, 1001), "rowwise"),         # rows off the grid: round-1 row-small kernel
    ((4200, 1001), "columnwise"),      # 1001 columns: scalar column tile on the 48-row geometry
    ((3500, 1225), "rowwise"),         # rows of 1225: row stream with two chunks (the one-wave-per-row form needs aligned bases)
    ((42000, 100), "columnwise"),      # C = 100: the periodic form's workspace bound, scalar tile
    ((1100, 4100), "rowwise"),         # ragged long rows, scalar row stream
]


@pytest.mark.parametrize("shape,orient", MISALIGNED_STREAMING)
def test_streaming_size_bases_off_the_16_byte_grid(shape, orient, dev):
    """Streaming-size tensors whose base pointers are 4 bytes off the 16-byte grid: every round-2 form requires aligned bases and
    must hand over to the round-1 kernels ON THE PLAN GEOMETRY OF ROUND 2 (rows per block, column mode for short rows in many
    layers, workspace bounds) -- same results, bit for bit in q / out / max|q|."""
    import learned_quantization_amd as lq
    rng = np.random.default_rng(stable_seed(shape, orient, "misaligned"))
    n = int(np.prod(shape))
    Pn = rng.normal(0, 0.05, size=shape).astype(np.float32)
    dn = (rng.normal(0, 1, size=shape) * 10.0 ** rng.uniform(-9, -2, size=shape)).astype(np.float32)
    sn = rng.uniform(1e-3, 3e-2, size=O.scale_shape(shape, orient)).astype(np.float32)
    base_p = torch.empty(n + 1, device=dev)
    base_d = torch.empty(n + 1, device=dev)
    base_o = torch.empty(n + 1, device=dev)
    base_p[1:].copy_(torch.from_numpy(Pn.reshape(-1)))
    base_d[1:].copy_(torch.from_numpy(dn.reshape(-1)))
    P, dy, out_buf = base_p[1:].view(shape), base_d[1:].view(shape), base_o[1:].view(shape)
    assert P.data_ptr() % 16 == 4 and P.is_contiguous()
    s = _t(sn, dev)
    lam = 3e-2
    q_o, out_o = O.fq_forward(Pn, sn)
    _, ds_o, im = O.nq_backward(Pn, sn, lam, dn, return_intermediates=True)
    out, q = lq.fq_forward(P, s, q_dtype=torch.int32)
    np.testing.assert_array_equal(out.cpu().numpy(), out_o)
    np.testing.assert_array_equal(q.cpu().numpy(), q_o.astype(np.int32))
    ds, parts = lq.fq_scale_grad(P, s, dy, lam, return_parts=True)
    np.testing.assert_array_equal(parts[0].cpu().numpy(), np.asarray(im["maxvalue"], np.float32).reshape(-1))
    np.testing.assert_allclose(ds.cpu().numpy(), ds_o, rtol=1e-5, atol=1e-30)
    out2, ds2 = lq.fq_fwd_bwd_fused(P, s, dy, lam, out=out_buf)
    np.testing.assert_array_equal(out2.cpu().numpy(), out_o)
    np.testing.assert_allclose(ds2.cpu().numpy(), ds_o, rtol=1e-5, atol=1e-30)
