"""CPU tests of the host logic and of the C-ABI library surface (no compute launches: no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import learned_quantization_amd as lq
from learned_quantization_amd import _hip
from oracle import lq_oracle as O

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _hip.load()
    header = open(os.path.join(ROOT, "include", "lq_hip.h")).read()
    declared = set(re.findall(r"\b(lq_[a-z0-9_]+)\s*\(", header))
    declared -= {"lq_status", "lq_qdtype", "lq_adam_mode"}
    assert len(declared) >= 16
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/lq_hip.h but not exported"
    assert declared == set(_hip.SIGNATURES), "python binding table and header disagree"
    assert lib.lq_version() == 3
    assert lib.lq_status_string(-3) == b"LQ_EWORKSPACE"


def test_argument_validation_without_gpu():
    """Every entry point validates before touching the device: bad calls return an error code, no launch."""
    lib = _hip.load()
    assert lib.lq_fq_forward(None, None, None, None, 0, 1, 1, 1, None) == -1
    assert b"NULL" in lib.lq_last_error()
    assert lib.lq_fq_forward(None, None, None, None, 0, 0, 1, 1, None) == -1
    assert b"positive" in lib.lq_last_error()
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    assert lib.lq_fq_forward(p, p, None, None, 0, 1, 1, 16, None) == -1             # neither out nor q
    assert lib.lq_fq_forward(p, p, p, p, 9, 1, 1, 16, None) == -1                    # bad q_dtype
    assert lib.lq_fq_forward(p + 2, p, p, None, 0, 1, 1, 16, None) == -4             # misaligned
    assert lib.lq_fq_scale_grad(p, p, p, 0.0, p, None, None, 0, 1, 1, 16, None) == -3   # no workspace
    assert lib.lq_fq_scale_grad(p, p, p, 0.0, p, None, p, 8, 1, 1, 16, None) == -3      # workspace too small
    assert b"too small" in lib.lq_last_error()
    assert lib.lq_scale_adam_step(p, p, p, p, 4, 1e-4, 0.9, 0.999, 1e-7, 0, 0.0, 0, None) == -1   # step is 1-based
    assert lib.lq_q_absmax_over_axis(p, p, p, 2, 2, 2, 1, 1, 16, None) == -1         # geometry mismatch
    assert lib.lq_workspace_bytes(0, 1, 1) == 0


@pytest.mark.parametrize("desc", [(256, 3, 50176), (1, 1, 38535168), (1, 784, 128), (784, 128, 1), (9, 64, 128),
                                  (1, 3, 3 * 512 * 512), (1, 1, 10), (1000, 7, 2), (1, 1000000, 10)])
def test_workspace_is_small_relative_to_tensor(desc):
    lib = _hip.load()
    n = desc[0] * desc[1] * desc[2]
    ws = lib.lq_workspace_bytes(*desc)
    assert ws >= 256 and ws % 4 == 0
    if n >= 1 << 16:
        assert ws <= n * 4, f"workspace {ws} B for {n} elements"


def test_descriptor_and_scale_shape_agree_with_oracle():
    for shape in [(784, 128), (128, 10), (3, 3, 64, 128), (7, 7, 3, 64), (10,), (256, 3, 224, 224)]:
        for orient in lq.ORIENTATIONS:
            if len(shape) == 1 and orient != "scalar":
                continue
            ss = lq.scale_shape(shape, orient)
            assert ss == O.scale_shape(shape, orient)
            assert lq.group_descriptor(shape, ss) == O.group_descriptor(shape, ss)
    with pytest.raises(ValueError, match="Invalid scaler application: diagonal"):
        lq.scale_shape((3, 3), "diagonal")
    with pytest.raises(ValueError):
        lq.group_descriptor((4, 4), (2, 2))
    with pytest.raises(ValueError):
        lq.group_descriptor((4, 4), (3, 1))
    with pytest.raises(ValueError):
        lq.group_descriptor((0, 4), (1,))


def test_descriptor_is_the_flat_index_rule():
    rng = np.random.default_rng(0)
    for shape, orient in [((5, 7), "rowwise"), ((5, 7), "columnwise"), ((2, 3, 4, 5), "channelwise"), ((2, 3, 4, 5), "columnwise")]:
        s = rng.uniform(1, 2, size=lq.scale_shape(shape, orient)).astype(np.float32)
        outer, G, inner = lq.group_descriptor(shape, s.shape)
        full = np.broadcast_to(s, shape).reshape(-1)
        i = np.arange(full.size)
        np.testing.assert_array_equal(full, s.reshape(-1)[(i // inner) % G])


def test_layers_construct_on_cpu_and_fail_loudly_on_call():
    lq.reset_layer_names()
    d1 = lq.CustomDenseLayer(seed=42, units=128, penalty_threshold=1e-10, orientation="rowwise",
                             initializer=lq.RandomNormal(seed=42), name="ignored", regularizer=None, input_shape=784)
    d2 = lq.CustomDenseLayer(seed=42, units=10, penalty_threshold=1e-10, orientation="columnwise",
                             initializer=lq.RandomNormal(seed=42), name="ignored", regularizer=None, input_shape=128)
    assert (d1.name, d2.name) == ("custom_dense_layer", "custom_dense_layer_1")     # Keras auto-names
    assert tuple(d1.W.shape) == (784, 128) and tuple(d1.b.shape) == (128,)
    assert tuple(d1.nested_q_w_layer.scale.shape) == (784, 1)
    assert tuple(d2.nested_q_w_layer.scale.shape) == (1, 10)
    assert tuple(d1.nested_q_b_layer.scale.shape) == (1,) and d1.nested_q_b_layer.orientation == "scalar"
    assert float(d1.nested_q_w_layer.scale[0, 0]) == np.float32(np.finfo(np.float32).eps * 100)
    assert abs(float(d1.W.std()) - 0.05) < 0.005                                     # RandomNormal stddev 0.05
    c = lq.CustomConv2DLayer(seed=1, penalty_threshold=1e-11, orientation="channelwise", initializer=lq.RandomNormal(seed=1),
                             filters=32, kernel_size=(3, 3), strides=(1, 1), padding="same", name="n",
                             regularizer=lq.l2(1e-4), input_shape=3)
    assert c.name == "custom_conv2d_layer" and c.padding == "SAME"
    assert tuple(c.kernel.shape) == (3, 3, 3, 32) and tuple(c.nested_q_k_layer.scale.shape) == (1, 1, 3, 1)
    nb = lq.CustomConv2DLayerNoBias(seed=1, penalty_threshold=None, orientation="scalar", initializer=lq.RandomNormal(seed=1),
                                    filters=4, kernel_size=(1, 1), strides=(1, 1), padding="valid", name="n",
                                    regularizer=None, input_shape=3)
    assert not hasattr(nb, "b") and not hasattr(nb, "nested_q_b_layer")
    assert len(lq.scale_parameters(torch.nn.ModuleList([d1, d2, c, nb]))) == 7
    with pytest.raises(ValueError, match="Invalid scaler application"):
        lq.CustomQuantizedScaleLayer(1e-3, None, "diagonal").build((3, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        d1(torch.zeros(2, 784))                                                      # product path never falls back
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lq.my_custom_gradient(torch.zeros(4), torch.ones(1), 1e-3)


def test_trained_weights_initialisation(mnist_weights):
    lq.reset_layer_names()
    d = lq.CustomDenseLayer(seed=0, units=128, penalty_threshold=1e-10, orientation="scalar", initializer=None, name="n",
                            regularizer=None, trained_weights=[mnist_weights["W1"], mnist_weights["b1"]], input_shape=784)
    np.testing.assert_array_equal(d.W.detach().numpy(), mnist_weights["W1"])
    np.testing.assert_array_equal(d.b.detach().numpy(), mnist_weights["b1"])
    with pytest.raises(ValueError):
        lq.CustomDenseLayer(seed=0, units=7, penalty_threshold=1e-10, orientation="scalar", initializer=None, name="n",
                            regularizer=None, trained_weights=[mnist_weights["W1"], mnist_weights["b1"]], input_shape=784)


def test_reference_import_paths_exist():
    from learned_quantization_amd.nested_quantization_layer import custom_layers as NQ
    from learned_quantization_amd.custom_loss_terms import custom_layers as CL
    from learned_quantization_amd.custom_loss_terms import custom_loss_functions as CF
    import inspect
    assert list(inspect.signature(NQ.my_custom_gradient).parameters) == ["parameter", "scale", "penalty_threshold"]
    assert list(inspect.signature(CL.my_custom_gradient).parameters) == ["parameter", "scale"]
    assert list(inspect.signature(NQ.CustomQuantizedScaleLayer.__init__).parameters)[1:4] == ["penalty_threshold", "initializer", "orientation"]
    assert list(inspect.signature(CL.CustomQuantizedScaleLayer.__init__).parameters)[1:4] == ["penalty_rate", "initializer", "orientation"]
    assert list(inspect.signature(NQ.CustomDenseLayer.__init__).parameters)[1:9] == [
        "seed", "units", "penalty_threshold", "orientation", "initializer", "name", "regularizer", "trained_weights"]
    assert list(inspect.signature(NQ.CustomConv2DLayer.__init__).parameters)[1:12] == [
        "seed", "penalty_threshold", "orientation", "initializer", "filters", "kernel_size", "strides", "padding", "name",
        "regularizer", "trained_weights"]
    for cls in (CF.SCCEMaxBin, CF.SCCEDifference, CF.SCCEInverse):
        assert list(inspect.signature(cls.__init__).parameters)[1:5] == ["layers", "penalty_rate", "log_dir", "l2_lambda"]
        assert hasattr(cls, "compute_total_loss")
    assert hasattr(CF.SCCEMaxBin, "compute_maxbin_penalty") and hasattr(CF.SCCEDifference, "compute_difference_penalty")
    assert hasattr(CF.SCCEInverse, "compute_inverse_penalty")
    assert NQ.MinValueConstraint(0.5).get_config() == {"min_value": 0.5}


def test_product_code_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under learned_quantization_amd/ may import or call it."""
    pkg = os.path.join(ROOT, "learned_quantization_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"
                assert "lq_oracle" not in src, f"{f} references the oracle"


def test_sparse_categorical_crossentropy_matches_oracle():
    rng = np.random.default_rng(0)
    logits = rng.normal(size=(6, 10)).astype(np.float32)
    p = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
    p[0, 3] = 0.0                                                   # clipped to 1e-7
    y = np.array([3, 1, 4, 1, 5, 9])
    got = lq.sparse_categorical_crossentropy(torch.tensor(y), torch.tensor(p)).numpy()
    np.testing.assert_allclose(got, O.sparse_categorical_crossentropy(y, p), rtol=1e-6)
    assert got[0] == pytest.approx(-np.log(1e-7), rel=1e-5)


def test_crossentropy_of_a_softmax_output_is_computed_from_the_logits():
    """Keras 2.11 (keras/backend.py sparse_categorical_crossentropy -> _get_logits): the output of a softmax activation carries
    its logits and the loss is -log_softmax(logits)[y], unclipped.  Saturated logits (the reference feeds raw 0..255 pixels;
    its nets saturate at initialisation) must still back-propagate softmax - onehot, where clip-then-log gives exactly 0."""
    from learned_quantization_amd.losses import softmax
    logits = torch.tensor([[60.0, 0.0, -40.0, 5.0], [0.0, 90.0, 1.0, 2.0], [0.3, 0.1, -0.2, 0.0]], requires_grad=True)
    y = torch.tensor([1, 0, 2])                                     # samples 0 and 1: p[y] < 1e-7, i.e. clipped in the other branch
    p = softmax(logits, dim=1)
    assert p._keras_logits is logits
    loss = lq.sparse_categorical_crossentropy(y, p)
    want = -torch.log_softmax(logits.detach().double(), dim=1)[torch.arange(3), y]
    np.testing.assert_allclose(loss.detach().numpy(), want.numpy(), rtol=1e-6)
    assert loss[0] > 16.2 and loss[1] > 16.2                         # beyond -log(1e-7) = 16.1: nothing was clipped
    loss.sum().backward()
    onehot = torch.zeros(3, 4)
    onehot[torch.arange(3), y] = 1.0
    np.testing.assert_allclose(logits.grad.numpy(), (torch.softmax(logits.detach(), 1) - onehot).numpy(), rtol=1e-5, atol=1e-7)
    assert float(logits.grad[0].abs().max()) > 0.99                 # saturated sample: gradient alive
    # the clip-and-log branch for probabilities that do not come from softmax(): zero gradient once clipped (Keras does the same)
    lg2 = logits.detach().clone().requires_grad_(True)
    p2 = torch.softmax(lg2, dim=1)
    lq.sparse_categorical_crossentropy(y, p2).sum().backward()
    assert float(lg2.grad[0].abs().max()) == 0.0 and float(lg2.grad[2].abs().max()) > 0.0
    # every model of the package ends in losses.softmax
    import inspect
    from learned_quantization_amd import models
    for cls in (models.MNISTDense, models.CIFARCNN, models.ResNet18Like, models.ResNet50Like):
        assert "return softmax(" in inspect.getsource(cls.forward)


def test_library_is_not_handed_out_before_the_device_selftest_passed(monkeypatch):
    """_hip.load(): `_lib` is published only after the self-test ran and passed; a failure is remembered and re-raised by
    every later load(); a load during stream capture (or without a visible GPU) defers the test instead of cancelling it."""
    from learned_quantization_amd import _hip
    saved = (_hip._lib, _hip._pending, _hip._selftest_error)
    try:
        _hip._lib, _hip._pending, _hip._selftest_error = None, None, None
        calls = []
        monkeypatch.setattr(_hip, "_device_selftest", lambda lib: calls.append(1) or False)      # deferred
        a = _hip.load()
        assert _hip._lib is None and _hip._pending is a
        assert _hip.load() is a and len(calls) == 2                                              # retried, same handle

        def failing(lib):
            _hip._selftest_error = "device self-test failed (simulated)"
            raise RuntimeError(_hip._selftest_error)
        monkeypatch.setattr(_hip, "_device_selftest", failing)
        with pytest.raises(RuntimeError, match="self-test failed"):
            _hip.load()
        monkeypatch.setattr(_hip, "_device_selftest", lambda lib: True)
        with pytest.raises(RuntimeError, match="self-test failed"):                              # remembered
            _hip.load()
        assert _hip._lib is None
        _hip._selftest_error = None
        assert _hip.load() is a and _hip._lib is a                                               # passes -> published
    finally:
        _hip._lib, _hip._pending, _hip._selftest_error = saved


def test_missing_extension_fails_loudly():
    """No silent fallback: if liblq_hip.so is absent every op raises (checked in a fresh interpreter)."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r);\n"
            "import torch, learned_quantization_amd as lq\n"
            "lq._hip.use_library('/nonexistent/liblq_hip.so')\n"
            "try:\n    lq.fq_forward(torch.ones(4), torch.ones(1))\nexcept RuntimeError as e:\n    print('RAISED', e)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert "RAISED" in out.stdout and "HIP extension not built" in out.stdout, out.stdout + out.stderr


def test_product_loader_reads_no_environment():
    """The loader opens the in-tree library whatever the environment says (LQ_HIP_LIB used to swap the
    library under every op; LQ_SKIP_SELFTEST used to waive the device self-test); development builds are chosen in code."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from learned_quantization_amd import _hip; print(_hip.LIB_PATH)" % ROOT)
    env = dict(os.environ, LQ_HIP_LIB="/nonexistent/liblq_hip.so", LQ_SKIP_SELFTEST="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert out.stdout.strip().endswith(os.path.join("learned_quantization_amd", "csrc", "liblq_hip.so")), out.stdout + out.stderr
    src = open(os.path.join(ROOT, "learned_quantization_amd", "_hip.py")).read()
    assert "os.environ" not in src and "getenv" not in src


def test_keras_adam_brings_loaded_moments_into_the_parameters_element_order():
    """optimizer state saved from a model with another kernel storage keeps the checkpoint's strides through
    torch's load_state_dict; the flat-memory update must not pair it with the wrong weights."""
    from learned_quantization_amd.optim import _adopt_layout
    w = torch.randn(3, 3, 4, 8)
    p_oihw = w.permute(3, 2, 0, 1).contiguous().permute(2, 3, 1, 0)          # HWIO shape, OIHW memory
    m, v = torch.randn(3, 3, 4, 8), torch.rand(3, 3, 4, 8)                   # a checkpoint written by an HWIO-contiguous model
    st = {"m": m.clone(), "v": v.clone()}
    _adopt_layout(st, p_oihw)
    for k, ref in (("m", m), ("v", v)):
        assert st[k].stride() == p_oihw.stride() and torch.equal(st[k], ref)                      # same values index by index
        flat = torch.as_strided(st[k], (st[k].numel(),), (1,))                                    # ... and what the kernel walks
        assert torch.equal(flat, ref.permute(3, 2, 0, 1).reshape(-1))
    st2 = {"m": st["m"], "v": st["v"]}
    _adopt_layout(st2, p_oihw)
    assert st2["m"] is st["m"], "already in the parameter's order: left alone"
    with pytest.raises(ValueError, match="state 'm'"):
        _adopt_layout({"m": torch.zeros(3, 3, 4, 7), "v": v}, p_oihw)


def test_export_arrays_are_c_contiguous_whatever_the_storage():
    """a 1x1 kernel stored OIHW comes back from .cpu().numpy() F-contiguous; numpy would pickle it in Fortran order."""
    from learned_quantization_amd.export import _host
    k = torch.arange(12, dtype=torch.int8).view(1, 1, 3, 4)
    stored = k.permute(3, 2, 0, 1).contiguous().permute(2, 3, 1, 0)
    raw = stored.numpy()
    assert not raw.flags["C_CONTIGUOUS"]
    h = _host(stored)
    assert h.flags["C_CONTIGUOUS"] and np.array_equal(h, k.numpy()) and h.tobytes() == k.numpy().tobytes()


def test_fast_division_sequence_is_exact_on_cpu(tmp_path):
    """tests/tools/check_fast_div.c: the uniform-divisor division of the kernels == IEEE '/' for all 2^23
    mantissas x 3 exponents x both signs, per divisor (16 fixed + 8 random divisors here; 1516 were run once)."""
    import subprocess
    exe = str(tmp_path / "check_fast_div")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", exe,
                           os.path.join(ROOT, "tests", "tools", "check_fast_div.c"), "-lm"])
    res = subprocess.run([exe, "8", "2024"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and " 0 mismatches" in res.stdout, res.stdout + res.stderr


def test_invariant_divisor_division_is_exact_on_cpu(tmp_path):
    """tests/tools/check_fastdiv.cpp: csrc/lq_fastdiv.hpp (the header the flat streaming kernels take an element's group from)
    == the CPU's `/` and `%` for 34 fixed divisors x 2 M dividends each, 2000 random divisors, and the edges of the quotient steps."""
    import subprocess
    exe = str(tmp_path / "check_fastdiv")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "learned_quantization_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "tools", "check_fastdiv.cpp")])
    res = subprocess.run([exe, "2000", "2025"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and " 0 mismatches" in res.stdout, res.stdout + res.stderr


def test_fragment_index_arithmetic_is_exact_on_cpu(tmp_path):
    """tests/tools/check_frag.cpp: csrc/lq_frag.hpp (where a (row block, tile, group) partial of the batch's column traversals lives,
    and which fragments a finalize block reads) == a brute-force walk over the columns, for every group width 1..64 x 16 group counts."""
    import subprocess
    exe = str(tmp_path / "check_frag")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "learned_quantization_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "tools", "check_frag.cpp")])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and " 0 mismatches" in res.stdout, res.stdout + res.stderr


def test_batch_abi_validation_without_gpu():
    """lq_batch_* argument checks that return before any HIP call."""
    import ctypes
    lib = _hip.load()
    handle = ctypes.c_void_p()
    assert lib.lq_batch_create(None, 1, ctypes.byref(handle)) == -1
    arr = (_hip.TensorDesc * 1)()
    assert lib.lq_batch_create(arr, 0, ctypes.byref(handle)) == -1
    assert lib.lq_batch_create(arr, 100000, ctypes.byref(handle)) == -1
    assert b"1..256" in lib.lq_last_error()
    arr[0] = _hip.TensorDesc(None, None, None, None, None, None, None, 1, 1, 16, 1e-3, 0.0)
    assert lib.lq_batch_create(arr, 1, ctypes.byref(handle)) == -1            # NULL P/s/out
    assert lib.lq_batch_forward(None, None) == -1
    assert lib.lq_batch_scale_grad(None, None, None, 0, None) == -1
    assert lib.lq_batch_workspace_bytes(None) == 0
    assert lib.lq_batch_destroy(None) == 0
    assert lib.lq_selftest_ratio_division(1, 0, 1, None, None) == -1


def test_shipped_library_reads_no_environment(tmp_path):
    """The development knobs (LQ_TUNE_*: traversal plan, partial layout, summation order) exist only in `make dev` builds
    (-DLQ_DEV_KNOBS, tools/): the shipped library does not import getenv at all, and its plan is a pure function of the
    descriptor -- lq_workspace_bytes is the same with every knob of the sources set."""
    import subprocess
    import sys
    src_dir = os.path.join(os.path.dirname(_hip.LIB_PATH))
    knobs = set()
    for f in os.listdir(src_dir):
        if f.endswith((".hip", ".hpp")):
            knobs |= set(re.findall(r'"(LQ_TUNE_[A-Z0-9_]+)"', open(os.path.join(src_dir, f)).read()))
    assert len(knobs) >= 15, knobs
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", _hip.LIB_PATH], text=True)
    assert "getenv" not in undefined, "the shipped library must not read the environment"
    probe = ("import ctypes, json; lib = ctypes.CDLL(%r); "
             "lib.lq_workspace_bytes.restype = ctypes.c_size_t; lib.lq_workspace_bytes.argtypes = [ctypes.c_int64] * 3; "
             "print(json.dumps([lib.lq_workspace_bytes(*d) for d in "
             "[(256, 3, 50176), (1, 1, 38535168), (1, 784, 128), (784, 128, 1), (9, 64, 128), (32768, 1001, 1), (1, 65536, 512), "
             "(1, 1048576, 32), (602112, 64, 1), (256, 2048, 49), (1, 8192, 4100), (12845056, 3, 1)]]))") % _hip.LIB_PATH
    base = subprocess.check_output([sys.executable, "-c", probe], text=True, env={k: v for k, v in os.environ.items() if not k.startswith("LQ_TUNE_")})
    for value in ("0", "1", "3", "128"):
        env = dict(os.environ, **{k: value for k in knobs})
        assert subprocess.check_output([sys.executable, "-c", probe], text=True, env=env) == base, f"LQ_TUNE_* = {value} changed the plan"


def test_memory_order_descriptor_matches_the_logical_groups():
    """descriptor.memory_descriptor: for any dense permutation of the axes, element i IN MEMORY belongs to group
    (i // inner) % G -- checked against the logical index along the scale's axis, element by element."""
    import itertools
    from learned_quantization_amd.descriptor import group_descriptor, memory_descriptor, memory_order, scale_shape
    rng = np.random.default_rng(0)
    for shape in [(3, 3, 4, 5), (1, 1, 6, 7), (2, 5), (7, 1, 3, 2), (4,)]:
        for perm in itertools.permutations(range(len(shape))):
            mem = torch.empty([shape[a] for a in perm])                       # memory axes, slowest first
            t = mem.permute([perm.index(a) for a in range(len(shape))])       # logical view of that memory
            assert tuple(t.shape) == shape
            order = memory_order(t.shape, t.stride())
            assert order is not None and [shape[a] for a in order if shape[a] != 1] == [shape[a] for a in perm if shape[a] != 1]
            for orient in ("rowwise", "columnwise", "channelwise", "scalar"):
                try:
                    ss = scale_shape(shape, orient) if orient == "scalar" or {"rowwise": 0, "columnwise": 1, "channelwise": 2}[orient] < len(shape) else None
                except IndexError:
                    ss = None
                if ss is None:
                    continue
                outer, G, inner = memory_descriptor(t.shape, t.stride(), ss)
                assert outer * G * inner == t.numel()
                # logical group index of every element, laid out in memory order
                axis = [i for i, d in enumerate(ss) if d != 1]
                idx = np.indices(shape)[axis[0]] if axis and len(ss) == len(shape) else np.zeros(shape, dtype=np.int64)
                flat_mem = np.transpose(idx, perm).reshape(-1)
                want = (np.arange(t.numel()) // inner) % G
                assert np.array_equal(flat_mem, want), (shape, perm, orient)
                if t.is_contiguous():
                    assert (outer, G, inner) == group_descriptor(shape, ss)
    # not a dense permutation: gaps, broadcast strides
    assert memory_order((4, 5), (10, 1)) is None and memory_order((4, 5), (0, 1)) is None
    assert memory_descriptor((4, 5), (10, 1), (4, 1)) is None


def test_conv_kernel_storage_is_a_layout_not_a_shape():
    import learned_quantization_amd as lq
    for st in ("oihw", "hwio"):
        lq.reset_layer_names()
        layer = lq.CustomConv2DLayer(seed=0, penalty_threshold=1e-3, orientation="channelwise", initializer=lq.RandomNormal(seed=3),
                                     filters=6, kernel_size=(3, 3), strides=(1, 1), padding="same", name="c", regularizer=None,
                                     input_shape=4, kernel_storage=st)
        assert tuple(layer.kernel.shape) == (3, 3, 4, 6) and tuple(layer.nested_q_k_layer.scale.shape) == (1, 1, 4, 1)
        assert layer.kernel.is_contiguous() == (st == "hwio") and layer.kernel.permute(3, 2, 0, 1).is_contiguous() == (st == "oihw")
        if st == "oihw":
            ref = layer.kernel.detach().clone()
        else:
            assert torch.equal(layer.kernel.detach(), ref), "same values from the same initializer for either storage"
    with pytest.raises(ValueError):
        lq.CustomConv2DLayer(seed=0, penalty_threshold=1e-3, initializer=lq.RandomNormal(seed=3), filters=6, kernel_storage="ohwi")
    with lq.default_kernel_storage("hwio"):
        m = lq.build_model("cifar", mode="nq", value=1e-11, seed=1, orientation="channelwise")
    assert all(p.is_contiguous() for p in m.parameters())
    m = lq.build_model("cifar", mode="nq", value=1e-11, seed=1, orientation="channelwise")
    assert any(not p.is_contiguous() for p in m.parameters()) and all(p.dim() == 4 for p in m.parameters() if not p.is_contiguous())
