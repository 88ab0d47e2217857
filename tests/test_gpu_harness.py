"""GPU tests of the "next" rows: end-to-end training step (f-3), integer export (f-1), tracking statistics (f-2)."""
import os
import zipfile

import numpy as np
import pytest
import torch

from oracle import lq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("config,mode,loss,orient", [
    ("mnist", "nq", None, "rowwise"), ("mnist", "cl", "difference", "columnwise"),
    ("cifar", "nq", None, "channelwise"), ("cifar", "cl", "maxbin", "rowwise"), ("cifar", "cl", "inverse", "scalar"),
])
def test_training_steps_run_and_respect_invariants(dev, config, mode, loss, orient, tmp_path):
    from learned_quantization_amd.train import Trainer, synthetic_batch
    value = 1e-3 if mode == "nq" else 1e-2
    tr = Trainer(config, mode, value, orient, loss, device=dev, log_dir=str(tmp_path))
    x, y = synthetic_batch(config, 16, dev, torch.Generator(device=dev).manual_seed(0))
    scales0 = [s.detach().clone() for s in tr.scale_opt.param_groups[0]["params"]]
    losses = [float(tr.step(x, y)) for _ in range(3)]
    assert all(np.isfinite(l) for l in losses)
    scales = tr.scale_opt.param_groups[0]["params"]
    assert all(float(s.min()) >= O.SCALE_MIN for s in scales)               # MinValueConstraint after every step
    moved = sum(int((a != b).any()) for a, b in zip(scales0, scales))
    assert moved > 0                                                           # the scales do learn
    if mode == "nq":
        # nested-quantization scale gradient is <= 0 -> scales can only grow
        assert all(bool((b >= a).all()) for a, b in zip(scales0, scales))
    l, acc = tr.evaluate(x, y)
    assert np.isfinite(l) and 0.0 <= acc <= 1.0


def test_resnet18_like_forward_backward_small_images(dev, tmp_path):
    from learned_quantization_amd.train import Trainer
    tr = Trainer("imagenette", "nq", 1e-11, "channelwise", None, device=dev, log_dir=str(tmp_path))
    x = torch.rand(4, 3, 64, 64, device=dev) * 255.0
    y = torch.randint(0, 10, (4,), device=dev)
    p = tr.model(x)
    assert p.shape == (4, 10) and torch.allclose(p.sum(1), torch.ones(4, device=dev), atol=1e-5)
    assert np.isfinite(float(tr.step(x, y)))
    stem = tr.custom_layers[0]
    assert stem.kernel.grad is not None and stem.nested_q_k_layer.scale.grad is not None


def test_export_matches_reference_format(dev, tmp_path):
    import learned_quantization_amd as lq
    lq.reset_layer_names()
    m = lq.build_model("cifar", mode="nq", value=1e-11, seed=1, orientation="channelwise", device=dev)
    with torch.no_grad():
        for l in lq.custom_layers_of(m):
            l.nested_q_k_layer.scale.fill_(0.01)
            l.nested_q_b_layer.scale.fill_(0.02)
    sizes = lq.save_compress_parameters(m, str(tmp_path))
    w = np.load(tmp_path / "weights.npy", allow_pickle=True).item()           # our own file: a pickled dict like the reference's
    assert "custom_conv2d_layer/W" in w and "custom_conv2d_layer_5/b" in w
    for l in lq.custom_layers_of(m):
        k, b = l.kernel.detach().cpu().numpy(), l.b.detach().cpu().numpy()
        assert w[l.name + "/W"].dtype == np.int8 and w[l.name + "/W"].shape == k.shape      # HWIO, like the reference
        np.testing.assert_array_equal(w[l.name + "/W"], O.export_int8(k, np.full((1, 1, k.shape[2], 1), 0.01, np.float32)))
        np.testing.assert_array_equal(w[l.name + "/b"], O.export_int8(b, np.array([0.02], np.float32)))
    with zipfile.ZipFile(tmp_path / "weights.zip") as z:
        assert z.namelist() == ["weights.npy"] and z.getinfo("weights.npy").compress_type == zipfile.ZIP_DEFLATED
    lines = open(tmp_path / "file_sizes.log").read().splitlines()
    assert lines[0].startswith("Weights size: ") and lines[1].startswith("Compressed weights size: ") and lines[0].endswith(" MB")
    assert sizes["zip_mb"] < sizes["weights_mb"]
    sc = np.load(tmp_path / "scales.npz")
    assert sc["custom_conv2d_layer/W_scale"].shape == (1, 1, 3, 1)


def test_tracking_callback_logs(dev, tmp_path):
    import learned_quantization_amd as lq
    lq.reset_layer_names()
    layer = lq.CustomConv2DLayer(seed=1, penalty_threshold=1e-11, orientation="channelwise", initializer=lq.RandomNormal(seed=3),
                                 filters=8, kernel_size=(3, 3), strides=(1, 1), padding="same", name="n", regularizer=None,
                                 input_shape=4, device=dev)
    with torch.no_grad():
        layer.nested_q_k_layer.scale.fill_(0.01)
        layer.nested_q_b_layer.scale.fill_(0.02)
    cb = lq.NestedScaleTrackingCallback(layer, str(tmp_path))
    cb.on_train_begin()
    st = cb.on_epoch_end(0)
    cb.on_epoch_end(1)
    cb.on_train_end()
    k, b = layer.kernel.detach().cpu().numpy(), layer.b.detach().cpu().numpy()
    qk = O.quantized_integers(k, np.full((1, 1, 4, 1), 0.01, np.float32))
    qb = O.quantized_integers(b, np.array([0.02], np.float32))
    assert st["unique_k"] == len(np.unique(qk)) and st["unique_b"] == len(np.unique(qb))
    np.testing.assert_array_equal(st["max_k"].numpy(), np.max(np.abs(qk), axis=1).flatten())   # custom_callbacks.py:98-99
    assert st["max_b"] == float(np.max(np.abs(qb)))
    name = "Kernel_custom_conv2d_layer_(3, 3, 4, 8)"
    text = open(tmp_path / "on_epoch_end" / f"Number_of_unique_{name}.log").read().splitlines()
    assert text == ["Epoch 0", str(st["unique_k"]), "Epoch 1", str(st["unique_k"])]             # parser format of plot_scripts.py:13-38
    mx = open(tmp_path / "on_epoch_end" / f"Max_{name}.log").read().splitlines()
    assert mx[0] == "Epoch 0" and len(mx) == 2 * (1 + 3 * 4 * 8)
    q_lines = open(tmp_path / "on_train_end" / f"Quantized_{name}_Columnwise-scaler_(1, 1, 4, 1).log").read().splitlines()
    assert len(q_lines) == k.size and float(q_lines[0]) == qk.flatten()[0]
    u_lines = open(tmp_path / "on_train_begin" / f"Unique_initial_quantized_{name}_Columnwise-scaler_(1, 1, 4, 1).log").read().splitlines()
    assert len(u_lines) == len(np.unique(qk)) and ", " in u_lines[0]
    acc = lq.AccuracyLossTrackingCallBack(str(tmp_path))
    acc.on_epoch_end(0, {"val_accuracy": 0.5, "val_loss": 1.0, "accuracy": 0.4, "loss": 1.2})
    assert open(tmp_path / "accuracy" / "val_accuracy.log").read() == "Epoch 0\n0.5\n"


def test_capturable_scale_adam_and_graphed_step(dev, tmp_path):
    """K6 with the step counter on the device == host-step K6; the whole step replays from a hipGraph."""
    import learned_quantization_amd as lq
    rng = np.random.default_rng(0)
    p1 = torch.nn.Parameter(torch.full((32,), lq.SCALE_INIT, device=dev))
    p2 = torch.nn.Parameter(torch.full((32,), lq.SCALE_INIT, device=dev))
    for p in (p1, p2):
        p.lq_constraint = lq.MinValueConstraint(lq.SCALE_INIT)
    o1, o2 = lq.ScaleAdam([p1], lr=1e-4), lq.ScaleAdam([p2], lr=1e-4, capturable=True)
    for _ in range(5):
        g = torch.tensor((rng.normal(size=32) * 1e-2).astype(np.float32), device=dev)
        p1.grad, p2.grad = g.clone(), g.clone()
        o1.step()
        o2.step()
        assert torch.equal(p2, p1)          # one device arithmetic for both forms (k_adam_dev): bit-identical
    from learned_quantization_amd.train import Trainer, synthetic_batch
    tr = Trainer("cifar", "nq", 1e-3, "channelwise", None, device=dev, log_dir=str(tmp_path), graph=True)
    x, y = synthetic_batch("cifar", 32, dev, torch.Generator(device=dev).manual_seed(0))
    s0 = tr.custom_layers[0].nested_q_k_layer.scale.detach().clone()
    losses = [float(tr.step_graphed(x, y)) for _ in range(4)]
    assert all(np.isfinite(l) for l in losses) and tr.graph is not None
    s1 = tr.custom_layers[0].nested_q_k_layer.scale.detach()
    assert bool((s1 >= s0).all()) and float(s1.min()) >= O.SCALE_MIN


@pytest.mark.parametrize("orient,batched", [("rowwise", False), ("rowwise", True), ("columnwise", False)])
def test_graphed_step_equals_eager_step(dev, tmp_path, orient, batched):
    """The whole step from a hipGraph == the eager step, parameter for parameter (dense model: rocBLAS is deterministic; the
    scale optimizer forms its bias corrections on the device in both forms)."""
    from learned_quantization_amd.train import Trainer, synthetic_batch
    x, y = synthetic_batch("mnist", 32, dev, torch.Generator(device=dev).manual_seed(0))
    res = []
    for graph in (False, True):
        tr = Trainer("mnist", "nq", 1e-3, orient, None, device=dev, log_dir=str(tmp_path), graph=graph, batched=batched, seed=7)
        tr.model.eval()
        step = tr.step_graphed if graph else tr.step
        for _ in range(4 + (0 if graph else 3)):          # step_graphed runs 3 eager warm-up steps before it captures
            step(x, y)
        torch.cuda.synchronize()
        res.append({n: p.detach().clone() for n, p in tr.model.named_parameters()})
    for n in res[0]:
        assert torch.equal(res[0][n], res[1][n]), n


def test_row_stream_kernels_inside_a_capture(dev, tmp_path):
    """Row-wise conv kernels have rows of kw*ci*co >= 1024 elements: the row-stream traversal, which the library launches through
    hipExtLaunchKernelGGL (lq_profile_events hook, NULL events here) -- it must be capturable like a plain launch."""
    from learned_quantization_amd.train import Trainer, synthetic_batch
    tr = Trainer("cifar", "nq", 1e-3, "rowwise", None, device=dev, log_dir=str(tmp_path), graph=True)
    x, y = synthetic_batch("cifar", 16, dev, torch.Generator(device=dev).manual_seed(0))
    losses = [float(tr.step_graphed(x, y)) for _ in range(3)]
    assert all(np.isfinite(l) for l in losses) and tr.graph is not None
    assert all(float(s.min()) >= O.SCALE_MIN for s in tr.scale_opt.param_groups[0]["params"])


def test_experiment_driver_end_to_end(dev, tmp_path, mnist_weights):
    """The reference's main() flow on synthetic data: log tree, callbacks, export -- from scratch and post-training (PTQ)."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    base = ["--seed", "42", "--epochs", "2", "--steps-per-epoch", "3", "--batch", "16", "--log-root", str(tmp_path)]
    runs = [["--config", "cifar", "--orientation", "channelwise", "--training", "from_scratch"],
            ["--config", "cifar", "--orientation", "rowwise", "--training", "from_scratch", "--custom_loss", "difference", "--batched"],
            ["--config", "mnist", "--orientation", "rowwise", "--training", "post_training", "--value", "1e-10",
             "--baseline-weights", os.path.join(root, "tests", "golden", "mnist_baseline_weights.npz")]]
    for extra in runs:
        res = subprocess.run([sys.executable, "-m", "learned_quantization_amd.experiment"] + base + extra, cwd=root,
                             capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stdout + res.stderr
        out = json.loads(res.stdout.strip().splitlines()[-1])
        d = out["log_dir"]
        assert f"{extra[3]}_{extra[5]}" in d and "seed_42" in d and "_lr_0.0001_pr_" in d
        for f in ("model_structure.log", "weights.npy", "weights.zip", "file_sizes.log", "scales.npz",
                  "accuracy/val_accuracy.log", "loss/train_loss.log"):
            assert os.path.exists(os.path.join(d, f)), f
        assert len(os.listdir(os.path.join(d, "on_epoch_end"))) > 0 and len(os.listdir(os.path.join(d, "on_train_end"))) > 0
        assert open(os.path.join(d, "loss", "val_loss.log")).read().startswith("Epoch 0\n")
        assert out["export"]["zip_mb"] > 0 and np.isfinite(out["final"]["loss"])
        if extra[1] == "mnist":
            # PTQ from the shipped baseline: the first layer's weights really are the baseline's
            w = np.load(os.path.join(d, "weights.npy"), allow_pickle=True).item()
            assert w["custom_dense_layer/W"].shape == mnist_weights["W1"].shape


def test_functional_sanity_quantization_emerges_without_accuracy_loss(dev, mnist_weights, tmp_path):
    """SURVEY 8c item 3 (coarse end-to-end sanity): from the shipped MNIST baseline weights, on a synthetic task labelled
    by the baseline itself, the nested quantization layer collapses the integer range for lambda > 0 (thesis chapter4.tex:
    123-127: tens of integers), leaves it untouched for lambda = 0 (no quantisation pressure), and the quantised model
    tracks the accuracy of the unquantised run."""
    import learned_quantization_amd as lq
    from learned_quantization_amd.train import Trainer
    W1, b1, W2, b2 = (torch.tensor(mnist_weights[k], device=dev) for k in ("W1", "b1", "W2", "b2"))

    def teacher(x):
        return ((torch.relu(torch.flatten(x, 1) @ W1 + b1)) @ W2 + b2).argmax(1)

    def images(n, g):
        m = (torch.rand(n, 1, 28, 28, device=dev, generator=g) < 0.19).float()
        return m * torch.rand(n, 1, 28, 28, device=dev, generator=g)

    results = {}
    for lam in (0.0, 1e-8):
        tr = Trainer("mnist", "nq", lam, "rowwise", None, device=dev, seed=42, batched=True, log_dir=str(tmp_path))
        with torch.no_grad():
            tr.model.dense_1.W.copy_(W1); tr.model.dense_1.b.copy_(b1); tr.model.dense_2.W.copy_(W2); tr.model.dense_2.b.copy_(b2)
        g = torch.Generator(device=dev).manual_seed(0)
        xv = images(2048, g)
        yv = teacher(xv)
        _, acc0 = tr.evaluate(xv, yv)
        assert acc0 > 0.99                                   # floor() at the initial scale 1.19e-5 is nearly lossless
        for _ in range(600):
            x = images(32, g)
            tr.step(x, teacher(x))
        uniq = int(lq.q_unique(tr.model.dense_1.W.data, tr.model.dense_1.nested_q_w_layer.scale.data)[0].numel())
        results[lam] = (uniq, tr.evaluate(xv, yv)[1], float(tr.model.dense_1.nested_q_w_layer.scale.min()))
    assert results[0.0][0] > 10000 and results[0.0][2] == pytest.approx(lq.SCALE_INIT)     # lambda = 0: scales never move
    assert results[1e-8][0] < 300                                                            # lambda > 0: a few dozen/hundred integers
    assert abs(results[1e-8][1] - results[0.0][1]) < 0.05                                    # same accuracy as the unquantised run


@pytest.mark.parametrize("kind", ["maxbin", "difference", "inverse"])
def test_nq_layer_with_loss_term_scale_gradient_is_the_sum_of_both_pieces(dev, kind, tmp_path):
    """BASELINE.json configs[3] ("nested quantization + custom_loss_terms penalty").  EXTENSION WITHOUT A REFERENCE CALL SITE:
    the reference's loss-term variant zeroes the op's scale gradient (CL-L:61-62) and its NQ variant is never trained with an
    SCCE* loss.  A layer that has penalty_threshold set AND is handed to a loss term gets
        ds = hand-written NQ gradient (NQ-L:62-118, from the dy the op received) + d(rate * penalty)/ds (CL-F:75-275),
    which is what TensorFlow's autodiff would add up too.  Checked against the oracle's two pieces."""
    import learned_quantization_amd as lq
    from oracle import lq_oracle_f64 as O64
    lam, rate = 2e-3, 0.25
    lq.reset_layer_names()
    layer = lq.CustomDenseLayer(seed=0, units=24, penalty_threshold=lam, orientation="rowwise", initializer=lq.RandomNormal(seed=1),
                                name="d", regularizer=None, input_shape=40, device=dev, penalty_rate=rate)
    with torch.no_grad():
        layer.nested_q_w_layer.scale.uniform_(1e-3, 1e-2)
        layer.nested_q_b_layer.scale.uniform_(1e-3, 1e-2)
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(16, 40, device=dev, generator=g)
    wgt = torch.randn(16, 24, device=dev, generator=g) * 1e-3
    qw = layer.nested_q_w_layer(layer.W)
    qb = layer.nested_q_b_layer(layer.b)
    qw.retain_grad()
    qb.retain_grad()
    task = ((x @ qw + qb) * wgt).sum()
    cls = {"maxbin": lq.SCCEMaxBin, "difference": lq.SCCEDifference, "inverse": lq.SCCEInverse}[kind]
    pen = getattr(cls([layer], rate, str(tmp_path)), f"compute_{kind}_penalty")()
    (task + rate * pen).backward()
    W, b = layer.W.detach().cpu().numpy(), layer.b.detach().cpu().numpy()
    sw, sb = layer.nested_q_w_layer.scale.detach().cpu().numpy(), layer.nested_q_b_layer.scale.detach().cpu().numpy()
    l64 = [(W, sw, O.group_descriptor(W.shape, sw.shape), b, sb, O.group_descriptor(b.shape, sb.shape))]
    e = O64.penalty_grads(kind, l64, rate)[0]
    for nested, P, s, dy, pds, pabs, nm in ((layer.nested_q_w_layer, W, sw, qw.grad, e["dsK"], e["dsK_abs"], "W"),
                                            (layer.nested_q_b_layer, b, sb, qb.grad, e["dsb"], e["dsb_abs"], "b")):
        _, nq = O.nq_backward(P, s, lam, dy.cpu().numpy())
        want = nq.reshape(-1).astype(np.float64) + pds
        got = nested.scale.grad.cpu().numpy().reshape(-1).astype(np.float64)
        bound = 1e-5 * np.abs(nq.reshape(-1)) + 1e-5 * pabs + 2.0 ** -23 * np.abs(want)
        assert np.all(np.abs(got - want) <= bound), f"{kind} {nm}: {np.abs(got - want).max()} vs {bound.min()}"
        assert np.all(nq <= 0)                               # the NQ piece alone is <= 0; the sum need not be


@pytest.mark.parametrize("kind", ["maxbin", "difference"])
def test_nqcl_trainer_batched_equals_unbatched(dev, kind, tmp_path):
    """mode "nqcl" through the Trainer: the batched path (NQ ds from lq_batch_scale_grad, penalty ds ADDED by
    lq_batch_penalty_grads | LQ_PENALTY_ACCUMULATE_DS) must give the step of the per-tensor autograd path."""
    from learned_quantization_amd.train import Trainer, synthetic_batch
    x, y = synthetic_batch("mnist", 32, dev, torch.Generator(device=dev).manual_seed(0))
    res = []
    for batched in (False, True):
        tr = Trainer("mnist", "nqcl", (1e-3, 0.05), "rowwise", kind, device=dev, log_dir=str(tmp_path), batched=batched)
        tr.model.eval()
        for _ in range(2):
            tr.step(x, y)
        res.append([s.detach().clone() for s in tr.scale_opt.param_groups[0]["params"]])
    for a, b in zip(*res):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5)
    assert any(bool((s != O.SCALE_INIT).any()) for s in res[0])
