"""Thin counterpart of the reference's ``experiment.py`` drivers (SURVEY f-3) on synthetic data.

  python -m learned_quantization_amd.experiment --config cifar --seed 42 --orientation channelwise \
         --training from_scratch [--custom_loss maxbin] [--value 1e-11] [--epochs 2 --steps-per-epoch 20 --batch 128]

Flags ``--seed --orientation --training [--custom_loss]`` and the log tree
``logs/<orientation>_<training>/seed_<seed>/<timestamp>_lr_<lr>_pr_<value>/`` mirror
/root/reference/MNIST/nested_quantization_layer/experiment.py:252-300 and
/root/reference/CIFAR-10/custom_loss_terms/experiment.py:289-302; the run does what the reference's ``main`` does
around ``model.fit``: structure log (utils/log_scripts.py:10-58), one tracking callback per custom layer + the
accuracy/loss callback (experiment.py:61-81), per-epoch validation, the Imagenette LR schedule
(IMAGENETTE/.../experiment.py:67-78), the integer export at the end (utils/log_scripts.py:61-97).
``post_training`` needs ``--baseline-weights file.npz`` (arrays W1,b1,W2,b2 for ``--config mnist``): the reference's
PTQ entry (MNIST/.../experiment.py:70-118) with the weights of its shipped ``baseline_model.keras``.
No dataset is reachable offline: images are U[0,255), labels uniform (so accuracies are chance level).
"""
from __future__ import annotations

import argparse
import json
import os
import time

import numpy as np
import torch

from . import layers as L
from .export import save_compress_parameters
from .tracking import AccuracyLossTrackingCallBack, NestedScaleTrackingCallback
from .train import Trainer, synthetic_batch


def scheduler(epoch: int, lr: float) -> float:
    """IMAGENETTE/nested_quantization_layer/experiment.py:67-78."""
    if epoch == 40:
        return lr * 0.5
    if epoch == 60:
        return lr * 0.2
    return lr


def log_model_structure(model: torch.nn.Module, folder_name: str, filename: str = "model_structure.log") -> None:
    """utils/log_scripts.py:10-58: layers with the shapes of their parameters and scalers."""
    os.makedirs(folder_name, exist_ok=True)
    with open(os.path.join(folder_name, filename), "w") as f:
        f.write("\n" + "-" * 80 + "\n")
        f.write("MODEL STRUCTURE\n")
        for i, (name, layer) in enumerate(model.named_modules()):
            if name == "" or isinstance(layer, L.CustomQuantizedScaleLayer):
                continue
            f.write(f"\nLAYER {i}: {name} ({type(layer).__name__})\n")
            if getattr(layer, "b", None) is not None:
                f.write(f"  - Bias with shape: {tuple(layer.b.shape)}\n")
            if hasattr(layer, "nested_q_b_layer"):
                f.write(f"  - Bias scaler with shape: {tuple(layer.nested_q_b_layer.scale.shape)}\n")
            if hasattr(layer, "W"):
                f.write(f"  - Weight matrix with shape: {tuple(layer.W.shape)}\n")
            if hasattr(layer, "nested_q_w_layer"):
                f.write(f"  - Weight matrix scaler with shape: {tuple(layer.nested_q_w_layer.scale.shape)}\n")
            if hasattr(layer, "kernel"):
                f.write(f"  - Kernel with shape: {tuple(layer.kernel.shape)}\n")
            if hasattr(layer, "nested_q_k_layer"):
                f.write(f"  - Kernel scaler with shape: {tuple(layer.nested_q_k_layer.scale.shape)}\n")
        f.write("-" * 80 + "\n")


def fit(tr: Trainer, epochs: int, steps_per_epoch: int, batch: int, callbacks, lr: float, use_scheduler: bool = False,
        val_steps: int = 2, seed: int = 42):
    """model.fit counterpart: callbacks get Keras-style ``logs`` dicts."""
    g = torch.Generator(device=tr.device).manual_seed(seed)
    val = [synthetic_batch(tr.config, batch, tr.device, g) for _ in range(val_steps)]
    for cb in callbacks:
        if hasattr(cb, "on_train_begin"):
            cb.on_train_begin()
    history = []
    for epoch in range(epochs):
        if use_scheduler:
            lr = scheduler(epoch, lr)
            for opt in (tr.opt,):
                for grp in opt.param_groups:
                    grp["lr"] = lr
            if hasattr(tr.scale_opt, "param_groups") and tr.batch is None:
                for grp in tr.scale_opt.param_groups:
                    grp["lr"] = lr
            elif tr.batch is not None:
                tr.batch.hyper["lr"] = lr
        run_loss, run_acc = 0.0, 0.0
        for _ in range(steps_per_epoch):
            x, y = synthetic_batch(tr.config, batch, tr.device, g)
            loss = tr.step(x, y)
            run_loss += float(loss.detach())
        with torch.no_grad():
            tr.model.eval()
            p = tr.model(x)
            run_acc = float((p.argmax(1) == y).float().mean())
        vl, va = zip(*(tr.evaluate(vx, vy) for vx, vy in val))
        logs = {"loss": run_loss / steps_per_epoch, "accuracy": run_acc, "val_loss": float(np.mean(vl)),
                "val_accuracy": float(np.mean(va)), "lr": lr}
        history.append(logs)
        for cb in callbacks:
            cb.on_epoch_end(epoch, logs)
    for cb in callbacks:
        if hasattr(cb, "on_train_end"):
            cb.on_train_end()
    return history


def main(argv=None):
    ap = argparse.ArgumentParser(description="Train model for one penalty value in this scenario (synthetic data).")
    ap.add_argument("--config", choices=["mnist", "cifar", "imagenette"], default="cifar")
    ap.add_argument("--seed", type=int, required=True, help="Random seed for reproducibility.")
    ap.add_argument("--orientation", type=str, required=True, choices=["rowwise", "columnwise", "channelwise", "scalar"])
    ap.add_argument("--training", type=str, required=True, choices=["post_training", "from_scratch"])
    ap.add_argument("--custom_loss", type=str, default=None, choices=["maxbin", "difference", "inverse"],
                    help="custom_loss_terms variant; omitted = nested_quantization_layer variant")
    ap.add_argument("--value", type=float, default=None, help="penalty_threshold (default 1e-11) or penalty_rate (default 1e-7)")
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--steps-per-epoch", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--baseline-weights", default=None)
    ap.add_argument("--log-root", default="logs")
    ap.add_argument("--batched", action="store_true")
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X (no CPU fallback)")
    dev = torch.device("cuda", 0)
    mode = "cl" if args.custom_loss else "nq"
    value = args.value if args.value is not None else (1e-7 if mode == "cl" else 1e-11)
    scenario_dir = os.path.join(args.log_root, f"{args.orientation}_{args.training}", f"seed_{args.seed}")
    log_dir = os.path.join(scenario_dir, f"{time.strftime('%Y-%m-%d_%H-%M-%S')}_lr_{args.lr}_pr_{value}")
    os.makedirs(log_dir, exist_ok=True)

    tr = Trainer(args.config, mode, value, args.orientation, args.custom_loss, lr=args.lr, seed=args.seed, device=dev,
                 log_dir=log_dir, batched=args.batched)
    if args.training == "post_training":
        if args.config != "mnist" or not args.baseline_weights:
            raise SystemExit("post_training needs --config mnist --baseline-weights <npz with W1,b1,W2,b2> "
                             "(only the MNIST baseline ships with the reference)")
        w = np.load(args.baseline_weights)
        with torch.no_grad():
            tr.model.dense_1.W.copy_(torch.from_numpy(w["W1"]))
            tr.model.dense_1.b.copy_(torch.from_numpy(w["b1"]))
            tr.model.dense_2.W.copy_(torch.from_numpy(w["W2"]))
            tr.model.dense_2.b.copy_(torch.from_numpy(w["b2"]))
    log_model_structure(tr.model, log_dir)
    callbacks = [NestedScaleTrackingCallback(layer, log_dir) for layer in tr.custom_layers]
    callbacks.append(AccuracyLossTrackingCallBack(log_dir))
    t0 = time.perf_counter()
    history = fit(tr, args.epochs, args.steps_per_epoch, args.batch, callbacks, args.lr,
                  use_scheduler=(args.config == "imagenette"), seed=args.seed)
    sizes = save_compress_parameters(tr.model, log_dir)
    stats = callbacks[0].stats()
    print(json.dumps({"log_dir": log_dir, "epochs": args.epochs, "seconds": time.perf_counter() - t0,
                      "final": history[-1], "export": sizes,
                      "first_layer_unique_integers": stats["unique_k"], "first_layer_max_abs_q": float(stats["max_k"].max())}))


if __name__ == "__main__":
    main()
