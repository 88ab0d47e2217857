"""Scale / quantisation tracking statistics (SURVEY f-2) in the reference's log format.

Mirrors /root/reference/CIFAR-10/nested_quantization_layer/custom_components/custom_callbacks.py:
``NestedScaleTrackingCallback`` (:9-208) and ``AccuracyLossTrackingCallBack`` (:211-237).  File names,
directory layout (``on_epoch_end`` / ``on_train_end`` / ``on_train_begin``) and the
``Epoch N`` + one-value-per-line format are kept so the reference's plot_scripts.py parsers
(process_file_logged_per_epoch, plot_scripts.py:13-38) keep working.

The statistics that define the Pareto axes are computed on the GPU -- ``floor(P/s)`` by the HIP kernel K1,
``max|q|`` over axis 1 by ``lq_q_absmax_over_axis``, unique values/counts by ``lq_q_minmax`` + ``lq_q_histogram``
(LDS-privatised integer histogram) -- and only the
results cross PCIe.  The reference's per-epoch dump of every raw kernel value (:52-66) is opt-in
(``dump_values=True``): it is tens of MB of text per epoch and not needed by the Pareto plots.
"""
from __future__ import annotations

import os

import torch

from . import ops


def _shape(t) -> str:
    return "(" + ", ".join(str(int(d)) for d in t.shape) + ("," if t.dim() == 1 else "") + ")"


class NestedScaleTrackingCallback:
    def __init__(self, layer, log_dir, dump_values: bool = False):
        self.layer = layer
        self.dump_values = dump_values
        if hasattr(layer, "kernel"):
            self.param, self.qk_layer, kernel_name = layer.kernel, layer.nested_q_k_layer, f"Kernel_{layer.name}"
        else:
            self.param, self.qk_layer, kernel_name = layer.W, layer.nested_q_w_layer, "Weights"
        self.qb_layer = layer.nested_q_b_layer
        self.k_scale, self.b_scale = self.qk_layer.scale, self.qb_layer.scale
        bias_name = f"Bias_{layer.name}" if hasattr(layer, "kernel") else "Bias"
        ks_name, bs_name = self.qk_layer.scale_name, self.qb_layer.scale_name
        ksh, bsh = _shape(self.param), _shape(layer.b)
        kss, bss = _shape(self.k_scale), _shape(self.b_scale)
        e, t, b0 = (os.path.join(log_dir, d) for d in ("on_epoch_end", "on_train_end", "on_train_begin"))
        for d in (e, t, b0):
            os.makedirs(d, exist_ok=True)
        self.log_file_path_kernel = f"{e}/{kernel_name}_{ksh}.log"
        self.log_file_path_biases = f"{e}/{bias_name}_{bsh}.log"
        self.log_file_path_k_scale = f"{e}/{kernel_name}_{ksh}_{ks_name}_{kss}.log"
        self.log_file_path_b_scale = f"{e}/{bias_name}_{bsh}_{bs_name}_{bss}.log"
        self.log_file_path_qk_epoch = f"{e}/Number_of_unique_{kernel_name}_{ksh}.log"
        self.log_file_path_qb_epoch = f"{e}/Number_of_unique_{bias_name}_{bsh}.log"
        self.log_file_path_qk_epoch_max = f"{e}/Max_{kernel_name}_{ksh}.log"
        self.log_file_path_qb_epoch_max = f"{e}/Max_{bias_name}_{bsh}.log"
        self.log_file_path_qk = f"{t}/Quantized_{kernel_name}_{ksh}_{ks_name}_{kss}.log"
        self.log_file_path_qb = f"{t}/Quantized_{bias_name}_{bsh}_{bs_name}_{bss}.log"
        self.log_file_path_qk_unique = f"{t}/Unique_quantized_{kernel_name}_{ksh}_{ks_name}_{kss}.log"
        self.log_file_path_qb_unique = f"{t}/Unique_quantized_{bias_name}_{bsh}_{bs_name}_{bss}.log"
        self.log_file_path_qk_initial = f"{b0}/Initial_Quantized_{kernel_name}_{ksh}__{ks_name}__{kss}.log"
        self.log_file_path_qb_initial = f"{b0}/Initial_Quantized_{bias_name}_{bsh}_{bs_name}_{bss}.log"
        self.log_file_path_qk_initial_unique = f"{b0}/Unique_initial_quantized_{kernel_name}_{ksh}_{ks_name}_{kss}.log"
        self.log_file_path_qb_initial_unique = f"{b0}/Unique_initial_quantized_{bias_name}_{bsh}_{bs_name}_{bss}.log"

    # ---- statistics (device side)
    def stats(self):
        """{'unique_k', 'unique_b', 'max_k' (max|q| over axis 1), 'max_b'} -- custom_callbacks.py:84-129."""
        axis = 1 if self.param.dim() > 1 else 0
        b = self.layer.b.data
        return {
            "unique_k": int(ops.q_unique(self.param.data, self.k_scale.data)[0].numel()),
            "unique_b": int(ops.q_unique(b, self.b_scale.data)[0].numel()),
            "max_k": ops.q_absmax_over_axis(self.param.data, self.k_scale.data, axis).flatten().cpu(),
            # global max|q| of the bias (custom_callbacks.py:123): the same kernel, the bias viewed as (1, n, 1)
            "max_b": float(ops.q_absmax_over_axis(b.reshape(1, -1), self.b_scale.data.reshape(1, 1), 1)),
        }

    @staticmethod
    def _append(path, header, values):
        with open(path, "a") as f:
            if header is not None:
                f.write(header)
            for v in values:
                f.write(f"{v}\n")

    def on_epoch_end(self, epoch, logs=None):
        hdr = f"Epoch {epoch}\n"
        if self.dump_values:
            self._append(self.log_file_path_kernel, hdr, self.param.detach().flatten().cpu().numpy())
            self._append(self.log_file_path_biases, hdr, self.layer.b.detach().flatten().cpu().numpy())
        self._append(self.log_file_path_k_scale, hdr, self.k_scale.detach().flatten().cpu().numpy())
        self._append(self.log_file_path_b_scale, hdr, self.b_scale.detach().flatten().cpu().numpy())
        st = self.stats()
        self._append(self.log_file_path_qk_epoch, hdr, [st["unique_k"]])
        self._append(self.log_file_path_qk_epoch_max, hdr, st["max_k"].numpy())
        self._append(self.log_file_path_qb_epoch, hdr, [st["unique_b"]])
        self._append(self.log_file_path_qb_epoch_max, hdr, [st["max_b"]])
        return st

    def _dump_quantized(self, path_k, path_b, path_ku, path_bu):
        for param, scale, p_all, p_unique in ((self.param, self.k_scale, path_k, path_ku),
                                              (self.layer.b, self.b_scale, path_b, path_bu)):
            q = ops.quantized_integers(param.data, scale.data, torch.float32).flatten()
            self._append(p_all, None, q.cpu().numpy())
            u, c = ops.q_unique(param.data, scale.data)                 # HIP histogram, not a device sort
            with open(p_unique, "a") as f:
                for value, count in zip(u.cpu().numpy().astype("float32"), c.cpu().numpy()):
                    f.write(f"{value}, {count}\n")

    def on_train_end(self, logs=None):
        self._dump_quantized(self.log_file_path_qk, self.log_file_path_qb, self.log_file_path_qk_unique,
                             self.log_file_path_qb_unique)

    def on_train_begin(self, logs=None):
        self._dump_quantized(self.log_file_path_qk_initial, self.log_file_path_qb_initial,
                             self.log_file_path_qk_initial_unique, self.log_file_path_qb_initial_unique)


class AccuracyLossTrackingCallBack:
    """custom_callbacks.py:211-237: train/val accuracy and loss, one value per epoch."""

    def __init__(self, log_dir, accuracy_file="accuracy.log", loss_file="loss.log"):
        os.makedirs(f"{log_dir}/accuracy", exist_ok=True)
        os.makedirs(f"{log_dir}/loss", exist_ok=True)
        self.accuracy_log_file_path = f"{log_dir}/accuracy/train_{accuracy_file}"
        self.val_accuracy_log_file_path = f"{log_dir}/accuracy/val_{accuracy_file}"
        self.loss_log_file_path = f"{log_dir}/loss/train_{loss_file}"
        self.val_loss_log_file_path = f"{log_dir}/loss/val_{loss_file}"

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        for path, key in ((self.val_accuracy_log_file_path, "val_accuracy"), (self.val_loss_log_file_path, "val_loss"),
                          (self.accuracy_log_file_path, "accuracy"), (self.loss_log_file_path, "loss")):
            with open(path, "a") as f:
                f.write(f"Epoch {epoch}\n")
                f.write(f"{logs.get(key)}\n")
