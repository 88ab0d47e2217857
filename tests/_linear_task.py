"""The product Trainer with the NETWORK taken out of the loop (test infrastructure).

``LinearTaskTrainer`` differentiates a fixed linear functional of the fake-quantised tensors,

    task(step) = sum_i < q_i , c_i[step] >            q_i = floor(P_i / s_i) * s_i   (NQ-L:55-60)

instead of the model's loss: the upstream gradient that reaches every fake-quant op is then EXACTLY ``c_i[step]``
(d<q,c>/dq = 1 * c: one exact multiplication, no reduction), with no convolution / GEMM library between the
objective and the op.  Everything else of a training step is the product's own code, unchanged: ``quantize_all`` /
the per-tensor ops, the STE pass-through dP = dy (NQ-L:118), K2+K3, the data-parallel bucket and its exchange, exact
mode B's recompute from P.grad, penalty injection, the regularisers' gradients, both optimizers.

Why it exists: comparisons of whole training steps THROUGH MIOpen inherit its run-to-run noise (weight
gradients reduced with atomics; solver choice that depends on the process's history) and Adam turns an ulp of a gradient
into up to a whole step -- such tests can only carry tolerances on noise.  The equivalences the product actually
claims (mode B == single process, storage "oihw" == "hwio", batched == per-tensor, one-rank data parallel == plain)
are statements about THIS code, so they are asserted here bit for bit on injected gradients.
"""
import torch

from learned_quantization_amd.train import Trainer


def make_coefficients(trainer, steps, seed=1234, lo=-12.0, hi=-2.0):
    """Per step and custom layer a pair (c_kernel, c_bias) in the parameter's LOGICAL shape (Dense (in, out), conv HWIO),
    contiguous, generated on the CPU from ``seed`` (identical whatever the storage / process / rank): N(0,1) times a
    log-uniform magnitude 10^U(lo, hi), so that ratios |dy|/|out| straddle the thresholds in use (1e-11 .. 1e-3)."""
    g = torch.Generator().manual_seed(seed)
    dev = trainer.device
    out = []
    for _ in range(steps):
        per_layer = []
        for layer in trainer.custom_layers:
            cs = []
            for p in layer._regularized():          # (kernel|W, b) or (kernel,)
                shape = tuple(p.shape)
                c = torch.randn(shape, generator=g) * torch.pow(10.0, torch.empty(shape).uniform_(lo, hi, generator=g))
                cs.append(c.to(dev))
            per_layer.append(tuple(cs))
        out.append(per_layer)
    return out


class LinearTaskTrainer(Trainer):
    """``Trainer`` whose objective is the linear functional above (+ the regularisers, as ``Trainer.loss`` adds them).
    ``x`` and ``y`` of ``step`` are ignored.  ``coefficients`` = ``make_coefficients(...)``; call n of ``step`` uses entry
    ``n % len``."""

    def __init__(self, *a, **kw):
        if kw.get("graph"):
            raise ValueError("LinearTaskTrainer selects its coefficients on the host: eager steps only")
        super().__init__(*a, **kw)
        self.coefficients = None
        self._n = 0

    def _objective(self, x, y):
        coeffs = self.coefficients[self._n % len(self.coefficients)]
        self._n += 1
        total = None
        for layer, cs in zip(self.custom_layers, coeffs):
            w, qb = layer.quantized_parameters()            # the layer's own route to its fake-quantised tensors
            ck = cs[0]
            if w.dim() == 4:                                # conv: the layer hands out the OIHW-shaped tensor MIOpen would consume
                ck = ck.permute(3, 2, 0, 1)
            t = (w * ck).sum()
            if qb is not None:
                t = t + (qb * cs[1]).sum()
            total = t if total is None else total + t
        if self.loss_obj is not None and self.batch is None:
            # per-tensor path of the loss-term modes: the penalty goes through autograd as in Trainer.loss (CL-F:58); the
            # batched path injects its gradients after backward (Trainer._backward_phase)
            total = total + self.loss_obj.penalty_rate * self.loss_obj._penalty()
        return self._with_regularizers(total)


def snapshot(trainer):
    return {n: p.detach().clone() for n, p in trainer.model.named_parameters()}


def max_abs_diff(a, b):
    return max(float((a[k].double() - b[k].double()).abs().max()) for k in a)
