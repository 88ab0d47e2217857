"""Multi-tensor batching of the per-step fake-quant work (SURVEY f-4) on top of ``lq_batch_*``.

A training step of the reference quantises every custom layer's kernel AND bias through its own
``my_custom_gradient`` call (custom_layers.py:263-268, 338-350): 4 / 12 / 40 tiny ops forward and as many
backward for the MNIST / CIFAR-10 / Imagenette models.  None of them depends on an activation, so

  * ``FakeQuantBatch.quantize_all()`` runs ALL forwards in one launch before the model forward and hands each
    layer its quantised tensors;
  * its autograd node receives ALL upstream gradients at the end of the backward pass and computes every scale
    gradient in two launches (traversal + finalize), writing ``scale.grad`` directly (no per-scale
    accumulate kernels) and passing ``dy`` through unchanged as ``∂P`` (STE, custom_layers.py:118);
  * ``BatchedScaleAdam.step()`` updates every scale (Adam + MinValueConstraint, custom_layers.py:158) in one launch.

Semantics per tensor are those of the single-tensor ops (same device code; results bit-identical).
Gradient accumulation across several backward passes is not supported in this mode (``scale.grad`` is
overwritten each backward), which matches the reference's one-backward-per-step training loop.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import torch

from . import _hip
from .descriptor import memory_descriptor
from .layers import CustomDenseLayer, _ConvBase, custom_layers_of


class _Entry:
    __slots__ = ("layer", "slot", "param", "nested", "out", "ds", "m", "v", "desc", "out_oihw", "dp", "conv", "shape", "pstride", "pdev",
                 "nq")


class FakeQuantBatch:
    def __init__(self, model_or_layers, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-7, mode: str = "keras",
                 oihw: bool = True, hwio_out: bool = True, autograd: bool = True):
        """``autograd=True``: ``quantize_all()`` is ONE autograd node (2 x tensors inputs, one output per tensor) and everything happens
        inside ``loss.backward()``.  Convenient -- and, for 40 tensors, 0.4-0.6 ms of autograd-engine work per step (an 80-input node,
        40 output wrappers, an AccumulateGrad per parameter that adds into the data-parallel bucket with a launch of its own).
        ``autograd=False`` (the trainer's form): the fake-quantised tensors are persistent LEAVES -- autograd only delivers their
        gradients (``leaf.grad``, stolen from the producer: no launch) -- and ``finish_backward()`` after ``loss.backward()`` runs
        the scale-gradient launches and hands ``dP = dy`` (custom_layers.py:118) to the parameters: by reference where a parameter
        has no gradient yet, with one fused add over all others (bucket views, regularised kernels).
        ``oihw``: conv kernels of nested-quantization layers get an OIHW companion output (what MIOpen consumes).
        ``hwio_out=False``: where the LDS-tile kernel writes that companion, the HWIO output is not materialised at all -- in a
        training step the convolution is the kernel's only consumer (custom_layers.py:340-348), so the forward moves 8 bytes
        per element instead of 12; ``quantize_all()`` then returns the HWIO tensor as a permuted VIEW of the companion."""
        layers = custom_layers_of(model_or_layers) if isinstance(model_or_layers, torch.nn.Module) else list(model_or_layers)
        self.layers = layers
        self.autograd = bool(autograd)
        self._external_grads = False
        self._fused_opt = None               # BatchedScaleAdam(fused=True): the scale-gradient finalize applies the Adam step itself
        self.defer_scale_grads = False       # exact data-parallel mode: backward skips ds, scale_grads_from_param_grads() follows
        self.entries: List[_Entry] = []
        for layer in layers:
            if isinstance(layer, _ConvBase):
                pairs = [(0, layer.kernel, layer.nested_q_k_layer)]
                if layer._has_bias:
                    pairs.append((1, layer.b, layer.nested_q_b_layer))
            elif isinstance(layer, CustomDenseLayer):
                pairs = [(0, layer.W, layer.nested_q_w_layer), (1, layer.b, layer.nested_q_b_layer)]
            else:
                raise TypeError(f"not a custom layer: {type(layer).__name__}")
            for slot, param, nested in pairs:
                e = _Entry()
                e.layer, e.slot, e.param, e.nested = layer, slot, param, nested
                _hip.require_device_f32(param.data, "parameter")
                _hip.require_device_f32(nested.scale.data, "scale")
                # a dense permutation of the logical axes (conv kernels stored OIHW, layers.py kernel_storage) is described in
                # memory order; outputs and gradients carry the parameter's strides
                e.desc = memory_descriptor(tuple(param.shape), param.data.stride(), tuple(nested.scale.shape))
                if e.desc is None:
                    raise ValueError("parameters must be dense (contiguous, or a permutation of a contiguous array)")
                e.out = torch.empty_like(param.data)
                e.shape = tuple(param.shape)
                e.pstride = tuple(param.data.stride())
                e.pdev = param.device
                e.nq = nested.penalty_threshold is not None
                # write straight into an existing gradient buffer (e.g. a DataParallel bucket view) when there is one
                g = nested.scale.grad
                if g is not None and g.is_contiguous():
                    e.ds = g                       # external buffer (bucket view): zeroed by its owner, never set to None
                    self._external_grads = True
                else:
                    e.ds = torch.zeros_like(nested.scale.data)
                e.m = torch.zeros_like(nested.scale.data)
                e.v = torch.zeros_like(nested.scale.data)
                # conv kernels of nested-quantization layers: the forward launch also emits the OIHW tensor MIOpen consumes and
                # the scale-gradient launch reads MIOpen's OIHW weight gradient, writing dP back in HWIO order (lq_hip.h:
                # lq_fq_forward_oihw) -- no transposition launches around the convolutions
                e.out_oihw = e.dp = e.conv = None
                if (oihw and slot == 0 and isinstance(layer, _ConvBase) and nested.penalty_threshold is not None
                        and param.data.is_contiguous()):          # HWIO-stored kernels only: an OIHW-stored one needs no companion
                    kh, kw, ci, co = (int(d) for d in param.shape)
                    e.conv = (kh * kw, ci, co)
                    e.out_oihw = torch.empty((co, ci, kh, kw), dtype=torch.float32, device=param.device)
                    e.dp = torch.empty_like(param.data)
                    if (not hwio_out and param.data_ptr() % 16 == 0
                            and _hip.load().lq_conv_tile_supported(kh * kw, ci, co, *e.desc) == 1):
                        e.out = None               # the companion is the only forward output of this tensor
                self.entries.append(e)
        if not self.entries:
            raise ValueError("no custom layers")
        self.device = self.entries[0].param.device
        lib = _hip.load()
        n = len(self.entries)
        arr = (_hip.TensorDesc * n)()
        for i, e in enumerate(self.entries):
            lam = e.nested.penalty_threshold
            c = getattr(e.nested.scale, "lq_constraint", None)
            arr[i] = _hip.TensorDesc(e.param.data_ptr(), e.nested.scale.data_ptr(), None, _hip.ptr(e.out), e.ds.data_ptr(),
                                     e.m.data_ptr(), e.v.data_ptr(), e.desc[0], e.desc[1], e.desc[2],
                                     float("nan") if lam is None else float(lam),
                                     float(c.min_value) if c is not None else float("-inf"),
                                     _hip.ptr(e.out_oihw), _hip.ptr(e.dp), *(e.conv or (0, 0, 0)))
        handle = ctypes.c_void_p()
        _hip.check(lib.lq_batch_create(arr, n, ctypes.byref(handle)), "lq_batch_create")
        self._handle = handle
        self._ptrs = (ctypes.c_void_p * n)()
        self.ws = torch.empty(lib.lq_batch_workspace_bytes(handle), dtype=torch.uint8, device=self.device)
        self.hyper = dict(lr=lr, betas=betas, eps=eps, mode=mode)
        self._data_ptrs = [(e.param.data_ptr(), e.nested.scale.data_ptr()) for e in self.entries]
        # per-step host work is kept to list lookups: the flat (param, scale, ...) argument list of the autograd node and,
        # per layer, which outputs are its kernel and its bias
        self._flat = [t for e in self.entries for t in (e.param, e.nested.scale)]
        slots = {}
        for i, e in enumerate(self.entries):
            slots.setdefault(id(e.layer), [e.layer, None, None])[1 + e.slot] = i
        self._layer_slots = list(slots.values())
        self._oihw_idx = [i for i, e in enumerate(self.entries) if e.out_oihw is not None]     # extra autograd outputs, in this order
        self._oihw_pos = {i: len(self.entries) + k for k, i in enumerate(self._oihw_idx)}
        if not self.autograd:
            # persistent leaves over the static output buffers (the forward launch rewrites the memory under them; autograd never
            # looks at their values, only routes gradients to them)
            self._leaf = [None if e.out is None else e.out.detach().requires_grad_(True) for e in self.entries]
            self._leaf_o = [None if e.out_oihw is None else e.out_oihw.detach().requires_grad_(True) for e in self.entries]
            self._all_leaves = [t for t in self._leaf + self._leaf_o if t is not None]
            self._last_ptrs = [0] * n
            # no companions at all (kernels stored OIHW, dense models): what quantize_all() hands to the layers never changes
            self._static_pre = self._static_outs = None
            if not self._oihw_idx and all(lf is not None for lf in self._leaf):
                self._static_outs = list(self._leaf)
                self._static_pre = [(layer.__dict__, (None if ik is None else self._leaf[ik], None if ib is None else self._leaf[ib], None))
                                    for layer, ik, ib in self._layer_slots]

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                _hip.load().lq_batch_destroy(h)
            except Exception:
                pass
            self._handle = None

    def _check_pointers(self):
        for e, (pp, sp) in zip(self.entries, self._data_ptrs):
            if e.param.data_ptr() != pp or e.nested.scale.data_ptr() != sp:
                raise RuntimeError("a parameter or scale was re-allocated after the batch was built "
                                   "(e.g. model.to(device)); rebuild the FakeQuantBatch")

    # ------------------------------------------------------------------ forward
    def quantize_all(self):
        """One launch: fake-quantise every kernel/bias; layers pick the results up in their next call."""
        self._check_pointers()
        if not self.autograd:
            return self._quantize_all_leaves()
        outs = _BatchFn.apply(self, *self._flat)
        n = len(self.entries)
        for layer, ik, ib in self._layer_slots:
            # plain instance attribute (not a Parameter/Module): written through __dict__, nn.Module.__setattr__ costs ~3 us
            layer.__dict__["_q_pre"] = (None if ik is None else outs[ik], None if ib is None else outs[ib],
                                        outs[self._oihw_pos[ik]] if ik in self._oihw_pos else None)
        return outs[:n]

    def _quantize_all_leaves(self):
        _hip.check(_hip.load().lq_batch_forward(self._handle, _hip.stream_ptr(self.device)), "lq_batch_forward")
        self._awaiting_finish = True
        for lf in self._all_leaves:
            lf.grad = None
        if self._static_pre is not None:                   # every tensor has its own HWIO-shaped leaf: the hand-outs never change
            for d, pre in self._static_pre:
                d["_q_pre"] = pre
            return self._static_outs
        outs = [lf if lf is not None else lo.permute(2, 3, 1, 0)      # the HWIO-shaped view of a companion-only kernel
                for lf, lo in zip(self._leaf, self._leaf_o)]
        for layer, ik, ib in self._layer_slots:
            layer.__dict__["_q_pre"] = (None if ik is None else outs[ik], None if ib is None else outs[ib],
                                        self._leaf_o[ik] if ik is not None else None)
        return outs

    def finish_backward(self):
        """``autograd=False``: call after ``loss.backward()``.  Reads the gradients autograd left on the fake-quantised leaves, runs
        the scale-gradient launches (unless ``defer_scale_grads``: exact data-parallel mode computes them after the exchange) and
        gives every parameter its ``dP = dy`` (custom_layers.py:118)."""
        if self.autograd:
            raise RuntimeError("finish_backward() belongs to FakeQuantBatch(autograd=False)")
        if not getattr(self, "_awaiting_finish", False):
            # a second call would add dP to the parameters' gradients (bucket views) once more
            raise RuntimeError("finish_backward() without a quantize_all() since the last call: one backward pass per forward")
        self._awaiting_finish = False
        if self._static_pre is not None and not self.defer_scale_grads:
            dps = self._leaf_grads_core()
        else:
            dys = [None if lf is None else lf.grad for lf in self._leaf] + [self._leaf_o[i].grad for i in self._oihw_idx]
            dps = self._backward_core(dys)
        add_to, add_from = [], []
        for e, dp in zip(self.entries, dps):
            p = e.param
            if p.grad is None:
                p.grad = dp                                # by reference: no launch (what AccumulateGrad does with a fresh gradient)
            else:
                add_to.append(p.grad)                      # a data-parallel bucket view, or the gradient a regulariser left there
                add_from.append(dp)
        if add_to:
            torch._foreach_add_(add_to, add_from)          # one fused launch for all of them

    def _leaf_grads_core(self):
        """The backward of every tensor when the upstream gradients are the leaves' own ``.grad`` (no companions): autograd's
        AccumulateGrad has already made them float32 tensors of the leaf's shape and strides on the leaf's device (its layout
        contract; assignment to ``.grad`` checks shape, dtype and device too), so of the per-tensor checks of ``_backward_core`` only
        the element order is looked at; the pointer table is rewritten only where an address changed (the caching allocator hands a steady-state step the same blocks again)."""
        ptrs, last, entries = self._ptrs, self._last_ptrs, self.entries
        dps = []
        for i, lf in enumerate(self._leaf):
            d = lf.grad
            if d is None:                                  # nothing consumed this tensor: dP = 0
                d = torch.zeros_like(entries[i].param.data)
            elif d.stride() != entries[i].pstride:         # a hand-assigned gradient in another element order (autograd's own obey the leaf's)
                d = _hip.require_device_f32(d, "dy", like=entries[i].param.data)
            a = d.data_ptr()
            if a != last[i]:
                ptrs[i] = a
                last[i] = a
            dps.append(d)
        ext = self._external_grads
        for e in self.entries:
            if e.nq:
                g = e.nested.scale
                if g.grad is not None and not ext:
                    raise RuntimeError("FakeQuantBatch: scale gradients must be None before backward "
                                       "(gradient accumulation over several backward passes is not supported in batched mode)")
        self._scale_grad_call(False)
        for e in self.entries:
            if e.nq:
                e.nested.scale.grad = e.ds                 # written in place by the kernel: no accumulate launch
        return dps

    def _scale_grad_call(self, oihw: bool):
        """lq_batch_scale_grad(_oihw), or -- a fused optimizer attached -- lq_batch_scale_grad_step: same launches, the finalize
        also applies the scales' Adam step."""
        lib = _hip.load()
        opt = self._fused_opt
        sp = _hip.stream_ptr(self.device)
        if opt is None:
            fn = lib.lq_batch_scale_grad_oihw if oihw else lib.lq_batch_scale_grad
            _hip.check(fn(self._handle, self._ptrs, _hip.ptr(self.ws), self.ws.numel(), sp), "lq_batch_scale_grad")
            return
        if opt._applied:
            # the finalize of a scale-gradient pass applies Adam: a second pass before step() (gradient accumulation, a retried
            # backward) would update the scales twice and the weights once
            raise RuntimeError("FakeQuantBatch: the fused scale update of this step has already been applied; call the optimizer's "
                               "step() first, or build BatchedScaleAdam(fused=False) (fused=True excludes gradient accumulation)")
        step, step_dev = opt._advance()
        h = self.hyper
        md = {"keras": _hip.LQ_ADAM_KERAS, "torch": _hip.LQ_ADAM_TORCH}[h["mode"]]
        _hip.check(lib.lq_batch_scale_grad_step(self._handle, self._ptrs, 1 if oihw else 0, _hip.ptr(self.ws), self.ws.numel(),
                                                h["lr"], h["betas"][0], h["betas"][1], h["eps"], int(step or 0), _hip.ptr(step_dev),
                                                md, sp), "lq_batch_scale_grad_step")
        opt._applied = True

    # ------------------------------------------------------------------ backward of every tensor (both modes)
    def _backward_core(self, dys):
        """``dys``: upstream gradient of every HWIO-shaped output, then of every OIHW companion (``None`` where nothing consumed
        it).  Launches the scale-gradient pass (unless deferred), sets ``scale.grad`` and returns ``dP`` per tensor."""
        batch = self
        n = len(batch.entries)
        # a conv kernel with an OIHW companion: its consumer (the convolution) used the companion, so the gradient arrives there,
        # in OIHW order; a consumer that used the HWIO output instead is served by the plain path below
        dys = list(dys)
        oihw_used = False
        for i in batch._oihw_idx:
            d_o = dys[batch._oihw_pos[i]]
            if d_o is not None:
                if dys[i] is not None:
                    raise RuntimeError("FakeQuantBatch: both the HWIO and the OIHW output of one conv kernel received a gradient")
                oihw_used = True
        if batch.defer_scale_grads:          # dP == dy (custom_layers.py:118); ds follows after the all-reduce
            dps = []
            for i, e in enumerate(batch.entries):
                d = dys[i]
                if d is None and i in batch._oihw_pos and dys[batch._oihw_pos[i]] is not None:
                    d = dys[batch._oihw_pos[i]].permute(2, 3, 1, 0)
                dps.append(d if d is not None else torch.zeros_like(e.param.data))
            return dps
        keep = []
        gathered = set()
        if not batch._oihw_idx:
            # no companions (kernels stored OIHW, dense models): the per-step host work is one pass of cheap checks -- this loop runs
            # every eager step for every tensor (tools/bench_weights.py us_per_step_batched_leaves)
            ptrs = batch._ptrs
            f32 = torch.float32
            for i, e in enumerate(batch.entries):
                d = dys[i]
                if d is None:
                    d = torch.zeros_like(e.param.data)
                elif d.dtype is not f32 or d.shape != e.shape or d.stride() != e.pstride or d.device != e.pdev:
                    d = _hip.require_device_f32(d, "dy", like=e.param.data)      # the full checks, a relayout where needed
                if e.nq and e.nested.scale.grad is not None and not batch._external_grads:
                    raise RuntimeError("FakeQuantBatch: scale gradients must be None before backward "
                                       "(gradient accumulation over several backward passes is not supported in batched mode)")
                keep.append(d)
                ptrs[i] = d.data_ptr()
            batch._scale_grad_call(False)
            for e in batch.entries:
                if e.nq:
                    e.nested.scale.grad = e.ds                     # written in place by the kernel: no accumulate launch
            return keep
        for i, e in enumerate(batch.entries):
            d = dys[i]
            if d is None and i in batch._oihw_pos and dys[batch._oihw_pos[i]] is not None:
                d = dys[batch._oihw_pos[i]]
                gathered.add(i)
            elif d is None:
                d = torch.zeros_like(e.param.data)
            elif oihw_used and i in batch._oihw_pos:
                raise RuntimeError("FakeQuantBatch: conv kernels must all be consumed through the same layout in one step")
            d = _hip.require_device_f32(d, "dy", like=None if i in gathered else e.param.data)
            keep.append(d)
            batch._ptrs[i] = d.data_ptr()
        for e in batch.entries:
            if e.nested.penalty_threshold is not None and e.nested.scale.grad is not None and not batch._external_grads:
                # the kernel OVERWRITES its gradient buffer: a second backward without zero_grad(set_to_none=True)
                # would silently drop the first gradient -- refuse instead (before anything is launched)
                raise RuntimeError("FakeQuantBatch: scale gradients must be None before backward "
                                   "(gradient accumulation over several backward passes is not supported in batched mode)")
        batch._scale_grad_call(oihw_used)
        dps = []
        for i, (e, d) in enumerate(zip(batch.entries, keep)):
            dps.append(e.dp if i in gathered else d)               # dP is dy itself (custom_layers.py:118), in HWIO order
            if e.nested.penalty_threshold is not None:
                e.nested.scale.grad = e.ds                         # written in place by the kernel: no accumulate launch
            # STE-only: zeros_like(scale) (CL custom_layers.py:62) -- no scale gradient from the op
        assert len(dps) == n
        return dps

    # ------------------------------------------------------------------ exact data-parallel mode (ddp.py, mode B)
    def scale_grads_from_param_grads(self):
        """ds of every nested-quantization tensor from its parameter's CURRENT gradient, in two launches.  After the
        data-parallel all-reduce ``P.grad`` is the global-batch dy (dP == dy, custom_layers.py:118), so this yields the
        single-device large-batch scale gradient, identical on every rank."""
        lib = _hip.load()
        keep = []
        for i, e in enumerate(self.entries):
            g = e.param.grad
            if g is None:
                if e.nested.penalty_threshold is not None:
                    raise RuntimeError("scale_grads_from_param_grads: a quantised parameter has no gradient")
                self._ptrs[i] = None
                continue
            g = _hip.require_device_f32(g, "parameter gradient", like=e.param.data)
            keep.append(g)
            self._ptrs[i] = g.data_ptr()
        self._scale_grad_call(False)
        for e in self.entries:
            if e.nested.penalty_threshold is not None:
                e.nested.scale.grad = e.ds

    # ------------------------------------------------------------------ custom loss terms
    _KINDS = {"maxbin": 0, "difference": 1, "inverse": 2}

    def inject_penalty_grads(self, kind: str, penalty_rate: float, accumulate_ds: bool = False):
        """Adds d(penalty_rate * penalty)/dP to every ``P.grad`` and writes d(...)/ds to every ``scale.grad`` in 2-4
        launches.  Call after ``loss.backward()`` of the task loss alone: the reference's objective is
        ``mean(SCCE) + penalty_rate * penalty`` (custom_loss_functions.py:58), so the coefficient of tensor i's term
        ``mean(...)_i`` is the constant ``penalty_rate * numel_i / sum(numel)`` (:110-116) and no autograd node per tensor
        is needed.  Equivalent to differentiating ``SCCE*.compute_total_loss`` (tests/test_gpu_batch.py).
        ``accumulate_ds``: add the penalty's scale gradient to what the ds buffers already hold (nested-quantization layers
        trained with a loss term: LQ_PENALTY_ACCUMULATE_DS)."""
        lib = _hip.load()
        n = len(self.entries)
        normalizer = float(sum(e.param.numel() for e in self.entries))
        coeff = (ctypes.c_float * n)(*[float(penalty_rate) * e.param.numel() / normalizer for e in self.entries])
        grads = (ctypes.c_void_p * n)()
        for i, e in enumerate(self.entries):
            if kind != "inverse":
                g = e.param.grad
                if g is None:
                    g = e.param.grad = torch.zeros_like(e.param.data)
                if not _hip.same_layout(g, e.param.data):       # a hand-assigned gradient: bring it into the parameter's element order
                    g = e.param.grad = torch.empty_like(e.param.data).copy_(g)
                grads[i] = g.data_ptr()
        kind_flag = self._KINDS[kind] | (_hip.LQ_PENALTY_ACCUMULATE_DS if accumulate_ds else 0)
        _hip.check(lib.lq_batch_penalty_grads(self._handle, kind_flag, coeff, grads, _hip.ptr(self.ws), self.ws.numel(),
                                              _hip.stream_ptr(self.device)), "lq_batch_penalty_grads")
        for e in self.entries:
            e.nested.scale.grad = e.ds

    # ------------------------------------------------------------------ optimizer
    def scale_adam_step(self, step: Optional[int] = None, step_dev: Optional[torch.Tensor] = None):
        lib = _hip.load()
        h = self.hyper
        md = {"keras": _hip.LQ_ADAM_KERAS, "torch": _hip.LQ_ADAM_TORCH}[h["mode"]]
        _hip.check(lib.lq_batch_scale_adam(self._handle, h["lr"], h["betas"][0], h["betas"][1], h["eps"],
                                           int(step or 0), _hip.ptr(step_dev), md, _hip.stream_ptr(self.device)),
                   "lq_batch_scale_adam")


class _BatchFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, batch: FakeQuantBatch, *tensors):
        lib = _hip.load()
        _hip.check(lib.lq_batch_forward(batch._handle, _hip.stream_ptr(batch.device)), "lq_batch_forward")
        ctx.batch = batch
        ctx.set_materialize_grads(False)        # an output nobody consumed arrives as None in backward, not as a zero tensor
        # fresh tensor objects over the static buffers: every HWIO output, then the OIHW companions of the conv kernels
        # (a tensor without a materialised HWIO output hands out the HWIO-shaped permuted view of its companion)
        return tuple(e.out.detach() if e.out is not None else e.out_oihw.detach().permute(2, 3, 1, 0) for e in batch.entries) \
            + tuple(batch.entries[i].out_oihw.detach() for i in batch._oihw_idx)

    @staticmethod
    def backward(ctx, *dys):
        batch: FakeQuantBatch = ctx.batch
        dps = batch._backward_core(dys)
        grads = [None]
        for dp in dps:
            grads.extend((dp, None))       # (dP, d scale): the scale gradient is written in place by the kernel, never through autograd
        return tuple(grads)


class BatchedScaleAdam:
    """Optimizer facade over ``FakeQuantBatch.scale_adam_step`` (K6 for every scale in one launch)."""

    def __init__(self, batch: FakeQuantBatch, capturable: bool = False, fused: bool = False):
        """``fused``: the batch's scale-gradient finalize applies this optimizer's step in the same launch
        (lq_batch_scale_grad_step) and ``step()`` only acknowledges it.  For training steps in which nothing reads or changes the
        scale gradients between backward and ``step()`` -- no loss term, no exchange of ds (single process, or exact data-parallel
        mode B where ds is computed after the exchange); every tensor must be a nested-quantization one.  Exactly ONE
        scale-gradient pass per ``step()``: a second one raises (it would apply Adam to the scales twice) -- no gradient accumulation."""
        self.batch = batch
        self.capturable = capturable
        self._applied = False
        if fused:
            if any(e.nested.penalty_threshold is None for e in batch.entries):
                raise ValueError("a fused scale update needs nested-quantization layers throughout (every scale gets its gradient "
                                 "from the batch's own scale-gradient pass)")
            if batch._external_grads:
                raise ValueError("a fused scale update cannot be combined with gradient buffers that are exchanged (bucket views)")
            batch._fused_opt = self
        self._step = 0
        self._step_t = torch.zeros(1, dtype=torch.int64, device=batch.device) if capturable else None
        self.param_groups = [{"params": [e.nested.scale for e in batch.entries]}]

    def zero_grad(self, set_to_none: bool = True):
        for e in self.batch.entries:
            e.nested.scale.grad = None

    def _advance(self):
        """(host step, device step tensor) of the update that is about to be applied."""
        if self.capturable:
            self._step_t += 1
            return None, self._step_t
        self._step += 1
        return self._step, None

    @torch.no_grad()
    def step(self):
        if self.batch._fused_opt is self:
            if not self._applied:
                raise RuntimeError("BatchedScaleAdam(fused=True).step() without a backward pass of the batch since the last step")
            self._applied = False            # the finalize of this step's scale-gradient pass has already updated the scales
            return
        # only scales that actually received a gradient this step are updated by Adam in the reference too; with the
        # nested-quantization op every scale does.  STE-only scales are updated from the loss-term gradients.
        for e in self.batch.entries:
            g = e.nested.scale.grad
            if g is None:
                e.ds.zero_()
            elif g.data_ptr() != e.ds.data_ptr():
                e.ds.copy_(g)                                      # loss-term gradients arrive in autograd-owned tensors
        step, step_dev = self._advance()
        self.batch.scale_adam_step(step=step, step_dev=step_dev)
