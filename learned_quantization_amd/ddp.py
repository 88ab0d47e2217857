"""Data-parallel exchange step: one process per GPU, RCCL over xGMI via torch.distributed.

The reference has no distribution of any kind; this is new design.
Every rank holds a full replica (parameters + scales) and a shard of the minibatch.  Per step
there is ONE all-reduce over one flat fp32 bucket (model gradients || learned-scale gradients,
2.2 MB for the CIFAR CNN, 44.7 MB for the ResNet-18-like net); scale gradients are a few KB and
ride in the same bucket instead of paying their own latency-bound collective.

The nested-quantization scale gradient is a NON-linear function of dy (thresholding,
custom_layers.py:97-111), so two modes exist:

  mode "A" (default, the north-star wording): each rank computes ds from its local dy; ds is
      averaged with everything else.  Differs from a single-device run at the global batch.
  mode "B" (exact): because dP == dy exactly (custom_layers.py:118), the averaged P.grad IS the
      global-batch dy.  ds is recomputed AFTER the all-reduce from (P, s, P.grad) and is not
      communicated: bit-identical on all ranks and equal to the single-device large-batch result.
      Weight regularisers must stay out of P.grad until then (apply them in the optimizer).

The exchange is bucketed (25 MiB sub-buckets, last layers first) and launched from gradient hooks so that it
overlaps the remaining backward; xGMI is point-to-point (7 links x ~153 GB/s per GPU), a ring all-reduce is bound
by one link (44.7 MB -> ~0.5 ms), so overlapping it matters more than on a switched fabric.

Works with any backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" for the CPU tests of the
bucket logic.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .descriptor import memory_order


# ---------------------------------------------------------------------------------------------------------------------
# Collective control decisions.  Which FORM of the step a rank runs (one hipGraph holding the RCCL all-reduce, or graph / eager
# all-reduce / graph, or plain eager launches) decides the SEQUENCE of collectives it issues; if one rank fell back on its own
# because its capture failed while the others kept the captured form, the ranks would wait for each other forever.  So the
# outcome of every capture attempt is agreed on over a host-side (gloo) group before anybody proceeds: all ranks keep their
# graphs, or all drop them, or -- on an error that is not a refused capture -- all raise.
CAPTURED, REFUSED, FATAL = 1, 0, -1


class CaptureRefused(RuntimeError):
    """The stack refused to record the collective into a hipGraph (raised by the attempt passed to ``capture_with_agreement``)."""


_control_groups = {}


def control_group(group=None):
    """A gloo group over the ranks of ``group`` for host-side agreement (None while torch.distributed is not initialised; the
    group itself when it already is gloo).  COLLECTIVE on first use: every rank must call it at the same point of the program
    (Trainer.__init__, bench.py right after init_process_group)."""
    if not dist.is_initialized():
        return None
    if dist.get_backend(group) == "gloo":
        return group if group is not None else dist.group.WORLD
    key = id(group) if group is not None else None
    if key not in _control_groups:
        import os
        if os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: the host name of a container need not resolve
        ranks = dist.get_process_group_ranks(group) if group is not None else None
        try:
            _control_groups[key] = dist.new_group(ranks=ranks, backend="gloo")
        except Exception:                   # no usable gloo transport (the same on every rank of a node): agree over the product group
            _control_groups[key] = _PRODUCT_GROUP
    return _control_groups[key]


_PRODUCT_GROUP = "product-group"      # control_group() could not build a gloo group: decisions travel over the default (RCCL) group


def agree(code: int, ctl_group) -> int:
    """MIN over the ranks of an integer decision code: every rank returns the same value."""
    if ctl_group is None:
        return code
    if ctl_group is _PRODUCT_GROUP:
        t = torch.tensor([int(code)], dtype=torch.int64, device=torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())
    t = torch.tensor([int(code)], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=ctl_group)
    return int(t.item())


def capture_with_agreement(attempt: Callable[[], None], ctl_group, health_check: Optional[Callable[[], None]] = None) -> bool:
    """Runs ``attempt()`` (this rank's capture) and agrees with the other ranks on what happened.

    True: every rank captured -- keep the graphs.  False: at least one rank's capture was refused (``CaptureRefused``) -- EVERY rank
    must drop whatever it captured and take the fallback form; before returning False ``health_check()`` (an eager collective on
    the product communicator with a bounded wait) must pass on every rank, else all raise: a communicator that a failed capture left
    unusable is a reason to stop with a message, not to hang in the next all-reduce.  Any other exception in ``attempt`` is an error
    of the step itself (argument validation, shapes): it is re-raised on the rank that saw it and the other ranks raise too."""
    err = None
    try:
        attempt()
        code = CAPTURED
    except CaptureRefused as e:
        code, err = REFUSED, e
    except Exception as e:                  # noqa: BLE001 -- reported to the other ranks first, then re-raised
        code, err = FATAL, e
    agreed = agree(code, ctl_group)
    if code == FATAL:
        raise err
    if agreed == FATAL:
        raise RuntimeError("capture_with_agreement: another rank failed while recording the step (not a refused capture); stopping "
                           "on every rank")
    if agreed == CAPTURED:
        return True
    ok = CAPTURED
    herr = None
    if health_check is not None:
        try:
            health_check()
        except Exception as e:              # noqa: BLE001
            ok, herr = FATAL, e
    if agree(ok, ctl_group) != CAPTURED:
        raise RuntimeError("the collective could not be recorded into a hipGraph on every rank and the communicator does not answer "
                           "afterwards; rerun with --no-graph-collectives (train) / --exchange sync (bench)") from (herr or err)
    return False


class GradBucket:
    """Flat fp32 bucket over a fixed parameter list; grads become views into it.  Every view starts on a 256-byte
    boundary (the kernels' float4 paths need 16-byte aligned gradient buffers, lq_hip.h; one-element scales would
    otherwise shift everything behind them); the padding stays zero and rides through the all-reduce."""

    ALIGN = 64          # floats

    def __init__(self, params: Sequence[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.sizes = [p.numel() for p in self.params]
        self.offsets = []
        off = 0
        for n in self.sizes:
            self.offsets.append(off)
            off += -(-n // self.ALIGN) * self.ALIGN
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.views = []
        for p, n, o in zip(self.params, self.sizes, self.offsets):
            # the view carries the parameter's own strides (a conv kernel shaped HWIO and stored OIHW keeps that order in the
            # bucket): gradient and parameter are then read by the kernels with one descriptor
            v = self.flat[o:o + n].view_as(p) if p.is_contiguous() else torch.as_strided(self.flat, p.shape, p.stride(), o)
            self.views.append(v)
            p.grad = v            # autograd accumulates in place into the bucket

    def zero_(self):
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is not v:   # an optimizer's set_to_none=True replaced it
                p.grad = v

    def gather_(self):
        """Copies grads that were re-allocated elsewhere back into the bucket."""
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v

    def all_reduce_mean_(self, group=None, async_op: bool = False):
        world = dist.get_world_size(group)
        if world == 1:
            return None
        # pre-divide: sum of (g / N) -- keeps magnitudes bounded and is what DDP does
        self.flat.div_(world)
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def _avg_op(group=None):
    """RCCL averages inside the collective (ReduceOp.AVG); gloo has no AVG: pre-divide and SUM there."""
    try:
        return dist.ReduceOp.AVG if dist.get_backend(group) == "nccl" else None
    except Exception:
        return None


class DataParallel:
    """Wraps a module for data-parallel training.

    usage per step:  dp.zero_grad(); loss.backward(); dp.sync_gradients(); optimizer.step()

    The flat bucket is cut into sub-buckets of ``bucket_mb`` MiB in reverse parameter order; a sub-bucket's
    all-reduce is launched (asynchronously, RCCL's own stream) from a post-accumulate-grad hook as soon as every
    parameter in it has its gradient, so the exchange of the late layers overlaps the backward of the early ones.
    ``sync_gradients`` launches whatever is left and waits.  ``overlap=False`` = one all-reduce per sub-bucket after
    backward.  On RCCL the mean is taken by the collective itself (``ReduceOp.AVG``: no extra launch per bucket).

    One backward per ``zero_grad()`` -- the reference's training loop.  For gradient accumulation run the earlier
    backward passes under ``with dp.no_sync():`` (gradients accumulate in the bucket, nothing is exchanged); a second
    backward that would hit an already exchanged bucket raises instead of silently leaving the replicas diverged.

    With the multi-tensor batch (``FakeQuantBatch``) every dP is handed over at the very end of the backward pass -- by ONE
    autograd node, or (``autograd=False``) by ``finish_backward()`` after it -- so buckets that hold quantised parameters or
    scales complete last and nothing of their exchange overlaps (``attach_batch`` holds them for ``exchange()``); buckets of
    the ordinary layers (BN, plain Dense) still do.

    mode "B": the nested layers are told to skip their local scale gradient in backward (it would be discarded);
    ``sync_gradients`` recomputes every ds from the all-reduced P.grad -- through ``batch.scale_grads_from_param_grads()``
    (two launches for all tensors) when a ``FakeQuantBatch`` is attached with ``attach_batch``, else tensor by tensor.

    ``force_collectives=True`` runs the collectives even in a one-rank group (rehearsal of the RCCL path on one GPU).
    """

    def __init__(self, module: torch.nn.Module, mode: str = "A", group=None,
                 scale_grad_fn: Optional[Callable] = None, broadcast: bool = True, bucket_mb: float = 25.0,
                 overlap: bool = True, force_collectives: bool = False):
        if mode not in ("A", "B"):
            raise ValueError("mode must be 'A' or 'B'")
        self.module = module
        self.mode = mode
        self.group = group
        self._scale_grad_fn = scale_grad_fn
        self._batch = None
        params = [p for p in module.parameters() if p.requires_grad]
        self.scales = [p for p in params if getattr(p, "lq_is_scale", False)]
        self.others = [p for p in params if not getattr(p, "lq_is_scale", False)]
        # mode B never communicates ds: scales stay outside the bucket
        self.bucket = GradBucket(params if mode == "A" else self.others)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._collectives = dist.is_initialized() and (self.world > 1 or force_collectives)
        self._avg = _avg_op(group) if self._collectives else None
        if broadcast and dist.is_initialized() and self._collectives:
            for t in list(module.parameters()) + list(module.buffers()):
                d = t.data
                if not d.is_contiguous():     # a dense permutation (conv kernels stored OIHW): the contiguous view of the same memory
                    order = memory_order(d.shape, d.stride())
                    d = d.permute(order) if order is not None else d
                dist.broadcast(d, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if mode == "B":
            for _, nested in self._nq_pairs():
                nested.defer_scale_grad = True        # backward skips the local K2+K3 launch (ops._NestedQuantFn)
        # ---- sub-buckets (contiguous ranges of the flat buffer), last parameters first
        self._ranges: List[Tuple[int, int]] = []
        self._param_bucket = {}
        self._pending_count: List[int] = []
        self._handles: List = []
        self._launched: List[bool] = []
        self.overlap = overlap and self._collectives
        self._hooks_on = True
        self._synced = False
        self._hold = set()                # sub-buckets that only exchange() may launch (attach_batch)
        limit = max(int(bucket_mb * (1 << 20) / 4), 1)
        offsets = self.bucket.offsets
        hi = self.bucket.flat.numel()
        cur_lo, cur_params = hi, []
        for idx in range(len(self.bucket.params) - 1, -1, -1):
            cur_lo = offsets[idx]
            cur_params.append(idx)
            if hi - cur_lo >= limit or idx == 0:
                b = len(self._ranges)
                self._ranges.append((cur_lo, hi))
                for i in cur_params:
                    self._param_bucket[i] = b
                self._pending_count.append(len(cur_params))
                hi, cur_params = cur_lo, []
        self._remaining = list(self._pending_count)
        self._launched = [False] * len(self._ranges)
        # the hooks are registered in every configuration: with overlap they launch a sub-bucket's all-reduce as soon as it is
        # complete; without (graphed steps, overlap=False) they still count arrivals, so a second backward into an already
        # exchanged bucket raises in every mode
        for i, p in enumerate(self.bucket.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))

    def __call__(self, *a, **k):
        return self.module(*a, **k)

    def attach_batch(self, batch) -> None:
        """Mode B with a FakeQuantBatch: ds of every tensor is recomputed by the batch (two launches)."""
        self._batch = batch
        if self.mode == "B":
            batch.defer_scale_grads = True
        if not getattr(batch, "autograd", True):
            # FakeQuantBatch(autograd=False) hands dP to its parameters in finish_backward(), after loss.backward() has returned: a
            # sub-bucket that holds one of them must not be exchanged from a hook -- a regularised kernel's hook fires for the
            # regulariser's gradient alone -- but by exchange(), which the step calls after finish_backward()
            managed = {id(e.param) for e in batch.entries} | {id(e.nested.scale) for e in batch.entries}
            self._hold = {self._param_bucket[i] for i, p in enumerate(self.bucket.params) if id(p) in managed}

    def no_sync(self):
        """Context manager: backward passes inside accumulate into the bucket without exchanging anything."""
        dp = self

        class _NoSync:
            def __enter__(self_inner):
                dp._hooks_on = False

            def __exit__(self_inner, *exc):
                dp._hooks_on = True
                return False
        return _NoSync()

    def _make_hook(self, i):
        def hook(_param):
            if not self._hooks_on:
                return
            b = self._param_bucket[i]
            if self._launched[b] or self._synced or self._remaining[b] <= 0:
                raise RuntimeError("DataParallel: a gradient arrived for a bucket that was already exchanged -- a second "
                                   "backward without dp.zero_grad().  For gradient accumulation run the earlier backward "
                                   "passes under `with dp.no_sync():`")
            self._remaining[b] -= 1
            if self._remaining[b] == 0 and self.overlap and b not in self._hold:
                self._launch(b)
        return hook

    def _launch(self, b, async_op: bool = True):
        lo, hi = self._ranges[b]
        for i, p in enumerate(self.bucket.params):        # a grad re-allocated elsewhere goes back into the bucket first
            if self._param_bucket[i] == b and p.grad is not None and p.grad.data_ptr() != self.bucket.views[i].data_ptr():
                self.bucket.views[i].copy_(p.grad)
                p.grad = self.bucket.views[i]
        chunk = self.bucket.flat[lo:hi]
        if self._avg is not None:
            h = dist.all_reduce(chunk, op=self._avg, group=self.group, async_op=async_op)
        else:
            chunk.div_(self.world)
            h = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            self._handles.append(h)
        self._launched[b] = True

    def begin_step(self):
        """Host-side reset of the per-step state (what zero_grad() does besides zeroing the bucket); a captured step
        replays the zeroing kernel itself and calls this before every replay."""
        self._remaining = list(self._pending_count)
        self._launched = [False] * len(self._ranges)
        self._handles = []
        self._synced = False

    def zero_grad(self):
        self.bucket.zero_()
        self.begin_step()
        if self.mode == "B":
            for s in self.scales:
                s.grad = None

    def _nq_pairs(self) -> List[Tuple[torch.nn.Parameter, torch.nn.Module]]:
        """(parameter, nested layer) pairs of every nested-quantization op in the module."""
        pairs = []
        for m in self.module.modules():
            for pname, lname in (("W", "nested_q_w_layer"), ("kernel", "nested_q_k_layer"), ("b", "nested_q_b_layer")):
                if hasattr(m, pname) and hasattr(m, lname):
                    nested = getattr(m, lname)
                    if getattr(nested, "penalty_threshold", None) is not None and nested.scale is not None:
                        pairs.append((getattr(m, pname), nested))
        return pairs

    def exchange(self, capture_safe: bool = False):
        """The all-reduce part of ``sync_gradients``.  ``capture_safe``: synchronous collectives on the current stream and
        no host-side gathering (for recording the step into a hipGraph: gradients are the static bucket views then)."""
        if self._collectives:
            if not capture_safe:
                self.bucket.gather_()
            for b in range(len(self._ranges)):
                if not self._launched[b]:                  # parameters that got no gradient this step, or overlap=False
                    self._launch(b, async_op=not capture_safe)
            for h in self._handles:
                h.wait()
            self._handles = []
        self._synced = True

    def health_check(self, seconds: float = 60.0):
        """One eager all-reduce on the product communicator with a bounded wait (after a refused capture)."""
        if not self._collectives:
            return
        import datetime
        t = torch.ones(1, device=self.bucket.flat.device)
        h = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        h.wait(timeout=datetime.timedelta(seconds=seconds))
        if self.bucket.flat.is_cuda:
            torch.cuda.synchronize(self.bucket.flat.device)
        if int(t.item()) != self.world:
            raise RuntimeError(f"health check: all-reduce over {self.world} ranks returned {float(t)}")

    def recompute_scale_grads(self):
        """Mode B: ds from the all-reduced P.grad, which IS the global-batch dy (dP == dy, custom_layers.py:118)."""
        if self.mode != "B":
            return
        if self._batch is not None and self._scale_grad_fn is None:
            self._batch.scale_grads_from_param_grads()
            return
        fn = self._scale_grad_fn
        if fn is None:
            from . import ops
            fn = ops.fq_scale_grad
        for p, nested in self._nq_pairs():
            nested.scale.grad = fn(p.data, nested.scale.data, p.grad, nested.penalty_threshold)

    def sync_gradients(self):
        self.exchange()
        self.recompute_scale_grads()
