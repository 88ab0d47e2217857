"""End-to-end training harness for the BASELINE configs (SURVEY f-3): the thin PyTorch counterpart of the
reference's ``experiment.py`` drivers, on synthetic data (no dataset is reachable offline).

  python -m learned_quantization_amd.train --config cifar --mode nq --value 1e-11 --orientation channelwise \
        --batch 256 --steps 50 --warmup 10 [--loss maxbin] [--ddp-mode A|B] [--seed 42]
  (N GPUs: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... -m learned_quantization_amd.train ...)

Mirrors (reference file:line under /root/reference):
  seeds, lr 1e-4 Adam, batch sizes       CIFAR-10/nested_quantization_layer/experiment.py:435-443, 550-572
  CL mode ``loss=obj.compute_total_loss``  CIFAR-10/custom_loss_terms/experiment.py:436-465
  regulariser terms added to the loss     custom_layers.py:327 (Keras ``add_weight(regularizer=...)``)
  constraint after the optimizer step     custom_layers.py:158

Per step: forward (custom layers -> K1), loss, backward (-> K2+K3 / K5 backward), one bucketed all-reduce
(DataParallel, RCCL) when world_size > 1, torch Adam on the ordinary parameters, ScaleAdam (K6: Adam +
MinValueConstraint in one launch) on the learned scales.  Prints one JSON line with images/s.
"""
from __future__ import annotations

import argparse
import json
import os
import time
from typing import Optional

import torch
import torch.distributed as dist

from . import layers as L
from . import _hip
from .ddp import CaptureRefused, DataParallel, capture_with_agreement, control_group
from .losses import SCCEDifference, SCCEInverse, SCCEMaxBin, sparse_categorical_crossentropy
from .models import INPUT_SHAPES, build_model
from .optim import KerasAdam, ScaleAdam, non_scale_parameters, scale_parameters

LOSSES = {"maxbin": SCCEMaxBin, "difference": SCCEDifference, "inverse": SCCEInverse}


def synthetic_batch(config: str, batch: int, device, generator: Optional[torch.Generator] = None):
    """Images are raw 0..255 floats like the reference feeds them (experiment.py:512-513); uniform labels."""
    c, h, w = INPUT_SHAPES[config]
    x = torch.rand(batch, c, h, w, device=device, generator=generator) * 255.0
    y = torch.randint(0, 10, (batch,), device=device, generator=generator)
    return x, y


class Trainer:
    """One training configuration: model, loss, optimizers, optional multi-tensor batch, hipGraph and data-parallel wrapper.

    ``mode``: "nq" (nested quantization: ``value`` = penalty_threshold), "cl" (custom loss terms: ``value`` =
    penalty_rate, ``loss`` names the term), "nqcl" (BASELINE.json configs[3], "nested quantization + custom_loss_terms
    penalty": ``value`` = (penalty_threshold, penalty_rate); the scale gradient is the hand-written NQ gradient PLUS the
    penalty's -- an extension without a reference call site, the reference never combines the two: CL-L:61-62).

    A step is two phases: ``_backward_phase`` (zero the gradients, fake-quantise, forward, loss, backward, penalty
    injection) and ``_update_phase`` (exact-mode scale gradients, both optimizers); between them the data-parallel
    exchange.  ``graph=True`` records the phases into hipGraphs: one graph for the whole step on one GPU; with
    ``world_size > 1`` ONE graph that contains the RCCL all-reduce as well (RCCL kernels are capturable; the default on
    backend "nccl": 118.7 k against 109.6 k images/s for the CIFAR step on a one-rank communicator, profiles/r02_e2e.jsonl),
    or graph(backward) -> eager bucketed all-reduce -> graph(update) (``graph_collectives=False``, other backends, and the
    fallback when the capture of the collective fails).
    """

    def __init__(self, config="cifar", mode="nq", value=1e-11, orientation="channelwise", loss: Optional[str] = None,
                 lr=1e-4, seed=42, device=None, ddp_mode="A", log_dir="logs", graph=False, batched=False,
                 bucket_mb: float = 25.0, overlap: bool = True, graph_collectives: Optional[bool] = None,
                 force_collectives: bool = False, kernel_storage: str = "oihw"):
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        torch.manual_seed(seed)
        L.reset_layer_names()
        self.config = config
        self.mode = mode
        # conv kernels shaped HWIO like the reference's, stored in the order MIOpen consumes (layers.py kernel_storage): the
        # fake-quantised kernel goes to the convolution as written and its weight gradient is dP
        self.model = build_model(config, mode=mode, value=value, seed=seed, orientation=orientation, device=self.device,
                                 kernel_storage=kernel_storage)
        self.model.to(self.device)
        self.custom_layers = L.custom_layers_of(self.model)
        self.loss_obj = None
        self.loss_kind = loss
        self.penalty_rate = value[1] if mode == "nqcl" else value
        if mode in ("cl", "nqcl"):
            if loss not in LOSSES:
                raise ValueError(f"mode {mode!r} needs --loss maxbin|difference|inverse")
            self.loss_obj = LOSSES[loss](self.custom_layers, self.penalty_rate, log_dir)   # custom_loss_terms/experiment.py:436-455
        elif loss is not None:
            raise ValueError("a loss term needs mode 'cl' or 'nqcl'")
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        use_dp = self.world > 1 or (force_collectives and dist.is_initialized())
        # Exact mode B reads the all-reduced P.grad as the global-batch dy (custom_layers.py:118): everything else that has a
        # gradient with respect to P or s -- the loss term of "nqcl", weight regularisers -- is applied AFTER the scale
        # gradients were recomputed (it is a function of the parameters only, identical on every rank: no exchange needed).
        self._after_exchange = use_dp and ddp_mode == "B"
        if self._after_exchange and mode == "nqcl" and not batched:
            raise ValueError("mode 'nqcl' with ddp_mode 'B' needs batched=True: the loss term's gradients are injected after the "
                             "exact scale gradients were recomputed, which the per-tensor autograd path cannot order")
        # graphed steps exchange after backward (hooks cannot launch collectives from inside a capture that is replayed
        # without them): overlap is a property of the eager step
        self.dp = DataParallel(self.model, mode=ddp_mode, bucket_mb=bucket_mb, overlap=overlap and not graph,
                               force_collectives=force_collectives) if use_dp else None
        # Keras 2.11 Adam (lr 1e-4, eps 1e-7 outside the bias correction) for every ordinary parameter in one launch
        self.opt = KerasAdam(non_scale_parameters(self.model), lr=lr, eps=1e-7, capturable=graph)
        self.batch = None
        if batched:
            # one launch for every fake-quant forward, two for every scale gradient, one for every scale update
            from .batch import BatchedScaleAdam, FakeQuantBatch
            if self.dp is not None:
                self.dp.zero_grad()          # makes scale.grad the bucket views the batch will write into
            # the convolutions consume the OIHW companions only; autograd=False: the fake-quantised tensors are leaves and
            # _backward_phase calls finish_backward() itself (0.4-0.6 ms less autograd-engine work per eager step for 40 tensors)
            self.batch = FakeQuantBatch(self.model, lr=lr, hwio_out=False, autograd=False)
            # nothing touches ds between its computation and the scales' update when there is no loss term and ds is not
            # exchanged (one process, or exact mode B): the finalize then applies the Adam step itself (one launch fewer)
            fused = self.loss_obj is None and (self.dp is None or ddp_mode == "B")
            self.scale_opt = BatchedScaleAdam(self.batch, capturable=graph, fused=fused)
            if self.dp is not None:
                self.dp.attach_batch(self.batch)
        else:
            self.scale_opt = ScaleAdam(scale_parameters(self.model), lr=lr, capturable=graph)
        self.regularized = [l for l in self.custom_layers if l.regularizer is not None]
        self.graph = None
        self.graph_update = None
        if graph_collectives is None:          # default: capture the collective where the backend can be captured
            graph_collectives = bool(use_dp and dist.get_backend() == "nccl")
        self.graph_collectives = graph_collectives
        # host-side agreement on capture outcomes (collective creation: every rank builds its Trainer at the same point)
        self._control = control_group() if (use_dp and graph and graph_collectives) else None
        self.graph_note = None
        self._want_graph = graph

    def loss(self, y, p):
        if self.loss_obj is not None and self.batch is None:
            per_sample = self.loss_obj.compute_total_loss(y, p)
        else:
            per_sample = sparse_categorical_crossentropy(y, p)
        return self._with_regularizers(per_sample.mean())

    def _with_regularizers(self, total):
        """Keras adds the regulariser losses to the objective (custom_layers.py:327).  Exact mode B: their VALUE is added here,
        their gradients follow in ``_update_phase``, after the scale gradients were recomputed from the pure task-loss P.grad."""
        for layer in self.regularized:
            r = layer.regularization_loss()
            total = total + (r.detach() if self._after_exchange else r)
        return total

    def _objective(self, x, y):
        """The scalar that is differentiated: task loss of the model's output (+ loss term, + regularisers)."""
        return self.loss(y, self.model(x))

    # ---------------------------------------------------------------- the two phases of a step
    def _backward_phase(self, x, y):
        self.model.train()
        if self.dp is not None:
            self.dp.zero_grad()
        else:
            self.opt.zero_grad(set_to_none=True)
            self.scale_opt.zero_grad(set_to_none=True)
        if self.batch is not None:
            self.batch.quantize_all()
        loss = self._objective(x, y)
        loss.backward()
        if self.batch is not None:
            self.batch.finish_backward()         # scale-gradient launches; dP = dy to every quantised parameter
        if self.batch is not None and self.loss_obj is not None and not self._after_exchange:
            self._inject_penalty()
        return loss

    def _inject_penalty(self):
        # batched custom-loss-terms mode: the task loss went through autograd, the penalty gradients are injected
        # by the batch kernels (identical on every rank, so adding them before the all-reduce changes nothing);
        # "nqcl": the penalty's ds is ADDED to the nested-quantization ds the batch has just written
        self.batch.inject_penalty_grads(self.loss_kind, self.penalty_rate, accumulate_ds=(self.mode == "nqcl"))

    def _update_phase(self):
        if self.dp is not None:
            self.dp.recompute_scale_grads()          # mode B only: ds from the pure task-loss P.grad
        if self._after_exchange:
            if self.batch is not None and self.loss_obj is not None:
                self._inject_penalty()
            if self.regularized:
                with self.dp.no_sync():              # a second backward into the exchanged bucket, on purpose
                    reg = self.regularized[0].regularization_loss()
                    for layer in self.regularized[1:]:
                        reg = reg + layer.regularization_loss()
                    reg.backward()
        self.opt.step()
        self.scale_opt.step()

    def step(self, x, y):
        loss = self._backward_phase(x, y)
        if self.dp is not None:
            self.dp.exchange()
        self._update_phase()
        return loss

    def _capture(self, fn):
        g = torch.cuda.CUDAGraph()
        # with a process group alive its watchdog THREAD polls events while this thread captures: "thread_local" keeps
        # that legal (the default "global" mode makes every other thread's HIP call an error during the capture)
        mode = "thread_local" if self.dp is not None else "global"
        with torch.cuda.graph(g, capture_error_mode=mode):
            out = fn()
        return g, out

    def _capture_whole_step(self) -> bool:
        """ONE graph: backward phase, RCCL all-reduce (synchronous collectives on the capturing stream), update phase.
        The outcome is AGREED ON by all ranks (ddp.capture_with_agreement, over the gloo control group): True when every rank
        captured; False -- on every rank, with the trainer left ready for the split form -- when the stack refused to record the
        collective on at least one of them (an error raised while the exchange was being recorded, or when the capture was
        closed) and the communicator still answers an eager all-reduce afterwards.  Errors of the step itself (argument
        validation, shapes: anything raised in the backward or update phase, any ``LQError``) are re-raised, on every rank."""
        phase = ["backward"]

        def whole():
            loss = self._backward_phase(self._x, self._y)
            phase[0] = "exchange"
            self.dp.exchange(capture_safe=True)
            phase[0] = "update"
            self._update_phase()
            phase[0] = "end"                 # what is raised from here on comes from closing the capture
            return loss

        def attempt():
            try:
                self.graph, self._loss = self._capture(whole)
            except _hip.LQError:
                raise
            except RuntimeError as e:
                if phase[0] in ("exchange", "end"):
                    raise CaptureRefused(f"{e!r}"[:300]) from e
                raise

        try:
            ok = capture_with_agreement(attempt, self._control, health_check=self.dp.health_check)
        except BaseException:
            self.graph = None
            raise
        if ok:
            return True
        self.graph = None
        self.graph_collectives = False
        self.graph_note = "the collective could not be captured on every rank: graph / eager all-reduce / graph"
        torch.cuda.synchronize(self.device)
        self.dp.begin_step()                 # arrival counters of the hooks that ran during the abandoned capture
        if hasattr(self.scale_opt, "_applied"):
            self.scale_opt._applied = False  # the fused scale update was recorded, not executed
        return False

    def step_graphed(self, x, y):
        """The training step as hipGraph launches.  These steps are launch-bound (hundreds of small kernels); capture
        removes the host from the loop.  First call: 3 eager warm-up steps on a side stream, then capture.
        One GPU: ONE graph (forward, loss, backward, both optimizers).  Data parallel: graph(backward phase) -> eager
        bucketed RCCL all-reduce -> graph(update phase); or, with ``graph_collectives``, one graph including the
        all-reduce (synchronous collectives on the capturing stream)."""
        if self.graph is None:
            self._x = torch.empty_like(x)
            self._y = torch.empty_like(y)
            self._x.copy_(x)
            self._y.copy_(y)
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(3):
                    self.step(self._x, self._y)
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            if self.dp is None:
                self.opt.zero_grad(set_to_none=True)
                self.scale_opt.zero_grad(set_to_none=True)
                self.graph, self._loss = self._capture(lambda: self.step(self._x, self._y))
            elif self.graph_collectives and self._capture_whole_step():
                pass
            else:
                self.graph, self._loss = self._capture(lambda: self._backward_phase(self._x, self._y))
                self.dp.exchange()
                self.graph_update, _ = self._capture(self._update_phase)
        self._x.copy_(x)
        self._y.copy_(y)
        if self.dp is not None:
            self.dp.begin_step()
        self.graph.replay()
        if self.graph_update is not None:
            self.dp.exchange()
            self.graph_update.replay()
        return self._loss

    @torch.no_grad()
    def evaluate(self, x, y):
        self.model.eval()
        p = self.model(x)
        return float(sparse_categorical_crossentropy(y, p).mean()), float((p.argmax(1) == y).float().mean())


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=list(INPUT_SHAPES), default="cifar")
    ap.add_argument("--mode", choices=["nq", "cl", "nqcl"], default="nq")
    ap.add_argument("--value", type=float, default=1e-11, help="penalty_threshold (nq, nqcl) or penalty_rate (cl)")
    ap.add_argument("--rate", type=float, default=1e-7, help="penalty_rate of the loss term in mode nqcl")
    ap.add_argument("--value-coarse", type=float, default=None,
                    help="resnet50 only: threshold of the 3x3 kernels ('mixed' quantisation intensity); --value is the rest")
    ap.add_argument("--orientation", default="channelwise")
    ap.add_argument("--loss", choices=list(LOSSES), default=None)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--ddp-mode", choices=["A", "B"], default="A")
    ap.add_argument("--export-dir", default=None, help="write the reference's integer export here at the end")
    ap.add_argument("--graph", action="store_true", help="replay the step from hipGraphs (N > 1: graph / all-reduce / graph)")
    ap.add_argument("--graph-collectives", dest="graph_collectives", action="store_true", default=None,
                    help="N > 1: capture the RCCL all-reduce inside ONE graph (the default on backend nccl)")
    ap.add_argument("--no-graph-collectives", dest="graph_collectives", action="store_false",
                    help="N > 1: graph(backward) -> eager bucketed all-reduce -> graph(update)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed and run the collectives even with one rank (RCCL rehearsal on one GPU)")
    ap.add_argument("--bucket-mb", type=float, default=25.0)
    ap.add_argument("--backend", default="nccl", help="nccl = RCCL over xGMI (default); gloo + --share-gpu rehearses N>1 on one GPU")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--batched", action="store_true", help="multi-tensor launches for all fake-quant ops of a step (lq_batch_*)")
    ap.add_argument("--kernel-storage", choices=["oihw", "hwio"], default="oihw",
                    help="memory order of the conv kernels behind their HWIO shape (layers.py)")
    ap.add_argument("--channels-last", action="store_true",
                    help="feed NHWC-strided batches (torch.channels_last): MIOpen's fp32 igemm kernels are NHWC; measured "
                         "+17 %% on the ResNet-18-like config, -17 %% on the small CIFAR CNN")
    args = ap.parse_args(argv)

    # read by the HSA runtime when it initialises (the first torch.cuda call below): the host driver only supports dmabuf IPC
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("the training harness needs an MI355X (no CPU fallback)")
    if args.share_gpu:
        if args.backend == "nccl":
            raise SystemExit("--share-gpu needs --backend gloo (RCCL cannot put two ranks on one GPU)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if world == 1:
            for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29531"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(k, v)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    rank = dist.get_rank() if use_dist else 0

    value = args.value
    if args.mode == "nqcl":
        value = (args.value, args.rate)
    elif args.config == "resnet50" and args.value_coarse is not None:
        value = (args.value_coarse, args.value)
    tr = Trainer(args.config, args.mode, value, args.orientation, args.loss, seed=args.seed, device=dev,
                 ddp_mode=args.ddp_mode, graph=args.graph, batched=args.batched, bucket_mb=args.bucket_mb,
                 graph_collectives=args.graph_collectives, force_collectives=args.force_dist, kernel_storage=args.kernel_storage)
    do_step = tr.step_graphed if args.graph else tr.step
    g = torch.Generator(device=dev).manual_seed(args.seed + rank)
    batches = [synthetic_batch(args.config, args.batch, dev, g) for _ in range(4)]
    if args.channels_last and args.config != "mnist":
        batches = [(x.contiguous(memory_format=torch.channels_last), y) for x, y in batches]
    for i in range(args.warmup):
        do_step(*batches[i % 4])
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = do_step(*batches[i % 4])
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if rank == 0:
        n_q = sum(p.numel() for l in tr.custom_layers for p in l._regularized())
        print(json.dumps({
            "metric": "images/sec end-to-end training step (synthetic data)", "config": args.config, "mode": args.mode,
            "value": world * args.batch * args.steps / dt, "unit": "images/s", "n_gpus": world,
            "ms_per_step": dt / args.steps * 1e3, "per_gpu_batch": args.batch, "orientation": args.orientation,
            "loss_term": args.loss, "quantized_elements": n_q, "final_loss": float(loss.detach()), "ddp_mode": args.ddp_mode,
            "hipgraph": bool(args.graph), "graph_collectives": bool(tr.graph_collectives and args.graph and use_dist),
            **({"graph_note": tr.graph_note} if tr.graph_note else {}), "batched": bool(args.batched),
            "backend": (args.backend if use_dist else None),
            "channels_last": bool(args.channels_last), "kernel_storage": args.kernel_storage}))
        if args.export_dir:
            from .export import save_compress_parameters
            print(json.dumps(save_compress_parameters(tr.model, args.export_dir)))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
