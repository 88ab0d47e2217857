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
from .ddp import DataParallel
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
    def __init__(self, config="cifar", mode="nq", value=1e-11, orientation="channelwise", loss: Optional[str] = None,
                 lr=1e-4, seed=42, device=None, ddp_mode="A", log_dir="logs", graph=False, batched=False):
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        torch.manual_seed(seed)
        L.reset_layer_names()
        self.config = config
        self.model = build_model(config, mode=mode, value=value, seed=seed, orientation=orientation, device=self.device)
        self.model.to(self.device)
        self.custom_layers = L.custom_layers_of(self.model)
        self.loss_obj = None
        self.loss_kind, self.penalty_rate = loss, value
        if mode == "cl":
            if loss not in LOSSES:
                raise ValueError("mode 'cl' needs --loss maxbin|difference|inverse")
            self.loss_obj = LOSSES[loss](self.custom_layers, value, log_dir)      # custom_loss_terms/experiment.py:436-455
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.dp = DataParallel(self.model, mode=ddp_mode) if self.world > 1 else None
        if graph and self.world > 1:
            raise ValueError("graph capture of the whole step is single-GPU only (the all-reduce stays eager)")
        # Keras 2.11 Adam (lr 1e-4, eps 1e-7 outside the bias correction) for every ordinary parameter in one launch
        self.opt = KerasAdam(non_scale_parameters(self.model), lr=lr, eps=1e-7, capturable=graph)
        self.batch = None
        if batched:
            # one launch for every fake-quant forward, two for every scale gradient, one for every scale update
            from .batch import BatchedScaleAdam, FakeQuantBatch
            if self.dp is not None:
                self.dp.zero_grad()          # makes scale.grad the bucket views the batch will write into
            self.batch = FakeQuantBatch(self.model, lr=lr)
            self.scale_opt = BatchedScaleAdam(self.batch, capturable=graph)
        else:
            self.scale_opt = ScaleAdam(scale_parameters(self.model), lr=lr, capturable=graph)
        self.regularized = [l for l in self.custom_layers if l.regularizer is not None]
        self.graph = None
        self._want_graph = graph

    def loss(self, y, p):
        if self.loss_obj is not None and self.batch is None:
            per_sample = self.loss_obj.compute_total_loss(y, p)
        else:
            per_sample = sparse_categorical_crossentropy(y, p)
        total = per_sample.mean()
        for layer in self.regularized:                     # Keras adds regulariser losses to the objective
            total = total + layer.regularization_loss()
        return total

    def step(self, x, y):
        self.model.train()
        if self.dp is not None:
            self.dp.zero_grad()
        else:
            self.opt.zero_grad(set_to_none=True)
            self.scale_opt.zero_grad(set_to_none=True)
        if self.batch is not None:
            self.batch.quantize_all()
        loss = self.loss(y, self.model(x))
        loss.backward()
        if self.batch is not None and self.loss_obj is not None:
            # batched custom-loss-terms mode: the task loss went through autograd, the penalty gradients are injected
            # by the batch kernels (identical on every rank, so adding them before the all-reduce changes nothing)
            self.batch.inject_penalty_grads(self.loss_kind, self.penalty_rate)
        if self.dp is not None:
            self.dp.sync_gradients()
        self.opt.step()
        self.scale_opt.step()
        return loss

    def step_graphed(self, x, y):
        """The whole training step (forward, loss, backward, both optimizers) as ONE hipGraph launch.
        The step is launch-bound at these model sizes (hundreds of small kernels); capture removes the
        host from the loop.  First call: 3 eager warm-up steps on a side stream, then capture."""
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
            self.graph = torch.cuda.CUDAGraph()
            self.opt.zero_grad(set_to_none=True)
            self.scale_opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(self.graph):
                self._loss = self.step(self._x, self._y)
        self._x.copy_(x)
        self._y.copy_(y)
        self.graph.replay()
        return self._loss

    @torch.no_grad()
    def evaluate(self, x, y):
        self.model.eval()
        p = self.model(x)
        return float(sparse_categorical_crossentropy(y, p).mean()), float((p.argmax(1) == y).float().mean())


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=list(INPUT_SHAPES), default="cifar")
    ap.add_argument("--mode", choices=["nq", "cl"], default="nq")
    ap.add_argument("--value", type=float, default=1e-11, help="penalty_threshold (nq) or penalty_rate (cl)")
    ap.add_argument("--orientation", default="channelwise")
    ap.add_argument("--loss", choices=list(LOSSES), default=None)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--ddp-mode", choices=["A", "B"], default="A")
    ap.add_argument("--export-dir", default=None, help="write the reference's integer export here at the end")
    ap.add_argument("--graph", action="store_true", help="capture the whole step in a hipGraph (single GPU)")
    ap.add_argument("--backend", default="nccl", help="nccl = RCCL over xGMI (default); gloo + --share-gpu rehearses N>1 on one GPU")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--batched", action="store_true", help="multi-tensor launches for all fake-quant ops of a step (lq_batch_*)")
    ap.add_argument("--channels-last", action="store_true",
                    help="feed NHWC-strided batches (torch.channels_last): MIOpen's fp32 igemm kernels are NHWC; measured "
                         "+17 %% on the ResNet-18-like config, -17 %% on the small CIFAR CNN")
    args = ap.parse_args(argv)

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
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    rank = dist.get_rank() if world > 1 else 0

    tr = Trainer(args.config, args.mode, args.value, args.orientation, args.loss, seed=args.seed, device=dev,
                 ddp_mode=args.ddp_mode, graph=args.graph, batched=args.batched)
    do_step = tr.step_graphed if args.graph else tr.step
    g = torch.Generator(device=dev).manual_seed(args.seed + rank)
    batches = [synthetic_batch(args.config, args.batch, dev, g) for _ in range(4)]
    if args.channels_last and args.config != "mnist":
        batches = [(x.contiguous(memory_format=torch.channels_last), y) for x, y in batches]
    for i in range(args.warmup):
        do_step(*batches[i % 4])
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = do_step(*batches[i % 4])
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if rank == 0:
        n_q = sum(p.numel() for l in tr.custom_layers for p in l._regularized())
        print(json.dumps({
            "metric": "images/sec end-to-end training step (synthetic data)", "config": args.config, "mode": args.mode,
            "value": world * args.batch * args.steps / dt, "unit": "images/s", "n_gpus": world,
            "ms_per_step": dt / args.steps * 1e3, "per_gpu_batch": args.batch, "orientation": args.orientation,
            "loss_term": args.loss, "quantized_elements": n_q, "final_loss": float(loss), "ddp_mode": args.ddp_mode,
            "hipgraph": bool(args.graph), "batched": bool(args.batched),
            "channels_last": bool(args.channels_last)}))
        if args.export_dir:
            from .export import save_compress_parameters
            print(json.dumps(save_compress_parameters(tr.model, args.export_dir)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
