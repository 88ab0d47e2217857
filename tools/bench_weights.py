#!/usr/bin/env python3
"""Weight-shaped sweeps: the per-step set of quantised tensors of the BASELINE configs
C1 (MNIST 2-dense, 4 tensors, 101 770 el), C2 (CIFAR CNN, 12 tensors, 287 008 el), C3 (ResNet-18-like, 40 tensors,
11 171 712 el) -- latency regime.  Reports microseconds per step (forward + scale gradient + scale update of the whole
set) for the multi-tensor batch (lq_batch_*: 4 launches) and for the per-tensor entry points (4 launches per tensor),
for all four orientations.  Weights ~ N(0, 0.05) seed 42, dy ~ N(0, 1e-3), scales at the reference's init value.

    python tools/bench_weights.py [--steps 200] > profiles/r01_weight_sweeps.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import learned_quantization_amd as lq  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (LQ_HIP_LIB -> _hip.use_library)


def timed(fn, steps, dev, reset=None):
    if reset is not None:
        reset()                  # every timed loop starts from the same scales and Adam state (the loops update them)
    for _ in range(10):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--only", default=None, help="config:orientation, e.g. mnist:columnwise")
    ap.add_argument("--abi-only", action="store_true", help="time only the raw batch ABI (profiling runs)")
    ap.add_argument("--companion-only", action="store_true", help="conv kernels emit the OIHW companion only (hwio_out=False: what the trainer uses)")
    ap.add_argument("--kernel-storage", default="hwio", choices=["hwio", "oihw"],
                    help="memory order of the conv kernels (layers.py): hwio = the LDS-tile companion path of the *_oihw column; "
                         "oihw = stored as MIOpen consumes them, plain streaming launches (the *_oihw column then equals the plain one)")
    ap.add_argument("--ablate", type=int, default=0, help="development library only (LQ_HIP_LIB=.../liblq_hip_dev.so): lq_dev_set_ablate mask")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = []
    if args.ablate:
        import ctypes as _ct
        _lib = lq._hip.load()
        _lib.lq_dev_set_ablate.restype = _ct.c_int
        _lib.lq_dev_set_ablate.argtypes = [_ct.c_int]
        assert _lib.lq_dev_set_ablate(args.ablate) == 0
    for config, lam in (("mnist", 1e-10), ("cifar", 1e-11), ("imagenette", 1e-11), ("resnet50", 1e-11)):
        for orient in (("rowwise", "columnwise", "scalar") if config == "mnist" else ("rowwise", "columnwise", "channelwise", "scalar")):
            if args.only and args.only != f"{config}:{orient}":
                continue
            lq.reset_layer_names()
            # resnet50 (BASELINE configs[4], 108 tensors, 23.5 M el): "mixed" thresholds, 1e-10 on the 3x3 kernels, 1e-11 elsewhere
            model = lq.build_model(config, mode="nq", value=(1e-10, lam) if config == "resnet50" else lam, seed=42,
                                   orientation=orient, device=dev, kernel_storage=args.kernel_storage)
            batch = lq.FakeQuantBatch(model, hwio_out=not args.companion_only)
            opt = lq.BatchedScaleAdam(batch)
            g = torch.Generator(device=dev).manual_seed(42)
            dys = [torch.empty_like(e.param.data).normal_(generator=g) * 1e-3 for e in batch.entries]      # the parameter's strides
            n_el = sum(e.param.numel() for e in batch.entries)

            def batched_step():
                opt.zero_grad()
                outs = batch.quantize_all()
                torch.autograd.backward(outs, dys)
                opt.step()

            # the trainer's form (round 4): fake-quantised tensors are leaves, autograd only delivers their gradients (emulated here by
            # assigning them, which is what AccumulateGrad does with a fresh gradient), finish_backward() does the rest
            batch_l = opt_l = None
            if not args.abi_only:
                batch_l = lq.FakeQuantBatch(model, hwio_out=not args.companion_only, autograd=False)
                opt_l = lq.BatchedScaleAdam(batch_l)
                params_l = [e.param for e in batch_l.entries]

            def batched_leaf_step():
                opt_l.zero_grad()
                for p in params_l:
                    p.grad = None
                batch_l.quantize_all()
                for i, d in enumerate(dys):
                    lo = batch_l._leaf_o[i]
                    if lo is not None:
                        lo.grad = dys_o[i]
                    else:
                        batch_l._leaf[i].grad = d
                batch_l.finish_backward()
                opt_l.step()

            single_opt = lq.ScaleAdam([e.nested.scale for e in batch.entries], lr=1e-4)

            def per_tensor_step():
                for e, d in zip(batch.entries, dys):
                    lq.fq_forward(e.param.data, e.nested.scale.data)
                    e.nested.scale.grad = lq.fq_scale_grad(e.param.data, e.nested.scale.data, d, e.nested.penalty_threshold)
                single_opt.step()

            # raw C-ABI cost without autograd/python per-tensor glue
            lib = lq._hip.load()
            sp = lq._hip.stream_ptr(dev)
            import ctypes
            ptrs = (ctypes.c_void_p * len(dys))(*[d.data_ptr() for d in dys])

            def batched_abi_only():
                lib.lq_batch_forward(batch._handle, sp)
                lib.lq_batch_scale_grad(batch._handle, ptrs, batch.ws.data_ptr(), batch.ws.numel(), sp)
                lib.lq_batch_scale_adam(batch._handle, 1e-4, 0.9, 0.999, 1e-7, 1, None, 0, sp)

            # the training path proper: the convolutions consumed the OIHW companions, so MIOpen's weight gradients arrive in
            # OIHW order and the scale-gradient launch gathers them (and writes dP back in HWIO order)
            dys_o = [d.permute(3, 2, 0, 1).contiguous() if e.out_oihw is not None else d for e, d in zip(batch.entries, dys)]
            ptrs_o = (ctypes.c_void_p * len(dys))(*[d.data_ptr() for d in dys_o])

            def batched_abi_oihw():
                lib.lq_batch_forward(batch._handle, sp)
                lib.lq_batch_scale_grad_oihw(batch._handle, ptrs_o, batch.ws.data_ptr(), batch.ws.numel(), sp)
                lib.lq_batch_scale_adam(batch._handle, 1e-4, 0.9, 0.999, 1e-7, 1, None, 0, sp)

            def batched_abi_fused():            # what the trainer's nq step launches: forward, traversal, finalize + Adam
                lib.lq_batch_forward(batch._handle, sp)
                lib.lq_batch_scale_grad_step(batch._handle, ptrs_o, 1, batch.ws.data_ptr(), batch.ws.numel(), 1e-4, 0.9, 0.999, 1e-7, 1, None, 0, sp)

            state0 = [(e.nested.scale.data.clone(), e.m.clone(), e.v.clone()) for e in batch.entries]

            def reset():
                for e, (s0, m0, v0) in zip(batch.entries, state0):
                    e.nested.scale.data.copy_(s0)
                    e.m.copy_(m0)
                    e.v.copy_(v0)

            row = {"config": config, "orientation": orient, "tensors": len(batch.entries), "elements": n_el,
                   "companion_only": bool(args.companion_only), "kernel_storage": args.kernel_storage,
                   "us_per_step_batched_abi": timed(batched_abi_only, args.steps, dev, reset),
                   "us_per_step_batched_abi_oihw": timed(batched_abi_oihw, args.steps, dev, reset),
                   "us_per_step_batched_abi_oihw_fused_update": timed(batched_abi_fused, args.steps, dev, reset)}
            if not args.abi_only:
                row["us_per_step_batched_autograd"] = timed(batched_step, args.steps, dev, reset)
                for p in params_l:
                    p.grad = None
                row["us_per_step_batched_leaves"] = timed(batched_leaf_step, args.steps, dev, reset)
                row["us_per_step_per_tensor"] = timed(per_tensor_step, args.steps, dev, reset)
            rows.append(row)
            print(json.dumps(rows[-1]), flush=True)
            del batch, model


if __name__ == "__main__":
    main()
