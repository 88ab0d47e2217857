"""Development probe: end-to-end step rate of the harness under torch/MIOpen knobs (not part of the product path).
usage: python tools/e2e_knobs.py <config> <batch> <benchmark 0|1> <graph 0|1> [channels_last 0|1]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learned_quantization_amd.train import Trainer, synthetic_batch

config, batch, bench, graph = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cl = int(sys.argv[5]) if len(sys.argv) > 5 else 0
torch.backends.cudnn.benchmark = bool(bench)
dev = torch.device("cuda", 0)
tr = Trainer(config, "nq", 1e-11, "channelwise", None, device=dev, graph=bool(graph), batched=True)
step = tr.step_graphed if graph else tr.step
g = torch.Generator(device=dev).manual_seed(1)
bs = [synthetic_batch(config, batch, dev, g) for _ in range(4)]
if cl:
    bs = [(x.contiguous(memory_format=torch.channels_last), y) for x, y in bs]
for i in range(10):
    step(*bs[i % 4])
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for i in range(n):
    step(*bs[i % 4])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{config} bs{batch} benchmark={bench} graph={graph} channels_last={cl}: {batch * n / dt:9.0f} images/s  {dt / n * 1e3:7.3f} ms/step", flush=True)
