"""Dev: inference throughput of the eval path (network forward with the fused vote aggregation + prediction
parsing + batched HIP NMS + per-class lists), bs=32, T=1024 by default."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd.p2rnet import P2RConfig, default_config, METHODS
from pose2room_amd.p2rnet.synthetic import make_batch
B, T = int(os.environ.get('B', 32)), int(os.environ.get('T', 1024))
dev = torch.device('cuda:0')
cfg = P2RConfig(default_config('test', data={'num_frames': T}, test={'remove_far_box': False}), device=dev)   # random weights: the far-box filter would reject every box
torch.manual_seed(42)
net = METHODS.get('P2RNet')(cfg).to(dev).eval()
batch = make_batch(B, T, seed=1, device=dev)
with torch.no_grad():
    for _ in range(3): net.generate(batch, eval=True)
    gc.collect(); gc.freeze()
    for name, fn in (('network only (generate_end_points)', lambda: net.generate_end_points(batch)),
                     ('generate: network + parsing + NMS + lists', lambda: net.generate(batch, eval=True))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f'{name}: {dt * 1e3:.1f} ms/batch = {B / dt:.0f} samples/s (bs={B}, T={T})')
