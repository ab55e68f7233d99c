"""Dev: report every host-synchronising call inside one train step (torch.cuda.set_sync_debug_mode)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pose2room_amd.p2rnet.synthetic import make_batch
dev = torch.device('cuda:0')
trainer, cfg = bench.build_trainer(dev, 1024, 1)
batch = make_batch(32, 1024, seed=1234, device=dev)
for _ in range(2): trainer.train_step(dict(batch))
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
import traceback
_orig = warnings.showwarning
def show(message, category, filename, lineno, file=None, line=None):
    if 'synchroniz' in str(message):
        st = [f for f in traceback.extract_stack() if '/repo/' in f.filename and 'find_syncs' not in f.filename]
        print('SYNC:', str(message)[:80], '<-', ' | '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in st[-3:]))
warnings.showwarning = show
warnings.simplefilter('always')
trainer.train_step(dict(batch))
torch.cuda.set_sync_debug_mode("default")
