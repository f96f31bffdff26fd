"""Stand-alone timing of the frozen MobileNetV3-Large forward (config 4's dominant family): 256 images of 128x128x3
uint8 per chain, as the step driver launches it.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from serl_amd.networks.mobilenet import Backbone

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = W = 128
bb = Backbone.synthetic(0, device="cuda")
imgs = torch.from_numpy(np.random.default_rng(0).integers(0, 256, size=(n, H, W, 3), dtype=np.uint8)).cuda()
out = torch.empty((n,) + tuple(bb.feature_shape(H, W)), dtype=torch.float32, device="cuda")
bb.forward(imgs, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    bb.forward(imgs, out=out)
e1.record()
torch.cuda.synchronize()
print(f"mbv3 forward, {n} images: {e0.elapsed_time(e1) / 10:.3f} ms per chain")
