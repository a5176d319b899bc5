"""Forward + backward of RepVGG-A0's last two RepBlocks (192 -> 1280 @ 14 stride 2, 1280 -> 1280 @ 7) at batch 256: the launches the
big-tile gather-conv serves.  Meant to run under rocprofv3 (kernel trace / PMC), prints nothing but a checksum."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import holocron_amd as h  # noqa: E402

torch.manual_seed(0)
b1 = h.models.RepBlock(192, 1280, 2, False).cuda().train()
b2 = h.models.RepBlock(1280, 1280, 1, True).cuda().train()
x = torch.randn(256, 192, 14, 14, device="cuda").requires_grad_(True)
for _ in range(int(os.environ.get("ITERS", "4"))):
    out = b2(b1(x))
    out.float().mean().backward()
torch.cuda.synchronize()
print("checksum", float(out.float().abs().mean()))
