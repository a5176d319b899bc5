"""torch.profiler view of three eager headline steps: which aten ops launch the small copy / fill kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import holocron_amd as h
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = h.models.repvgg_a0(num_classes=10).to(dev).train()
opt = h.optim.AdaBelief(m.parameters(), lr=1e-3)
x = torch.rand(256, 3, 224, 224, device=dev)
t = torch.randint(0, 10, (256,), device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(m(x), t, label_smoothing=0.1)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
rows = prof.key_averages()
print(rows.table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=70))
