"""Weight repack of repvgg_a0 (hc_pack_conv_weights_multi, once per optimizer step): the full launch and the launch restricted to
one stage's blocks, to see where its time is."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import holocron_amd as h


def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


torch.manual_seed(0)
m = h.models.repvgg_a0(num_classes=10).cuda().train()
x = torch.rand((256, 3, 224, 224), device="cuda")
m(x)                                         # geometry-dependent weight images are chosen by the first forward
torch.cuda.synchronize()
blocks = [(si, bi, b) for si, st in enumerate(m.features) for bi, b in enumerate(st)]


def stale(sel):
    for si, bi, b in blocks:
        if sel(si, bi):
            b._hc.packed_key = None


def run(sel):
    def f():
        stale(sel)
        m._hc_pack_table = None
        m._pack_all()
    return f


# the table upload is part of f() here (host side); time only the kernel through events around _pack_all with a cached table
def run_cached(sel):
    stale(sel); m._hc_pack_table = None; m._pack_all(); torch.cuda.synchronize()
    def f():
        stale(sel)
        m._pack_all()
    return f


def kernel_only(sel):
    """the launch alone, table cached: no Python between the launches"""
    from holocron_amd import _lib
    stale(sel); m._hc_pack_table = None; m._pack_all(); torch.cuda.synchronize()
    c = m._hc_pack_table
    lib = _lib.load()
    return lambda: lib.hc_pack_conv_weights_multi(c[1].data_ptr(), c[2], c[3], _lib.stream())


nel = sum(p.numel() for n, p in m.named_parameters() if p.dim() == 4)
t = timeit(kernel_only(lambda s, b: True), 50)
print(f"kernel only, all blocks {t:8.1f} us  ({12.0 * nel / t * 1e-6:.2f} TB/s on 2 x 6 B per weight)")
t = timeit(kernel_only(lambda s, b: s == 4 and b == 1), 50)
print(f"kernel only, 1280 x 1280 block {t:8.1f} us")
print(f"all blocks            {timeit(run_cached(lambda s, b: True)):8.1f} us")
for si in range(5):
    nb = len(m.features[si])
    print(f"stage {si} ({nb:2d} blocks)    {timeit(run_cached(lambda s, b, si=si: s == si)):8.1f} us")
print(f"stage 3 block 0 only  {timeit(run_cached(lambda s, b: s == 3 and b == 0)):8.1f} us")
print(f"stage 3 blocks 1..14  {timeit(run_cached(lambda s, b: s == 3 and b > 0)):8.1f} us")
print(f"stage 4 block 0 only  {timeit(run_cached(lambda s, b: s == 4 and b == 0)):8.1f} us")
print(f"stage 4 block 1 only  {timeit(run_cached(lambda s, b: s == 4 and b == 1)):8.1f} us")
