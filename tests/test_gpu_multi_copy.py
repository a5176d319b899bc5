"""GPU: hc_multi_copy (gradient-bucket pack / unpack, holocron_amd/parallel.py::_copy_all) is bit-exact against torch's cast-copy
+ scaling for every dtype pair, for views at unaligned offsets of a flat buffer, and across the piece / launch boundaries."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sdt,ddt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.float32)])
@pytest.mark.parametrize("scale", [1.0, 0.125, 1.0 / 3.0])
def test_multi_copy_matches_torch(sdt, ddt, scale):
    from holocron_amd import _lib, parallel
    g = torch.Generator(device="cuda").manual_seed(5)
    # sizes: scalars, odd tails, one piece exactly, one piece + 5, a tensor of 9 pieces, and enough tensors for three launches
    sizes = [1, 3, 10, 48, 1000, _lib.HC_MULTI_COPY_PIECE, _lib.HC_MULTI_COPY_PIECE + 5, 9 * _lib.HC_MULTI_COPY_PIECE + 3] + [257] * 150
    flat = torch.zeros(sum(sizes) + 7, dtype=ddt, device="cuda")
    src, dst, off = [], [], 1                                  # views start at element 1: nothing is 16-byte aligned by luck
    for n in sizes:
        src.append(torch.randn(n + 1, generator=g, device="cuda").to(sdt)[1:])      # the sources are off by one element too
        dst.append(flat[off:off + n])
        off += n
    assert parallel._hip_copy_all(dst, src, scale) is True
    torch.cuda.synchronize()
    for d, s in zip(dst, src):
        want = (s.float() * scale).to(ddt)
        assert torch.equal(d, want), (d.numel(), float((d.float() - want.float()).abs().max()))
    assert float(flat[0]) == 0.0 and float(flat[off:].abs().sum()) == 0.0         # nothing outside the views was written


def test_copy_all_takes_the_hip_path_and_falls_back():
    from holocron_amd import parallel
    a = [torch.randn(100, device="cuda"), torch.randn(7, 3, device="cuda")]
    b = [torch.empty(100, device="cuda"), torch.empty(7, 3, device="cuda")]
    parallel._copy_all(b, a, 0.5)
    assert all(torch.equal(x, y * 0.5) for x, y in zip(b, a))
    # a non-contiguous destination does not qualify: the torch path gives the same result
    nc = [torch.empty(3, 7, device="cuda").t()]
    assert parallel._hip_copy_all(nc, [a[1]], 1.0) is False
    parallel._copy_all(nc, [a[1]], 0.5)
    assert torch.equal(nc[0], a[1] * 0.5)
    # fp16 is not a wire format of the kernel
    h = [torch.empty(100, device="cuda", dtype=torch.float16)]
    assert parallel._hip_copy_all(h, [a[0]], 1.0) is False
