"""unimm_host_mask_pack (the host half of the input path: include/unimm_hip.h, ABI 18) against numpy's packbits, every dtype
the reference's masks come in (int64: utils/data_utils.py:300, bool / uint8 / int32 / float32), ragged widths, threads."""
import numpy as np
import pytest
import torch

from unimm_amd import lib


def _want(m):
    T = m.shape[-1]
    nw = (T + 31) // 32
    bits = np.packbits(m.numpy() != 0, axis=-1, bitorder="little")
    pad = np.zeros(m.shape[:-1] + (nw * 4,), dtype=np.uint8)
    pad[..., :bits.shape[-1]] = bits
    return pad.view("<u4").reshape(m.shape[:-1] + (nw,))


@pytest.mark.parametrize("dtype", [torch.int64, torch.bool, torch.uint8, torch.int32, torch.float32])
@pytest.mark.parametrize("T", [256, 37, 64, 1, 33])
def test_host_mask_pack_matches_packbits(dtype, T):
    g = torch.Generator().manual_seed(T)
    m = torch.rand((6, 5, T), generator=g) < 0.4
    m = m.float() * 0.25 if dtype == torch.float32 else m.to(dtype)
    got = lib.host_mask_pack(m).numpy().view(np.uint32)
    assert (got == _want(m)).all()


def test_host_mask_pack_threads_out_buffer_and_errors():
    g = torch.Generator().manual_seed(0)
    m = (torch.rand((64, 256, 256), generator=g) < 0.5).to(torch.int64)          # large enough to be split over threads
    want = _want(m)
    out = torch.empty((64, 256, 8), dtype=torch.int32)
    for th in (0, 1, 3, 16):
        out.zero_()
        assert lib.host_mask_pack(m, out=out, threads=th) is out
        assert (out.numpy().view(np.uint32) == want).all(), th
    mt = m.transpose(1, 2)                                                         # non-contiguous input
    assert (lib.host_mask_pack(mt).numpy().view(np.uint32) == _want(mt.contiguous())).all()
    with pytest.raises(lib.UnimmHipError):
        lib.host_mask_pack(m, out=torch.empty((3,), dtype=torch.int32))
    with pytest.raises(lib.UnimmHipError):
        lib.host_mask_pack(m.to(torch.float64))


def test_host_copy_threads():
    g = torch.Generator().manual_seed(1)
    a = torch.randn((50, 37, 2048), generator=g)                                   # 15 MB: split over threads
    for th in (0, 1, 3, 8):
        b = torch.zeros_like(a)
        assert lib.host_copy(b, a, threads=th) is b and torch.equal(a, b)
    small = torch.randn(100)
    assert torch.equal(lib.host_copy(torch.zeros(100), small), small)              # small / strided / other dtypes: torch's copy_
    at = a.transpose(0, 1)
    assert torch.equal(lib.host_copy(torch.zeros(at.shape), at), at)
