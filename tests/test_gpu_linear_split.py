"""gigl_linear's split-precision path (three bf16 planes per fp32 operand, six bf16 MFMAs per k-step) keeps fp32-class
accuracy: error vs an fp64 product bounded like the exact fp32 MFMA kernel's (a few 1e-7 of sum |a||w|)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,k,n,act", [(1000, 200, 256, 1), (5000, 512, 47, 0), (700, 1536, 256, 0), (513, 36, 64, 0),
                                       (4096, 256, 128, 1), (640, 4, 3, 0)])
def test_split_linear_matches_fp64(m, k, n, act):
    from gigl_amd.engine import HipEngine
    eng = HipEngine(0)
    g = torch.Generator().manual_seed(m + k + n)
    a = torch.randn(m, k, generator=g) * torch.exp(torch.randn(m, 1, generator=g) * 2)  # rows of very different scale
    w = torch.randn(n, k, generator=g) / np.sqrt(k)
    w[0] *= 1e-4
    b = torch.randn(n, generator=g)
    dev = eng.device
    md = torch.tensor([m], dtype=torch.int32, device=dev)
    got = eng.linear(a.to(dev), w.to(dev), b.to(dev), md, m, act).cpu().double()
    ref = a.double() @ w.double().T + b.double()
    if act:
        ref = ref.clamp(min=0)
    scale = a.double().abs() @ w.double().abs().T + b.double().abs()
    err = ((got - ref).abs() / scale).max().item()
    assert err < 4e-7, err
    # asymmetric operands catch transposed tiles: identity-like A
    eye = torch.zeros(m, k)
    eye[torch.arange(min(m, k)), torch.arange(min(m, k))] = 1.0
    got = eng.linear(eye.to(dev), w.to(dev), None, md, m, 0).cpu()
    assert torch.equal(got[: min(m, k)], w.T[: min(m, k)].contiguous()) and not got[min(m, k):].any()
    eng.close()
