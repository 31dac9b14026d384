"""The allocation-free entry point (jova_nonsat_fwd_noalloc, include/jova_hip.h) against the entry point it is a
variant of (jova_nonsat_fwd), through the C ABI: bit-identical outputs, derivative factors and evaluation counts on
two calls in a row, the second with a smaller input (the built-in scratch keeps the first call's contents), and the
size limit is an error, not a fault.  Collected last (file name), so a failure here hides no other test."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(x, ep=1e-4, max_iter=100):
    from jova_amd._lib import P, call, lib, stream
    n = x.numel()
    res = []
    for name in ("jova_nonsat_fwd", "jova_nonsat_fwd_noalloc"):
        out = torch.full_like(x, float("nan"))
        dfac = torch.full_like(x, float("nan"))
        n_eval = torch.full((3,), -1, dtype=torch.int32, device=x.device)
        if name == "jova_nonsat_fwd":
            ws = torch.empty(int(lib().jova_nonsat_workspace_bytes(n)), dtype=torch.uint8, device=x.device)
            call(name, P(x), n, ep, max_iter, P(out), P(dfac), P(n_eval), P(ws), stream())
        else:
            call(name, P(x), n, ep, max_iter, P(out), P(dfac), P(n_eval), stream())
        torch.cuda.synchronize()
        res.append((out.cpu(), dfac.cpu(), n_eval.cpu().tolist()))
    return res


def test_noalloc_matches_workspace_entry_point_twice(device):
    from jova_amd._lib import lib
    g = torch.Generator().manual_seed(11)
    max_n = int(lib().jova_nonsat_noalloc_max_n())
    big = (torch.randn(46 * 128 * 194, generator=g) * 3).to(device)          # Davis-sized [A_max, B, g]
    small = (torch.randn(7 * 5 * 37, generator=g) * 0.5).to(device)          # odd size: the 4-byte path
    assert big.numel() <= max_n
    for x in (big, small):
        (o0, d0, e0), (o1, d1, e1) = _both(x)
        assert e0 == e1 and e0[2] == 0 and e0[0] >= 2, (e0, e1)
        assert torch.equal(o0, o1) and torch.equal(d0, d1)
        assert torch.isfinite(o1).all() and torch.isfinite(d1).all()


def test_noalloc_rejects_sizes_beyond_the_scratch(device):
    from jova_amd._lib import P, call, lib, stream
    max_n = int(lib().jova_nonsat_noalloc_max_n())
    x = torch.zeros(max_n + 4, device=device)
    out, dfac = torch.empty_like(x), torch.empty_like(x)
    n_eval = torch.zeros(3, dtype=torch.int32, device=device)
    with pytest.raises(RuntimeError, match="exceed the built-in scratch"):
        call("jova_nonsat_fwd_noalloc", P(x), x.numel(), 1e-4, 100, P(out), P(dfac), P(n_eval), stream())
    # the limit itself is accepted
    call("jova_nonsat_fwd_noalloc", P(x), max_n, 1e-4, 100, P(out), P(dfac), P(n_eval), stream())
    torch.cuda.synchronize()
    assert n_eval.cpu().tolist()[2] == 0 and float(out[:max_n].abs().max()) == 0.0
