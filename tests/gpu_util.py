"""Helpers for the -m gpu parity tests (HIP path vs the CPU oracle)."""
import ctypes as C

import numpy as np
import torch

import comic_amd._lib as L

DEV = 'cuda:0'
# Parity bar (BASELINE.json north_star): fp32 results within 1e-3 relative of the oracle,
# measured as max|a-b| / max|b| per tensor; integer / index outputs bit-exact.
F32_RTOL = 1e-3


def lib():
    return L.load()


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def rel_err(got, ref):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    denom = np.abs(ref).max() + 1e-30
    return float(np.abs(got - ref).max() / denom)


def elementwise_excess(got, ref, tol):
    """Element-wise criterion beside the max-norm one: |a - b| <= tol*|b| + tol*rms(b) for EVERY element.  (With
    max|b| as the absolute term the bound would be implied by the max-norm check; the root mean square is the typical
    magnitude of the tensor, so an element may not hide behind one large outlier of its tensor.)  Returns the largest
    |a - b| / (tol*|b| + tol*rms(b)); <= 1 passes."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    if not got.size:
        return 0.0
    bound = tol * np.abs(ref) + tol * (np.sqrt(np.mean(ref * ref)) + 1e-30)
    return float((np.abs(got - ref) / bound).max())


def assert_close(got, ref, tol=F32_RTOL, name='', elementwise=None):
    """Max-norm bound at `tol`; comparisons at the north star's fp32 bar (tol == F32_RTOL) are also held to the
    element-wise criterion (elementwise=None).  The bf16 plans (3e-2 bounds) and the self-comparisons at 1e-5 keep the
    max-norm form: their tolerances were set for it."""
    if elementwise is None:
        elementwise = tol == F32_RTOL
    e = rel_err(got, ref)
    assert np.isfinite(np.asarray(got, np.float64)).all(), '%s: non-finite values' % name
    assert e <= tol, '%s: rel err %.3e > %.1e (max|ref| %.3e)' % (name, e, tol, np.abs(ref).max())
    if elementwise:
        x = elementwise_excess(got, ref, tol)
        assert x <= 1.0, '%s: element-wise |a-b| is %.2f x (tol*|b| + tol*rms(b)), tol %.1e' % (name, x, tol)
    return e


def stream():
    return torch.cuda.current_stream().cuda_stream


def sync():
    torch.cuda.synchronize()


# Host arrays handed to an asynchronous kernel by raw pointer must stay alive until the
# kernel has run: a temporary `dev(x).data_ptr()` frees its block right away and the next
# temporary re-uses the same address.  P() parks the tensor until the test ends.
_KEEP = []


def P(a, dtype=None):
    t = dev(a, dtype)
    _KEEP.append(t)
    return t.data_ptr()


def release():
    sync()
    _KEEP.clear()
