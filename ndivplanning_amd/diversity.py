"""Normalized-diversification loss -- mirror of the reference's `diversity.py`
(same four public names, same argument meaning), computed by the gfx950 kernel
`k_ndiv` behind `ndp_ndiv_fwd_bwd` (include/ndp.h).

Reference behaviour kept (diversity.py:8-41):
  * `compute_pairwise_divergence(recodes, codes)` takes N, k from `codes`,
    squeezes both inputs and views them as [N, k, -1];
  * the row-sum denominator is a constant for the gradient (`.detach()`);
  * the sub-gradient at zero distance (diagonal, coincident samples) is 0;
  * K == 1 gives NaN (0/0).
Only `recodes` is differentiable here -- the reference never differentiates the
noise (train_gan.py:193-196 passes the raw noise tensor).
"""
import torch

from . import _capi

HINGE_ALPHA = 0.8


def _as_nkc(t, n, k, name):
    _capi.require_gpu_f32(t, name)
    return torch.squeeze(t).reshape(n, k, -1).contiguous()


class _NDivLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, z):
        lib = _capi.load()
        n, k, cx = x.shape
        cz = z.shape[2]
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        partials = _capi.empty(lib.ndp_ndiv_partials(n, k), x)
        if z.device != x.device:
            raise _capi.NdpError("recodes are on %s, codes on %s" % (x.device, z.device))
        with _capi.on_device(x):
            _capi.check(lib.ndp_ndiv_fwd_bwd(_capi.ptr(x), cx, _capi.ptr(z), cz, n, k, 1.0, _capi.ptr(loss),
                                             _capi.ptr(grad), _capi.ptr(partials), _capi.stream_ptr()),
                        "ndp_ndiv_fwd_bwd")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        (grad,) = ctx.saved_tensors
        return grad * grad_out, None


def compute_pairwise_divergence(recodes, codes):
    """sum relu(0.8 * z_tilde - x_tilde) over all rows and pairs (diversity.py:36-41)."""
    n, k = codes.size(0), codes.size(1)
    if not 1 <= k <= _capi.MAX_SAMPLES:
        raise _capi.NdpError("num_sample=%d outside 1..%d" % (k, _capi.MAX_SAMPLES))
    z = _as_nkc(codes.detach(), n, k, "codes")
    x = _as_nkc(recodes, n, k, "recodes")
    return _NDivLoss.apply(x, z)


def _pairwise_matrices(z):
    """[N,K,K] distances of a [N,K,C] tensor via the same kernel arithmetic is not
    needed on the training path; the helpers below exist for API completeness and
    are plain tensor expressions on the GPU (no autograd guarantees beyond torch's)."""
    _capi.require_gpu_f32(z, "z")
    return torch.linalg.vector_norm(z[:, :, None, :] - z[:, None, :, :], ord=2, dim=3)


def compute_pairwise(z):
    """All-pairs L2 distance inside each row: [N,K,C] -> [N,K,K] (diversity.py:8-9)."""
    return _pairwise_matrices(z)


def compute_pair_distance(z, weight=None):
    """Row-normalised pairwise distance, denominator detached (diversity.py:12-19)."""
    d = compute_pairwise(z)
    if weight is not None:
        d = compute_pairwise(weight) * d
    return d / torch.sum(d, dim=2)[..., None].detach()


def compute_pair_unnormal_distance(z, weight=None):
    """Un-normalised variant (diversity.py:22-29; unused by the training path)."""
    d = compute_pairwise(z)
    if weight is not None:
        d = compute_pairwise(weight) * d
    return d
