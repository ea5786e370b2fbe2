"""GMM2D on the MI355X kernels, interface of the reference's ``Transformer/model/components/gmm2d.py:7-217``
(2-D Gaussian mixture: Cholesky sampling, log-density, grid-search mode).  The live Transformer path never
instantiates it; it is provided as a stand-alone op because the north star names it.

Randomness: the reference draws ``torch.randn`` noise and a ``Categorical`` component inside ``rsample``; a HIP
kernel cannot reproduce torch's generator streams, so ``rsample`` takes the noise from torch's device generator
(plumbing) or from the caller (``noise=`` / ``components=``), and the kernel does the arithmetic
``mu_c + L_c z_c`` -- bit-comparable with the reference once the draws are pinned.
"""
from __future__ import annotations

import ctypes as C

import torch

from ... import _lib
from ..._lib import MmtError


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise MmtError(f"GMM2D: {name} must be a device tensor (no CPU fallback)")
    if t.dtype != torch.float32:
        raise MmtError(f"GMM2D: {name} has dtype {t.dtype}, expected torch.float32")
    return t.contiguous()


class GMM2D:
    """:param log_pis: [..., N]  :param mus: [..., N*2] (or [..., N, 2])  :param log_sigmas: same  :param corrs: [..., N]"""

    def __init__(self, log_pis, mus, log_sigmas, corrs):
        log_pis, corrs = _dev(log_pis, "log_pis"), _dev(corrs, "corrs")
        self.components = log_pis.shape[-1]
        self.dimensions = 2
        self.device = log_pis.device
        self.batch_shape = tuple(log_pis.shape[:-1])
        self.mus = self.reshape_to_components(_dev(mus, "mus"))
        self.log_sigmas = self.reshape_to_components(_dev(log_sigmas, "log_sigmas"))
        N = self.components
        if tuple(self.mus.shape) != self.batch_shape + (N, 2) or tuple(self.log_sigmas.shape) != self.batch_shape + (N, 2) \
                or tuple(corrs.shape) != self.batch_shape + (N,):
            raise MmtError(f"GMM2D: inconsistent shapes log_pis {tuple(log_pis.shape)} mus {tuple(self.mus.shape)} "
                           f"log_sigmas {tuple(self.log_sigmas.shape)} corrs {tuple(corrs.shape)}")
        if not 1 <= N <= 64:
            raise MmtError("GMM2D: 1 <= components <= 64")
        self.corrs = corrs
        self._R = 1
        for s in self.batch_shape:
            self._R *= s
        if self._R == 0:
            raise MmtError("GMM2D: empty batch")
        e = lambda *shape: torch.empty(self.batch_shape + shape, dtype=torch.float32, device=self.device)
        self.log_pis, self.sigmas, self.one_minus_rho2, self.L, self._cov = e(N), e(N, 2), e(N), e(N, 2, 2), e(N, 2, 2)
        lib = _lib.load()
        _lib.check(lib.mmt_gmm2d_prepare(_p(log_pis), _p(self.log_sigmas), _p(corrs), self._R, N, _p(self.log_pis),
                                         _p(self.sigmas), _p(self.one_minus_rho2), _p(self.L), _p(self._cov), _stream()),
                   "mmt_gmm2d_prepare")

    @classmethod
    def from_log_pis_mus_cov_mats(cls, log_pis, mus, cov_mats):
        cov = _dev(cov_mats, "cov_mats")
        if cov.shape[-2:] != (2, 2):
            raise MmtError("GMM2D: cov_mats must end in (2, 2)")
        rn = cov.numel() // 4
        log_sigmas = torch.empty(cov.shape[:-2] + (2,), dtype=torch.float32, device=cov.device)
        corrs = torch.empty(cov.shape[:-2], dtype=torch.float32, device=cov.device)
        _lib.check(_lib.load().mmt_gmm2d_from_cov(_p(cov), rn, _p(log_sigmas), _p(corrs), _stream()), "mmt_gmm2d_from_cov")
        return cls(log_pis, mus, log_sigmas, corrs)

    def reshape_to_components(self, tensor):
        if tensor.dim() == 5 or (tensor.shape[-1] == 2 and tensor.dim() >= 2 and tensor.shape[-2] == self.components):
            return tensor
        return tensor.reshape(list(tensor.shape[:-1]) + [self.components, self.dimensions])

    # ------------------------------------------------------------------ densities
    def _log_prob(self, value, want_logp, want_reg):
        value = _dev(value, "value")
        if tuple(value.shape) != self.batch_shape + (2,):
            raise MmtError(f"GMM2D: value shape {tuple(value.shape)}, expected {self.batch_shape + (2,)}")
        logp = torch.empty(self.batch_shape, dtype=torch.float32, device=self.device) if want_logp else None
        reg = torch.empty(self.batch_shape + (self.components,), dtype=torch.float32, device=self.device) if want_reg else None
        _lib.check(_lib.load().mmt_gmm2d_log_prob(_p(self.log_pis), _p(self.mus), _p(self.log_sigmas), _p(self.sigmas),
                                                  _p(self.corrs), _p(self.one_minus_rho2), _p(value), self._R,
                                                  self.components, _p(logp), _p(reg), _stream()), "mmt_gmm2d_log_prob")
        return logp, reg

    def log_prob(self, value):
        """gmm2d.py:136-163: logsumexp_n(log pi_n + log N(value | mu_n, Sigma_n)) -> [...]"""
        return self._log_prob(value, True, False)[0]

    def reg_los(self, value):
        """gmm2d.py:107-133: despite its docstring the reference returns sqrt(|value - mu_n|^2 + 1e-8) -> [..., N]"""
        return self._log_prob(value, False, True)[1]

    # ------------------------------------------------------------------ sampling
    def rsample(self, sample_shape=torch.Size(), noise=None, components=None, generator=None):
        """gmm2d.py:79-105 -> sample_shape + [..., 2].  ``noise``: standard normal of shape sample_shape + mus.shape;
        ``components``: integer component draws of shape sample_shape + batch_shape (default: inverse-CDF draw from
        uniforms of torch's device generator)."""
        sample_shape = tuple(sample_shape)
        S = 1
        for s in sample_shape:
            S *= s
        N = self.components
        if noise is None:
            noise = torch.randn(sample_shape + tuple(self.mus.shape), device=self.device, generator=generator)
        noise = _dev(noise, "noise")
        if tuple(noise.shape) != sample_shape + tuple(self.mus.shape):
            raise MmtError(f"GMM2D.rsample: noise shape {tuple(noise.shape)}, expected {sample_shape + tuple(self.mus.shape)}")
        comp = u = None
        if components is not None:
            if tuple(components.shape) != sample_shape + self.batch_shape or not components.is_cuda:
                raise MmtError("GMM2D.rsample: components must be a device tensor of shape sample_shape + batch_shape")
            comp = components.to(torch.int32).contiguous()
        else:
            u = torch.rand(sample_shape + self.batch_shape, device=self.device, generator=generator)
        out = torch.empty(sample_shape + self.batch_shape + (2,), dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().mmt_gmm2d_rsample(_p(self.log_pis), _p(self.mus), _p(self.L), _p(noise), _p(comp), _p(u), S,
                                                 self._R, N, _p(out), None, _stream()), "mmt_gmm2d_rsample")
        return out

    def mode(self):
        """gmm2d.py:176-204: arg max of log_prob over the 0.01 mesh spanned by the component means (one component: the mean)."""
        if self.mus.shape[-2] > 1:
            if self.mus.dim() == 5 and self.mus.shape[0] != 1:
                raise MmtError("For taking the mode only one sample makes sense.")
            out = torch.empty(self.batch_shape + (2,), dtype=torch.float32, device=self.device)
            _lib.check(_lib.load().mmt_gmm2d_mode(_p(self.log_pis), _p(self.mus), _p(self.log_sigmas), _p(self.sigmas),
                                                  _p(self.corrs), _p(self.one_minus_rho2), self._R, self.components,
                                                  _p(out), _stream()), "mmt_gmm2d_mode")
            return out
        return torch.squeeze(self.mus, dim=-2)

    def get_for_node_at_time(self, n, t):
        return self.__class__(self.log_pis[:, n:n + 1, t:t + 1], self.mus[:, n:n + 1, t:t + 1],
                              self.log_sigmas[:, n:n + 1, t:t + 1], self.corrs[:, n:n + 1, t:t + 1])

    def get_covariance_matrix(self):
        """gmm2d.py:211-217 -> [..., N, 2, 2]"""
        return self._cov
