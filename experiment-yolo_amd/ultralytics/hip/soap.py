"""SOAP -- "Shampoo with Adam in the preconditioner's eigenbasis" (arXiv 2409.11321) -- as the reference trainer ships it
(engine/trainer.py:54-473) and builds it (``SOAP(g[2], lr=lr, betas=(momentum, 0.95), weight_decay=0.0)`` + the two other
parameter groups, :1156-1165): betas (momentum, 0.95), eps 1e-8, preconditioner refreshed every 10 steps, 1-D tensors (biases,
norm weights) run plain Adam, decoupled weight decay applied after the step, bias correction on, first call only seeds the
preconditioner.  Options the reference never switches on (merge_dims, precondition_1d, normalize_grads, channels_last) are not
carried.

This is the one optimizer of the hot path that is NOT a HIP kernel: per parameter it is a handful of small GEMMs, an eigh once and
a QR every tenth step -- host-driven torch ops over views of the flat parameter / gradient buffers, selectable by name as in the
reference, priced accordingly."""
from __future__ import annotations

import torch


class _State:
    __slots__ = ("step", "m", "v", "gg", "q")

    def __init__(self, g):
        self.step = 0
        self.m, self.v = torch.zeros_like(g), torch.zeros_like(g)
        # one Gram accumulator per dimension of a >= 2-D tensor; None = that dimension is left alone
        self.gg = [None] if g.dim() == 1 else [torch.zeros(n, n, device=g.device, dtype=g.dtype) for n in g.shape]
        self.q = None


def _rotate(t, bases, back):
    """Contract every dimension of ``t`` with its basis (or cycle it unchanged when it has none): after len(bases) contractions the
    dimensions are back in their original order.  Forward uses Q, back uses Q^T (reference project / project_back)."""
    for q in bases:
        if q is None:
            t = t.permute(*range(1, t.dim()), 0)
        else:
            t = torch.tensordot(t, q, dims=[[0], [1 if back else 0]])
    return t


def _gram_update(st, g, beta):
    if g.dim() == 1:
        return
    for k, acc in enumerate(st.gg):
        others = [d for d in range(g.dim()) if d != k]
        acc.lerp_(torch.tensordot(g, g, dims=[others, others]), 1 - beta)


def _basis_eigh(st):
    out = []
    for acc in st.gg:
        if acc is None:
            out.append(None)
            continue
        eye = 1e-30 * torch.eye(acc.shape[0], device=acc.device, dtype=acc.dtype)
        q = None
        if not acc.is_cuda:  # the reference's path: float32, float64 only when LAPACK gives up
            try:
                _, q = torch.linalg.eigh(acc + eye)
            except Exception:
                q = None
        if q is None:
            # On the GPU the float32 eigensolver returned bases that were not orthogonal for the rank-deficient Gram matrices of
            # the small detection-head layers (Q^T Q off by O(1)): projecting and projecting back then no longer cancel and the
            # layer's weights grew 10x per step.  float64 there -- these matrices are at most a few hundred wide.
            _, q = torch.linalg.eigh(acc.double() + eye.double())
            q = q.to(acc.dtype)
        out.append(torch.flip(q, [1]))  # descending eigenvalues
    return out


def _basis_qr(st):
    """One power iteration + QR per dimension; the second-moment estimate follows the re-sorted eigen directions."""
    out, v = [], st.v
    for k, (acc, q) in enumerate(zip(st.gg, st.q)):
        if acc is None:
            out.append(None)
            continue
        est = torch.diag(q.T @ acc @ q)
        order = torch.argsort(est, descending=True)
        v = v.index_select(k, order)
        if acc.is_cuda:
            q2, _ = torch.linalg.qr((acc @ q[:, order]).double())
            q2 = q2.to(acc.dtype)
        else:
            q2, _ = torch.linalg.qr(acc @ q[:, order])
        out.append(q2)
    st.v = v
    return out


def _refresh(st, g, beta, every):
    """reference update_preconditioner: momentum leaves the old basis, Gram matrices take the new gradient, the basis is created
    (eigh) or refreshed (QR every ``every`` steps), momentum enters the current basis again."""
    if st.q is not None:
        st.m = _rotate(st.m, st.q, back=True)
    _gram_update(st, g, beta)
    if st.q is None:
        st.q = _basis_eigh(st)
    if st.step > 0 and st.step % every == 0:
        st.q = _basis_qr(st)
    if st.step > 0:
        st.m = _rotate(st.m, st.q, back=False)


class Soap:
    """``params``: list of (tensor view, group index); ``step(grads, lr[3], weight_decay[3])`` updates the views in place."""

    def __init__(self, params, beta1=0.9, beta2=0.95, eps=1e-8, every=10, max_precond_dim=10000):
        self.params, self.beta1, self.beta2, self.eps, self.every = list(params), beta1, beta2, eps, every
        self.max_precond_dim = max_precond_dim
        self.state = {}

    @torch.no_grad()
    def step(self, grads, lr, weight_decay):
        for i, ((p, grp), g) in enumerate(zip(self.params, grads)):
            st = self.state.get(i)
            if st is None:
                st = self.state[i] = _State(g)
                st.gg = [a if (a is None or a.shape[0] <= self.max_precond_dim) else None for a in st.gg]
                _refresh(st, g, self.beta2, self.every)
                continue  # the first call only seeds the preconditioner: the current gradient never meets its own projection
            gp = _rotate(g, st.q, back=False)
            st.step += 1
            st.m.mul_(self.beta1).add_(gp, alpha=1 - self.beta1)
            st.v.mul_(self.beta2).add_(gp.square(), alpha=1 - self.beta2)
            size = lr[grp] * (1 - self.beta2 ** st.step) ** 0.5 / (1 - self.beta1 ** st.step)
            upd = _rotate(st.m / (st.v.sqrt() + self.eps), st.q, back=True)
            p.add_(upd, alpha=-size)
            if weight_decay[grp] > 0.0:
                p.add_(p, alpha=-lr[grp] * weight_decay[grp])
            _refresh(st, g, self.beta2, self.every)
