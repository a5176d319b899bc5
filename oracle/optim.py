"""Single-tensor restatement of the reference optimizer steps."""
import math

import torch


def adabelief_step(p, g, m, s, step, lr, beta1, beta2, eps, weight_decay=0.0, smax=None):
    """holocron/optim/adabelief.py:138-167 (``step`` is the count AFTER the increment).  In place."""
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    if weight_decay != 0:
        g = g.add(p, alpha=weight_decay)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    r = g - m
    s.mul_(beta2).addcmul_(r, r, value=1 - beta2)
    if smax is not None:
        torch.maximum(smax, s, out=smax)
        denom = (smax.sqrt() / math.sqrt(bc2)).add_(eps)
    else:
        denom = (s.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def lars_step(p, g, buf, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
    """holocron/optim/lars.py:108-133.  In place on p and g (weight decay lands in g, Q4); returns the
    momentum buffer (created from g when ``buf`` is None and momentum != 0)."""
    p_norm = torch.norm(p)
    denom = torch.norm(g)
    if weight_decay != 0:
        g.add_(p, alpha=weight_decay)
        denom = denom + weight_decay * p_norm
    local_lr = 1 if (p_norm == 0 or denom == 0) else p_norm / denom
    if momentum == 0:
        p.add_(g, alpha=float(-lr * local_lr))
        return None
    if buf is None:
        buf = g.clone()
    else:
        buf.mul_(momentum).add_(g, alpha=1 - dampening)
    d = g.add(buf, alpha=momentum) if nesterov else buf
    p.add_(d, alpha=float(-lr * local_lr))
    return buf


def adamp_step(p, g, m, s, step, lr, beta1, beta2, eps, weight_decay=0.0, delta=0.1, smax=None):
    """holocron/optim/adamp.py:164-200 (``step`` is the count AFTER the increment).  In place."""
    import torch.nn.functional as F
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    if weight_decay != 0:
        g = g.add(p, alpha=weight_decay)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    s.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    if smax is not None:
        torch.maximum(smax, s, out=smax)
        denom = (smax.sqrt() / math.sqrt(bc2)).add_(eps)
    else:
        denom = (s.sqrt() / math.sqrt(bc2)).add_(eps)
    pt = m / bc1 / denom
    if F.cosine_similarity(p.view(1, -1), g.view(1, -1)).max() < delta / math.sqrt(p.numel()):
        normalized = p / p.norm().add_(eps)
        pt -= (normalized * pt).sum() * normalized
    p.add_(pt, alpha=-lr)


def ademamix_step(p, g, m1, m2, nu, step, lr, beta1, beta2, beta3, alpha, eps, weight_decay=0.0):
    """holocron/optim/ademamix.py:176-200.  In place."""
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    if weight_decay != 0:
        g = g.add(p, alpha=weight_decay)
    m1.mul_(beta1).add_(g, alpha=1 - beta1)
    nu.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    m2.mul_(beta3).add_(g, alpha=1 - beta3)
    denom = (nu.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m1 / bc1 + alpha * m2, denom, value=-lr)


def lamb_step(p, g, m, s, lr, beta1, beta2, eps, weight_decay=0.0, scale_clip=(0.0, 10.0)):
    """holocron/optim/lamb.py:104-135 (the moments are NOT bias corrected there).  In place; returns local_lr."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    s.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    update = m / (s.sqrt() + eps)
    if weight_decay != 0:
        update = update.add(p, alpha=weight_decay)
    return _trust_ratio_update(p, update, lr, scale_clip)


def _trust_ratio_update(p, update, lr, scale_clip):
    p_norm = p.pow(2).sum().sqrt()
    update_norm = update.pow(2).sum().sqrt()
    phi_p = p_norm.clamp(*scale_clip)
    local_lr = 1 if phi_p == 0 or update_norm == 0 else phi_p / update_norm
    p.add_(update, alpha=-lr * float(local_lr))
    return float(local_lr)


def ralars_step(p, g, m, s, step, lr, beta1, beta2, eps, weight_decay=0.0, force_adaptive_momentum=False, scale_clip=(0, 10)):
    """holocron/optim/ralars.py:77-138 (``step`` is the count AFTER the increment).  In place; returns local_lr."""
    sma_inf = 2 / (1 - beta2) - 1
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    s.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    sma_t = sma_inf - 2 * step * (1 - bc2) / bc2
    if sma_t > 4:
        r_t = math.sqrt((sma_t - 4) * (sma_t - 2) * sma_inf / ((sma_inf - 4) * (sma_inf - 2) * sma_t))
        update = r_t * (m / bc1) / ((s / bc2).sqrt() + eps)
    elif force_adaptive_momentum:
        update = (m / bc1) / ((s / bc2).sqrt() + eps)
    else:
        update = m / bc1
    if weight_decay != 0:
        update = update.add(p, alpha=weight_decay)
    return _trust_ratio_update(p, update, lr, scale_clip)


def tadam_step(p, g, m, s, W, step, lr, beta1, beta2, eps, weight_decay=0.0, dof=None, smax=None):
    """holocron/optim/tadam.py:176-212 (``W`` is the one-element W_t state tensor).  In place."""
    dof_ = p.numel() if dof is None else dof
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    if weight_decay != 0:
        g = g.add(p, alpha=weight_decay)
    w_t = g.sub(m).pow_(2).div_(s.add(eps)).sum()
    w_t.add_(dof_).pow_(-1).mul_(dof_ + p.numel())
    m.mul_(W / (W + w_t)).addcdiv_(w_t * g, W + w_t)
    W.mul_((2 * beta1 - 1) / beta1)
    W.add_(w_t)
    s.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    if smax is not None:
        torch.maximum(smax, s, out=smax)
        denom = (smax.sqrt() / math.sqrt(bc2)).add_(eps)
    else:
        denom = (s.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def adan_step(p, g, prev_g, m, v, n, step, lr, beta1, beta2, beta3, eps, weight_decay=0.0, nmax=None):
    """holocron/optim/adan.py:164-199; ``prev_g`` is read and never written, like there.  In place."""
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    bc3 = 1 - beta3 ** step
    if weight_decay != 0:
        g = g.add(p, alpha=weight_decay)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    dg = g - prev_g
    v.mul_(beta2).add_(dg, alpha=1 - beta2)
    tmp = g + beta2 * dg
    n.mul_(beta3).addcmul_(tmp, tmp, value=1 - beta3)
    if nmax is not None:
        torch.maximum(nmax, n, out=nmax)
        denom = (nmax.sqrt() / math.sqrt(bc3)).add_(eps)
    else:
        denom = (n.sqrt() / math.sqrt(bc3)).add_(eps)
    p.add_((m / bc1 + beta2 * v / bc2) / denom, alpha=-lr)
    if weight_decay != 0:
        p /= 1 + weight_decay * lr


def lookahead_sync(fast, slow, sync_rate):
    """holocron/optim/wrapper.py:121-134.  In place."""
    if sync_rate > 0:
        slow.add_(fast - slow, alpha=sync_rate)
    fast.copy_(slow)
