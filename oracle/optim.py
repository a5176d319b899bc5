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
