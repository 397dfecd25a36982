"""Oracle losses (zs3/utils/loss.py:5-115) written from their closed forms.  Test infrastructure only."""
import torch
import torch.nn.functional as F


def cross_entropy_2d(logit, target, weight=None, ignore_index=255, batch_average=True):
    """loss.py:31-46: sum_i w[t_i] * -log softmax(z_i)[t_i] / sum_i w[t_i] over t_i != ignore, then / B.
    Evaluated with F.cross_entropy (what nn.CrossEntropyLoss dispatches to) so that the oracle is
    bit-compatible with the reference on CPU -- random-init training trajectories amplify 1e-7
    rounding differences by orders of magnitude per step.  `cross_entropy_2d_closed_form` is the
    same quantity written out; tests check the two agree."""
    loss = F.cross_entropy(logit, target.long(), weight=weight, ignore_index=ignore_index, reduction="mean")
    return loss / logit.shape[0] if batch_average else loss


def cross_entropy_2d_closed_form(logit, target, weight=None, ignore_index=255, batch_average=True):
    b, c = logit.shape[:2]
    t = target.long().reshape(-1)
    z = logit.permute(0, 2, 3, 1).reshape(-1, c)
    keep = t != ignore_index
    t, z = t[keep], z[keep]
    nll = torch.logsumexp(z, 1) - z.gather(1, t[:, None])[:, 0]
    w = torch.ones_like(nll) if weight is None else weight[t]
    loss = (w * nll).sum() / w.sum()
    return loss / b if batch_average else loss


class SegmentationLosses:
    def __init__(self, weight=None, size_average=True, batch_average=True, ignore_index=255, cuda=False):
        self.weight, self.batch_average, self.ignore_index = weight, batch_average, ignore_index

    def build_loss(self, mode="ce"):
        try:
            return {"ce": self.ce, "focal": self.focal, "ce_finetune": self.ce_finetune}[mode]
        except KeyError:
            raise NotImplementedError

    def ce(self, logit, target):
        return cross_entropy_2d(logit, target, self.weight, self.ignore_index, self.batch_average)

    def ce_finetune(self, logit, target):  # loss.py:48-60: no class weights
        return cross_entropy_2d(logit, target, None, self.ignore_index, self.batch_average)

    def focal(self, logit, target, gamma=2, alpha=0.5):  # loss.py:62-81: focal on the *scalar* CE
        logpt = -cross_entropy_2d(logit, target, self.weight, self.ignore_index, False)
        pt = torch.exp(logpt)
        loss = -((1 - pt) ** gamma) * (alpha * logpt if alpha is not None else logpt)
        return loss / logit.shape[0] if self.batch_average else loss


def mmd_loss(gen, real, sigma=(2, 5, 10, 20, 40, 80)):
    """loss.py:99-115.  X=[gen;real]; E_ij = <x_i,x_j> - |x_i|^2/2 - |x_j|^2/2; s = [+1/N]*N ++ [-1/M]*M;
    loss = sqrt(sum_v sum_ij s_i s_j exp(E_ij / v)).  (The +1/N block sits on the first N rows, as in
    loss.py:92-97; call sites always have M == N.)"""
    x = torch.cat((gen, real), 0)
    m, n = gen.shape[0], real.shape[0]
    # the Gram product is formed BEFORE the squared norms, as in loss.py:101-102: autograd sums the two gradient
    # contributions to x in the order the ops were recorded, so the order decides the last bit of the gradient
    xx = x @ x.t()
    sq = (x * x).sum(1, keepdim=True)
    e = xx - 0.5 * sq - 0.5 * sq.t()
    s = torch.cat((torch.full((n, 1), 1.0 / n), torch.full((m, 1), -1.0 / m)), 0).to(x)
    ss = s @ s.t()
    total = 0
    for v in sigma:
        total = total + (ss * torch.exp(e / v)).sum()
    return torch.sqrt(total)


class GMMNLoss:
    def __init__(self, sigma=(2, 5, 10, 20, 40, 80), cuda=False):
        self.sigma = tuple(sigma)

    def build_loss(self):
        return lambda gen, real: mmd_loss(gen, real, self.sigma)
