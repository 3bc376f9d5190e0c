"""The one helper of the reference's dino/utils.py that the hot path uses: trunc_normal_
(dino/utils.py:482-520), the initialiser of cls_token / pos_embed / Linear weights."""
import math

import torch


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    """Fill `tensor` in place with N(mean, std^2) restricted to the ABSOLUTE interval [a, b]
    by inverse-CDF sampling, as the reference does (so with std=.02 and the default bounds it
    is effectively an untruncated normal)."""
    def cdf(v):
        return 0.5 * (1.0 + math.erf(v / math.sqrt(2.0)))

    lo, hi = cdf((a - mean) / std), cdf((b - mean) / std)
    with torch.no_grad():
        tensor.uniform_(2.0 * lo - 1.0, 2.0 * hi - 1.0).erfinv_()
        tensor.mul_(std * math.sqrt(2.0)).add_(mean).clamp_(min=a, max=b)
    return tensor
