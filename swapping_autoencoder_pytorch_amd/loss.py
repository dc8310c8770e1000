"""The one loss function of models/networks/loss.py that is on the hot path (:10-16)."""
from .stylegan2_op import softplus_mean


def gan_loss(pred, should_be_classified_as_real):
    """Non-saturating logistic loss, averaged per sample so that a mean over the (sharded) batch
    is a global-batch mean: F.softplus(-+pred).view(B, -1).mean(dim=1) as one kernel."""
    sign = -1.0 if should_be_classified_as_real else 1.0
    return softplus_mean(pred, sign)
