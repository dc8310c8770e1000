"""The one loss function of models/networks/loss.py that is on the hot path (:10-16)."""
import torch.nn.functional as F


def gan_loss(pred, should_be_classified_as_real):
    """Non-saturating logistic loss, averaged per sample so that a mean over the (sharded) batch
    is a global-batch mean."""
    sign = -1.0 if should_be_classified_as_real else 1.0
    return F.softplus(sign * pred).view(pred.size(0), -1).mean(dim=1)
