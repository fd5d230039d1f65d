"""ENS (Liu et al., ICLR 2017) -- MI-FGSM on the mean of several surrogates' logits.
Mirror of transferattack/ensemble/ens.py:31-36; the averaging lives in ``EnsembleModel`` (utils.py:82-105) when
all members sit on one GPU, or in ``transferattack_amd.dist.ShardedEnsemble`` when there is one member per rank:
then the logits are averaged by one RCCL all-reduce forward and the members' input-gradients are summed by a second
one backward (both inside ``ShardedEnsemble.forward``, so every attack class gets them), after which every rank of the
group runs the identical fused update."""
from ..attack import Attack


class ENS(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1.; surrogates e.g.
    --model='resnet50,vgg16,mobilenet_v2,inception_v3' (ens.py:27)."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., targeted=False,
                 random_start=False, norm='linfty', loss='crossentropy', device=None, attack='ENS', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self._schedule(alpha, epoch, decay)
