"""ENS (Liu et al., ICLR 2017) -- MI-FGSM on the mean of several surrogates' logits.
Mirror of transferattack/ensemble/ens.py:31-36; the averaging lives in ``EnsembleModel`` (utils.py:82-105),
or in ``transferattack_amd.dist.ShardedEnsemble`` when the members are spread over GPUs (RCCL all-reduce)."""
from ..attack import Attack


class ENS(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1.
    Example: python main.py --attack ens --model='resnet50,vgg16,mobilenet_v2,inception_v3'"""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., targeted=False,
                 random_start=False, norm='linfty', loss='crossentropy', device=None, attack='ENS', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self.alpha = alpha
        self.epoch = epoch
        self.decay = decay
