"""FGSM (Goodfellow et al., ICLR 2015) -- one step of size epsilon, no momentum.
Mirror of transferattack/gradient/fgsm.py:28-33."""
from ..attack import Attack


class FGSM(Attack):
    """Official arguments: epsilon=16/255.
    Example: python main.py --input_dir ./data --output_dir adv_data/fgsm/resnet50 --attack fgsm --model=resnet50
    """

    def __init__(self, model_name, epsilon=16/255, targeted=False, random_start=False, norm='linfty',
                 loss='crossentropy', device=None, attack='FGSM', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self.alpha = epsilon
        self.epoch = 1
        self.decay = 0
