#!/usr/bin/env python
"""End-to-end parity of BASELINE configs[1] (MI-FGSM / ResNet-50 / eps 16/255 / K=10) at small scale:
the product on MI355X (plain surrogate, and the bench's folded-BN + NHWC arrangement) against the oracle (the
reference's ATen CPU arithmetic) on the SAME seeded images and weights.  Reports the uint8 pixel mismatch of the final
adversarial images and the attack success rate on the surrogate and on two held-out victims.  One JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
N = int(os.environ.get("TA_PARITY_IMAGES", "32"))


def images(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (n, 3, 224, 224), generator=g, dtype=torch.uint8).float() / 255


def build(name, gain):
    """seeded surrogate; ``gain`` < 1 scales the last BatchNorm weight of every residual branch (a smoother,
    better-conditioned network) -- used only to show how the end-to-end divergence tracks conditioning"""
    from transferattack_amd import backbones
    from transferattack_amd.backbones import resnet
    torch.manual_seed(0)
    m = getattr(resnet, name)()
    for mod in m.modules():
        if isinstance(mod, resnet.Bottleneck):
            mod.bn3.weight.data.mul_(gain)
        if isinstance(mod, resnet.BasicBlock):
            mod.bn2.weight.data.mul_(gain)
    backbones.calibrate_batchnorm(m, 0)
    return m.eval()


def main():
    import fgsm_oracle as O
    import transferattack_amd as ta
    from transferattack_amd.utils import quantize_images, wrap_model
    n = int(os.environ.get("TA_PARITY_IMAGES", "16"))
    x = images(n, 0)
    out = {}
    for name, gain in (("resnet50", 1.0), ("resnet50", 0.1), ("resnet18", 1.0), ("resnet18", 0.25)):
        surrogate = build(name, gain)
        with torch.no_grad():
            label = O.logits_of(surrogate, x).argmax(1)
        trace = []
        delta_cpu = O.run_attack("mifgsm", surrogate, x, label, trace=trace)
        u8_cpu = O.quantize_u8(x + delta_cpu)
        base = ta.load_attack_class("mifgsm")
        gpu_model = build(name, gain)

        def load_model(self, model_name, gpu_model=gpu_model):
            for p in gpu_model.parameters():
                p.requires_grad_(False)
            return wrap_model(gpu_model.to("cuda"))

        atk = type("P", (base,), {"load_model": load_model})(model_name="x")
        grads = []
        orig = type(atk).get_grad

        def get_grad(self, loss, delta, orig=orig, grads=grads, **kw):
            grads.append(orig(self, loss, delta, **kw).cpu())
            return grads[-1].to("cuda")

        type(atk).get_grad = get_grad
        delta = atk(x, label)
        u8 = quantize_images(x, delta)
        flips = [round(100 * float((torch.sign(g) != torch.sign(t["grad"])).float().mean()), 3)
                 for g, t in zip(grads, trace)]
        with torch.no_grad():
            asr_cpu = float((O.logits_of(surrogate.cpu(), torch.from_numpy(u8_cpu).permute(0, 3, 1, 2).float() / 255).argmax(1) != label).float().mean())
            asr_gpu = float((O.logits_of(surrogate.cpu(), torch.from_numpy(u8).permute(0, 3, 1, 2).float() / 255).argmax(1) != label).float().mean())
        out["%s_gain%.2f" % (name, gain)] = {
            "uint8_mismatch_pct": round(float((u8 != u8_cpu).mean()) * 100, 3),
            "grad_sign_flips_pct_by_iteration": flips, "asr_surrogate_cpu": asr_cpu, "asr_surrogate_gpu": asr_gpu}
    print(json.dumps({"images": n, "attack": "mifgsm", "results": out}))


if __name__ == "__main__":
    main()
