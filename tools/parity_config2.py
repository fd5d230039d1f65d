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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
N = int(os.environ.get("TA_PARITY_IMAGES", "32"))


def images(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (n, 3, 224, 224), generator=g, dtype=torch.uint8).float() / 255


def main():
    import fgsm_oracle as O
    import transferattack_amd as ta
    from transferattack_amd import backbones
    from transferattack_amd.utils import quantize_images
    x = images(N, 0)
    surrogate = backbones.create("resnet50", seed=0, verbose=False)
    with torch.no_grad():
        label = O.logits_of(surrogate, x).argmax(1)            # "ground truth" = the surrogate's clean prediction
    t0 = time.time()
    delta_cpu = O.run_attack("mifgsm", surrogate, x, label)
    cpu_s = time.time() - t0
    u8_cpu = O.quantize_u8(x + delta_cpu)
    victims = {"resnet50(surrogate)": surrogate, "resnet18": backbones.create("resnet18", seed=1, verbose=False),
               "vgg16": backbones.create("vgg16", seed=2, verbose=False)}

    def asr(u8):
        xa = torch.from_numpy(u8).permute(0, 3, 1, 2).float() / 255
        out = {}
        with torch.no_grad():
            for name, m in victims.items():
                clean = O.logits_of(m, x).argmax(1)
                out[name] = round(float((O.logits_of(m, xa).argmax(1) != clean).float().mean()) * 100, 2)
        return out

    res = {"images": N, "cpu_oracle_seconds": round(cpu_s, 1), "asr_cpu_oracle": asr(u8_cpu)}
    for tag, env in (("gpu_plain", {"TA_FOLD_BN": "0", "TA_CHANNELS_LAST": "0"}),
                     ("gpu_foldbn_nhwc", {"TA_FOLD_BN": "1", "TA_CHANNELS_LAST": "1"})):
        os.environ.update(env)
        atk = ta.load_attack_class("mifgsm")(model_name="resnet50")
        delta = atk(x, label)
        u8 = quantize_images(x, delta)
        res[tag] = {"uint8_mismatch_pct": round(float((u8 != u8_cpu).mean()) * 100, 4),
                    "images_identical": int((u8.reshape(N, -1) == u8_cpu.reshape(N, -1)).all(1).sum()),
                    "max_abs_level_diff": int(abs(u8.astype(int) - u8_cpu.astype(int)).max()), "asr": asr(u8)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
