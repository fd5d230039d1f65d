"""Update-rule variants of the FGSM family on the HIP hooks (registry: transferattack_amd.attack_zoo)."""
