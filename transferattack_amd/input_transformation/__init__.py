"""Input-transformation attacks: HIP transforms behind the reference's ``transform`` hook (registry: attack_zoo)."""
