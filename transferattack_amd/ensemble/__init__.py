"""Multi-surrogate attacks on EnsembleModel / the sharded ensembles of transferattack_amd.dist (registry: attack_zoo)."""
