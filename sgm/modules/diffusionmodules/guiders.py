from ccedit_amd.sampling import IdentityGuider, VanillaCFG, VanillaCFGTV2V  # noqa: F401
