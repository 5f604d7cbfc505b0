from ccedit_amd.sampling import EpsScaling  # noqa: F401
