from ccedit_amd.sampling import EpsWeighting  # noqa: F401
