from ccedit_amd.network import ControlledUNetModel3DTV2V, ControlNet2D  # noqa: F401
