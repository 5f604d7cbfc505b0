from ccedit_amd.engine import VideoDiffusionEngineTV2V  # noqa: F401
