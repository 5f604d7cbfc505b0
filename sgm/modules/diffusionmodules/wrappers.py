from ccedit_amd.network import IdentityWrapper, OpenAIWrapperControlLDM3DTV2V  # noqa: F401

OPENAIUNETWRAPPERCONTROLLDM3DTV2V = "sgm.modules.diffusionmodules.wrappers.OpenAIWrapperControlLDM3DTV2V"
