from ccedit_amd.network import (Downsample, Downsample3D, ResBlock, ResBlock3D, TimestepEmbedSequential,  # noqa: F401
                                UNetModel, UNetModel3D, Upsample3D)
