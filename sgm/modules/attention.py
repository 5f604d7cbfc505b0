from ccedit_amd.network import (BasicTransformerBlock, BasicTransformerSingleLayerBlock, CrossAttention,  # noqa: F401
                                FeedForward, SpatialTransformer, SpatialTransformer3D, SpatialTransformer3DCA)
