"""Vector quantisation: one codebook (`VQEmbedding`) and the product quantiser over channel groups
(`DVQEmbedding`), both backed by the HIP nearest / gather / EMA kernels."""
from . import vq_embedding as _impl

VQEmbedding = _impl.VQEmbedding
DVQEmbedding = _impl.DVQEmbedding
SingleVQEmbedding = _impl.SingleVQEmbedding

__all__ = ("VQEmbedding", "DVQEmbedding", "SingleVQEmbedding")
