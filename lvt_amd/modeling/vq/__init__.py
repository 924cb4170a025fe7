from .vq_embedding import DVQEmbedding, VQEmbedding

__all__ = ["DVQEmbedding", "VQEmbedding"]
