"""Embedding / LM head of the stand-in (replicated, vLLM's names)."""
import torch

DEFAULT_VOCAB_PADDING_SIZE = 64


class VocabParallelEmbedding(torch.nn.Module):
    def __init__(self, num_embeddings: int, embedding_dim: int, params_dtype=None, org_num_embeddings=None,
                 padding_size: int = DEFAULT_VOCAB_PADDING_SIZE, quant_config=None, prefix: str = ""):
        super().__init__()
        from vllm.model_executor.layers.linear import _param
        self.weight = _param(num_embeddings, embedding_dim, params_dtype)

    def forward(self, input_ids):
        return self.weight[input_ids]


class ParallelLMHead(VocabParallelEmbedding):
    def __init__(self, num_embeddings: int, embedding_dim: int, bias: bool = False, params_dtype=None, org_num_embeddings=None,
                 padding_size: int = DEFAULT_VOCAB_PADDING_SIZE, quant_config=None, prefix: str = ""):
        super().__init__(num_embeddings, embedding_dim, params_dtype, org_num_embeddings, padding_size, quant_config, prefix)

    def tie_weights(self, embed_tokens: VocabParallelEmbedding):
        return embed_tokens
