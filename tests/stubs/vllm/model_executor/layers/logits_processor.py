import torch


class LogitsProcessor(torch.nn.Module):
    def __init__(self, vocab_size: int, org_vocab_size: int = None, scale: float = 1.0, logits_as_input: bool = False,
                 soft_cap=None):
        super().__init__()
        self.scale, self.org_vocab_size = scale, org_vocab_size or vocab_size

    def forward(self, lm_head, hidden_states, sampling_metadata=None, embedding_bias=None):
        return (hidden_states @ lm_head.weight.T).float()[..., :self.org_vocab_size] * self.scale
