class MLPSpeculatorConfig:
    model_type = "mlp_speculator"

    def __init__(self, vocab_size: int = 32000, emb_dim: int = 4096, inner_dim: int = 0, n_predict: int = 3, top_k_tokens_per_head=None,
                 n_candidates: int = 5, tie_weights: bool = False, scale_input: bool = False, **kwargs):
        self.vocab_size, self.emb_dim, self.inner_dim, self.n_predict = vocab_size, emb_dim, inner_dim, n_predict
        self.tie_weights, self.scale_input = tie_weights, scale_input
        self.num_lookahead_tokens = n_predict
        for k, v in kwargs.items():
            setattr(self, k, v)
