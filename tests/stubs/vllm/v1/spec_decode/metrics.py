class SpecDecodingStats:
    def __init__(self, num_spec_tokens: int):
        self.num_spec_tokens = num_spec_tokens
        self.num_drafts = self.num_draft_tokens = self.num_accepted_tokens = 0
        self.num_accepted_tokens_per_pos = [0] * num_spec_tokens

    def observe_draft(self, num_draft_tokens: int, num_accepted_tokens: int):
        self.num_drafts += 1
        self.num_draft_tokens += num_draft_tokens
        self.num_accepted_tokens += num_accepted_tokens
        assert num_accepted_tokens <= self.num_spec_tokens
        for i in range(num_accepted_tokens):
            self.num_accepted_tokens_per_pos[i] += 1


class SpecDecodingLogging:
    def __init__(self):
        self.num_drafts, self.accepted_tokens_per_pos_lists, self.logged = [], [], 0

    def observe(self, stats: SpecDecodingStats):
        self.num_drafts.append(stats.num_drafts)
        self.accepted_tokens_per_pos_lists.append(list(stats.num_accepted_tokens_per_pos))

    def log(self, log_fn=None):
        import numpy as np
        np.sum(np.array(self.accepted_tokens_per_pos_lists), axis=0)     # ragged lists would raise here
        self.logged += 1
