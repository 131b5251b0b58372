"""RejectionSampler of the stand-in, greedy rows only: out int32 [B, max_spec_len + 1] filled with
-1; for position p < n_i write argmax(target_logits[row]); stop after the first draft != argmax; if none was rejected
write the bonus token at position n_i.  parse_output drops -1 and ids >= vocab."""
import torch

MAX_SPEC_LEN = 32
PLACEHOLDER_TOKEN_ID = -1


class RejectionSampler(torch.nn.Module):
    calls = 0

    def forward(self, metadata, draft_probs, target_logits, bonus_token_ids, sampling_metadata):
        RejectionSampler.calls += 1
        assert draft_probs is None and sampling_metadata.all_greedy
        n = metadata.num_draft_tokens
        B = len(n)
        out = torch.full((B, max(n) + 1), PLACEHOLDER_TOKEN_ID, dtype=torch.int32, device=target_logits.device)
        arg = target_logits.argmax(dim=-1).tolist()
        draft = metadata.draft_token_ids.tolist()
        bonus = bonus_token_ids.reshape(-1).tolist()
        at = 0
        for i in range(B):
            ok = True
            for p in range(n[i]):
                out[i, p] = arg[at + p]
                if draft[at + p] != arg[at + p]:
                    ok = False
                    break
            if ok:
                out[i, n[i]] = bonus[i]
            at += n[i]
        return out

    @staticmethod
    def parse_output(output_token_ids: torch.Tensor, vocab_size: int):
        rows = output_token_ids.cpu().tolist()
        return [[t for t in row if t != PLACEHOLDER_TOKEN_ID and t < vocab_size] for row in rows]
