"""RejectionSampler of the stand-in (draft_probs is always None here): out int32 [B, max_spec_len + 1]
filled with -1.  Greedy rows (all_greedy, or temperature == -1): for position p < n_i write argmax(target_logits[row]);
stop after the first draft != argmax.  Random rows: p = softmax(logits / T); accept the draft iff p[draft] >= u
(u ~ U[0,1) f64: one torch.rand over all draft positions, then the requests that own a generator re-draw theirs with
it — requests without draft tokens draw nothing), else emit argmax(p / q) with p[draft] := 0, q ~ Exp(1) (one
[B, V] f32 exponential_, re-drawn per request with a generator).  If nothing was rejected the bonus token goes to
position n_i.  parse_output drops -1 and ids >= vocab."""
import torch

MAX_SPEC_LEN = 32
PLACEHOLDER_TOKEN_ID = -1
GREEDY_TEMPERATURE = -1


def generate_uniform_probs(num_tokens, num_draft_tokens, generators, device):
    u = torch.rand((num_tokens,), dtype=torch.float64, device=device)
    start = 0
    for i, n in enumerate(num_draft_tokens):
        if n == 0:
            continue
        g = generators.get(i)
        if g is not None:
            u[start:start + n].uniform_(generator=g)
        start += n
    return u


def generate_recovery_noise(batch, vocab, num_draft_tokens, generators, device):
    q = torch.empty((batch, vocab), dtype=torch.float32, device=device)
    q.exponential_()
    for i, g in generators.items():
        if num_draft_tokens[i] > 0:
            q[i].exponential_(generator=g)
    return q


class RejectionSampler(torch.nn.Module):
    calls = 0

    def forward(self, metadata, draft_probs, target_logits, bonus_token_ids, sampling_metadata):
        RejectionSampler.calls += 1
        assert draft_probs is None
        sm = sampling_metadata
        n = metadata.num_draft_tokens
        B = len(n)
        dev = target_logits.device
        out = torch.full((B, max(n) + 1), PLACEHOLDER_TOKEN_ID, dtype=torch.int32, device=dev)
        draft = metadata.draft_token_ids.tolist()
        bonus = bonus_token_ids.reshape(-1).tolist()
        temp = None if sm.all_greedy else sm.temperature.tolist()
        if temp is not None:
            u = generate_uniform_probs(len(draft), n, sm.generators, dev).tolist()
            q = generate_recovery_noise(B, target_logits.shape[-1], n, sm.generators, dev)
        at = 0
        for i in range(B):
            ok = True
            for p in range(n[i]):
                row = target_logits[at + p]
                if temp is None or temp[i] == GREEDY_TEMPERATURE:
                    tok = int(row.float().argmax())
                    out[i, p] = tok
                    ok = draft[at + p] == tok
                else:
                    x = (row / temp[i]).to(row.dtype)
                    prob = torch.softmax(x.float(), dim=-1)
                    if float(prob[draft[at + p]]) >= u[at + p]:
                        out[i, p] = draft[at + p]
                    else:
                        prob = prob.clone()
                        prob[draft[at + p]] = 0.0
                        out[i, p] = int((prob / q[i]).argmax())
                        ok = False
                if not ok:
                    break
            if ok:
                out[i, n[i]] = bonus[i]
            at += n[i]
        return out

    @staticmethod
    def parse_output(output_token_ids: torch.Tensor, vocab_size: int):
        rows = output_token_ids.cpu().tolist()
        return [[t for t in row if t != PLACEHOLDER_TOKEN_ID and t < vocab_size] for row in rows]
