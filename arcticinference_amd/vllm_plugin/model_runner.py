"""vLLM-side glue of the spec-decode hot path (mirror of
/root/reference/arctic_inference/vllm/model_runner.py:526-744 and plugins.py:37-63), written against vLLM
0.9.2's V1 GPUModelRunner.  vLLM is not present in the build container: this module is exercised only
through its vLLM-free parts (runner_logic.py, the engine); everything here is constructed lazily.

Differences from the reference, all on the MI355X side of the boundary:
  * suffix proposals for the whole batch are ONE device round trip (SuffixCache.speculate_batch) instead of a
    Python loop of host tree walks;
  * the LSTM draft consumes `last_token` / `hidden_index` produced on the device by the acceptance kernel
    when the runner uses ops.rejection_sample; when vLLM's own RejectionSampler produced the tokens the
    reference's index arithmetic (arctic_proposer.py:133-147) is applied unchanged.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from .runner_logic import MAX_SPEC_LEN, arctic_max_spec_tokens, merge_proposals, min_suffix_score, suffix_query


class ArcticProposer:
    """Interface of the reference's ArcticProposer (spec_dec/arctic_proposer.py:30-166) over the HIP speculator."""

    def __init__(self, speculator):
        self.model = speculator
        self.input_hidden_dim = speculator.input_hidden_dim

    def prepare_hidden_states(self, sample_hidden_states: torch.Tensor, sampled_token_ids, spec_decode_metadata):
        """Returns (hidden_states, hidden_index): the gather itself is fused into the draft kernel."""
        assert sample_hidden_states.shape[-1] == self.input_hidden_dim, "hidden_states shape mismatch"
        sampled = np.asarray(sampled_token_ids)
        if sampled.shape[-1] == 1:
            return sample_hidden_states, None
        gen_lens = (sampled != -1).sum(axis=1)
        n = np.asarray(spec_decode_metadata.num_draft_tokens) + 1
        idx = (gen_lens - 1) + np.cumsum(n) - n           # arctic_proposer.py:138-145
        return sample_hidden_states, torch.from_numpy(idx.astype(np.int32)).to(sample_hidden_states.device)

    def propose(self, context_token_ids, previous_hidden_states, num_predict_tokens: int, hidden_index=None):
        assert num_predict_tokens > 0
        ids = torch.as_tensor(np.asarray(context_token_ids), device=previous_hidden_states.device)
        out = self.model.generate_proposals(ids, previous_hidden_states, num_predict_tokens, hidden_index=hidden_index)
        return out.cpu().numpy()


def ArcticLSTMSpeculatorForVllm(*, vllm_config, prefix: str = ""):
    """Model-registry constructor with the reference's signature (arctic_speculator.py:414)."""
    from ..speculator import ArcticLSTMSpeculator, LSTMSpeculatorConfig
    hf = vllm_config.model_config.hf_config
    cfg = LSTMSpeculatorConfig(vocab_size=hf.vocab_size, input_hidden_dim=hf.input_hidden_dim, inner_dim=hf.inner_dim,
                               emb_dim=hf.emb_dim, proj_dim=hf.proj_dim, n_predict=hf.n_predict,
                               num_lookahead_tokens=hf.num_lookahead_tokens, tie_weights=hf.tie_weights,
                               tie_lstm_embs=hf.tie_lstm_embs, scale_input=hf.scale_input,
                               method=getattr(hf, "method", "sum_rnn"))
    from vllm.distributed import parallel_state
    sp = getattr(parallel_state, "_SP", None)
    tp = parallel_state._TP
    grp = sp if (sp is not None and sp.world_size > tp.world_size) else tp   # SpeculatorTPInit, vocab_parallel_embedding.py:20-35
    return ArcticLSTMSpeculator(cfg, max_num_seqs=vllm_config.scheduler_config.max_num_seqs, tp_size=grp.world_size,
                                tp_rank=tp.rank % grp.world_size, tp_group=grp.device_group)


def ArcticMLPSpeculatorForVllm(*, vllm_config, prefix: str = ""):
    """Model-registry constructor with the reference's signature (arctic_speculator.py:112)."""
    from ..speculator import ArcticMLPSpeculator, MLPSpeculatorConfig
    hf = vllm_config.model_config.hf_config
    cfg = MLPSpeculatorConfig(vocab_size=hf.vocab_size, emb_dim=hf.emb_dim, inner_dim=hf.inner_dim, n_predict=hf.n_predict,
                              num_lookahead_tokens=hf.num_lookahead_tokens, tie_weights=hf.tie_weights,
                              scale_input=hf.scale_input)
    from vllm.distributed import parallel_state
    sp = getattr(parallel_state, "_SP", None)
    tp = parallel_state._TP
    grp = sp if (sp is not None and sp.world_size > tp.world_size) else tp   # SpeculatorTPInit, vocab_parallel_embedding.py:20-35
    return ArcticMLPSpeculator(cfg, max_num_seqs=vllm_config.scheduler_config.max_num_seqs, tp_size=grp.world_size,
                               tp_rank=tp.rank % grp.world_size, tp_group=grp.device_group)


def build_bootstrap_patches():
    """EngineCoreProc / WorkerBase patches that make the plugin load in every process (plugins.py:37-63)."""
    import vllm.plugins
    from vllm.v1.engine.core import EngineCoreProc
    from vllm.v1.worker.worker_base import WorkerBase

    from ..patching import ArcticPatch

    class EngineCoreProcPatch(ArcticPatch[EngineCoreProc]):
        _orig_run_engine_core = EngineCoreProc.run_engine_core

        @staticmethod
        def run_engine_core(*args, **kwargs):
            vllm.plugins.load_general_plugins()
            return EngineCoreProcPatch._orig_run_engine_core(*args, **kwargs)

    class WorkerBasePatch(ArcticPatch[WorkerBase]):
        _orig_init = WorkerBase.__init__

        def __init__(self, *args, **kwargs):
            build_model_runner_patch().apply_patch()   # after the fork: touches the GPU
            return self._orig_init(*args, **kwargs)

    return [EngineCoreProcPatch, WorkerBasePatch]


_runner_patch = None


def build_model_runner_patch():
    global _runner_patch
    if _runner_patch is not None:
        return _runner_patch
    from vllm.v1.worker.gpu_model_runner import GPUModelRunner

    from ..patching import ArcticPatch
    from ..suffix_cache import SuffixCache, SuffixSpecResult

    class GPUModelRunnerPatch(ArcticPatch[GPUModelRunner]):
        _orig_init = GPUModelRunner.__init__
        _orig_propose_draft_token_ids = getattr(GPUModelRunner, "propose_draft_token_ids", None)

        def __init__(self, vllm_config, *args, **kwargs):
            self._orig_init(vllm_config, *args, **kwargs)
            sc = self.speculative_config
            self._suffix_cache = None
            if sc is not None and (getattr(sc, "enable_suffix_decoding", False) or sc.method == "suffix"):
                self._suffix_cache = SuffixCache(sc.suffix_cache_max_depth)     # model_runner.py:155-156

        def _update_suffix_cache(self, sampled_token_ids: List[List[int]]) -> None:
            seen = set()
            new_ids, new_prompts = [], []
            for i, sampled in enumerate(sampled_token_ids):
                req_id = self.input_batch.req_ids[i]
                seen.add(req_id)
                if not sampled:
                    continue
                if not self._suffix_cache.has_cached_prompt(req_id):
                    index = self.input_batch.req_id_to_index[req_id]
                    n = self.input_batch.num_prompt_tokens[index]
                    new_ids.append(req_id)
                    new_prompts.append(self.input_batch.token_ids_cpu[index, :n])
            if new_ids:
                self._suffix_cache.cache_prompts(new_ids, new_prompts)          # trees built on host threads
            # the per-request update_response loop of the reference (:657-673) as one native call, same order
            upd = [(self.input_batch.req_ids[i], sampled) for i, sampled in enumerate(sampled_token_ids) if sampled]
            if upd:
                import numpy as np
                self._suffix_cache.update_responses(
                    [r for r, _ in upd], np.fromiter((t for _, s in upd for t in s), dtype=np.int32),
                    np.fromiter((len(s) for _, s in upd), dtype=np.int32, count=len(upd)))
            for req_id in self._suffix_cache.cached_prompt_ids():                # model_runner.py:675-678
                if req_id not in seen:
                    self._suffix_cache.evict_prompt(req_id)

        def propose_suffix_draft_token_ids(self, sampled_token_ids, spec_token_ids=None):
            cfg = self.speculative_config
            results = [SuffixSpecResult() for _ in sampled_token_ids]
            ids, pats, kws, where = [], [], [], []
            for i, sampled in enumerate(sampled_token_ids):
                spec_ids = spec_token_ids[i] if spec_token_ids is not None else []
                if not sampled:
                    continue
                start = self.input_batch.num_tokens_no_spec[i]
                end = start + len(sampled)
                if end >= self.max_model_len:
                    self.input_batch.token_ids_cpu[i, start:self.max_model_len] = sampled[:self.max_model_len - start]
                    continue
                self.input_batch.token_ids_cpu[i, start:end] = sampled
                q = suffix_query(self.input_batch.token_ids_cpu[i], end, spec_ids, self.max_model_len,
                                 cfg.suffix_cache_max_depth, cfg.suffix_max_spec_factor, cfg.suffix_max_spec_offset,
                                 cfg.suffix_min_token_prob)
                if q is None:
                    continue
                ids.append(self.input_batch.req_ids[i])
                pats.append(q[0])
                kws.append(q[1])
                where.append(i)
            if ids:
                res = self._suffix_cache.speculate_batch(
                    ids, pats, [k["max_spec_tokens"] for k in kws], [k["max_spec_factor"] for k in kws],
                    [k["max_spec_offset"] for k in kws], [k["min_token_prob"] for k in kws], [True] * len(ids))
                for i, r in zip(where, res):
                    results[i] = r
            return results

        def propose_arctic_draft_token_ids(self, scheduler_output, sampled_token_ids, previous_hidden_states=None,
                                           hidden_index=None):
            last_tokens, ends = [], []
            for i, sampled in enumerate(sampled_token_ids):
                if not sampled:
                    if self.speculative_config.enable_suffix_decoding:
                        return [[]] * len(sampled_token_ids)                     # model_runner.py:616-618
                    req_id = self.input_batch.req_ids[i]
                    st = self.requests[req_id]
                    seq_len = st.num_computed_tokens + scheduler_output.num_scheduled_tokens[req_id]
                    sampled = [st.get_token_id(seq_len)]
                start = self.input_batch.num_tokens_no_spec[i]
                ends.append(start + len(sampled_token_ids[i]))
                last_tokens.append(sampled[-1])
            k = arctic_max_spec_tokens(self.speculative_config.num_speculative_tokens, ends, self.max_model_len)
            if k <= 0:
                return [[] for _ in sampled_token_ids]
            out = self.drafter.propose(last_tokens, previous_hidden_states, k, hidden_index=hidden_index).tolist()
            return [o if s else [] for o, s in zip(out, sampled_token_ids)]

        def propose_draft_token_ids(self, scheduler_output, sampled_token_ids, original_sampled_token_ids,
                                    sampling_metadata, hidden_states, sample_hidden_states, aux_hidden_states,
                                    spec_decode_metadata, attn_metadata):
            sc = self.speculative_config
            if sc and sc.disable_by_batch_size and len(self.input_batch.req_ids) > sc.disable_by_batch_size:
                return [[] for _ in sampled_token_ids]
            suffix_ids = None
            remaining = list(sampled_token_ids)
            if self._suffix_cache is not None:
                results = self.propose_suffix_draft_token_ids(remaining)
                floor = min_suffix_score(sc.method, sc.num_speculative_tokens)
                suffix_ids = []
                for i, r in enumerate(results):
                    if r.score >= floor:
                        remaining[i] = []
                        suffix_ids.append(r.token_ids)
                    else:
                        suffix_ids.append([])
            model_ids = None
            if sc.method in ("arctic", "mlp_speculator"):
                hs, hidx = self.drafter.prepare_hidden_states(sample_hidden_states, original_sampled_token_ids,
                                                              spec_decode_metadata)
                model_ids = self.propose_arctic_draft_token_ids(scheduler_output, remaining, hs, hidx)
            elif sc.method != "suffix" and self._orig_propose_draft_token_ids is not None:
                model_ids = self._orig_propose_draft_token_ids(scheduler_output, remaining, sampling_metadata,
                                                               hidden_states, sample_hidden_states, aux_hidden_states,
                                                               spec_decode_metadata, attn_metadata)
            return merge_proposals(suffix_ids, model_ids)

    _runner_patch = GPUModelRunnerPatch
    return _runner_patch
