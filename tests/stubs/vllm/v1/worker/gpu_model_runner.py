"""`GPUModelRunner` of the stand-in: vLLM V1's per-worker bookkeeping (requests, input batch, input preparation with
SpecDecodeMetadata, KV-cache binding, stock execute_model) over the toy model.  Only what the ArcticInference
plugin patches or calls is here, under vLLM's names; behaviour is the minimum a test needs (tests/stubs/README.md)."""
import logging
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from vllm.attention.layer import AttentionMetadata
from vllm.config import CompilationLevel, get_current_vllm_config, set_current_vllm_config
from vllm.distributed.parallel_state import get_pp_group, get_tp_group
from vllm.forward_context import set_forward_context
from vllm.model_executor.model_loader import get_model
from vllm.v1.outputs import EMPTY_MODEL_RUNNER_OUTPUT, ModelRunnerOutput, SamplerOutput
from vllm.v1.sample.metadata import SamplingMetadata
from vllm.v1.sample.rejection_sampler import RejectionSampler
from vllm.v1.spec_decode.metadata import SpecDecodeMetadata

logger = logging.getLogger("vllm.v1.worker.gpu_model_runner")


@dataclass
class CachedRequestState:
    req_id: str
    prompt_token_ids: List[int]
    block_ids: List[int]
    num_computed_tokens: int = 0
    output_token_ids: List[int] = field(default_factory=list)
    sampling_params: Any = None           # an object with .temperature (0 = greedy) and .seed, or None (greedy)

    @property
    def num_tokens(self) -> int:
        return len(self.prompt_token_ids) + len(self.output_token_ids)

    def get_token_id(self, idx: int) -> int:
        n = len(self.prompt_token_ids)
        return self.prompt_token_ids[idx] if idx < n else self.output_token_ids[idx - n]


class InputBatch:
    def __init__(self, max_num_reqs: int, max_model_len: int, max_blocks: int, vocab_size: int):
        self.req_ids: List[str] = []
        self.req_id_to_index: Dict[str, int] = {}
        self.token_ids_cpu = np.zeros((max_num_reqs, max_model_len + 64), dtype=np.int32)
        self.num_tokens = np.zeros(max_num_reqs, dtype=np.int32)
        self.num_tokens_no_spec = np.zeros(max_num_reqs, dtype=np.int32)
        self.num_prompt_tokens = np.zeros(max_num_reqs, dtype=np.int32)
        self.num_computed_tokens_cpu = np.zeros(max_num_reqs, dtype=np.int32)
        self.block_table = np.zeros((max_num_reqs, max_blocks), dtype=np.int32)
        self.num_blocks = np.zeros(max_num_reqs, dtype=np.int32)
        self.generators: Dict[int, Any] = {}
        self.pooling_params: Dict[str, Any] = {}
        self.vocab_size = vocab_size
        self.temperature_cpu = np.full(max_num_reqs, -1.0, dtype=np.float32)     # -1: greedy (vLLM's GREEDY_TEMPERATURE)
        self.device = "cpu"
        self.extra_sampling = {}           # tests set metadata fields vLLM would fill from SamplingParams (e.g. min_tokens)

    @property
    def sampling_metadata(self) -> SamplingMetadata:
        """What _make_sampling_metadata builds for the current batch."""
        n = len(self.req_ids)
        t = self.temperature_cpu[:n]
        all_greedy = bool((t < 0).all())
        kw = dict(temperature=None if all_greedy else torch.from_numpy(t.copy()).to(self.device), all_greedy=all_greedy,
                  all_random=bool((t >= 0).all()) and n > 0, generators=self.generators, logit_bias=[None] * n)
        kw.update(self.extra_sampling)
        return SamplingMetadata(**kw)

    def add(self, st: CachedRequestState) -> None:
        i = len(self.req_ids)
        self.req_ids.append(st.req_id)
        self.req_id_to_index[st.req_id] = i
        toks = st.prompt_token_ids + st.output_token_ids
        self.token_ids_cpu[i, :len(toks)] = toks
        self.num_tokens[i] = self.num_tokens_no_spec[i] = len(toks)
        self.num_prompt_tokens[i] = len(st.prompt_token_ids)
        self.num_computed_tokens_cpu[i] = st.num_computed_tokens
        self.block_table[i, :len(st.block_ids)] = st.block_ids
        self.num_blocks[i] = len(st.block_ids)
        sp = st.sampling_params
        temp = float(getattr(sp, "temperature", 0.0) or 0.0)
        self.temperature_cpu[i] = temp if temp > 0 else -1.0
        self.generators.pop(i, None)
        if temp > 0 and getattr(sp, "seed", None) is not None:
            g = getattr(st, "generator", None)        # a resumed request keeps its generator
            if g is None:
                g = st.generator = torch.Generator(device=self.device).manual_seed(int(sp.seed))
            self.generators[i] = g

    def remove(self, req_id: str) -> None:
        """Drops a request and closes the gap (vLLM condenses the batch the same way: rows move up)."""
        i = self.req_id_to_index.pop(req_id)
        last = len(self.req_ids) - 1
        for arr in (self.token_ids_cpu, self.num_tokens, self.num_tokens_no_spec, self.num_prompt_tokens,
                    self.num_computed_tokens_cpu, self.block_table, self.num_blocks, self.temperature_cpu):
            arr[i:last] = arr[i + 1:last + 1]
        self.generators = {(k if k < i else k - 1): g for k, g in self.generators.items() if k != i}
        del self.req_ids[i]
        self.req_id_to_index = {r: k for k, r in enumerate(self.req_ids)}


class Sampler(torch.nn.Module):
    """Greedy rows: arg-max.  Random rows: softmax(logits / T) sampled as probs.div_(q).argmax() with q ~ Exp(1) — one
    [B, V] exponential_ from the default generator, rows of seeded requests re-drawn with their own generator."""
    calls = 0

    def forward(self, logits, sampling_metadata):
        Sampler.calls += 1
        sm = sampling_metadata
        greedy = logits.argmax(dim=-1)
        if sm.all_greedy:
            return SamplerOutput(sampled_token_ids=greedy.view(-1, 1).to(torch.int32), logprobs_tensors=None)
        temp = sm.temperature.to(logits.device)
        probs = torch.softmax(logits.float() / torch.where(temp < 0, torch.ones_like(temp), temp).unsqueeze(1), dim=-1)
        q = torch.empty_like(probs)
        q.exponential_()
        for i, g in sm.generators.items():
            q[i].exponential_(generator=g)
        rand = probs.div_(q).argmax(dim=-1)
        out = torch.where(temp < 0, greedy, rand)
        return SamplerOutput(sampled_token_ids=out.view(-1, 1).to(torch.int32), logprobs_tensors=None)


class GPUModelRunner:
    def __init__(self, vllm_config, device):
        self.vllm_config = vllm_config
        self.model_config = vllm_config.model_config
        self.parallel_config = vllm_config.parallel_config
        self.scheduler_config = vllm_config.scheduler_config
        self.cache_config = vllm_config.cache_config
        self.compilation_config = vllm_config.compilation_config
        self.speculative_config = vllm_config.speculative_config
        self.device = torch.device(device)
        self.max_model_len = self.model_config.max_model_len
        self.max_num_tokens = self.scheduler_config.max_num_batched_tokens
        self.max_num_reqs = self.scheduler_config.max_num_seqs
        self.hidden_size = self.model_config.get_hidden_size()
        self.block_size = self.cache_config.block_size
        self.is_multimodal_model = self.model_config.is_multimodal_model
        self.uses_mrope = self.model_config.uses_mrope
        self.use_aux_hidden_state_outputs = False
        self.use_cuda_graph = self.compilation_config.level == CompilationLevel.PIECEWISE
        self.cudagraph_batch_sizes = sorted(self.compilation_config.cudagraph_capture_sizes)
        self.full_cuda_graph = self.compilation_config.full_cuda_graph
        self.input_ids = torch.zeros(self.max_num_tokens + 64, dtype=torch.int64, device=self.device)
        self.positions = torch.zeros(self.max_num_tokens + 64, dtype=torch.int64, device=self.device)
        self.inputs_embeds = None
        self.requests: Dict[str, CachedRequestState] = {}
        max_blocks = (self.max_model_len + self.block_size - 1) // self.block_size + 4
        self.input_batch = InputBatch(self.max_num_reqs, self.max_model_len, max_blocks, self.model_config.get_vocab_size())
        self.input_batch.device = self.device
        self.sampler = Sampler()
        self.kv_caches: List[torch.Tensor] = []
        self.dummy_runs: List[tuple] = []      # (num_tokens, which model, tp world size, is_profile) — tests read it
        self.execute_dummy_runs = False        # tests on the GPU set it: _dummy_run then really runs the model
        # full-graph mode: everything attention reads lives in persistent buffers (a captured graph holds their addresses)
        self._md = None
        if self.full_cuda_graph:
            dev = self.device
            self._md = {"qsl": torch.zeros(self.max_num_reqs + 1, dtype=torch.int32, device=dev),
                        "seq": torch.zeros(self.max_num_reqs, dtype=torch.int32, device=dev),
                        "bt": torch.zeros(self.max_num_reqs, max_blocks, dtype=torch.int32, device=dev),
                        "slots": torch.full((self.max_num_tokens + 64,), -1, dtype=torch.int64, device=dev)}
        if self.speculative_config is not None:
            # vLLM's constructor knows its own methods only (the plugin hides "arctic" / "suffix" from it)
            if self.speculative_config.method != "ngram":
                raise ValueError(f"Unknown speculative decoding method: {self.speculative_config.method}")
            self.drafter = None
            self.rejection_sampler = RejectionSampler()

    # ---- state ------------------------------------------------------------------------------------
    def _update_states(self, scheduler_output) -> None:
        ib = self.input_batch
        for rid in scheduler_output.finished_req_ids:
            self.requests.pop(rid, None)
            if rid in ib.req_id_to_index:
                ib.remove(rid)
        # requests that are not scheduled this step leave the batch (preempted / waiting); they come back as "resumed"
        for rid in list(ib.req_ids):
            if rid not in scheduler_output.num_scheduled_tokens:
                ib.remove(rid)
        for new in scheduler_output.scheduled_new_reqs:
            st = CachedRequestState(new.req_id, list(new.prompt_token_ids), list(new.block_ids), new.num_computed_tokens,
                                    sampling_params=getattr(new, "sampling_params", None))
            self.requests[new.req_id] = st
            ib.add(st)
        for c in scheduler_output.scheduled_cached_reqs:
            st = self.requests[c.req_id]
            st.num_computed_tokens = c.num_computed_tokens
            if c.resumed_from_preemption:
                st.block_ids = list(c.new_block_ids)
                ib.add(st)
            else:
                st.block_ids.extend(c.new_block_ids)
                i = ib.req_id_to_index[c.req_id]
                ib.block_table[i, ib.num_blocks[i]:ib.num_blocks[i] + len(c.new_block_ids)] = c.new_block_ids
                ib.num_blocks[i] += len(c.new_block_ids)
            i = ib.req_id_to_index[c.req_id]
            ib.num_computed_tokens_cpu[i] = c.num_computed_tokens
        for rid, spec in scheduler_output.scheduled_spec_decode_tokens.items():
            i = ib.req_id_to_index[rid]
            start = int(ib.num_tokens_no_spec[i])
            ib.token_ids_cpu[i, start:start + len(spec)] = spec
            ib.num_tokens[i] = start + len(spec)

    def _prepare_inputs(self, scheduler_output):
        ib = self.input_batch
        B = len(ib.req_ids)
        n_sched = np.array([scheduler_output.num_scheduled_tokens[r] for r in ib.req_ids], dtype=np.int32)
        T = int(n_sched.sum())
        qsl = np.zeros(B + 1, dtype=np.int32)
        np.cumsum(n_sched, out=qsl[1:])
        req_of = np.repeat(np.arange(B), n_sched)
        pos = ib.num_computed_tokens_cpu[:B][req_of] + (np.arange(T) - qsl[:-1][req_of])
        toks = ib.token_ids_cpu[req_of, pos]
        slots = ib.block_table[req_of, pos // self.block_size].astype(np.int64) * self.block_size + pos % self.block_size
        seq_lens = (ib.num_computed_tokens_cpu[:B] + n_sched).astype(np.int32)
        dev = self.device
        self.input_ids[:T] = torch.from_numpy(toks.astype(np.int64)).to(dev)
        self.positions[:T] = torch.from_numpy(pos.astype(np.int64)).to(dev)
        if self._md is None:
            meta = AttentionMetadata(num_actual_tokens=T, max_query_len=int(n_sched.max()),
                                     query_start_loc=torch.from_numpy(qsl).to(dev), max_seq_len=int(seq_lens.max()),
                                     seq_lens=torch.from_numpy(seq_lens).to(dev),
                                     block_table=torch.from_numpy(ib.block_table[:B].copy()).to(dev),
                                     slot_mapping=torch.from_numpy(slots).to(dev), query_start_loc_cpu=qsl, seq_lens_cpu=seq_lens)
        else:
            meta = self._fill_persistent_metadata(qsl, seq_lens, ib.block_table[:B], slots, int(n_sched.max()))
        attn_metadata = {name: meta for name in self.compilation_config.static_forward_context}
        n_draft = np.array([len(scheduler_output.scheduled_spec_decode_tokens.get(r, ())) for r in ib.req_ids], dtype=np.int32)
        if n_draft.sum() == 0:
            logits_indices = torch.from_numpy(qsl[1:] - 1).to(dev).long()
            spec = None
        else:
            n_samp = n_draft + 1
            cu_samp = np.cumsum(n_samp)
            # sampled rows of the model output: the last n_draft + 1 tokens of every request
            samp_req = np.repeat(np.arange(B), n_samp)
            within = np.arange(int(cu_samp[-1])) - np.repeat(cu_samp - n_samp, n_samp)
            li = qsl[1:][samp_req] - n_samp[samp_req] + within
            is_bonus = np.zeros(len(li), dtype=bool)
            is_bonus[cu_samp - 1] = True
            target = np.nonzero(~is_bonus)[0]
            draft_ids = toks[li[target] + 1]                     # the token AFTER a verifying row is the draft it verifies
            spec = SpecDecodeMetadata(draft_token_ids=torch.from_numpy(draft_ids.astype(np.int32)).to(dev),
                                      num_draft_tokens=n_draft.tolist(),
                                      cu_num_draft_tokens=torch.from_numpy(np.cumsum(n_draft).astype(np.int32)).to(dev),
                                      target_logits_indices=torch.from_numpy(target.astype(np.int32)).to(dev),
                                      bonus_logits_indices=torch.from_numpy((cu_samp - 1).astype(np.int32)).to(dev),
                                      logits_indices=torch.from_numpy(li.astype(np.int32)).to(dev))
            logits_indices = spec.logits_indices.long()
        # vLLM 0.9.2's FlashAttention builder: a batch can run inside a full graph iff it is decode-only
        attention_cuda_graphs = bool(self.full_cuda_graph and int(n_sched.max()) == 1)
        return attn_metadata, attention_cuda_graphs, logits_indices, spec, n_sched

    def _fill_persistent_metadata(self, qsl, seq_lens, block_table, slots, max_query_len, max_seq_len=None):
        """Writes a batch into the persistent buffers and neutralises what lies behind it (requests without tokens, slot -1)
        — a graph captured for more requests / tokens than this step has reads those entries too."""
        md, B, T = self._md, len(seq_lens), len(slots)
        md["qsl"][:B + 1].copy_(torch.from_numpy(np.ascontiguousarray(qsl)), non_blocking=True)
        md["qsl"][B + 1:].fill_(int(qsl[-1]))
        md["seq"][:B].copy_(torch.from_numpy(np.ascontiguousarray(seq_lens)))
        md["seq"][B:].fill_(0)
        md["bt"][:B].copy_(torch.from_numpy(np.ascontiguousarray(block_table)))
        md["slots"][:T].copy_(torch.from_numpy(np.ascontiguousarray(slots)))
        md["slots"][T:].fill_(-1)
        return AttentionMetadata(num_actual_tokens=T, max_query_len=max_query_len, query_start_loc=md["qsl"][:B + 1],
                                 max_seq_len=int(max_seq_len if max_seq_len is not None else seq_lens.max()),
                                 seq_lens=md["seq"][:B], block_table=md["bt"][:B], slot_mapping=md["slots"][:T],
                                 query_start_loc_cpu=qsl, seq_lens_cpu=seq_lens)

    # ---- the stock step (what the plugin's execute_model replaces) ----------------------------------
    @torch.inference_mode()
    def execute_model(self, scheduler_output, intermediate_tensors=None):
        self._update_states(scheduler_output)
        if not scheduler_output.total_num_scheduled_tokens:
            return EMPTY_MODEL_RUNNER_OUTPUT
        attn_metadata, _, logits_indices, spec, _ = self._prepare_inputs(scheduler_output)
        n = scheduler_output.total_num_scheduled_tokens
        with set_forward_context(attn_metadata, self.vllm_config, num_tokens=n):
            hidden = self.model(input_ids=self.input_ids[:n], positions=self.positions[:n], intermediate_tensors=None,
                                inputs_embeds=None)
        sample_hidden = hidden[logits_indices]
        logits = self.model.compute_logits(sample_hidden, None)
        sm = self.input_batch.sampling_metadata
        if spec is None:
            out = self.sampler(logits=logits, sampling_metadata=sm)
        else:
            out = self.sampler(logits=logits[spec.bonus_logits_indices.long()], sampling_metadata=sm)
            out.sampled_token_ids = self.rejection_sampler(spec, None, logits[spec.target_logits_indices.long()],
                                                           out.sampled_token_ids, sm)
        ids = out.sampled_token_ids
        valid = ids.tolist() if ids.shape[-1] == 1 else self.rejection_sampler.parse_output(ids, self.input_batch.vocab_size)
        ib = self.input_batch
        for i, req_id in enumerate(ib.req_ids):
            st = self.requests[req_id]
            if st.num_computed_tokens + scheduler_output.num_scheduled_tokens[req_id] < st.num_tokens:
                valid[i].clear()
        for i, toks in enumerate(valid):
            if not toks:
                continue
            start = int(ib.num_tokens_no_spec[i])
            ib.token_ids_cpu[i, start:start + len(toks)] = toks
            ib.num_tokens_no_spec[i] = ib.num_tokens[i] = start + len(toks)
            self.requests[ib.req_ids[i]].output_token_ids.extend(toks)
        spec_ids = None
        if self.speculative_config:
            spec_ids = self.propose_draft_token_ids(scheduler_output, valid, sm, hidden, sample_hidden, None, spec, attn_metadata)
        return ModelRunnerOutput(req_ids=ib.req_ids, req_id_to_index=ib.req_id_to_index, sampled_token_ids=valid,
                                 spec_token_ids=spec_ids, logprobs=None, prompt_logprobs_dict={})

    def propose_draft_token_ids(self, scheduler_output, sampled_token_ids, sampling_metadata, hidden_states,
                                sample_hidden_states, aux_hidden_states, spec_decode_metadata, attn_metadata):
        """vLLM's 8-argument form.  "ngram": repeat the last token twice (enough to be a recognisable proposer)."""
        return [[s[-1], s[-1]] if s else [] for s in sampled_token_ids]

    # ---- model / cache / graphs -----------------------------------------------------------------------
    def load_model(self) -> None:
        self.model = get_model(vllm_config=self.vllm_config)
        drafter = getattr(self, "drafter", None)
        if drafter is not None and hasattr(drafter, "load_model"):
            drafter.load_model(self.model)

    def initialize_kv_cache(self, kv_cache_config) -> None:
        """kv_cache_config: (num_blocks, dtype).  One [2, num_blocks, block_size, Hkv, D] tensor per attention layer of
        THIS runner's model, bound through the static forward context (layer name -> module)."""
        num_blocks, dtype = kv_cache_config
        self.kv_caches = []
        for name, mod in self.compilation_config.static_forward_context.items():
            t = torch.zeros(2, num_blocks, self.block_size, mod.num_kv_heads, mod.head_size, dtype=dtype, device=self.device)
            mod.kv_cache = [t]
            self.kv_caches.append(t)

    def profile_run(self) -> None:
        self._dummy_run(self.max_num_tokens, is_profile=True)

    def _dummy_run(self, num_tokens: int, capture_attn_cudagraph: bool = False, skip_eplb: bool = False,
                   is_profile: bool = False):
        which = "shift" if (getattr(self, "shift_model", None) is not None and self.model is self.shift_model) else "base"
        self.dummy_runs.append((num_tokens, which, get_tp_group().world_size, is_profile))
        if not self.execute_dummy_runs:
            return None
        # vLLM's dummy batch: min(num_tokens, max_num_reqs) requests sharing the tokens evenly (the last takes the rest)
        num_reqs = min(num_tokens, self.max_num_reqs)
        per = np.full(num_reqs, num_tokens // num_reqs, dtype=np.int32)
        per[-1] += num_tokens % num_reqs
        attn_metadata = None
        if capture_attn_cudagraph:
            assert self._md is not None, "capture_attn_cudagraph needs full_cuda_graph"
            qsl = np.zeros(num_reqs + 1, dtype=np.int32)
            np.cumsum(per, out=qsl[1:])
            meta = self._fill_persistent_metadata(qsl, per.copy(), np.zeros((num_reqs, self._md["bt"].shape[1]), np.int32),
                                                  np.full(num_tokens, -1, np.int64), int(per.max()),
                                                  max_seq_len=self.max_model_len)
            attn_metadata = {name: meta for name in self.compilation_config.static_forward_context}
        with set_forward_context(attn_metadata, self.vllm_config, num_tokens=num_tokens):
            return self.model(input_ids=self.input_ids[:num_tokens], positions=self.positions[:num_tokens],
                              intermediate_tensors=None, inputs_embeds=None)

    def capture_model(self) -> None:
        for n in reversed(self.cudagraph_batch_sizes):
            self._dummy_run(n)

    # ---- small hooks the real runner has ---------------------------------------------------------------
    def get_dp_padding(self, num_tokens: int):
        return 0, None

    def maybe_setup_kv_connector(self, scheduler_output) -> None:
        pass

    def maybe_wait_for_kv_save(self) -> None:
        pass

    def get_finished_kv_transfers(self, scheduler_output):
        return None, None

    def kv_connector_no_forward(self, scheduler_output):
        return EMPTY_MODEL_RUNNER_OUTPUT

    def eplb_step(self) -> None:
        pass

    def _get_prompt_logprobs_dict(self, hidden_states, scheduler_output):
        return {}

    def apply_grammar_bitmask(self, scheduler_output, logits) -> None:
        pass

    def _get_nans_in_logits(self, logits):
        return {}
