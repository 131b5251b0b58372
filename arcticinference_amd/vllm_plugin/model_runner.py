"""vLLM-side glue of the spec-decode + shift-parallel hot path: the GPUModelRunner patch.

Mirror of /root/reference/arctic_inference/vllm/model_runner.py (same patched method names, arguments and
behaviour; every method cites the lines it replaces), written against vLLM 0.9.2's V1 GPUModelRunner and
routed to the MI355X kernels of this package:

  * acceptance: all-greedy batches go through ops.rejection_sample on the model's [T, V] logits in place
    (SpecDecodeMetadata.target_logits_indices / bonus_logits_indices are row indices into it: no gathered
    copies, no separate bonus sampler launch), which also leaves `last_token` and the hidden-state row of
    every request on the device (arctic_proposer.py:133-147 fused);
  * the draft model is enqueued right behind the acceptance, before the step's host sync, whenever the
    previous step used it (see HotPathEngine.step for the rule); its tokens are dropped if suffix decoding
    takes a request of the step (the reference's behaviour, :616-618);
  * suffix proposals for the whole batch are ONE device round trip (SuffixCache.speculate_batch) instead of a
    Python loop of host tree walks; prompt trees are built on a host thread while the prompt is prefilled;
  * verify attention: UlyssesAttentionPatch (vllm_plugin/ulysses.py) reads the step geometry published here
    (step_context) and calls aic_verify_attention_ex.

vLLM is not installed in the build container or on the GPU box: tests/stubs/vllm (signatures of the patched
classes only) stands in for it in tests/test_vllm_plugin*.py, which drive every method below.
"""
from __future__ import annotations

import contextlib
import copy
import logging
import time
from typing import Any, List, Optional

import numpy as np
import torch

from . import step_context
from .runner_logic import (INDEXING_REFERENCE, MAX_SPEC_LEN, arctic_max_spec_tokens, hip_acceptance_kind, merge_proposals,
                           min_suffix_score, proposal_end_index, proposal_indexing, rewrite_sampled, suffix_query)

logger = logging.getLogger(__name__)

ARCTIC_METHODS = ("arctic", "suffix", "mlp_speculator")

# ---------------------------------------------------------------------------------------------------
# shift-parallel mode switch (model_runner.py:54-87)
# ---------------------------------------------------------------------------------------------------
SP_TP_MODE: Optional[bool] = None


def is_shift_parallel_mode() -> bool:
    return SP_TP_MODE is True


@contextlib.contextmanager
def set_shift_parallel_mode(mode: Optional[bool]):
    """While active with mode=True, vLLM's tensor-parallel group IS the full SP x TP group (`_TP := _SP_TP`), so a
    model built or run inside it is the TP = SP*TP replica; mode=False pins the original TP group (the Ulysses
    model inside a shift-capable runner); None leaves everything alone.  Re-entrant; restores on exit."""
    if mode is None:
        yield
        return
    from vllm.distributed import parallel_state
    global SP_TP_MODE
    if not is_shift_parallel_mode():
        # first switch away from plain mode: remember the TP group vLLM built
        assert not getattr(parallel_state, "_TP_STATE_PATCHED", False)
        parallel_state._ORIG_TP = parallel_state._TP
    saved_mode, saved_tp = SP_TP_MODE, parallel_state.get_tp_group()
    SP_TP_MODE = mode
    parallel_state._TP = parallel_state._SP_TP if mode else parallel_state._ORIG_TP
    try:
        yield
    finally:
        SP_TP_MODE = saved_mode
        parallel_state._TP = saved_tp


# ---------------------------------------------------------------------------------------------------
# random draws of the acceptance step, as vLLM's rejection sampler makes them (vllm==0.9.2, v1/sample/rejection_sampler.py,
# recalled: generate_uniform_probs / sample_recovered_tokens; call site model_runner.py:405-411)
# ---------------------------------------------------------------------------------------------------
def draw_uniform_probs(num_tokens: int, num_draft_tokens, generators: dict, device) -> torch.Tensor:
    """u ~ U[0,1), float64, one per draft position: ONE draw over all positions from the default generator, then the
    positions of every request that owns a generator re-drawn with it; a request without draft tokens draws nothing."""
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


def draw_recovery_noise(batch: int, vocab: int, num_draft_tokens, generators: dict, device) -> torch.Tensor:
    """q ~ Exp(1), float32 [batch, vocab], ONE distribution per request: a whole-matrix draw from the default generator,
    then the row of every seeded request that has draft tokens re-drawn with its generator."""
    q = torch.empty((batch, vocab), dtype=torch.float32, device=device)
    q.exponential_()
    for i, g in generators.items():
        if num_draft_tokens[i] > 0:
            q[i].exponential_(generator=g)
    return q


# ---------------------------------------------------------------------------------------------------
# draft-model registry constructors (plugins.py:96-109; ctor signature arctic_speculator.py:112,414)
# ---------------------------------------------------------------------------------------------------
def _speculator_tp_group():
    """SpeculatorTPInit (vocab_parallel_embedding.py:20-35): the draft LM head is sharded over max(TP, SP) ranks."""
    from vllm.distributed import parallel_state
    sp = getattr(parallel_state, "_SP", None)
    tp = parallel_state._TP
    grp = sp if (sp is not None and sp.world_size > tp.world_size) else tp
    return grp.world_size, tp.rank % grp.world_size, (grp.device_group if grp.world_size > 1 else None)


def ArcticLSTMSpeculatorForVllm(*, vllm_config, prefix: str = ""):
    from ..speculator import LSTMSpeculatorConfig, lstm_family_speculator
    hf = vllm_config.model_config.hf_config
    method = getattr(hf, "method", "sum_rnn")          # the reference's default (arctic_speculator.py:441)
    cfg = LSTMSpeculatorConfig(vocab_size=hf.vocab_size, input_hidden_dim=hf.input_hidden_dim, inner_dim=hf.inner_dim,
                               emb_dim=hf.emb_dim, proj_dim=hf.proj_dim, n_predict=hf.n_predict,
                               num_lookahead_tokens=hf.num_lookahead_tokens, tie_weights=hf.tie_weights,
                               tie_lstm_embs=getattr(hf, "tie_lstm_embs", True), scale_input=hf.scale_input, method=method)
    size, rank, group = _speculator_tp_group()
    # "sum_lstm" -> the LSTM kernels; "sum_rnn" -> the same head on the MLP-speculator kernels (stacked stages refused)
    return lstm_family_speculator(cfg, max_num_seqs=vllm_config.scheduler_config.max_num_seqs, tp_size=size, tp_rank=rank,
                                  tp_group=group)


def ArcticMLPSpeculatorForVllm(*, vllm_config, prefix: str = ""):
    from ..speculator import ArcticMLPSpeculator, MLPSpeculatorConfig
    hf = vllm_config.model_config.hf_config
    cfg = MLPSpeculatorConfig(vocab_size=hf.vocab_size, emb_dim=hf.emb_dim, inner_dim=hf.inner_dim, n_predict=hf.n_predict,
                              num_lookahead_tokens=hf.num_lookahead_tokens, tie_weights=hf.tie_weights,
                              scale_input=hf.scale_input)
    size, rank, group = _speculator_tp_group()
    return ArcticMLPSpeculator(cfg, max_num_seqs=vllm_config.scheduler_config.max_num_seqs, tp_size=size, tp_rank=rank,
                               tp_group=group)


def speculator_for_architecture(arch: str):
    """The three names plugins.py:96-109 registers.  "MLPVariantSpeculatorPreTrainedModel" is the LSTM class there."""
    return {"ArcticMLPSpeculatorPreTrainedModel": ArcticMLPSpeculatorForVllm,
            "ArcticLSTMSpeculatorPreTrainedModel": ArcticLSTMSpeculatorForVllm,
            "MLPVariantSpeculatorPreTrainedModel": ArcticLSTMSpeculatorForVllm}[arch]


# ---------------------------------------------------------------------------------------------------
# bootstrap patches (plugins.py:37-63)
# ---------------------------------------------------------------------------------------------------
def build_bootstrap_patches():
    """EngineCoreProc / WorkerBase patches that make the plugin load in every process."""
    import vllm.plugins
    from vllm.v1.engine.core import EngineCoreProc
    from vllm.v1.worker.worker_base import WorkerBase

    from ..patching import ArcticPatch

    class EngineCoreProcPatch(ArcticPatch[EngineCoreProc]):
        _orig_run_engine_core = EngineCoreProc.run_engine_core

        @staticmethod
        def run_engine_core(*args, **kwargs):
            vllm.plugins.load_general_plugins()       # the EngineCore process is spawned: load the plugin there too
            return EngineCoreProcPatch._orig_run_engine_core(*args, **kwargs)

    class WorkerBasePatch(ArcticPatch[WorkerBase]):
        _orig_init = WorkerBase.__init__

        def __init__(self, *args, **kwargs):
            # the runner patch touches the GPU runtime: applied in the worker, after the fork (plugins.py:54-63); this is
            # also the first place the native library is loaded and asked for a device
            from .. import _native
            if _native.lib().aic_device_count() <= 0:
                logger.warning("ArcticInference (MI355X build): no HIP device visible in this worker; kernels will refuse "
                               "to run (AIC_ERR_NO_DEVICE)")
            patch = build_model_runner_patch()
            from vllm.v1.worker.gpu_model_runner import GPUModelRunner
            if "execute_model" not in vars(GPUModelRunner).get("_arctic_patches", {}):
                patch.apply_patch()
            return self._orig_init(*args, **kwargs)

    return [EngineCoreProcPatch, WorkerBasePatch]


# ---------------------------------------------------------------------------------------------------
# GPUModelRunner patch
# ---------------------------------------------------------------------------------------------------
_runner_patch = None


def build_model_runner_patch():
    global _runner_patch
    if _runner_patch is not None:
        return _runner_patch
    from vllm.attention.layer import Attention
    from vllm.config import CompilationLevel
    from vllm.distributed import parallel_state
    from vllm.distributed.kv_transfer import get_kv_transfer_group, has_kv_transfer_group
    from vllm.distributed.parallel_state import get_pp_group, get_tp_group
    from vllm.forward_context import set_forward_context
    from vllm.model_executor.model_loader import get_model
    from vllm.utils import round_up
    from vllm.v1.outputs import EMPTY_MODEL_RUNNER_OUTPUT, ModelRunnerOutput
    from vllm.v1.sample.rejection_sampler import RejectionSampler
    from vllm.v1.worker.gpu_model_runner import GPUModelRunner, logger

    from .. import ops
    from ..patching import ArcticPatch
    from ..suffix_cache import SuffixCache, SuffixSpecResult
    from .arctic_proposer import ArcticProposer

    class GPUModelRunnerPatch(ArcticPatch[GPUModelRunner]):
        _orig_init = GPUModelRunner.__init__
        _orig_initialize_kv_cache = GPUModelRunner.initialize_kv_cache
        _orig_prepare_inputs = GPUModelRunner._prepare_inputs
        _orig_profile_run = GPUModelRunner.profile_run
        _orig_load_model = GPUModelRunner.load_model
        _orig_propose_draft_token_ids = GPUModelRunner.propose_draft_token_ids

        # ---- construction (model_runner.py:99-156) ---------------------------------------------------
        def __init__(self, vllm_config, device):
            pc = vllm_config.parallel_config
            self.use_ulysses = getattr(pc, "ulysses_sequence_parallel_size", 1) > 1
            if self.use_ulysses and vllm_config.compilation_config.pass_config.enable_sequence_parallelism:
                raise ValueError("Ulysses sequence parallelism is incompatible with native sequence parallelism. Set "
                                 "enable_sequence_parallelism to False in the pass config to use Ulysses.")
            # vLLM's constructor would try to build its own drafter for a method it does not know: hide ours
            arctic_sc = None
            sc = vllm_config.speculative_config
            if sc is not None and sc.method in ARCTIC_METHODS:
                arctic_sc, vllm_config.speculative_config = sc, None
            self._orig_init(vllm_config, device)
            self._suffix_cache = None
            self._arctic_early = None          # draft-model output enqueued ahead of the host sync (or None)
            self._arctic_suffix_won_last = False
            self._arctic_rej = None            # ops.RejectionResult of the current step (greedy HIP path)
            self.shift_model = None
            self.shift_parallel_threshold = 0
            if arctic_sc is not None:
                self.vllm_config.speculative_config = arctic_sc
                self.speculative_config = arctic_sc
                if get_pp_group().is_last_rank:
                    if arctic_sc.method in ("arctic", "mlp_speculator"):
                        self.drafter = ArcticProposer(self.vllm_config)
                    elif arctic_sc.method != "suffix":
                        raise ValueError(f"Unknown speculative decoding method: {arctic_sc.method}")
                    self.rejection_sampler = RejectionSampler()
            sc = self.speculative_config
            if sc is not None and getattr(sc, "enable_suffix_decoding", False):
                if sc.method not in ARCTIC_METHODS:
                    raise ValueError("Suffix decoding is only supported with the 'arctic', 'mlp_speculator' or 'suffix' "
                                     "spec decoding methods.")
                self._suffix_cache = SuffixCache(sc.suffix_cache_max_depth)

        # ---- profile / inputs (model_runner.py:158-176) ----------------------------------------------
        def profile_run(self) -> None:
            self._orig_profile_run()
            if self.shift_model is not None:
                base, self.model = self.model, self.shift_model      # compile / warm the TP replica as well
                try:
                    with set_shift_parallel_mode(True):
                        self._dummy_run(self.max_num_tokens, is_profile=True)
                finally:
                    self.model = base

        def _prepare_inputs(self, *args, **kwargs):
            out = self._orig_prepare_inputs(*args, **kwargs)
            attn_metadata, logits_indices = out[0], out[2]
            for meta in attn_metadata.values():      # SwiftKV stops prefill tokens early: it needs these inside the model
                meta.swiftkv_logits_indices = logits_indices
            return out

        # ---- Ulysses wrapper of the model's forward (model_runner.py:178-216) ------------------------
        def monkeypatch_forward(self):
            sp = parallel_state._SP
            sp_size, sp_rank, group = sp.world_size, sp.rank_in_group, sp.device_group
            inner = self.model.forward
            key = "inputs_embeds" if self.is_multimodal_model else "input_ids"
            hidden = self.hidden_size

            def ulysses_forward(*args, **kwargs):
                x, pos = kwargs[key], kwargs["positions"]
                total = x.shape[0]
                n = total // sp_size
                lo = n * sp_rank
                kwargs[key] = x[lo:lo + n]                 # this rank's token slice
                kwargs["positions"] = pos[..., lo:lo + n] if pos.dim() > 1 else pos[lo:lo + n]
                with set_shift_parallel_mode(False):
                    out = inner(*args, **kwargs)
                if out.size(0) != n:
                    assert out.size(0) == total            # SwiftKV gathers inside the model (llama_swiftkv.py:250-252)
                    return out
                full = torch.empty((total, hidden), dtype=out.dtype, device=out.device)
                from ..dist_utils import all_gather_into_tensor
                all_gather_into_tensor(full, out.contiguous(), group=group)     # C6
                return full

            self.model.forward = ulysses_forward

        # ---- one engine step (model_runner.py:218-524) -----------------------------------------------
        @torch.inference_mode()
        def execute_model(self, scheduler_output, intermediate_tensors=None):
            self._update_states(scheduler_output)
            self._arctic_note_new_requests(scheduler_output)
            if not scheduler_output.total_num_scheduled_tokens:
                if not has_kv_transfer_group():
                    return EMPTY_MODEL_RUNNER_OUTPUT
                return self.kv_connector_no_forward(scheduler_output)

            (attn_metadata, attention_cuda_graphs, logits_indices, spec_decode_metadata,
             num_scheduled_tokens_np) = self._prepare_inputs(scheduler_output)
            n_sched = scheduler_output.total_num_scheduled_tokens
            use_shift = bool(self.use_ulysses and self.shift_model is not None and n_sched <= self.shift_parallel_threshold)
            n_input = self._arctic_padded_tokens(n_sched, use_shift)
            num_pad, num_tokens_across_dp = self.get_dp_padding(n_input)
            n_input += num_pad

            model_kwargs = self._arctic_model_inputs(scheduler_output, n_sched, n_input, intermediate_tensors)
            skip_cuda_graphs = self.full_cuda_graph and not attention_cuda_graphs
            step_context.publish(num_scheduled_tokens_np, use_shift)    # q_len per request: the attention patch's split lists
            try:
                with set_forward_context(attn_metadata, self.vllm_config, num_tokens=n_input,
                                         num_tokens_across_dp=num_tokens_across_dp, skip_cuda_graphs=skip_cuda_graphs):
                    self.maybe_setup_kv_connector(scheduler_output)
                    model = self.shift_model if use_shift else self.model
                    with set_shift_parallel_mode(use_shift):
                        model_output = model(**model_kwargs)
                    self.maybe_wait_for_kv_save()
                    finished_sending, finished_recving = self.get_finished_kv_transfers(scheduler_output)
            finally:
                step_context.clear()

            if self.use_aux_hidden_state_outputs:
                hidden_states, aux_hidden_states = model_output
            else:
                hidden_states, aux_hidden_states = model_output, None

            # pipeline stages other than the last hand their activations on (external_launcher: broadcast the logits)
            broadcast_pp = (self.parallel_config.distributed_executor_backend == "external_launcher"
                            and len(get_pp_group().ranks) > 0)
            sample_hidden_states = None
            if not get_pp_group().is_last_rank:
                if not broadcast_pp:
                    return hidden_states
                get_pp_group().send_tensor_dict(hidden_states.tensors, all_gather_group=get_tp_group())
                logits = None
            else:
                if self.input_batch.pooling_params:
                    return self._pool(hidden_states, n_sched, num_scheduled_tokens_np, finished_sending, finished_recving)
                sample_hidden_states = hidden_states[logits_indices]
                logits = self.model.compute_logits(sample_hidden_states, None)
            if broadcast_pp:
                data = get_pp_group().broadcast_tensor_dict({"logits": logits.contiguous()} if logits is not None else {},
                                                            src=len(get_pp_group().ranks) - 1)
                logits = data["logits"]
            if scheduler_output.grammar_bitmask is not None:
                self.apply_grammar_bitmask(scheduler_output, logits)

            sampling_metadata = self.input_batch.sampling_metadata
            sampler_output = self._arctic_sample(logits, sampling_metadata, spec_decode_metadata, sample_hidden_states)

            num_nans_in_logits = {}
            import vllm.envs as envs
            if envs.VLLM_COMPUTE_NANS_IN_LOGITS:
                num_nans_in_logits = self._get_nans_in_logits(logits)

            # partial prefills: the sampled token is not a real sample — rewind that request's generator, drop the token
            discard = []
            for i, req_id in enumerate(self.input_batch.req_ids):
                st = self.requests[req_id]
                if st.num_computed_tokens + scheduler_output.num_scheduled_tokens[req_id] < st.num_tokens:
                    gen = self.input_batch.generators.get(i)
                    if gen is not None:
                        gen.set_offset(gen.get_offset() - 4)
                    discard.append(i)

            lt = sampler_output.logprobs_tensors
            logprobs_lists = lt.tolists() if lt is not None else None            # first host sync of the step
            prompt_logprobs_dict = self._get_prompt_logprobs_dict(hidden_states[:n_sched], scheduler_output)

            sampled = sampler_output.sampled_token_ids
            if sampled.shape[-1] == 1:
                valid = sampled.tolist()
            else:
                valid = self.rejection_sampler.parse_output(sampled, self.input_batch.vocab_size)
            for i in discard:
                valid[i].clear()

            # the runner keeps the sampled tokens itself (the scheduler does not send them back)
            ib = self.input_batch
            for i, ids in enumerate(valid):
                if not ids:
                    continue
                start = int(ib.num_tokens_no_spec[i])
                end = start + len(ids)
                assert end <= self.max_model_len, (
                    f"Sampled token IDs exceed the max model length. Total number of tokens: {end} > max_model_len: "
                    f"{self.max_model_len}")
                ib.token_ids_cpu[i, start:end] = ids
                ib.num_tokens_no_spec[i] = end
                ib.num_tokens[i] = end
                self.requests[ib.req_ids[i]].output_token_ids.extend(ids)

            if self._suffix_cache is not None:
                self._update_suffix_cache(valid)
            if not self.speculative_config:
                spec_token_ids = None
            else:
                spec_token_ids = self.propose_draft_token_ids(scheduler_output, valid, sampler_output.sampled_token_ids,
                                                              sampling_metadata, hidden_states, sample_hidden_states,
                                                              aux_hidden_states, spec_decode_metadata, attn_metadata)
            self._arctic_early = self._arctic_rej = None

            if has_kv_transfer_group():
                get_kv_transfer_group().clear_connector_metadata()
            self.eplb_step()
            return ModelRunnerOutput(req_ids=ib.req_ids, req_id_to_index=ib.req_id_to_index, sampled_token_ids=valid,
                                     spec_token_ids=spec_token_ids, logprobs=logprobs_lists,
                                     prompt_logprobs_dict=prompt_logprobs_dict, pooler_output=[],
                                     finished_sending=finished_sending, finished_recving=finished_recving,
                                     num_nans_in_logits=num_nans_in_logits)

        # ---- pieces of the step ----------------------------------------------------------------------
        def _arctic_note_new_requests(self, scheduler_output) -> None:
            """A request's prompt is known when its prefill is first scheduled: start its prompt tree on a host thread
            now, so that it exists by the time the first token is sampled (the reference builds it on the engine thread
            at that point, :664-671).  Requests that come back from preemption take the synchronous path of
            _update_suffix_cache."""
            if self._suffix_cache is None:
                return
            for new in getattr(scheduler_output, "scheduled_new_reqs", ()) or ():
                if not self._suffix_cache.has_cached_prompt(new.req_id):
                    self._suffix_cache.cache_prompt_async(new.req_id, new.prompt_token_ids)

        def _arctic_padded_tokens(self, n_sched: int, use_shift: bool) -> int:
            """Token count the model runs on (model_runner.py:240-263): SP steps are padded to a multiple of SP (and to a
            graph size per rank), graph-sized steps to the graph size, eager steps to TP when vLLM's own SP pass is on."""
            graphs = self.use_cuda_graph
            if self.use_ulysses and not use_shift:
                sp = self.parallel_config.ulysses_sequence_parallel_size
                n = round_up(n_sched, sp)
                if graphs and n // sp <= self.cudagraph_batch_sizes[-1]:
                    n = self.vllm_config.pad_for_cudagraph(n // sp) * sp
                return n
            if graphs and n_sched <= self.cudagraph_batch_sizes[-1]:
                return self.vllm_config.pad_for_cudagraph(n_sched)
            tp = self.vllm_config.parallel_config.tensor_parallel_size
            if self.compilation_config.pass_config.enable_sequence_parallelism and tp > 1:
                return round_up(n_sched, tp)
            return n_sched

        def _arctic_model_inputs(self, scheduler_output, n_sched: int, n_input: int, intermediate_tensors) -> dict:
            """Keyword arguments of the model call (model_runner.py:269-308)."""
            mm_embeds = []
            if self.is_multimodal_model:
                self._execute_mm_encoder(scheduler_output)
                mm_embeds = self._gather_mm_embeddings(scheduler_output)
            if self.is_multimodal_model and get_pp_group().is_first_rank:
                ids = self.input_ids[:n_sched]
                emb = self.model.get_input_embeddings(ids, mm_embeds) if mm_embeds else self.model.get_input_embeddings(ids)
                self.inputs_embeds[:n_sched].copy_(emb)
                input_ids, inputs_embeds = None, self.inputs_embeds[:n_input]
            else:
                input_ids, inputs_embeds = self.input_ids[:n_input], None
            positions = self.mrope_positions[:, :n_input] if self.uses_mrope else self.positions[:n_input]
            if get_pp_group().is_first_rank:
                intermediate_tensors = None
            else:
                intermediate_tensors = self.sync_and_slice_intermediate_tensors(n_input, intermediate_tensors, True)
            return dict(input_ids=input_ids, positions=positions, intermediate_tensors=intermediate_tensors,
                        inputs_embeds=inputs_embeds)

        def _arctic_sample(self, logits, sampling_metadata, spec_decode_metadata, sample_hidden_states):
            """Sampling + acceptance (model_runner.py:381-412).  Verify steps whose sampling needs nothing beyond a
            temperature run on the HIP acceptance kernels, reading the target rows of `logits` in place through
            SpecDecodeMetadata.target_logits_indices (no gathered copy):
              * all-greedy batches: ONE launch that also takes the bonus rows' arg-max (no separate sampler call);
              * batches with random rows (temperature only; greedy rows may be mixed in): the bonus tokens come from
                vLLM's sampler as in the reference (:394-399), then the uniforms and the Exp(1) recovery noise are drawn
                exactly as vLLM's rejection sampler draws them — same calls, same order, per-request generators — and
                aic_rejection_random accepts.
            Anything else (top-k / top-p / min-p, penalties, bias, min_tokens, masks, logprobs: runner_logic.
            hip_acceptance_kind) keeps vLLM's sampler + RejectionSampler."""
            if spec_decode_metadata is None:
                return self.sampler(logits=logits, sampling_metadata=sampling_metadata)
            assert logits is not None
            md = spec_decode_metadata
            sm = sampling_metadata
            kind = hip_acceptance_kind(sm) if logits.is_cuda else None
            if kind == "greedy":
                from vllm.v1.outputs import SamplerOutput
                max_spec = max(int(max(md.num_draft_tokens)), 1)
                rej = ops.rejection_sample(logits, md.draft_token_ids, md.cu_num_draft_tokens, None, max_spec,
                                           target_row_index=md.target_logits_indices.to(torch.int64),
                                           bonus_row_index=md.bonus_logits_indices.to(torch.int64))
                self._arctic_rej = rej
                self._arctic_maybe_draft_early(rej, sample_hidden_states, md.num_draft_tokens)
                return SamplerOutput(sampled_token_ids=rej.output_token_ids, logprobs_tensors=None)
            # indexing with a tensor copies: in-place edits of the sampler do not reach `logits`
            bonus_logits = logits[md.bonus_logits_indices]
            out = self.sampler(logits=bonus_logits, sampling_metadata=sampling_metadata)
            if kind == "random":
                n_draft = md.num_draft_tokens
                max_spec = max(int(max(n_draft)), 1)
                uniform = draw_uniform_probs(int(md.draft_token_ids.numel()), n_draft, sm.generators, logits.device)
                noise = draw_recovery_noise(len(n_draft), logits.shape[-1], n_draft, sm.generators, logits.device)
                rej = ops.rejection_sample(logits, md.draft_token_ids, md.cu_num_draft_tokens, out.sampled_token_ids, max_spec,
                                           temperature=sm.temperature, uniform_probs=uniform, exp_noise=noise,
                                           target_row_index=md.target_logits_indices.to(torch.int64))
                self._arctic_rej = rej
                self._arctic_maybe_draft_early(rej, sample_hidden_states, n_draft)
                out.sampled_token_ids = rej.output_token_ids
                return out
            target_logits = logits[md.target_logits_indices]
            out.sampled_token_ids = self.rejection_sampler(md, None, target_logits, out.sampled_token_ids, sampling_metadata)
            return out

        def _arctic_uses_draft_model(self) -> bool:
            sc = self.speculative_config
            return bool(sc and sc.method in ("arctic", "mlp_speculator") and getattr(self, "drafter", None) is not None
                        and not (sc.disable_by_batch_size and len(self.input_batch.req_ids) > sc.disable_by_batch_size))

        def _arctic_maybe_draft_early(self, rej, sample_hidden_states, num_draft_tokens) -> None:
            """Enqueue the draft model behind the acceptance kernel, before the host has seen the step's tokens: its
            inputs (last accepted token, hidden-state row) are on the device already.  Done when the previous step used
            the draft model (steps resemble their predecessor); a step in which suffix decoding then takes a request
            drops the result, as the reference's rule demands."""
            self._arctic_early = None
            if not self._arctic_uses_draft_model() or sample_hidden_states is None:
                return
            if self._suffix_cache is not None and self._arctic_suffix_won_last:
                return
            # the draft length is one value for the batch (:629-641) and depends on how many tokens each request accepts;
            # with every draft accepted it is smallest: only if even then it is the configured k is it known now
            nb = len(self.input_batch.req_ids)
            mode = proposal_indexing(self.speculative_config)
            worst_ends = [proposal_end_index(int(self.input_batch.num_tokens_no_spec[i]) + int(num_draft_tokens[i]) + 1,
                                             int(num_draft_tokens[i]) + 1, mode) for i in range(nb)]
            k = self.speculative_config.num_speculative_tokens
            if arctic_max_spec_tokens(k, worst_ends, self.max_model_len) != k:
                return                                   # a request is close to max_model_len: take the exact late path
            self._arctic_early = (k, self.drafter.propose_on_device(rej.last_token, sample_hidden_states, k,
                                                                    hidden_index=rej.hidden_index))

        # ---- suffix cache maintenance (model_runner.py:657-678) --------------------------------------
        def _update_suffix_cache(self, sampled_token_ids: List[List[int]]) -> None:
            ib, cache = self.input_batch, self._suffix_cache
            seen = set()
            upd_ids, upd_lens, upd_toks = [], [], []
            for i, sampled in enumerate(sampled_token_ids):
                req_id = ib.req_ids[i]
                seen.add(req_id)
                if not sampled:
                    continue
                if not cache.has_cached_prompt(req_id):
                    index = ib.req_id_to_index[req_id]
                    cache.cache_prompt(req_id, ib.token_ids_cpu[index, :ib.num_prompt_tokens[index]])
                upd_ids.append(req_id)
                upd_lens.append(len(sampled))
                upd_toks.extend(sampled)
            if upd_ids:
                # the per-request update_response loop as one native call, same order
                cache.update_responses(upd_ids, np.asarray(upd_toks, np.int32), np.asarray(upd_lens, np.int32))
            for req_id in cache.cached_prompt_ids():
                if req_id not in seen:
                    cache.evict_prompt(req_id)

        # ---- proposals (model_runner.py:526-655, :680-744) -------------------------------------------
        def propose_suffix_draft_token_ids(self, sampled_token_ids, spec_token_ids=None):
            cfg, ib = self.speculative_config, self.input_batch
            mode = proposal_indexing(cfg)
            results = [SuffixSpecResult() for _ in sampled_token_ids]
            ids, pats, kws, where = [], [], [], []
            for i, sampled in enumerate(sampled_token_ids):
                spec_ids = spec_token_ids[i] if spec_token_ids is not None else []
                if not sampled:
                    continue
                # execute_model has already appended this step's sampled ids to the row and advanced num_tokens_no_spec
                # (:469-486).  "reference" indexing adds len(sampled) to it a second time and writes the ids again, as
                # :698-709 do; "single_advance" takes the row as it is (runner_logic.py).
                start = int(ib.num_tokens_no_spec[i])
                end = proposal_end_index(start, len(sampled), mode)
                if mode == INDEXING_REFERENCE:
                    rewrite_sampled(ib.token_ids_cpu[i], start, sampled, self.max_model_len)
                if end >= self.max_model_len:
                    continue
                q = suffix_query(ib.token_ids_cpu[i], end, spec_ids, self.max_model_len, cfg.suffix_cache_max_depth,
                                 cfg.suffix_max_spec_factor, cfg.suffix_max_spec_offset, cfg.suffix_min_token_prob)
                if q is None:
                    continue
                ids.append(ib.req_ids[i])
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

        def propose_arctic_draft_token_ids(self, scheduler_output, sampled_token_ids, previous_hidden_states=None):
            ib = self.input_batch
            mode = proposal_indexing(self.speculative_config)
            last_tokens: List[int] = []
            k = self.speculative_config.num_speculative_tokens
            for i, sampled in enumerate(sampled_token_ids):
                n = len(sampled)
                if n == 0:
                    if self.speculative_config.enable_suffix_decoding:
                        return [[]] * len(sampled_token_ids)        # suffix decoding took a request: nobody drafts (:616-618)
                    req_id = ib.req_ids[i]
                    st = self.requests[req_id]
                    sampled = [st.get_token_id(st.num_computed_tokens + scheduler_output.num_scheduled_tokens[req_id])]
                start = int(ib.num_tokens_no_spec[i])      # the row already holds this step's sampled ids (see above)
                end = proposal_end_index(start, n, mode)
                k = min(k, self.max_model_len - end - 1)
                if k <= 0:
                    continue
                if mode == INDEXING_REFERENCE:
                    # :635-636: the last sampled id over [start, end), and the conditioning token read back from the row
                    # (for a request without a sampled id that is the row's last known token, not `sampled`)
                    ib.token_ids_cpu[i, start:end] = sampled[-1]
                    last_tokens.append(int(ib.token_ids_cpu[i, end - 1]))
                else:
                    last_tokens.append(int(sampled[-1]))
            if k <= 0:
                return [[] for _ in sampled_token_ids]
            early = self._arctic_early
            if previous_hidden_states is None:
                # the caller saw that this step's draft-model run is already in flight behind the acceptance kernel
                assert early is not None and early[0] == k and all(len(s) > 0 for s in sampled_token_ids)
                out = early[1].cpu().numpy()
            else:
                hs, hidx = (previous_hidden_states if isinstance(previous_hidden_states, tuple)
                            else (previous_hidden_states, None))
                out = self.drafter.propose(last_tokens, previous_hidden_states=hs, num_predict_tokens=k, hidden_index=hidx)
            drafts = out.tolist()
            for i, sampled in enumerate(sampled_token_ids):
                if not sampled:
                    drafts[i] = []
            return drafts

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
                won = False
                for i, r in enumerate(results):
                    if r.score >= floor:
                        remaining[i] = []          # taken by suffix decoding: no other proposer for this request
                        suffix_ids.append(r.token_ids)
                        won = won or bool(sampled_token_ids[i])
                    else:
                        suffix_ids.append([])
                self._arctic_suffix_won_last = won
            model_ids = None
            if sc.method == "suffix":
                pass
            elif sc.method in ("arctic", "mlp_speculator"):
                assert isinstance(self.drafter, ArcticProposer)
                early = self._arctic_early
                mode = proposal_indexing(sc)
                ends = [proposal_end_index(self.input_batch.num_tokens_no_spec[i], len(remaining[i]), mode)
                        for i in range(len(remaining))]
                usable = (early is not None and all(len(s) > 0 for s in remaining)
                          and arctic_max_spec_tokens(sc.num_speculative_tokens, ends, self.max_model_len) == early[0])
                prev = None
                if not usable:
                    prev = self.drafter.prepare_hidden_states(sample_hidden_states=sample_hidden_states,
                                                              sampled_token_ids=original_sampled_token_ids,
                                                              spec_decode_metadata=spec_decode_metadata, fused=True)
                model_ids = self.propose_arctic_draft_token_ids(scheduler_output, remaining, previous_hidden_states=prev)
            else:
                model_ids = self._orig_propose_draft_token_ids(scheduler_output, remaining, sampling_metadata, hidden_states,
                                                               sample_hidden_states, aux_hidden_states, spec_decode_metadata,
                                                               attn_metadata)
            return merge_proposals(suffix_ids, model_ids)

        # ---- model loading / graphs / KV binding (model_runner.py:746-867) ---------------------------
        def load_model(self) -> None:
            pc = self.vllm_config.parallel_config
            want_shift = getattr(pc, "enable_shift_parallel", False)
            shift_config = copy.deepcopy(self.vllm_config) if want_shift else None    # before vLLM mutates it
            self._orig_load_model()
            if getattr(self.parallel_config, "ulysses_sequence_parallel_size", 1) > 1:
                self.monkeypatch_forward()
            if not want_shift:
                self.shift_model, self.shift_parallel_threshold = None, 0
                return
            spc = shift_config.parallel_config
            spc.tensor_parallel_size *= spc.ulysses_sequence_parallel_size     # the replica is a plain TP = SP*TP model
            spc.ulysses_sequence_parallel_size = 1
            with set_shift_parallel_mode(True):
                self.shift_model = get_model(vllm_config=shift_config)
            self.shift_parallel_threshold = spc.shift_parallel_threshold
            if "SwiftKV" in type(self.model).__name__:
                # SwiftKV's decode half always runs in full TP: both models share the shift replica's decode runner,
                # whose graphs cover every decode size (model_runner.py:767-773)
                self.model.model.decode_runner = self.shift_model.model.decode_runner

        def capture_model(self) -> None:
            if not self.use_cuda_graph:
                logger.warning("Skipping CUDA graph capture. To turn on CUDA graph capture, set -O %s and ensure "
                               "`use_cudagraph` was not manually set to False", CompilationLevel.PIECEWISE)
                return
            from vllm.compilation.counter import compilation_counter
            compilation_counter.num_gpu_runner_capture_triggers += 1
            t0 = time.perf_counter()
            mem_free = lambda: torch.cuda.mem_get_info()[0] if torch.cuda.is_available() else 0
            free0 = mem_free()
            sp = self.parallel_config.ulysses_sequence_parallel_size
            warmups = self.vllm_config.compilation_config.cudagraph_num_of_warmups
            full = self.full_cuda_graph

            def capture(sizes, scale):
                # big shapes first so that small ones reuse their pool; `scale` tokens run per graph-sized slice
                for n in sizes:
                    for _ in range(warmups + 1):
                        self._dummy_run(n * scale, capture_attn_cudagraph=full, skip_eplb=True)

            # graph_capture (patched) holds the TP, PP and SP_TP communicators in capture mode: the RCCL collectives of
            # both replicas are recorded into the graphs
            with parallel_state.graph_capture(device=self.device):
                sizes = list(reversed(self.cudagraph_batch_sizes))
                base = [n for n in sizes if self.shift_parallel_threshold < n * sp <= self.max_num_tokens]
                logger.info("original model shapes %s", base)
                capture(base, sp)               # the Ulysses model only ever sees steps above the threshold
                if self.shift_model is not None:
                    keep, self.model = self.model, self.shift_model
                    try:
                        swiftkv = "SwiftKV" in type(self.model).__name__    # its decode runner needs every size
                        shift = [n for n in sizes if n <= self.shift_parallel_threshold or swiftkv]
                        logger.info("shift model shapes %s", shift)
                        with set_shift_parallel_mode(True):
                            capture(shift, 1)
                    finally:
                        self.model = keep
            logger.info("Graph capturing finished in %.0f secs, took %.2f GiB", time.perf_counter() - t0,
                        (free0 - mem_free()) / (1 << 30))

        def initialize_kv_cache(self, kv_cache_config) -> None:
            self._orig_initialize_kv_cache(kv_cache_config)
            if self.shift_model is None:
                return
            # the TP replica attends over the SAME cache tensors (each rank owns the same head slice in both layouts)
            bound = self.vllm_config.compilation_config.static_forward_context
            for mod in self.shift_model.modules():
                if isinstance(mod, Attention):
                    mod.kv_cache = bound[mod.layer_name].kv_cache

    _runner_patch = GPUModelRunnerPatch
    return _runner_patch
