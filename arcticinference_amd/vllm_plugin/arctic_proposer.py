"""ArcticProposer — the drafter object GPUModelRunnerPatch installs for method "arctic" / "mlp_speculator"
(/root/reference/arctic_inference/vllm/spec_dec/arctic_proposer.py:30-166): same constructor, load_model,
prepare_hidden_states and propose, over the HIP speculators of arcticinference_amd.speculator.

MI355X-side additions: `propose_on_device` (no host copy: the runner enqueues the draft before its host sync) and
`prepare_hidden_states(..., fused=True)`, which hands back the row INDEX instead of a gathered copy — the gather
is fused into the speculator's first kernel."""
from __future__ import annotations

import logging
import os
from typing import Optional

import numpy as np
import torch

logger = logging.getLogger(__name__)

SUPPORTED_ARCHITECTURES = ("ArcticMLPSpeculatorPreTrainedModel", "ArcticLSTMSpeculatorPreTrainedModel",
                           "MLPVariantSpeculatorPreTrainedModel")


def _skip_model_check() -> bool:
    # arctic_inference/envs.py: ARCTIC_INFERENCE_SKIP_SPEC_MODEL_CHECK
    return os.environ.get("ARCTIC_INFERENCE_SKIP_SPEC_MODEL_CHECK", "0").lower() in ("1", "true")


class ArcticProposer:
    def __init__(self, vllm_config):
        self.vllm_config = vllm_config
        self.speculative_config = vllm_config.speculative_config
        self.model = None
        self.device = None
        self.input_hidden_dim = None

    def load_model(self, model) -> None:
        """`model` is the target model vLLM just loaded (arctic_proposer.py:42-111).  Checks the draft checkpoint's
        architecture list and that it was trained for this base architecture, then builds the draft model through
        vLLM's loader (the plugin registered the architectures to this package's constructors)."""
        from vllm.config import VllmConfig
        from vllm.model_executor.model_loader import get_model
        draft_mc = self.speculative_config.draft_model_config
        archs = draft_mc.hf_config.architectures
        if not isinstance(archs, list):
            logger.error("Draft model architectures %s is not a list. ", archs)
            raise TypeError()
        if len(archs) != 1:
            logger.error("Draft model architectures %s does not have exactly one architecture. ", archs)
            raise ValueError()
        if archs[0] not in SUPPORTED_ARCHITECTURES:
            logger.error("Draft model architecture %s is not supported by Arctic Speculator. ", archs)
            raise ValueError()
        if not _skip_model_check():
            base = self.vllm_config.model_config.architectures[0]
            trained_for = getattr(draft_mc.hf_config, "base_model_archs", None)
            if trained_for is None:
                logger.error("Draft model config does not have base_model_archs attribute. "
                             "Set ARCTIC_INFERENCE_SKIP_SPEC_MODEL_CHECK=1 to skip this assertion.")
                assert False
            if base not in trained_for:
                logger.error("Draft model trained with base model architectures %s does not match the base model "
                             "architecture %s in the vLLM config. Set ARCTIC_INFERENCE_SKIP_SPEC_MODEL_CHECK=1 to skip "
                             "this assertion.", trained_for, base)
                assert False
        quant = VllmConfig._get_quantization_config(self.vllm_config.model_config, self.vllm_config.load_config)
        draft_pc = self.speculative_config.draft_parallel_config
        draft_pc.worker_cls = self.vllm_config.parallel_config.sd_worker_cls
        # built field by field: process groups hang off the configs under Ulysses and do not deep-copy
        draft_cfg = VllmConfig(model_config=draft_mc, quant_config=quant, parallel_config=draft_pc,
                               load_config=self.vllm_config.load_config, device_config=self.vllm_config.device_config)
        self.model = get_model(vllm_config=draft_cfg)
        self.device = next(model.parameters()).device
        self.input_hidden_dim = getattr(self.model, "input_hidden_dim", None) or self.model.emb_dim

    def prepare_hidden_states(self, sample_hidden_states: torch.Tensor, sampled_token_ids, spec_decode_metadata,
                              fused: bool = False):
        """The hidden state each request drafts from: the row of its last ACCEPTED token among the step's sampled rows
        (arctic_proposer.py:113-147).  `fused=True` returns (sample_hidden_states, int32 row index) instead of the
        gathered tensor."""
        if sample_hidden_states is not None:
            assert sample_hidden_states.shape[-1] == self.input_hidden_dim, (
                f"hidden_states shape mismatch: {sample_hidden_states.shape[-1]} != {self.input_hidden_dim}. "
                "Please make sure spec model is trained using the same base model.")
        if sampled_token_ids.shape[-1] == 1:
            return (sample_hidden_states, None) if fused else sample_hidden_states
        assert spec_decode_metadata is not None
        n = torch.as_tensor(np.asarray(spec_decode_metadata.num_draft_tokens), device=sampled_token_ids.device) + 1
        gen_lens = (sampled_token_ids != -1).sum(dim=1)
        idx = (gen_lens - 1) + torch.cumsum(n, 0) - n
        if fused:
            return sample_hidden_states, idx.to(torch.int32)
        return sample_hidden_states[idx]

    def propose_on_device(self, last_tokens: torch.Tensor, previous_hidden_states: torch.Tensor, num_predict_tokens: int,
                          hidden_index: Optional[torch.Tensor] = None) -> torch.Tensor:
        """int64 [B, k] on the device; nothing here waits for the GPU."""
        assert num_predict_tokens > 0, f"num_predict_tokens must be greater than 0, got {num_predict_tokens}."
        return self.model.generate_proposals(last_tokens, previous_hidden_states, num_predict_tokens,
                                             hidden_index=hidden_index)

    def propose(self, context_token_ids, previous_hidden_states: torch.Tensor, num_predict_tokens: int,
                hidden_index: Optional[torch.Tensor] = None) -> Optional[np.ndarray]:
        assert num_predict_tokens > 0, f"num_predict_tokens must be greater than 0, got {num_predict_tokens}."
        ids = torch.as_tensor(np.asarray(context_token_ids), device=previous_hidden_states.device)
        return self.propose_on_device(ids, previous_hidden_states, num_predict_tokens, hidden_index).cpu().numpy()
