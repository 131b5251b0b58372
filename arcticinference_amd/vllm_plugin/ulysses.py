"""Ulysses / shift-parallel patches of vLLM's state (mirror of /root/reference/arctic_inference/vllm/ulysses.py):

  UlyssesModelConfigPatch            head counts per rank, PP layer range under the DP x PP x SP x TP layout   (:56-90)
  UlyssesParallelStatePatch          initialize_model_parallel creating _SP, _SP_TP (and _SP_AA / _SP_AG when there are
                                     fewer kv heads than SP ranks); graph_capture holding SP_TP too            (:93-316)
  UlyssesWorkerProcPatch             teardown of those groups                                                 (:319-342)
  UlyssesMultiprocExecutorPatch      one worker per rank of PP x SP x TP                                      (:345-424)
  UlyssesAttentionPatch              head repartition around attention + the verify-attention route          (:427-519)
  PiecewiseCompileInterpreterPatch   subgraphs whose shape symbol appears at several argument positions      (:522-589)
  UlyssesFusedMoEPatch               FusedMoE.forward -> forward_impl (the custom op blocks the shift model)  (:592-599)

The group ALGEBRA is arcticinference_amd.ulysses.rank_groups() (pure, tested against the reference's tensor
reshapes on CPU); this module only turns its rank lists into vLLM GroupCoordinators.  The attention patch is
where this build differs from the reference on purpose: decode / verify steps (every request has at most
MAX_SPEC_LEN + 1 query tokens) do not go to vLLM's attention backend but to aic_verify_attention_ex, and the
copies around the all-to-alls are the fused HIP kernels of csrc/ulysses_pack.hip.
"""
from __future__ import annotations

from contextlib import contextmanager
from typing import Any, List, Optional

import torch

from ..ulysses import UlyssesAttention, rank_groups
from . import step_context
from .runner_logic import MAX_SPEC_LEN


def _flash_layout(kv_cache, num_kv_heads: int, head_size: int) -> bool:
    """[2, num_blocks, block_size, Hkv, D] (vLLM's FlashAttention / Triton-unified layout, llama_swiftkv.py:617)."""
    return (isinstance(kv_cache, torch.Tensor) and kv_cache.dim() == 5 and kv_cache.shape[0] == 2
            and kv_cache.shape[3] == num_kv_heads and kv_cache.shape[4] == head_size and kv_cache.shape[2] % 16 == 0)


def build_ulysses_patches():
    """ArcticPatch classes, in the order apply_shift_parallel_patches() installs them (ulysses.py:46-53)."""
    import threading
    import weakref
    from concurrent.futures import ThreadPoolExecutor

    import vllm.envs as envs
    from vllm.attention.layer import Attention
    from vllm.compilation.backends import PiecewiseCompileInterpreter
    from vllm.config import ModelConfig, get_current_vllm_config
    from vllm.distributed import parallel_state
    from vllm.distributed.device_communicators.shm_broadcast import MessageQueue
    from vllm.distributed.parallel_state import (destroy_distributed_environment, destroy_model_parallel,
                                                 get_world_group, init_model_parallel_group)
    from vllm.executor.multiproc_worker_utils import set_multiprocessing_worker_envs
    from vllm.forward_context import get_forward_context
    from vllm.model_executor.layers.fused_moe import FusedMoE
    from vllm.platforms import current_platform
    from vllm.utils import get_distributed_init_method, get_open_port, resolve_obj_by_qualname
    from vllm.v1.executor.multiproc_executor import MultiprocExecutor, WorkerProc

    from .. import ops
    from ..patching import ArcticPatch
    from .custom_ops import attention_op, call_attention
    from .model_runner import is_shift_parallel_mode

    # registered NOW, while the patches are built (eagerly, once per process): the first forward of a model may already
    # run under Dynamo, where creating a custom op is not something the tracer can do
    attention_op()

    # -----------------------------------------------------------------------------------------------
    class UlyssesModelConfigPatch(ArcticPatch[ModelConfig]):
        _orig_get_num_kv_heads = ModelConfig.get_num_kv_heads
        _orig_get_num_attention_heads = ModelConfig.get_num_attention_heads

        def get_num_kv_heads(self, parallel_config) -> int:
            return max(1, self._orig_get_num_kv_heads(parallel_config) // parallel_config.ulysses_sequence_parallel_size)

        def get_num_attention_heads(self, parallel_config) -> int:
            return max(1, self._orig_get_num_attention_heads(parallel_config) // parallel_config.ulysses_sequence_parallel_size)

        def get_layers_start_end_indices(self, parallel_config):
            from vllm.distributed.utils import get_pp_indices
            mtp = (self.hf_text_config.model_type == "deepseek_mtp" or self.hf_config.model_type == "mimo_mtp")
            layers = getattr(self.hf_text_config, "num_nextn_predict_layers" if mtp else "num_hidden_layers", 0)
            # global rank = ((dp * PP + pp) * SP + sp) * TP + tp
            inner = parallel_config.tensor_parallel_size * parallel_config.ulysses_sequence_parallel_size
            pp_rank = (parallel_config.rank // inner) % parallel_config.pipeline_parallel_size
            return get_pp_indices(layers, pp_rank, parallel_config.pipeline_parallel_size)

    # -----------------------------------------------------------------------------------------------
    class UlyssesParallelStatePatch(ArcticPatch[parallel_state]):
        _SP = None
        _SP_TP = None
        _SP_AA = None
        _SP_AG = None

        def initialize_model_parallel(tensor_model_parallel_size: int = 1, pipeline_model_parallel_size: int = 1,
                                      backend: Optional[str] = None) -> None:
            """Process groups of the layout ExternalDP x DP x PP x SP x TP (TP fastest).  Besides vLLM's TP / PP / DP / EP:
            SP (Ulysses all-to-all), SP_TP (the shift replica's tensor-parallel group, TP-major so that a rank owns the
            same attention heads in both layouts), and — fewer kv heads than SP ranks — SP_AA x SP_AG."""
            assert torch.distributed.is_initialized()
            world = torch.distributed.get_world_size()
            backend = backend or torch.distributed.get_backend(get_world_group().device_group)
            config = get_current_vllm_config()
            dp = config.parallel_config.data_parallel_size if config is not None else 1
            sp = config.parallel_config.ulysses_sequence_parallel_size
            tp, pp = tensor_model_parallel_size, pipeline_model_parallel_size
            num_kv_heads = config.model_config._orig_get_num_kv_heads(config.parallel_config)
            ranks = rank_groups(world, dp, pp, sp, tp, num_kv_heads=num_kv_heads)
            local_rank = get_world_group().local_rank

            def make(kind: str, name: str, **kw):
                return init_model_parallel_group(ranks[kind], local_rank, backend, group_name=name, **kw)

            for slot in ("_TP", "_PP", "_DP", "_EP", "_SP", "_SP_TP"):
                assert getattr(parallel_state, slot, None) is None, f"{slot[1:]} group is already initialized"
            parallel_state._TP = make("TP", "tp", use_message_queue_broadcaster=True)   # the only group with a broadcaster
            parallel_state._PP = make("PP", "pp")
            parallel_state._DP = make("DP", "dp")
            parallel_state._EP = make("EP", "ep")
            parallel_state._SP = make("SP", "sp")
            parallel_state._SP_TP = make("SP_TP", "sp_tp")
            if "SP_AA" in ranks:
                parallel_state._SP_AA = make("SP_AA", "sp_aa")
                parallel_state._SP_AG = make("SP_AG", "sp_ag")
            ps = parallel_state
            ps.logger.info("rank %s in world size %s is assigned as DP rank %s, PP rank %s, TP rank %s, EP rank %s, SP rank "
                           "%s, SP_TP rank %s", torch.distributed.get_rank(), world, ps._DP.rank_in_group,
                           ps._PP.rank_in_group, ps._TP.rank_in_group, ps._EP.rank_in_group, ps._SP.rank_in_group,
                           ps._SP_TP.rank_in_group)
            if local_rank == 0:
                ps.logger.info("UlyssesParallelStatePatch initialized:\n" + "\n".join(
                    f"  {k} {len(v[0])} ranks {v}" for k, v in ranks.items()))

        @contextmanager
        def graph_capture(device: torch.device):
            """vLLM's graph_capture plus the SP_TP communicator: the shift replica's all-reduces are captured too."""
            from vllm.distributed.parallel_state import GraphCaptureContext
            is_gpu = torch.device(device).type == "cuda"        # (the gloo tests enter this on the CPU)
            context = GraphCaptureContext(torch.cuda.Stream(device=device) if is_gpu else None)
            with parallel_state._TP.graph_capture(context), parallel_state._PP.graph_capture(context), \
                    parallel_state._SP_TP.graph_capture(context):
                yield context

    # -----------------------------------------------------------------------------------------------
    class UlyssesWorkerProcPatch(ArcticPatch[WorkerProc]):
        def destroy_model_parallel(self):
            for slot in ("_SP", "_SP_TP", "_SP_AA", "_SP_AG"):
                group = getattr(parallel_state, slot, None)
                if group:
                    group.destroy()
                setattr(parallel_state, slot, None)

        def shutdown(self):
            self.rpc_broadcast_mq = None
            self.worker_response_mq = None
            destroy_model_parallel()
            self.destroy_model_parallel()          # the Ulysses communicators
            destroy_distributed_environment()

    # -----------------------------------------------------------------------------------------------
    class UlyssesMultiprocExecutorPatch(ArcticPatch[MultiprocExecutor]):
        def _init_executor(self) -> None:
            """vLLM's multiprocess executor start-up with world_size = PP x SP x TP workers (ulysses.py:347-424)."""
            self._finalizer = weakref.finalize(self, self.shutdown)
            self.is_failed = False
            self.shutdown_event = threading.Event()
            self.failure_callback = None
            self.io_thread_pool = None
            pc = self.parallel_config
            self.world_size = pc.world_size
            tp, pp, sp = pc.tensor_parallel_size, pc.pipeline_parallel_size, pc.ulysses_sequence_parallel_size
            assert self.world_size == tp * pp * sp, (
                f"world_size ({self.world_size}) must be equal to the tensor_parallel_size ({tp}) x pipeline"
                f"_parallel_size ({pp}) x ulysses_sequence_parallel_size ({sp}).")
            set_multiprocessing_worker_envs(pc)
            # single node only: loopback rendezvous
            init_method = get_distributed_init_method("127.0.0.1", get_open_port())
            self.rpc_broadcast_mq = MessageQueue(self.world_size, self.world_size,
                                                 max_chunk_bytes=envs.VLLM_MQ_MAX_CHUNK_BYTES_MB * 1024 * 1024)
            handle = self.rpc_broadcast_mq.export_handle()
            pending: List[Any] = []
            ok = False
            try:
                for rank in range(self.world_size):
                    pending.append(WorkerProc.make_worker_process(vllm_config=self.vllm_config, local_rank=rank, rank=rank,
                                                                  distributed_init_method=init_method,
                                                                  input_shm_handle=handle))
                # all workers exist before any is waited for: init_device() synchronises across them
                self.workers = WorkerProc.wait_for_ready(pending)
                self.rpc_broadcast_mq.wait_until_ready()
                for w in self.workers:
                    w.worker_response_mq.wait_until_ready()
                self.start_worker_monitor()
                ok = True
            finally:
                if not ok:
                    self._ensure_worker_termination([w.proc for w in pending])
            if self.max_concurrent_batches > 1:
                # pipeline parallelism: one IO thread keeps the response order
                self.io_thread_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="mp_exec_io")
            self.output_rank = self._get_output_rank()

    # -----------------------------------------------------------------------------------------------
    class UlyssesAttentionPatch(ArcticPatch[Attention]):
        _orig_init = Attention.__init__
        _orig_forward = Attention.forward

        def __init__(self, num_heads, *args, **kwargs):
            sp = getattr(parallel_state, "_SP", None)
            self.sp_size = sp.world_size if sp is not None else 1
            self.sp_device_group = sp.device_group if sp is not None else None
            self.is_kv_replicated = False
            self._ulysses = None
            if self.sp_size > 1 and not is_shift_parallel_mode():     # the shift replica is a plain TP model
                num_heads //= self.sp_size
                num_kv_heads = kwargs["num_kv_heads"]
                kv_groups = None
                if num_kv_heads < self.sp_size:
                    self.is_kv_replicated = True
                    num_kv_heads = 1
                    aa, ag = getattr(parallel_state, "_SP_AA", None), getattr(parallel_state, "_SP_AG", None)
                    assert aa is not None and ag is not None, (
                        "UlyssesAttentionPatch requires SP_AA and SP_AG groups to be initialized.")
                    kv_groups = (aa.device_group, aa.world_size, ag.device_group, ag.world_size)
                else:
                    num_kv_heads //= self.sp_size
                kwargs["num_kv_heads"] = num_kv_heads
                self._ulysses_kv_groups = kv_groups
            return self._orig_init(num_heads, *args, **kwargs)

        # -- the verify-attention route -------------------------------------------------------------
        def _arctic_verify(self, query, key, value):
            """Decode / verify steps on the HIP kernel: K/V of the step's tokens go into the paged cache
            (aic_reshape_and_cache_flash_bulk, one layer), then aic_verify_attention_ex reads the cache once for all
            draft positions of a request.  Returns None when the step is not one this kernel serves (prefill chunks,
            another cache layout, no metadata): the caller then uses vLLM's backend."""
            if not query.is_cuda:
                return None
            # Full-graph capture (vLLM's full_cuda_graph; its default piecewise graphs cut at this op and never get here):
            # the launch is recorded with the geometry of the capture-time metadata — request count, token count,
            # max_query_len and a max_seq_len that bounds every later replay — and reads the per-request lengths, the block
            # table and the slots from the persistent device buffers vLLM refreshes before each replay.  The host-side
            # request partition of the step (step_context) would freeze ONE step's lists into the graph: not used here.
            capturing = torch.cuda.is_current_stream_capturing()
            ctx = get_forward_context()
            meta = ctx.attn_metadata
            if isinstance(meta, dict):
                meta = meta.get(self.layer_name)
            if meta is None or getattr(meta, "max_query_len", 1 << 30) > MAX_SPEC_LEN + 1:
                return None
            kv_cache = self.kv_cache[ctx.virtual_engine]
            if not _flash_layout(kv_cache, self.num_kv_heads, self.head_size) or self.head_size not in (64, 128):
                return None
            if getattr(self.impl, "alibi_slopes", None) is not None or getattr(self.impl, "logits_soft_cap", None):
                return None
            # gpt-oss-class layers: a sliding window shortens the token range, a per-head sink adds one term to the
            # soft-max normalisation (aic_verify_attention_win)
            window = int(getattr(self, "sliding_window", None) or 0)
            sinks = self._arctic_sinks()
            if sinks is False:
                return None
            fp8 = kv_cache.dtype in (torch.uint8, torch.float8_e4m3fn)
            if fp8 and self.head_size != 128:
                return None
            n = meta.num_actual_tokens
            k_cache, v_cache = kv_cache[0], kv_cache[1]
            if fp8:
                k_cache, v_cache = k_cache.view(torch.float8_e4m3fn), v_cache.view(torch.float8_e4m3fn)
            # write the new K/V (vLLM's backend would: unified_attention does both)
            if key is not None and value is not None:
                writer = getattr(self, "_arctic_kv_writer", None)
                if writer is None or writer._keep[0][0].data_ptr() != k_cache.data_ptr():
                    writer = ops.KvBulkWriter([k_cache], [v_cache], "fp8" if fp8 else "auto", [self._k_scale],
                                              [self._v_scale], self.num_kv_heads, self.head_size)
                    self._arctic_kv_writer = writer
                kw = self.num_kv_heads * self.head_size
                writer(key[:n].reshape(n, kw), value[:n].reshape(n, kw), meta.slot_mapping[:n])
            hq, D = self.num_heads, self.head_size
            q = query[:n].unflatten(-1, (hq, D)) if query.dim() == 2 else query[:n]
            out = torch.empty((query.shape[0], hq * D), dtype=query.dtype, device=query.device)
            batch = meta.seq_lens.numel()
            split = None
            if not capturing:
                split = step_context.request_split(hq // self.num_kv_heads, query.device)
                ql = step_context.q_lens()
                if ql is not None and len(ql) != batch:
                    split = None                               # the published step does not describe this call
            ops.verify_attention(q, k_cache, v_cache, meta.block_table, meta.seq_lens.to(torch.int32),
                                 meta.query_start_loc.to(torch.int32), int(meta.max_query_len), int(meta.max_seq_len),
                                 float(self.impl.scale), out=out[:n].view(n, hq, D), req_split=split,
                                 k_scale=self._k_scale if fp8 else None, v_scale=self._v_scale if fp8 else None,
                                 sliding_window=window, sinks=sinks)
            step_context.calls["verify"] += 1
            return out

        def _arctic_sinks(self):
            """The layer's attention sinks as the kernel wants them (contiguous f32, one per LOCAL query head), None when
            the layer has none, False when they cannot be mapped onto this rank's heads (the caller then uses vLLM's
            backend).  vLLM shards the parameter over TP only; under Ulysses this rank attends with heads
            [sp_rank * num_heads, (sp_rank + 1) * num_heads) of that shard (rank-major after the all-to-all)."""
            raw = getattr(self.impl, "sinks", None)
            if raw is None:
                raw = getattr(self, "sinks", None)
            if raw is None:
                return None
            cached = getattr(self, "_arctic_sinks_cache", None)
            if cached is not None and cached[0] is raw and cached[1] == raw._version:
                return cached[2]
            n = raw.numel()
            if n == self.num_heads:
                local = raw
            elif self.sp_size > 1 and n == self.num_heads * self.sp_size:
                from ..ulysses import sp_local_head_range
                h0, h1 = sp_local_head_range(n, self.sp_size, parallel_state._SP.rank_in_group)
                local = raw.reshape(-1)[h0:h1]
            else:
                return False
            out = local.detach().reshape(-1).to(torch.float32).contiguous()
            self._arctic_sinks_cache = (raw, raw._version, out)
            return out

        def _arctic_attend(self, query, key, value, **kwargs):
            out = self._arctic_verify(query, key, value)
            if out is None:
                step_context.calls["fallback"] += 1
                out = self._orig_forward(query, key, value, **kwargs)
            return out

        def forward(self, query, key, value, **kwargs):
            """What Dynamo traces: one opaque op (custom_ops.py).  Layers called with extra arguments (MLA's output_shape)
            are not served by the route and keep the direct call."""
            if kwargs or not getattr(self, "layer_name", None):
                return self._arctic_forward(query, key, value, **kwargs)
            return call_attention(query, key, value, self.layer_name)

        def _arctic_forward(self, query, key, value, **kwargs):
            """The eager body behind the op: head repartition around attention (ulysses.py:457-519) + the verify route."""
            if self.sp_size == 1 or is_shift_parallel_mode():
                return self._arctic_attend(query, key, value, **kwargs)
            ua = self._ulysses
            if ua is None:
                ua = self._ulysses = UlyssesAttention(self.sp_size, self.sp_device_group, self.num_heads,
                                                      self.num_kv_heads, self.head_size,
                                                      kv_groups=getattr(self, "_ulysses_kv_groups", None))
            return ua.forward(query, key, value, lambda q_, k_, v_: self._arctic_attend(q_, k_, v_, **kwargs))

    # -----------------------------------------------------------------------------------------------
    class PiecewiseCompileInterpreterPatch(ArcticPatch[PiecewiseCompileInterpreter]):
        def find_symbolic_shape(self, args):
            """The one free shape symbol of a subgraph's arguments (Ulysses makes it appear as N and N/SP*k forms)."""
            from torch._subclasses.fake_tensor import FakeTensor
            symbols = set()
            for x in args:
                if isinstance(x, FakeTensor):
                    for dim in x.shape:
                        if isinstance(dim, torch.SymInt):
                            symbols.update(dim.node.expr.free_symbols)
            assert len(symbols) == 1, f"Expected exactly one symbolic shape, but found {len(symbols)}: {symbols}"
            return next(iter(symbols))

        def call_module(self, target, args, kwargs):
            assert isinstance(target, str)
            # (patched classes lose super(): call torch.fx.Interpreter's method directly)
            output = torch.fx.Interpreter.call_module(self, target, args, kwargs)
            if target not in self.compile_submod_names:
                return output
            index = self.compile_submod_names.index(target)
            submod = self.fetch_attr(target)
            # vLLM assumes ONE SymInt argument carries the runtime shape; under Ulysses several may: pass all positions
            sym = self.find_symbolic_shape(args)
            sym_positions = [i for i, x in enumerate(args) if isinstance(x, torch.SymInt) and x == sym]
            general = self.vllm_backend.compiler_manager.compile(
                submod, args, self.compilation_config.inductor_compile_config, self.compilation_config, graph_index=index,
                num_graphs=len(self.compile_submod_names), runtime_shape=None)
            backend_cls = resolve_obj_by_qualname(current_platform.get_piecewise_backend_cls())
            self.module.__dict__[target] = backend_cls(submod, self.vllm_config, self.graph_pool, index,
                                                       len(self.compile_submod_names), sym_positions, general,
                                                       self.vllm_backend)
            from vllm.compilation.counter import compilation_counter
            compilation_counter.num_piecewise_capturable_graphs_seen += 1
            return output

    # -----------------------------------------------------------------------------------------------
    class UlyssesFusedMoEPatch(ArcticPatch[FusedMoE]):
        def forward(self, hidden_states: torch.Tensor, router_logits: torch.Tensor):
            return self.forward_impl(hidden_states, router_logits)      # not through the custom op

    return [UlyssesModelConfigPatch, UlyssesParallelStatePatch, UlyssesWorkerProcPatch, UlyssesMultiprocExecutorPatch,
            UlyssesAttentionPatch, PiecewiseCompileInterpreterPatch, UlyssesFusedMoEPatch]
