"""Test harness around tests/stubs/vllm (the stand-in for vLLM 0.9.2): installs the stand-in, applies the plugin, and
drives GPUModelRunner.execute_model with a minimal scheduler that follows vLLM V1's contract (new / cached request
data, scheduled spec-decode tokens, num_computed_tokens rewound by the number of rejected drafts)."""
import os
import sys
from typing import Dict, List, Optional

STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")


def _purge():
    for name in [m for m in sys.modules if m == "vllm" or m.startswith("vllm.")]:
        del sys.modules[name]
    # patch classes are built against the vllm modules that were imported then
    from arcticinference_amd.vllm_plugin import args, config, model_runner, swiftkv_model
    model_runner._runner_patch = None
    model_runner.SP_TP_MODE = None
    args._built = None
    config._classes = None
    swiftkv_model._CLASSES = None      # the SwiftKV model classes close over vllm modules too (second test of a process)


def install():
    """Fresh stand-in on sys.path (every test starts from unpatched classes)."""
    _purge()
    if STUBS not in sys.path:
        sys.path.insert(0, STUBS)
    import vllm
    assert vllm.__file__.startswith(STUBS)
    return vllm


def uninstall():
    _purge()
    if STUBS in sys.path:
        sys.path.remove(STUBS)


def load_plugin(worker: bool = True):
    """arctic_inference_plugin() the way vLLM calls it, then (worker=True) what WorkerBase.__init__ does in a worker."""
    from arctic_inference.vllm.plugins import arctic_inference_plugin
    arctic_inference_plugin()
    if worker:
        from vllm.config import VllmConfig
        from vllm.v1.worker.worker_base import WorkerBase
        WorkerBase(VllmConfig())


def init_single_process_groups(vllm_config):
    """World / TP / PP (and, patched, SP ...) groups of a one-process "world"."""
    import torch.distributed as dist
    from vllm.config import set_current_vllm_config
    from vllm.distributed import parallel_state
    if not dist.is_initialized():       # vLLM initialises torch.distributed even for one worker
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    parallel_state.reset_for_tests()
    parallel_state.init_world_group(0)
    set_current_vllm_config(vllm_config)
    pc = vllm_config.parallel_config
    parallel_state.initialize_model_parallel(pc.tensor_parallel_size, pc.pipeline_parallel_size)


class MiniScheduler:
    """Just enough of vLLM's V1 scheduler: all live requests are scheduled every step; a request's step covers its
    not-yet-computed tokens plus the draft tokens proposed for it last step."""

    def __init__(self, block_size: int, max_model_len: int, chunk: Optional[int] = None):
        self.block_size, self.max_model_len, self.chunk = block_size, max_model_len, chunk
        self.reqs: Dict[str, dict] = {}
        self.new: List[str] = []
        self.finished: set = set()
        self.next_block = 1            # block 0 stays unused (vLLM's null block)
        self.stats = dict(drafts=0, draft_tokens=0, accepted=0)

    def add(self, req_id: str, prompt: List[int], temperature: float = 0.0, seed: Optional[int] = None) -> None:
        from types import SimpleNamespace
        self.reqs[req_id] = dict(prompt=list(prompt), out=[], computed=0, spec=[], blocks=[], sent=False,
                                 sampling=SimpleNamespace(temperature=temperature, seed=seed))
        self.new.append(req_id)

    def finish(self, req_id: str) -> None:
        del self.reqs[req_id]
        self.finished.add(req_id)

    def _grow(self, r: dict, upto: int) -> List[int]:
        need = (upto + self.block_size - 1) // self.block_size - len(r["blocks"])
        fresh = list(range(self.next_block, self.next_block + max(need, 0)))
        self.next_block += len(fresh)
        r["blocks"].extend(fresh)
        return fresh

    def schedule(self):
        from vllm.v1.core.sched.output import CachedRequestData, NewRequestData, SchedulerOutput
        new, cached, num, spec = [], [], {}, {}
        for rid, r in self.reqs.items():
            total = len(r["prompt"]) + len(r["out"])
            n = total + len(r["spec"]) - r["computed"]
            if self.chunk is not None and r["computed"] < len(r["prompt"]):
                n = min(n, self.chunk)                      # chunked prefill
            fresh = self._grow(r, r["computed"] + n)
            num[rid] = n
            if r["spec"]:
                spec[rid] = list(r["spec"])
            if not r["sent"]:
                new.append(NewRequestData(rid, list(r["prompt"]), list(r["blocks"]), r["computed"], r["sampling"]))
                r["sent"] = True
            else:
                cached.append(CachedRequestData(rid, False, [], fresh, r["computed"]))
        out = SchedulerOutput(new, cached, num, sum(num.values()), spec, set(self.finished))
        self.finished = set()
        self._last = out
        return out

    def update(self, output) -> Dict[str, List[int]]:
        """Consumes a ModelRunnerOutput; returns the tokens each request emitted this step."""
        emitted = {}
        for i, rid in enumerate(output.req_ids):
            r = self.reqs[rid]
            toks = output.sampled_token_ids[i]
            n_sched = self._last.num_scheduled_tokens[rid]
            n_spec = len(self._last.scheduled_spec_decode_tokens.get(rid, ()))
            r["computed"] += n_sched
            if toks:
                rejected = n_spec - (len(toks) - 1)
                r["computed"] -= rejected                   # the KV of rejected drafts is overwritten later
                if n_spec:
                    self.stats["drafts"] += 1
                    self.stats["draft_tokens"] += n_spec
                    self.stats["accepted"] += len(toks) - 1
            r["out"].extend(toks)
            emitted[rid] = toks
            r["spec"] = list(output.spec_token_ids[i]) if (output.spec_token_ids is not None and toks) else []
        return emitted
