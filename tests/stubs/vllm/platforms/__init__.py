class _Platform:
    rocm = True

    def is_rocm(self) -> bool:
        return self.rocm

    def is_cuda(self) -> bool:
        return False

    def get_piecewise_backend_cls(self) -> str:
        return "vllm.compilation.backends.PiecewiseBackend"


current_platform = _Platform()
