class EngineCoreProc:
    ran = 0

    @staticmethod
    def run_engine_core(*args, **kwargs):
        EngineCoreProc.ran += 1
        return ("engine core", args, kwargs)
