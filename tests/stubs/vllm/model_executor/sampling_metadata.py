class SamplingMetadata:
    pass
