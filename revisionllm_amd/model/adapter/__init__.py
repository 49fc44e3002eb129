from .tensor_utils import pad_sequences_1d  # noqa: F401
