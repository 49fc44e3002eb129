"""Host utilities.  ``disable_torch_init`` keeps the reference's name (revisionllm/utils.py:93-98); here model
creation never runs torch's default initialisers (weights are packed straight into HBM), so it is a no-op."""


def disable_torch_init():
    return None
