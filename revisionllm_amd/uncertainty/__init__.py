"""Uncertainty scores of the grounding path (reference package ``revisionllm/uncertainty``)."""
