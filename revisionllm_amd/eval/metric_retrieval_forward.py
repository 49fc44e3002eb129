"""``python -m revisionllm.eval.metric_retrieval_forward`` under its reference name (revisionllm/eval/metric_retrieval_forward.py): the stage-1 / stage-2
log merge + R@k / mIoU of ``eval.metrics`` with that script's defaults (two retrieval runs: stage2_long_100 and stage2_long_33; buffer 0)."""
from .metrics import grounding_metrics_stream, load_predictions, main, merge_stage1_stage2, print_metrics  # noqa: F401

if __name__ == "__main__":
    main()
