"""``python -m revisionllm.eval.metric_retrieval_forward_chapters`` under its reference name (revisionllm/eval/metric_retrieval_forward_chapters.py):
``eval.metrics`` with the chapters script's defaults - one retrieval run, the merge at buffer -1 (unfiltered) and 0."""
from .metrics import grounding_metrics_stream, load_predictions, merge_stage1_stage2, print_metrics  # noqa: F401
from .metrics import main as _main


def main(argv=None):
    return _main(argv, chapters=True)


if __name__ == "__main__":
    main()
