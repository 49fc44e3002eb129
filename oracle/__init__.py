"""CPU oracle for the ReVisionLLM recursive temporal-grounding inference path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU (torch fp32 / numpy) restatement of the
reference algorithm, written from the reference's cited lines.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only
as the checker / the reported CPU baseline - never as the thing shipped.  The product package
``revisionllm_amd`` never imports ``oracle`` and has no CPU fallback: it raises when the HIP
library is missing.

Parity pin: the reference ships no tests, golden vectors or fixtures for this path (SURVEY.md
section 4), and the LLM arithmetic lives in third-party ``transformers==4.41.2`` (requirements.txt:10),
``torch.nn.MultiheadAttention`` and ``peft`` which are not vendored under /root/reference.  The oracle
is therefore pinned against OUTPUTS OF THE REFERENCE ITSELF RUN IN THE BUILD CONTAINER (torch 2.10,
transformers 5.15, fp32 CPU): ``tests/golden/make_goldens.py`` imports the reference modules from
/root/reference through the shims in ``tests/golden/ref_import.py`` and stores inputs' seeds and the
reference outputs as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every oracle
function against them.

Modules (each function cites the reference file:line it follows):
  adapter    sine position embedding, MultiheadAttention, T2V / self layers, ClipEncoder, Linear projector
  llama      RMSNorm, RoPE, attention with KV cache, SwiGLU block, full forward (HF Llama semantics)
  splice     prepare_inputs_labels_for_multimodal (video rows spliced into text embeddings)
  sampling   logits processors (temperature / top-k / top-p), token selection, the generate loop
  scores     get_entropy_statistics, _topk_pooling + cosine
  recursion  window cutting, hierarchy groups, answer -> window index mapping, hit test, stage-1 IoU
  clip_vit   CLIP towers of the feature extractors (ViT image encoder, causal text transformer),
             pinned by golden g11 from the reference's vendored clip/model.py
(the metric merge of f-1 is pinned directly by golden g10; there is no oracle module for it)
"""
