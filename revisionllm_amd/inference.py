"""``inference`` / ``inference_stage1`` with the reference's signatures and behaviour (revisionllm/inference.py:28-75,
126-166): build the Vicuna-v1 prompt, tokenise around ``<video>``, repeat the ids per batch row, ``model.generate``
with sampling at T = 0.05, decode, strip the stop string."""
import torch

from .constants import IMAGE_TOKEN_INDEX
from .conversation import SeparatorStyle, conv_templates
from .mm_utils import KeywordsStoppingCriteria, tokenizer_image_token


def _prompt_ids(query, tokenizer, batch):
    conv = conv_templates["v1"].copy()
    conv.append_message(conv.roles[0], query)
    conv.append_message(conv.roles[1], None)
    ids = tokenizer_image_token(conv.get_prompt(), tokenizer, IMAGE_TOKEN_INDEX, return_tensors="pt").unsqueeze(0)
    stop_str = conv.sep if conv.sep_style != SeparatorStyle.TWO else conv.sep2
    return ids.repeat(batch, 1), stop_str


def _decode(tokenizer, output_ids, input_ids, stop_str):
    n_in = input_ids.shape[1]
    n_diff = (input_ids.to(output_ids.device) != output_ids[:, :n_in]).sum().item()
    if n_diff > 0:
        print(f"[Warning] {n_diff} output_ids are not the same as the input_ids")
    outputs = tokenizer.batch_decode(output_ids[:, n_in:], skip_special_tokens=True)
    for i, o in enumerate(outputs):
        o = o.strip()
        if o.endswith(stop_str):
            o = o[:-len(stop_str)]
        outputs[i] = o.strip()
    return outputs


def _check_engine(model):
    """The host has just synchronised (decode): the cheapest place to notice that an in-kernel hand-off wait ever gave up
    (the affected outputs are NaN-poisoned; this turns them into an exception)."""
    eng = getattr(model, "engine", None)
    if eng is not None:
        eng.check_handoff_status()


def inference(model, image, query_feats, query, tokenizer, visual_memory=None, prefix_memory=None, return_list=False):
    if visual_memory is not None:
        query = query + "<memory>"
    input_ids, stop_str = _prompt_ids(query, tokenizer, image.shape[0])
    KeywordsStoppingCriteria([stop_str], tokenizer, input_ids)  # constructed, never used (inference.py:42)
    with torch.inference_mode():
        model_output = model.generate(input_ids, images=image, query_feats=query_feats, do_sample=True, temperature=0.05,
                                      num_beams=1, max_new_tokens=1024, use_cache=True, visual_memory=visual_memory,
                                      prefix_memory=prefix_memory, output_scores=True, return_dict_in_generate=True,
                                      output_hidden_states=True)
    outputs = _decode(tokenizer, model_output["sequences"], input_ids, stop_str)
    _check_engine(model)
    if len(outputs) == 1 and not return_list:
        outputs = outputs[0]
    return outputs, model_output


def inference_stage1(model, image, query, tokenizer):
    input_ids, stop_str = _prompt_ids(query, tokenizer, image.shape[0])
    with torch.inference_mode():
        output_ids = model.generate(input_ids, images=image, query_feats=None, do_sample=True, temperature=0.05, num_beams=1,
                                    max_new_tokens=1024, use_cache=True, visual_memory=None, prefix_memory=None)
    outputs = _decode(tokenizer, output_ids, input_ids, stop_str)
    _check_engine(model)
    return outputs
