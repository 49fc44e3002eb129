"""Host-side multimodal helpers with the reference's surface (revisionllm/mm_utils.py).

``tokenizer_image_token`` is on the hot path (inference.py:35); the rest is kept for API compatibility.
The tokenizer is duck-typed: ``tokenizer(text).input_ids`` and ``tokenizer.bos_token_id``.
"""
import base64
from io import BytesIO

import numpy as np
import torch

from .constants import DEFAULT_IMAGE_TOKEN, DEFAULT_MEMORY_TOKEN, IMAGE_TOKEN_INDEX, MEMORY_TOKEN_INDEX


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    """Tokenise ``prompt`` chunk-wise around ``<video>`` (and ``<memory>``), inserting the sentinel ids.

    mm_utils.py:22-75: every chunk is tokenised separately (each gets its own BOS); the BOS is kept once at the
    front and stripped from later chunks; one ``image_token_index`` joins consecutive chunks.  With a
    ``<memory>`` marker after the video the layout is [text0, -200, text1, -300, text2].
    """
    pieces = prompt.split(DEFAULT_IMAGE_TOKEN)
    with_memory = len(pieces) > 1 and DEFAULT_MEMORY_TOKEN in pieces[1]
    if with_memory:
        chunks = [tokenizer(pieces[0]).input_ids] + [tokenizer(m).input_ids for m in pieces[1].split(DEFAULT_MEMORY_TOKEN)]
    else:
        chunks = [tokenizer(p).input_ids for p in pieces]

    ids, skip = [], 0
    if chunks and len(chunks[0]) > 0 and chunks[0][0] == tokenizer.bos_token_id:
        skip = 1
        ids.append(chunks[0][0])
    elif getattr(tokenizer, "name", None) == "GLMTokenizer":
        skip = 2
        ids = list(chunks[0][:2])

    def joined(parts):
        # the separator carries ``skip`` leading dummies so that x[skip:] leaves exactly one sentinel
        sep = [image_token_index] * (skip + 1)
        seq = []
        for i, part in enumerate(parts):
            seq.append(part)
            if i + 1 < len(parts):
                seq.append(sep)
        return seq

    if with_memory:
        for part in joined(chunks[:2]):
            ids.extend(part[skip:])
        ids.append(MEMORY_TOKEN_INDEX)
        ids.extend(chunks[2])
    else:
        for part in joined(chunks):
            ids.extend(part[skip:])

    if return_tensors is None:
        return ids
    if return_tensors == "pt":
        return torch.tensor(ids, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")


def get_model_name_from_path(model_path):
    parts = model_path.strip("/").split("/")
    return parts[-2] + "_" + parts[-1] if parts[-1].startswith("checkpoint-") else parts[-1]


class KeywordsStoppingCriteria:
    """Stops when a keyword's ids (or its decoded text in the last few tokens) appear (mm_utils.py:89-112).
    ``inference()`` constructs it but never passes it to ``generate`` (inference.py:42) - kept for the API."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = keywords
        self.keyword_ids = []
        for kw in keywords:
            ids = tokenizer(kw).input_ids
            if len(ids) > 1 and ids[0] == tokenizer.bos_token_id:
                ids = ids[1:]
            self.keyword_ids.append(torch.tensor(ids))
        self.tokenizer = tokenizer
        self.start_len = input_ids.shape[1]

    def __call__(self, output_ids, scores=None, **kwargs) -> bool:
        assert output_ids.shape[0] == 1, "Only support batch size 1 (yet)"
        for kid in self.keyword_ids:
            kid = kid.to(output_ids.device)
            if output_ids.shape[1] >= kid.shape[0] and output_ids[0, -kid.shape[0]:].equal(kid):
                return True
        offset = min(output_ids.shape[1] - self.start_len, 3)
        text = self.tokenizer.batch_decode(output_ids[:, -offset:], skip_special_tokens=True)[0]
        return any(kw in text for kw in self.keywords)


def print_trainable_parameters(model):
    params = list(model.named_parameters()) if hasattr(model, "named_parameters") else []
    total = sum(p.numel() for _, p in params)
    train = sum(p.numel() for _, p in params if getattr(p, "requires_grad", False))
    print(f"trainable params: {train} || all params: {total} || trainable%: {100 * train / max(total, 1):.2f}")


def load_image_from_base64(image):
    from PIL import Image
    return Image.open(BytesIO(base64.b64decode(image)))


def process_images(images, image_processor, model_cfg):
    return image_processor(images, return_tensors="pt")["pixel_values"]


class VideoExtractor:
    """Uniform frame sampler for raw videos (mm_utils.py:126-174); demo only, needs ``decord`` (not in this image)."""

    def __init__(self, N=100):
        self.N = N

    def extract(self, data, start_end=None, sample_fps=0):
        try:
            import decord
        except ImportError as e:  # upstream of the accelerated path; pre-extracted CLIP features are the input here
            raise ImportError("VideoExtractor needs the 'decord' package to read raw videos") from e
        reader = decord.VideoReader(data["video"], num_threads=1)
        if start_end is None:
            total, start, end = len(reader), 0, len(reader) - 1
        else:
            start, end = int(start_end[0]), int(start_end[1])
            total = end - start + 1
        fps = reader.get_avg_fps()
        split = data.get("split", None)
        if split is not None:
            start, end = max(int(fps * split[0]), 0), min(int(fps * split[1]), total - 1)
        n = int((total * sample_fps) // fps) if sample_fps > 0 else self.N
        idx = np.linspace(start, end, n, dtype=np.int32)
        reader.skip_frames(1)
        frames = reader.get_batch(idx).asnumpy()
        return data["id"], torch.from_numpy(frames.transpose((0, 3, 1, 2))), idx
