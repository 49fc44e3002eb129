"""Frame / query feature extraction with the CLIP towers - mirror of ``ClipFeatureExtractor``
(revisionllm/data/feature_extraction/clip_extractor.py:13-54) on the HIP kernels.

Differences from the reference, all at the IO edge: frames are handed over as a tensor [T,3,H,W] (what its ``VideoLoader``
produces from ffmpeg; video decoding is not part of this build) and the CLIP weights come from ``ClipTowers`` (a loaded
checkpoint or synthetic).  The on-disk format of the results is what ``data.feature_store`` reads back (f-2).
"""
import math

import torch

from .clip_model import ClipTowers

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def preprocess(frames):
    """uint8 / float [T,3,H,W] in 0..255 -> (x / 255 - mean) / (std + 1e-8)   (Preprocessing, clip_extractor.py:76-97)."""
    x = frames.float() / 255.0
    mean = torch.tensor(CLIP_MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, device=x.device).view(1, 3, 1, 1)
    return (x - mean) / (std + 1e-8)


class ClipFeatureExtractor:
    def __init__(self, towers: ClipTowers, tokenizer=None):
        self.clip_extractor = towers
        self.tokenizer = tokenizer            # ClipTokenizer (needs CLIP's merge table); only encode_text uses it
        self.device = towers.device

    @torch.no_grad()
    def encode_video(self, frames, bsz=60):
        """frames [T,3,H,W] (0..255) -> f32 [T, d]: batches of ``bsz`` frames through ``encode_image`` (clip_extractor.py:22-37)."""
        x = preprocess(frames)
        out = [self.clip_extractor.encode_image(x[i * bsz:(i + 1) * bsz]) for i in range(int(math.ceil(len(x) / bsz)))]
        return torch.cat(out, 0) if out else torch.empty(0, self.clip_extractor.cfg["embed_dim"], device=self.device)

    @torch.no_grad()
    def encode_text(self, text_list, bsz=60, tokens=None):
        """-> (list of [L_j, d] token features = last_hidden_state[1 : len-1], list of [d] EOT features = pooler_output)
        (clip_extractor.py:39-54).  ``tokens`` [n,77] may be given instead of text (pre-tokenised queries)."""
        if tokens is None:
            if self.tokenizer is None:
                raise ValueError("encode_text needs a ClipTokenizer (CLIP's merge table) or pre-tokenised `tokens`")
            tokens = self.tokenizer.tokenize(text_list, context_length=self.clip_extractor.cfg["ctx"])
        feats, eots = [], []
        for i in range(int(math.ceil(len(tokens) / bsz))):
            t = tokens[i * bsz:(i + 1) * bsz]
            out = self.clip_extractor.encode_text(t)
            valid = (t != 0).sum(1).tolist()
            for j, n in enumerate(valid):
                eots.append(out["pooler_output"][j])
                feats.append(out["last_hidden_state"][j, 1:n - 1])
        return feats, eots
