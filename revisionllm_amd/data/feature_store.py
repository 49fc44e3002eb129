"""Reader for the pre-extracted CLIP features the drivers consume.

On-disk formats follow the reference's writers / readers: per-movie ``<movie>.npy`` arrays [ctx_l, 768]
(eval_nlq_retrieval_e2e2.py:246-247), or an LMDB whose values are ``np.savez_compressed`` blobs with key ``features``
(or ``memory_global``) for videos and ``token_features`` / ``cls_features`` for queries (e2e2.py:187-192,238-255;
writers data/feature_extraction/mad_clip_text_extractor.py:101-107).  The ``lmdb`` package is optional (not installed in this image):
without it ``data/mdb_reader.py`` reads the environment's ``data.mdb`` itself; a directory of ``<key>.npz`` files with the same keys is accepted too.

``stage_windows`` is the step right before the hot path: gather the window frames on the host into a PINNED staging
buffer and copy them to the GPU asynchronously on a side stream, so the H2D transfer of query i+1 overlaps the
recursion of query i (PCIe Gen5 x16 ~ 63 GB/s: a 290-window MAD movie is 290*250*768*2 B = 111 MB ~ 2 ms).
"""
import io
import os

import numpy as np
import torch


def _npz_bytes(blob):
    """Decode one ``np.savez[_compressed]`` blob.  ``allow_pickle=False``: feature files hold plain float arrays (the reference's
    writers add an ``allow_pickle`` boolean array, also plain), and a data directory must not be able to run code."""
    if blob is None:
        raise KeyError("key not found in the feature store")
    with io.BytesIO(bytes(blob)) as reader:
        d = np.load(reader, allow_pickle=False)
        return {k: d[k] for k in d.files}


class FeatureStore:
    def __init__(self, feat_folder, q_feat_dir=None, vis_feat_storage="npy"):
        self.feat_folder, self.q_feat_dir, self.kind = feat_folder, q_feat_dir, vis_feat_storage
        self._venv = self._qenv = None
        if vis_feat_storage == "lmdb":
            self._venv = self._open_lmdb(feat_folder)
        if q_feat_dir is not None and os.path.isfile(os.path.join(q_feat_dir, "data.mdb")):
            self._qenv = self._open_lmdb(q_feat_dir)

    @staticmethod
    def _open_lmdb(path):
        try:
            import lmdb
        except ImportError:      # the package is optional: the build's own reader of the data.mdb format does point lookups (data/mdb_reader.py)
            from . import mdb_reader as lmdb
        env = lmdb.open(path, readonly=True, create=False, max_readers=4096 * 8, readahead=False)
        return env.begin(buffers=True)

    def video(self, movie):
        """-> float array [ctx_l, 768]."""
        if self._venv is not None:
            d = _npz_bytes(self._venv.get(movie.encode()))
            return d["features"] if "features" in d else d["memory_global"]
        p = os.path.join(self.feat_folder, movie + ".npy")
        if os.path.exists(p):
            return np.load(p)
        d = dict(np.load(os.path.join(self.feat_folder, movie + ".npz"), allow_pickle=False))
        return d["features"] if "features" in d else d["memory_global"]

    def query(self, query_id):
        """-> (token_features [Lq, 768], cls_features [768])."""
        if self.q_feat_dir is None:
            return None, None
        if self._qenv is not None:
            d = _npz_bytes(self._qenv.get(query_id.encode()))
        else:
            d = dict(np.load(os.path.join(self.q_feat_dir, query_id + ".npz"), allow_pickle=False))
        return d["token_features"], d["cls_features"]


class StagedWindows:
    """One staged window tensor: ``tensor`` bf16 [W, num_frames, 768] on the device, valid once ``event`` has completed.
    ``wait(stream)`` orders a consumer stream after the copy and tells the caching allocator that the block is in use on that
    stream (the tensor was allocated on the staging stream)."""

    def __init__(self, tensor, event):
        self.tensor, self.event = tensor, event

    def wait(self, stream=None):
        stream = stream or torch.cuda.current_stream(self.tensor.device)
        stream.wait_event(self.event)
        self.tensor.record_stream(stream)
        return self.tensor

    def __iter__(self):          # ``dev, ev = stager.stage_windows(...)``
        return iter((self.tensor, self.event))


class WindowStager:
    """Pinned host staging + asynchronous H2D of the window tensor [W, num_frames, 768] (bf16 on the device, as the
    reference casts it: e2e2.py:303-306).  ``depth`` pinned buffers are used round-robin, each with the event of its last
    copy: a buffer is overwritten only after that copy has completed, so the video of query i+1 (or i+depth-1) can be staged
    while query i's copy is still in flight."""

    def __init__(self, device="cuda:0", depth=2, op_dtype=None):
        from .. import hip
        self.op_dtype = hip.op_dtype(op_dtype)     # the engine's operand type (fp16 by default; the reference casts to its model dtype)
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self._slots = [{"buf": None, "event": None} for _ in range(max(1, depth))]
        self._next = 0

    def stage_windows(self, features, frame_idx):
        """features [ctx_l, 768] numpy (any float dtype), frame_idx int32 [W, num_frames] -> ``StagedWindows`` (unpacks as
        ``(device bf16 tensor, event)``).  Consumers on another stream call ``.wait(stream)`` (or wait for the event and
        ``tensor.record_stream(stream)`` themselves)."""
        W, F = frame_idx.shape
        n = W * F * features.shape[1]
        slot = self._slots[self._next]
        self._next = (self._next + 1) % len(self._slots)
        if slot["event"] is not None:
            slot["event"].synchronize()          # the previous copy out of this buffer has finished reading it
        if slot["buf"] is None or slot["buf"].numel() < n:
            slot["buf"] = torch.empty(n, dtype=self.op_dtype).pin_memory()     # (its predecessor is idle: waited above)
        host = slot["buf"][:n].view(W, F, features.shape[1])
        src = torch.from_numpy(np.ascontiguousarray(features))
        host.copy_(src[torch.from_numpy(frame_idx.astype(np.int64))])
        with torch.cuda.stream(self.stream):
            dev = host.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        slot["event"] = ev
        return StagedWindows(dev, ev)
