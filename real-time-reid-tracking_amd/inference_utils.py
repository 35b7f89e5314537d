"""Host mirror of the evaluation script's post-processing on the HIP library.

* ``diminish_camera_bias(embeddings, cams, la=0.05)``   reid/inference_utils.py:5-15
* ``smooth_tracklets(embeddings, seqs, indices_valid)``   reid/inference_utils.py:18-27
* ``extract_descriptors(model_or_engine, images, flip=True)``: what ``inference_efficient`` + the averaging at
  reid/image_reid_inference.py:112-123,252-253 produce for one set of images - ``normalize((d(x) + d(hflip x)) / 2)`` with
  ``d = cat(normalize(emb), normalize(logits))``.  ``images`` is float32 [N,3,256,128] after the caller's transform
  (the script uses ImageNet mean/std, data_transforms.py:56-130).
"""
import numpy as np

from .engine import get_engine


def _np(a):
    return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)


def diminish_camera_bias(embeddings, cams, la=0.05, device=0):
    """Returns the de-biased, row-normalised embeddings (same container type as the input; the reference works in place and
    returns its argument).  Camera ids without rows are skipped (the reference's torch.inverse would raise on them)."""
    out = get_engine(device).cam_debias(_np(embeddings), _np(cams), la)
    if hasattr(embeddings, "detach"):
        import torch
        res = torch.from_numpy(out).to(embeddings.dtype)
        embeddings.copy_(res)
        return embeddings
    return out


def smooth_tracklets(embeddings, seqs, indices_valid, device=0):
    """Same arguments as the reference: rows of one tracklet (equal ``seqs``) that are ``indices_valid`` move to
    0.1 * row + 0.9 * tracklet mean.  In place for torch tensors (the reference assigns into its argument) and returned."""
    out = get_engine(device).smooth_tracklets(_np(embeddings), _np(seqs), _np(indices_valid), keep=0.1)
    if hasattr(embeddings, "detach"):
        import torch
        embeddings.copy_(torch.from_numpy(out).to(embeddings.dtype))
        return embeddings
    return out


def extract_descriptors(images, flip=True, device=0):
    return get_engine(device).descriptor_f32_nchw(_np(images), flip)
