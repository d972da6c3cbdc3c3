"""Swap recipes of the reference's evaluators on the HIP model facade.

* ``simple_swap``  = evaluation/simple_swapping_evaluator.py:38-76 (one content, one
  style, list of mix alphas) -- batched: B pairs at once.
* ``swapping_grid`` = evaluation/content_style_grid_generation_evaluator.py:36-99
  (N contents x M styles, guided-filter post-process with the content as guide) with the
  per-image passes (encode, extract_feat_from_image, Rselfcorr) computed once per image
  instead of once per pair, and the (content, style) pairs sharded over ranks:
  pair (i, j) belongs to rank ((i * M + j) mod world) -- image-parallel, no collective on
  the data path.
Image file I/O (PIL decode/resize/PNG) is left to the caller: tensors in, tensors out.
"""
import torch

from . import glue


def simple_swap(model, content, style, alphas=(1.0,)):
    """content, style: (B,3,H,W) in [-1,1] on the GPU.  Returns {alpha: image (B,3,H,W)}."""
    sp, gl_c = model(content, command="encode")
    fea_c, fea_c1 = model(content, command="extract_feat_from_image")
    fea_s, fea_s1 = model(style, command="extract_feat_from_image")
    fea_c = torch.cat((fea_c, model(fea_c1, command="Rselfcorr")), dim=1)
    fea_s = torch.cat((fea_s, model(fea_s1, command="Rselfcorr")), dim=1)
    corr = model(fea_s, fea_c, command="corrm")
    _, gl_w = model(style, corr, command="encode2")
    out = {}
    for alpha in alphas:
        code = glue.lerp(gl_c, gl_w, alpha)
        out[alpha] = model(sp, code, target=None, command="decode")
    return out


def to_uint8_image(img):
    """ToPILImage()((x.clamp(-1,1)+1)*0.5) quantisation (simple_swapping_evaluator.py:61-62):
    (B,3,H,W) -> (B,H,W,3) uint8 (mul 255, truncate)."""
    v = (img.clamp(-1.0, 1.0) + 1.0) * 0.5
    return (v * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()


def shard_pairs(n_content, n_style, rank=0, world=1):
    """Pairs (i, j) owned by ``rank``: round-robin over the row-major pair index."""
    return [(i, j) for i in range(n_content) for j in range(n_style) if (i * n_style + j) % world == rank]


def swapping_grid(model, contents, styles, rank=0, world=1, smooth=True, pair_batch=8):
    """contents (N,3,H,W), styles (M,3,H,W) on this rank's GPU (every rank holds all images:
    they are small; only the pair work is sharded).  Returns {(i, j): image (3,H,W)} for the
    pairs this rank owns."""
    pairs = shard_pairs(contents.shape[0], styles.shape[0], rank, world)
    need_c = sorted({i for i, _ in pairs})
    need_s = sorted({j for _, j in pairs})
    cache_c, cache_s = {}, {}
    for i in need_c:
        img = contents[i:i + 1]
        sp, _ = model(img, command="encode")
        f0, f1 = model(img, command="extract_feat_from_image")
        cache_c[i] = (sp, torch.cat((f0, model(f1, command="Rselfcorr")), dim=1))
    for j in need_s:
        img = styles[j:j + 1]
        f0, f1 = model(img, command="extract_feat_from_image")
        cache_s[j] = torch.cat((f0, model(f1, command="Rselfcorr")), dim=1)
    out = {}
    for k in range(0, len(pairs), pair_batch):
        chunk = pairs[k:k + pair_batch]
        sp = torch.cat([cache_c[i][0] for i, _ in chunk], 0)
        fc = torch.cat([cache_c[i][1] for i, _ in chunk], 0)
        fs = torch.cat([cache_s[j] for _, j in chunk], 0)
        st = torch.cat([styles[j:j + 1] for _, j in chunk], 0)
        ct = torch.cat([contents[i:i + 1] for i, _ in chunk], 0)
        corr = model(fs, fc, command="corrm")
        _, gl_w = model(st, corr, command="encode2")
        img = model(sp, gl_w, command="decode", target=ct if smooth else None)
        for n, (i, j) in enumerate(chunk):
            out[(i, j)] = img[n]
    return out
