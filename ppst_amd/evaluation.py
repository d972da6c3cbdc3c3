"""Swap recipes of the reference's evaluators on the HIP model facade.

* ``simple_swap``  = evaluation/simple_swapping_evaluator.py:38-76 (one content, one
  style, list of mix alphas) -- batched: B pairs at once.
* ``swapping_grid`` = evaluation/content_style_grid_generation_evaluator.py:36-99
  (N contents x M styles, guided-filter post-process with the content as guide) in two phases:
    1. per-IMAGE passes (encode, extract_feat, Rselfcorr), computed once per image instead of once per pair and
       sharded over ranks by image (image k of the N + M belongs to rank k mod world), batched;
    2. ONE exchange: all_gather of the spatial codes (4.2 MB per content image) and of fea || Rselfcorr (8.4 MB per
       image) -- RCCL on the GPU, the only data-path collective of inference;
    3. per-PAIR passes (corrm, encode2, decode + guided filter), pair (i, j) on rank ((i * M + j) mod world), batched.
  An 8 x 8 grid on 8 ranks costs each rank 2 image passes + 8 pair passes (~5.9 TFLOP) against 16 + 64 on one GPU.
The recipes take and return tensors.  The folder-level front ends at the end of this file (``evaluate_swap_files``,
``evaluate_grid_folder``) add what the reference's evaluators do around them: decode (PIL, a thread pool), resize + normalise
on the device (ppst_amd/imageio.py, Pillow-exact), the uint8 quantisation on the device (glue.tensor2im), PNG encode in the
thread pool (zlib releases the GIL: the encode of one batch overlaps the next batch's kernels), the reference's file names.
"""
import torch

from . import glue


def simple_swap(model, content, style, alphas=(1.0,)):
    """content, style: (B,3,H,W) in [-1,1] on the GPU.  Returns {alpha: image (B,3,H,W)}."""
    if content.shape == style.shape:
        # the encoder passes of the recipe (encode(content) + the E1 / E2 inside the two extract_feat_from_image) as one batch of
        # 3B images, the two generator feature passes as one of 2B: the same work, bit-identical (nothing depends on the batch size)
        B = content.shape[0]
        sp3, gl3 = model(torch.cat((content, content, style), 0), command="encode")
        sp, gl_c = sp3[:B], [g[:B] for g in gl3]
        _, fea, fea1 = model(sp3[B:], [g[B:] for g in gl3], command="extract_feat")
        fea = torch.cat((fea, model(fea1, command="Rselfcorr")), dim=1)
        fea_c, fea_s = fea[:B], fea[B:]
    else:
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


def shard_images(n_content, n_style, rank=0, world=1):
    """Images owned by ``rank``: image k (contents first, then styles) belongs to rank k mod world.
    Returns (content indices, style indices)."""
    own = [k for k in range(n_content + n_style) if k % world == rank]
    return [k for k in own if k < n_content], [k - n_content for k in own if k >= n_content]


def grid_image_pass(model, contents, styles, rank=0, world=1, image_batch=8):
    """Phase 1: the per-image passes of this rank's images, batched.  Returns
    (ci, sp (len(ci),256,h,w), fc (len(ci),512,h,w)), (si, fs (len(si),512,h,w)) -- NCHW-shaped like the commands return."""
    ci, si = shard_images(contents.shape[0], styles.shape[0], rank, world)

    def feats(imgs, want_sp):
        sps, fs = [], []
        for k in range(0, imgs.shape[0], image_batch):
            img = imgs[k:k + image_batch]
            sp, gl = model(img, command="encode")
            f0, f1 = model(sp, gl, command="extract_feat")[1:]
            fs.append(torch.cat((f0, model(f1, command="Rselfcorr")), dim=1).contiguous())
            if want_sp:
                sps.append(sp.contiguous())
        return (torch.cat(sps, 0) if sps else None), (torch.cat(fs, 0) if fs else None)
    sp_c, f_c = feats(contents[ci], True) if ci else (None, None)
    _, f_s = feats(styles[si], False) if si else (None, None)
    return (ci, sp_c, f_c), (si, f_s)


def _gather_rows(local, idx, total, world, like):
    """all_gather of per-image rows: ``local`` (len(idx), ...) of this rank -> (total, ...) table on every rank.
    One padded all_gather_into_tensor (RCCL when the tensors live on the GPU, gloo on the CPU)."""
    import torch.distributed as dist
    per = (total + world - 1) // world
    shape = tuple(like)
    buf = torch.zeros((per,) + shape[1:], device=shape[0], dtype=torch.float32)
    if local is not None:
        buf[:local.shape[0]] = local
    out = torch.empty((world * per,) + shape[1:], device=shape[0], dtype=torch.float32)
    dist.all_gather_into_tensor(out, buf)
    return out.view((world, per) + shape[1:])


def grid_exchange(local_c, local_s, n_content, n_style, world, gathered=None, like=None):
    """Phase 2: assemble the full tables {content i: (sp, fc)}, {style j: fs} from every rank's phase-1 output.
    ``gathered``: list over ranks of phase-1 outputs (single-process simulation of N ranks); otherwise the live
    process group is used.  Tensors keep the commands' NCHW shape; the storage layout is plain contiguous."""
    table_c, table_s = {}, {}
    if gathered is not None:
        for (ci, sp_c, f_c), (si, f_s) in gathered:
            for n, i in enumerate(ci):
                table_c[i] = (sp_c[n:n + 1], f_c[n:n + 1])
            for n, j in enumerate(si):
                table_s[j] = f_s[n:n + 1]
        return table_c, table_s
    (ci, sp_c, f_c), (si, f_s) = local_c, local_s
    if world == 1:
        return grid_exchange(None, None, n_content, n_style, 1, gathered=[(local_c, local_s)])
    ref = sp_c if sp_c is not None else (f_c if f_c is not None else f_s)
    if ref is not None:
        dev, h, w = ref.device, ref.shape[2], ref.shape[3]
    elif like is not None:          # a rank that owns no image (world > N + M) still takes part in both gathers, with zero rows
        dev, h, w = like
    else:
        raise RuntimeError("this rank owns no image: pass like=(device, h, w) of the code grid")
    # rows of one rank: its contents' (sp | fc) = 768 channels, its styles' fs = 512 channels
    nc_max = (n_content + world - 1) // world + 1
    ns_max = (n_style + world - 1) // world + 1
    pack_c = torch.cat((sp_c, f_c), 1) if ci else None
    g_c = _gather_rows(pack_c, ci, nc_max * world, world, (dev, 768, h, w))
    g_s = _gather_rows(f_s, si, ns_max * world, world, (dev, 512, h, w))
    for r in range(world):
        rc, rs = shard_images(n_content, n_style, r, world)
        for n, i in enumerate(rc):
            table_c[i] = (g_c[r, n:n + 1, :256], g_c[r, n:n + 1, 256:])
        for n, j in enumerate(rs):
            table_s[j] = g_s[r, n:n + 1]
    return table_c, table_s


def grid_pair_pass(model, contents, styles, table_c, table_s, rank=0, world=1, smooth=True, pair_batch=8):
    """Phase 3: this rank's (content, style) pairs, batched: corrm -> encode2 -> decode (+ guided filter with the content
    as guide, content_style_grid_generation_evaluator.py:81-93)."""
    pairs = shard_pairs(contents.shape[0], styles.shape[0], rank, world)
    out = {}
    for k in range(0, len(pairs), pair_batch):
        chunk = pairs[k:k + pair_batch]
        sp = torch.cat([table_c[i][0] for i, _ in chunk], 0)
        fc = torch.cat([table_c[i][1] for i, _ in chunk], 0)
        fs = torch.cat([table_s[j] for _, j in chunk], 0)
        st = torch.cat([styles[j:j + 1] for _, j in chunk], 0)
        ct = torch.cat([contents[i:i + 1] for i, _ in chunk], 0)
        corr = model(fs, fc, command="corrm")
        _, gl_w = model(st, corr, command="encode2")
        img = model(sp, gl_w, command="decode", target=ct if smooth else None)
        for n, (i, j) in enumerate(chunk):
            out[(i, j)] = img[n]
    return out


def swapping_grid(model, contents, styles, rank=0, world=1, smooth=True, pair_batch=8, image_batch=8):
    """contents (N,3,H,W), styles (M,3,H,W) on this rank's GPU (every rank holds all images: they are small; the image
    passes and the pair passes are sharded).  Returns {(i, j): image (3,H,W)} for the pairs this rank owns."""
    local_c, local_s = grid_image_pass(model, contents, styles, rank, world, image_batch)
    table_c, table_s = grid_exchange(local_c, local_s, contents.shape[0], styles.shape[0], world,
                                     like=(contents.device, contents.shape[2] // 8, contents.shape[3] // 8))
    return grid_pair_pass(model, contents, styles, table_c, table_s, rank, world, smooth, pair_batch)


# ------------------------------------------------------------------------------------------------- file front ends
_IMG_EXT = (".jpg", ".jpeg", ".png", ".bmp", ".webp")


def _decode(path):
    from PIL import Image
    import numpy as np
    return np.array(Image.open(path).convert("RGB"))


def load_images(paths, load_size=512, device="cuda", pool=None):
    """data/base_dataset.py:85-171 (scale_shortside + make_power_2 + ToTensor + Normalize) for a list of files: decode on
    host threads, everything else on the device.  Returns a list of (1,3,H,W) tensors (sizes may differ per file)."""
    from . import imageio
    arrs = list(pool.map(_decode, paths)) if pool is not None else [_decode(p) for p in paths]
    return [imageio.preprocess(torch.from_numpy(a[None]).to(device), load_size) for a in arrs]


def _save_png(args):
    from PIL import Image
    arr, path = args
    Image.fromarray(arr).save(path)
    return path


def save_images(images, paths, pool=None):
    """images: (B,3,H,W) in [-1,1] on the device -> PNG files.  util.tensor2im quantisation (util/util.py:98-131) on the
    device, one device-to-host copy of the uint8 batch, encode on the pool's threads (returns the futures when a pool is
    given: the caller overlaps them with the next batch and joins at the end)."""
    u8 = glue.tensor2im(images).cpu().numpy()
    jobs = [(u8[i], p) for i, p in enumerate(paths)]
    if pool is None:
        return [_save_png(j) for j in jobs]
    return [pool.submit(_save_png, j) for j in jobs]


def evaluate_swap_files(model, structure_path, texture_path, out_dir, alphas=(1.0,), load_size=512, device="cuda"):
    """evaluation/simple_swapping_evaluator.py:38-76: one structure image, one texture image, one output per mix alpha named
    <structure>_<texture>_<alpha %.2f>.png (ToPILImage quantisation of the clamped image, :61-62).  Returns the paths."""
    import os
    from PIL import Image
    os.makedirs(out_dir, exist_ok=True)
    c, s_ = load_images([os.path.expanduser(structure_path), os.path.expanduser(texture_path)], load_size, device)
    outs = simple_swap(model, c, s_, alphas)
    stem = lambda p: os.path.splitext(os.path.basename(p))[0]
    paths = []
    for alpha in alphas:
        path = os.path.join(out_dir, "%s_%s_%.2f.png" % (stem(structure_path), stem(texture_path), alpha))
        Image.fromarray(to_uint8_image(outs[alpha])[0].cpu().numpy()).save(path)
        paths.append(path)
    return paths


def evaluate_grid_folder(model, dataroot, out_dir, rank=0, world=1, smooth=True, load_size=512, device="cuda", workers=8,
                         pair_batch=8, image_batch=8):
    """evaluation/content_style_grid_generation_evaluator.py:36-99 over <dataroot>/content/* and <dataroot>/style/*: every
    (content, style) pair, guided-filter post-process with the content as guide; files land in <out_dir>/images/ under the
    reference's names (<content>_<style>.png, the inputs as <name>.png; util/html.py:51-75 -- the HTML index itself is not
    written).  All images must come out of the preprocessing at one size (the reference batches them the same way).
    Multi-GPU: every rank reads all inputs (they are small), computes its share (swapping_grid) and writes its own files."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    cdir, sdir = os.path.join(dataroot, "content"), os.path.join(dataroot, "style")
    ls = lambda d: sorted(os.path.join(d, f) for f in os.listdir(d) if f.lower().endswith(_IMG_EXT))
    cpaths, spaths = ls(cdir), ls(sdir)
    if not cpaths or not spaths:
        raise RuntimeError("need images under %s and %s" % (cdir, sdir))
    img_dir = os.path.join(out_dir, "images")
    os.makedirs(img_dir, exist_ok=True)
    stem = lambda p: os.path.splitext(os.path.basename(p))[0]
    with ThreadPoolExecutor(max_workers=workers) as pool:
        imgs = load_images(cpaths + spaths, load_size, device, pool)
        if len({tuple(t.shape) for t in imgs}) != 1:
            raise RuntimeError("images of different sizes after preprocessing: %s" % sorted({tuple(t.shape[2:]) for t in imgs}))
        contents, styles = torch.cat(imgs[:len(cpaths)], 0), torch.cat(imgs[len(cpaths):], 0)
        futures = []
        if rank == 0:   # the top row / first column of the reference's page: the inputs themselves
            futures += save_images(torch.cat((contents, styles), 0), [os.path.join(img_dir, stem(p) + ".png") for p in cpaths + spaths], pool)
        out = swapping_grid(model, contents, styles, rank, world, smooth, pair_batch, image_batch)
        keys = sorted(out)
        for k in range(0, len(keys), pair_batch):
            chunk = keys[k:k + pair_batch]
            names = [os.path.join(img_dir, "%s_%s.png" % (stem(cpaths[i]), stem(spaths[j]))) for i, j in chunk]
            futures += save_images(torch.stack([out[ij] for ij in chunk], 0), names, pool)
        written = [f.result() for f in futures]
    return written
