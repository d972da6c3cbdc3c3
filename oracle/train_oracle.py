"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the GAN train step.

Functional restatement (PyTorch CPU autograd over oracle/ppst_oracle.py) of
  * PPSTModel.compute_discriminator_losses   (models/ppst_model.py:105-138)
  * PPSTModel.compute_R1_loss                (models/ppst_model.py:140-159)
  * the D update of PPSTOptimizer.train_discriminator_one_step (optimizers/ppst_optimizer.py:96-130):
    loss = sum(v.mean()), backward, Adam(lr*c, betas**c) with c = R1_every/(1+R1_every).
Parameters that require grad are those whose key starts with ``D.``.
Parity: the discriminator forward is pinned by tests/golden/swap512.npz ("D"); the loss /
gradient composition is plain autograd over those pinned functions.
"""
import torch

import ppst_oracle as O


def d_inputs(sd, real, noise=None, training_stage=2, lambda_StyleCon=1.0):
    """The images the D step classifies: (real, rec (B/2), mix (B)) -- generated under no_grad,
    exactly as the reference's frozen G/E produce them (ppst_model.py:106-131)."""
    B = real.shape[0]
    assert B % 2 == 0, "Batch size must be even on each GPU."
    with torch.no_grad():
        sp = O.encoder_con(sd, real)
        gl, _ = O.encoder_col(sd, real)
        _, feas, feas1 = O.generator(sd, sp, gl, extract_features=True, noise=noise)
        sps = torch.cat((feas, O.rselfcorr(feas1)), dim=1)
        corrms = O.corrm(sps, O.swap(sps))
        corr_self = O.corrm(sps, sps)
        mix = None
        if lambda_StyleCon > 0.0:
            _, gl_w = O.encoder_col(sd, real, corrmatrix=corrms)
            mix = O.generator(sd, O.swap(sp), gl_w, noise=noise)
        _, gl2 = O.encoder_col(sd, real, corrmatrix=corr_self)
        nz = None if noise is None else {k: v[:B // 2] for k, v in noise.items()}
        rec = O.generator(sd, sp[:B // 2], [g[:B // 2] for g in gl2], noise=nz)
    return rec, mix


def d_losses(sd, real, rec, mix, size=512, lambda_GAN=1.0):
    """compute_image_discriminator_losses (ppst_model.py:68-92), LSGAN (loss.py:11-18)."""
    losses = {
        "D_real": O.gan_loss(O.discriminator(sd, real, size), True) * lambda_GAN,
        "D_rec": O.gan_loss(O.discriminator(sd, rec, size), False) * (0.5 * lambda_GAN),
    }
    if mix is not None:
        losses["D_mix"] = O.gan_loss(O.discriminator(sd, mix, size), False) * (0.5 * lambda_GAN)
    return losses


def d_step_grads(sd, real, rec, mix, size=512):
    """losses + d(sum of losses)/d(theta_D) by autograd.  Returns (losses, {key: grad})."""
    keys = [k for k in sd if k.startswith("D.") and sd[k].is_floating_point() and "kernel" not in k]
    p = {k: (sd[k].detach().clone().requires_grad_(True) if k in keys else sd[k]) for k in sd}
    losses = d_losses(p, real, rec, mix, size)
    total = sum(v.mean() for v in losses.values())
    grads = torch.autograd.grad(total, [p[k] for k in keys])
    return {k: float(v) for k, v in losses.items()}, dict(zip(keys, grads))


def r1_loss(sd, real, size=512, lambda_R1=10.0):
    """compute_R1_loss: 0.5*lambda*||d D(x).sum() / dx||^2 per sample."""
    x = real.detach().clone().requires_grad_(True)
    pred = O.discriminator(sd, x, size).sum()
    g, = torch.autograd.grad(pred, [x], create_graph=True, retain_graph=True)
    return g.pow(2).sum(dim=(1, 2, 3)) * (lambda_R1 * 0.5)


def r1_step_grads(sd, real, size=512, lambda_R1=10.0, R1_once_every=16):
    """The lazy-R1 iteration of train_discriminator_one_step (optimizers/ppst_optimizer.py:116-126):
    per-sample penalties and d(mean(penalty) * R1_once_every)/d(theta_D) by double backward."""
    keys = [k for k in sd if k.startswith("D.") and not k.endswith("Blur.kernel")]
    sd2 = dict(sd)
    for k in keys:
        sd2[k] = sd[k].detach().clone().requires_grad_(True)
    pen = r1_loss(sd2, real, size, lambda_R1)
    loss = pen.mean() * R1_once_every
    grads = torch.autograd.grad(loss, [sd2[k] for k in keys], allow_unused=True)
    return pen.detach(), {k: (torch.zeros_like(sd[k]) if g is None else g) for k, g in zip(keys, grads)}


def adam_reference(params, grads, state, lr, beta1, beta2, eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad) single step, functional."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    out = {}
    for k in params:
        m = state.setdefault("m." + k, torch.zeros_like(params[k]))
        v = state.setdefault("v." + k, torch.zeros_like(params[k]))
        m.mul_(beta1).add_(grads[k], alpha=1 - beta1)
        v.mul_(beta2).addcmul_(grads[k], grads[k], value=1 - beta2)
        bc1, bc2 = 1 - beta1 ** t, 1 - beta2 ** t
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
        out[k] = params[k] - (lr / bc1) * (m / denom)
    return out


# ------------------------------------------------------------------ generator step, forward only
def rscl_loss(feat_q, feat_k, feat_k0, queue, nce_T=0.07):
    """rsclLoss.forward (networks/rscl.py:42-64): one positive, n current-batch logits (all -10, see below) and
    the queue (+ feat_k0 columns) as negatives, cross entropy at temperature nce_T."""
    import torch.nn.functional as F
    l_pos = (feat_q * feat_k).sum(1, keepdim=True)
    q = torch.cat((queue, feat_k0.t()), dim=1)
    l_neg2 = feat_q @ q
    n = feat_q.shape[0]
    # quirk kept: after feat_q.view(1, -1, 2048) the reference builds eye(feat_q.size(0)) = eye(1), whose single True
    # broadcasts in masked_fill_ over the whole (1, n, n) block -- every current-batch negative becomes -10 (rscl.py:57-60)
    l_cur = torch.full((n, n), -10.0)
    logits = torch.cat((l_pos, l_cur, l_neg2), dim=1)
    return F.cross_entropy(logits / nce_T, torch.zeros(n, dtype=torch.long))


def g_losses(sd, real, mask, noise=None, lambda_L1=3.0, lambda_GAN=1.0, lambda_StyleCon=1.0, lambda_Maskwarp=10.0, nce_T=0.07,
             size=512):
    """PPSTModel.compute_generator_losses (models/ppst_model.py:161-235), training_stage 2, forward values only,
    lambda_Cycwarp = 0 (lpips is unpinned).  Returns (losses, metrics, queues after the enqueues)."""
    l1 = torch.nn.functional.l1_loss
    B = real.shape[0]
    queues = [sd["criterionNCE.queue_data_A%d" % i].clone() for i in range(4)]
    ptrs = [int(sd["criterionNCE.queue_ptr_A%d" % i]) for i in range(4)]
    losses, metrics = {}, {}
    with torch.no_grad():
        sp = O.encoder_con(sd, real)
        gl, _ = O.encoder_col(sd, real)
        _, feas, feas1 = O.generator(sd, sp, gl, extract_features=True, noise=noise)
        sps = torch.cat((feas, O.rselfcorr(feas1)), dim=1)
        corr = O.corrm(sps, O.swap(sps))
        corr_self = O.corrm(sps, sps)
        _, gl = O.encoder_col(sd, real, corrmatrix=corr_self)
        _, pro_ms, gl_w, pro_mw = O.encoder_col(sd, real, mask=mask, corrmatrix=corr)
        mask_warp = O.model_warp(mask, corr)
        losses["Mask_warp"] = l1(mask_warp, O.swap(mask)) * lambda_Maskwarp
        rec = O.generator(sd, sp, gl, noise=noise)
        losses["G_L1"] = l1(rec, real) * lambda_L1
        mix = O.generator(sd, O.swap(sp), gl_w, noise=noise)
        _, pro_3m, _, _ = O.encoder_col(sd, mix, mask=O.swap(mask))
        _, pro_2m, _, _ = O.encoder_col(sd, rec, mask=mask)
        sp_3 = O.encoder_con(sd, mix)
        nz = None if noise is None else {k: v[:B // 2] for k, v in noise.items()}
        cyc = O.generator(sd, O.swap(sp_3)[:B // 2], [g[:B // 2] for g in gl], noise=nz)
        metrics["L1_dist"] = l1(cyc, real[:B // 2])
        losses["G_L1_cyc"] = metrics["L1_dist"] * 3
        s1 = s2 = 0.0
        for lid in range(0, 12, 3):
            li = lid // 3
            key0, keyw = torch.cat(pro_ms[lid:lid + 3], 0), torch.cat(pro_mw[lid:lid + 3], 0)
            query, query_r = torch.cat(pro_3m[lid:lid + 3], 0), torch.cat(pro_2m[lid:lid + 3], 0)
            s1 = s1 + rscl_loss(query, keyw, key0, queues[li], nce_T)
            s2 = s2 + rscl_loss(query_r, key0, keyw, queues[li], nce_T)
            for keys in (key0[0:1], key0[1:2], key0[2:3], keyw[0:1], keyw[1:2], keyw[2:3]):   # dequeue_and_enqueue, batch 1
                queues[li][:, ptrs[li]:ptrs[li] + 1] = keys.t()
                ptrs[li] = (ptrs[li] + 1) % queues[li].shape[1]
        losses["G_styleContmix"] = s1 * lambda_StyleCon
        losses["G_styleContrec"] = s2 * lambda_StyleCon
        losses["G_GAN_rec"] = O.gan_loss(O.discriminator(sd, rec, size), True) * (lambda_GAN * 0.5)
        losses["G_GAN_mix"] = O.gan_loss(O.discriminator(sd, mix, size), True) * (lambda_GAN * 1.0)
    return losses, metrics, queues
