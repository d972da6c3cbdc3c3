"""PPSTModel facade on the HIP path: ``model(*args, command="<method>", **kw)``.

Mirrors the model-level plugin boundary of the reference (models/base_model.py:114-123
string dispatch; models/ppst_model.py commands encode / encode2 / extract_feat /
extract_feat_from_image / Rselfcorr / corrm / warp / decode), with the same argument
order and return structure (lists of four codes, ``corr`` as a (B,4096,4096) tensor), so the
evaluator recipes (evaluation/simple_swapping_evaluator.py:44-60,
content_style_grid_generation_evaluator.py:36-99) run unchanged on top of it.
State-dict keys are the reference's (``E1. E2. G. D.`` prefixes).
"""
import torch
from torch import nn

from . import glue, ops, weights
from .networks import create_network
from .networks.base_network import as_nchw, to_nhwc


class Options:
    """The flags that shape the hot path with the reference's defaults."""

    def __init__(self, **kw):
        d = dict(netE1="StyleGAN2Resnet", netE2="StyleGAN2Resnet", netG="StyleGAN2Resnet", netD="StyleGAN2",
                 spatial_code_ch=256, global_code_ch=2048, crop_size=512, lambda_GAN=1.0, match_kernel=1,
                 num_gpus=1, local_rank=1, isTrain=False, checkpoints_dir="./checkpoints", name="ppst",
                 resume_iter="latest", pretrained_name=None, training_stage=2, lambda_R1=10.0, lambda_L1=3.0,
                 lambda_StyleCon=1.0, lambda_Maskwarp=10.0, lambda_Cycwarp=0.0, nce_T=0.07)
        d.update(kw)
        self.__dict__.update(d)


class RsclQueues(nn.Module):
    """State + forward of rsclLoss (networks/rscl.py:17-90): four (2048, 128) key queues with their write pointers
    (checkpoint keys ``criterionNCE.queue_data_A{i}`` / ``queue_ptr_A{i}``)."""

    def __init__(self, opt, queue_size=128, dim=2048):
        super().__init__()
        self.opt, self.queue_size = opt, queue_size
        for i in range(4):
            q = torch.nn.functional.normalize(torch.randn(dim, queue_size), dim=0)
            self.register_buffer("queue_data_A%d" % i, q)
            self.register_buffer("queue_ptr_A%d" % i, torch.zeros(1, dtype=torch.long))

    def forward(self, feat_q, feat_k, feat_k0=None, layer=-1):
        queue = getattr(self, "queue_data_A%d" % layer)
        if feat_k0 is None:
            feat_k0 = feat_q[:0]
        return ops.rscl_loss(feat_q, feat_k, feat_k0, queue, getattr(self.opt, "nce_T", 0.07))

    def dequeue_and_enqueue(self, keys, layer=-1):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():      # concat_all_gather (rscl.py:9-15, 67-69)
            parts = [torch.ones_like(keys) for _ in range(dist.get_world_size())]
            dist.all_gather(parts, keys.contiguous())
            parts[dist.get_rank()] = keys
            keys = torch.cat(parts, dim=0)
        bs = keys.size(0)
        q, p = getattr(self, "queue_data_A%d" % layer), getattr(self, "queue_ptr_A%d" % layer)
        ptr = int(p)
        assert self.queue_size % bs == 0
        q[:, ptr:ptr + bs] = keys.t()
        p[0] = (ptr + bs) % self.queue_size


    def enqueue_all(self, keys_per_layer):
        """The 24 ``dequeue_and_enqueue(keys[i:i+1], layer)`` calls of one generator iteration (ppst_model.py:214-219:
        six keys for each of the four queues) with ONE collective: the reference all_gathers 24 separate [1, 2048]
        tensors (8 KB each, latency bound); here the keys of all layers travel as one [24, 2048] all_gather and are
        written in the reference's order (call i enqueues rank 0..world-1's i-th key).  keys_per_layer: list of
        (keys (n, C), layer)."""
        import torch.distributed as dist
        allk = torch.cat([k for k, _ in keys_per_layer], 0).contiguous()
        world = 1
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            world = dist.get_world_size()
            parts = [torch.empty_like(allk) for _ in range(world)]
            dist.all_gather(parts, allk)
            parts[dist.get_rank()] = allk
            allk = torch.stack(parts, 1)                       # (N, world, C): call i -> ranks in order
        else:
            allk = allk[:, None]
        assert self.queue_size % world == 0
        o = 0
        for keys, layer in keys_per_layer:
            q, p = getattr(self, "queue_data_A%d" % layer), getattr(self, "queue_ptr_A%d" % layer)
            ptr = int(p)
            for i in range(keys.shape[0]):
                q[:, ptr:ptr + world] = allk[o + i].t()
                ptr = (ptr + world) % self.queue_size
            p[0] = ptr
            o += keys.shape[0]


class PPSTModel(nn.Module):
    def __init__(self, opt=None, with_D=False, with_nce=False):
        super().__init__()
        self.opt = opt or Options()
        if with_nce:
            self.criterionNCE = RsclQueues(self.opt)
            self.register_buffer("num_discriminator_iters", torch.zeros(1, dtype=torch.long))
        self.E1 = create_network(self.opt, self.opt.netE1, "encoder_con")
        self.E2 = create_network(self.opt, self.opt.netE2, "encoder_col")
        self.G = create_network(self.opt, self.opt.netG, "generator")
        if with_D:
            self.D = create_network(self.opt, self.opt.netD, "discriminator")
        # 'random' draws N(0,1) per call like the reference's NoiseInjection (stylegan2_layers.py:388-390);
        # a dict '<Block>.<conv>' -> (B,1,H,W) pins the noise (parity tests); None forbids non-zero noise weights
        self.noise = "random"
        # True: the feature passes whose image nobody reads (extract_feat_from_image, the loss passes) skip ToRGB.  Off by
        # default: the benchmarked recipe does everything the reference's does, used or not.
        self.skip_unused_rgb = False
        # the Cycwarp term's metric (ppst_model.py:61 ``lpips.LPIPS(net='alex')``): inject a differentiable callable
        # (image_rec, real) -> tensor to train with lambda_Cycwarp > 0 (train_g.GeneratorTrainer.compute_generator_losses)
        # (kept out of nn.Module's registry: an injected lpips module must not add keys to the checkpoint contract)
        self.__dict__["perceptual_metric"] = None

    # BaseModel.forward (models/base_model.py:114-123)
    def forward(self, *args, command=None, **kwargs):
        if command is not None:
            method = getattr(self, command)
            assert callable(method), "[%s] is not a method of %s" % (command, type(self).__name__)
            return method(*args, **kwargs)
        raise ValueError(command)

    def load_weights(self, sd):
        own = self.state_dict()
        missing = [k for k in own if k not in sd]
        if missing:
            raise KeyError("state dict lacks %d keys, e.g. %s" % (len(missing), missing[:3]))
        self.load_state_dict({k: sd[k] for k in own}, strict=True)
        self._weights_changed()
        return self

    def _weights_changed(self):
        """Packed weights, style tables and the trainers' host-side scalar copies are stale after any external write."""
        for net in (self.E1, self.E2, self.G, getattr(self, "D", None)):
            if net is not None and hasattr(net, "_cache"):
                net._cache.clear()
        tr = self.__dict__.get("_trainer")
        if tr is not None:
            tr.invalidate()

    # BaseModel.save / BaseModel.load (models/base_model.py:33-112): same file layout
    # (<checkpoints_dir>/<name>/<iter>_checkpoint.pth + latest_checkpoint.pth symlink), same key
    # walk -- own keys only, D.* skipped at test time, missing keys skipped with a message.  A
    # shape mismatch asks the reference's user on stdin; here ``force`` decides (None: raise,
    # "all": copy the overlapping corner and zero the rest, the reference's "all" answer).
    def save(self, total_steps_so_far):
        import os
        savedir = os.path.join(self.opt.checkpoints_dir, self.opt.name)
        os.makedirs(savedir, exist_ok=True)
        checkpoint_name = "%dk_checkpoint.pth" % (total_steps_so_far // 1000)
        torch.save({k: v.detach().cpu() for k, v in self.state_dict().items()}, os.path.join(savedir, checkpoint_name))
        sympath = os.path.join(savedir, "latest_checkpoint.pth")
        if os.path.lexists(sympath):
            os.remove(sympath)
        os.symlink(checkpoint_name, sympath)
        return os.path.join(savedir, checkpoint_name)

    def load(self, checkpoint_path=None, force=None, verbose=True):
        import os
        opt = self.opt
        if checkpoint_path is None:
            name = opt.pretrained_name if (opt.isTrain and getattr(opt, "pretrained_name", None) is not None) else opt.name
            checkpoint_path = os.path.join(opt.checkpoints_dir, name, "%s_checkpoint.pth" % opt.resume_iter)
        if not os.path.exists(checkpoint_path):
            assert opt.isTrain, "In test mode, the checkpoint file must exist (%s)" % checkpoint_path
            if verbose:
                print("checkpoint %s does not exist! Training will start from scratch" % checkpoint_path)
            return False
        # weights_only: the loader will be aimed at third-party .pth files -- nothing from the file is executed (a checkpoint of
        # this layout is a flat dict of tensors, which the restricted unpickler takes)
        sd = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        with torch.no_grad():
            for name, own in self.state_dict().items():
                if not opt.isTrain and (name.startswith("D.") or name.startswith("Dpatch.")):
                    continue
                if name not in sd:
                    if verbose:
                        print("Key %s does not exist in checkpoint. Skipping..." % name)
                    continue
                param = sd[name]
                if own.shape != param.shape:
                    msg = "Key [%s]: Shape does not match the created model (%s) and loaded checkpoint (%s)" % (
                        name, str(own.shape), str(param.shape))
                    if force != "all":
                        raise ValueError(msg)
                    ms = [min(a, b) for a, b in zip(own.shape, param.shape)]
                    if len(ms) in (1, 2, 4):
                        lo = tuple(slice(0, m) for m in ms)
                        hi = tuple(slice(m, None) for m in ms)
                        own[lo].copy_(param[lo])
                        own[hi].zero_()
                    continue
                own.copy_(param)
        self._weights_changed()  # packed weights / style tables / host scalars are stale
        if verbose:
            print("checkpoint loaded from %s" % checkpoint_path)
        return True

    def per_gpu_initialize(self):
        pass

    def set_perceptual_metric(self, fn):
        self.__dict__["perceptual_metric"] = fn
        return self

    def swap(self, x):
        return glue.swap(x)

    # -- commands (models/ppst_model.py:264-387) ------------------------------
    def encode(self, image, extract_features=False, testtime=False):
        return self.E1(image), self.E2(image, extract_features=extract_features)[0]

    def encode2(self, image, corrmatrix):
        return self.E2(image, corrmatrix=corrmatrix)

    def extract_feat(self, spatial_code, global_code):
        return self.G(spatial_code, global_code, extract_features=True, noise=self.noise)

    def extract_feat_from_image(self, img):
        sp = self.E1(img)
        gl = self.E2(img)[0]
        _, fea, fea1 = self.G(sp, gl, extract_features=True, noise=self.noise, want_rgb=not self.skip_unused_rgb)
        return fea, fea1

    def Rselfcorr(self, fea):
        """(B,64,256,256) -> (B,256,64,64) (ppst_model.py:330-339)."""
        return as_nchw(ops.rselfcorr(to_nhwc(fea).contiguous()))

    def corrm(self, fea, fea0):
        """softmax(cos(fea0_i, fea_j)/0.01) over j -> (B, hw, hw) (ppst_model.py:341-364);
        fea = style/key features, fea0 = content/query features, both (B,512,h,w)."""
        mk = int(getattr(self.opt, "match_kernel", 1))
        k = to_nhwc(fea)
        q = to_nhwc(fea0)
        B, h, w, C = k.shape
        if mk == 1:
            kr, qr = k.reshape(B, h * w, C), q.reshape(B, h * w, C)
        else:
            # F.unfold matching (:345-347): k x k neighbourhoods as rows; the mean is taken over the first 256 unfolded ROWS
            # as the reference slices them.  An even k makes F.unfold return (h + 1)(w + 1) positions: the reference's warp
            # would not accept that matrix either.
            if mk < 1 or mk % 2 == 0:
                raise ValueError("match_kernel must be odd (got %d)" % mk)
            kr, qr = ops.unfold_rows(k, mk), ops.unfold_rows(q, mk)
        kn = ops.corr_prep(kr, 256)
        qn = ops.corr_prep(qr, 256)
        corr = ops.gemm_nt(qn, kn)
        return ops.softmax_rows_(corr, 0.01)

    def warp(self, fea, corr):
        """PPSTModel.warp (ppst_model.py:366-387)."""
        b, c, h, w = fea.shape
        H = corr.shape[1]
        if H != h * w:
            s = int(((h * w) / H) ** 0.5)
            patches = ops.unfold_patches(fea, s)
            return ops.fold_patches(ops.gemm_nn(corr, patches, mode="x3"), c, h, w, s)
        f = to_nhwc(fea).reshape(b, h * w, c)
        return as_nchw(ops.gemm_nn(corr, f.contiguous(), mode="x3").view(b, h, w, c))

    def decode(self, spatial_code, global_code, target=None):
        out = self.G(spatial_code, global_code, noise=self.noise)
        if target is not None:
            # GIFSmoothing(r=30, eps=(0.02*255)^2) with guide = target (ppst_model.py:290-305)
            return ops.guided_filter(glue.tensor2im(target), glue.tensor2im(out), 30, (0.02 * 255) ** 2)
        return out

    def discriminate(self, x):
        return self.D(x)

    def get_visuals_for_snapshot(self, real):
        """models/ppst_model.py:237-248.  The reference's body calls an undefined ``self.E`` (the SwapAE single encoder it was
        forked from) and cannot run; the INTENDED semantics, restated with the two encoders that replaced it: (sp, gl) =
        (E1(real), E2(real)[0]) as in ``encode`` (:264), ``rec = G(sp, gl)``, ``mix = G(sp, swap(gl))`` and ``layout`` = the
        3-component PCA picture of the spatial code (util.visualize_spatial_code, util/util.py:231-254: centre over all pixels,
        project on the first three principal axes, rescale to [-1, 1]) resized bilinearly to the image (util.resize2d_tensor
        :464-476).  During training at most 4 images (2 with several GPUs).  The PCA is host-side visualisation like the
        reference's (numpy / sklearn there, torch.linalg here); the sign of a principal axis is not defined, so ``layout`` is
        parity-unpinned by nature -- rec / mix are the path's own decode."""
        if getattr(self.opt, "isTrain", False):
            real = real[:2] if getattr(self.opt, "num_gpus", 1) > 1 else real[:4]
        with torch.no_grad():
            sp, gl = self.encode(real)
            rec = self.G(sp, gl, noise=self.noise)
            mix = self.G(sp, [self.swap(g) for g in gl], noise=self.noise)
            X = sp.detach().float().permute(0, 2, 3, 1).reshape(-1, sp.shape[1]).cpu().double()
            X = X - X.mean(0, keepdim=True)
            B, _, h, w = sp.shape
            try:
                _, _, Vh = torch.linalg.svd(X, full_matrices=False)
                Z = (X @ Vh[:3].t()).reshape(B, h, w, 3).permute(0, 3, 1, 2)
                Z = (Z - Z.min()) / (Z.max() - Z.min()) * 2 - 1
                layout = torch.nn.functional.interpolate(Z.float(), real.shape[-2:], mode="bilinear", align_corners=False).to(real.device)
            except RuntimeError:
                layout = torch.zeros(B, 3, real.shape[2], real.shape[3], device=real.device)
        return {"real": real, "layout": layout, "rec": rec, "mix": mix}

    # ---- train-step commands (models/ppst_model.py:68-235).  Like the reference's, they return loss TENSORS that carry the
    # graph: ``sum(v.mean() for v in losses.values()).backward()`` (optimizers/ppst_optimizer.py:86-88, :110-111, :121-123)
    # leaves d/d(theta) in ``p.grad`` of the networks' parameters, so a restated PPSTOptimizer runs on top of this facade
    # unchanged (ppst_amd/train_g.py:PPSTOptimizer).  ONE composition per command: the generator iteration is
    # GeneratorTrainer.compute_generator_losses (autograd blocks over HIP kernels, ppst_amd/autograd.py), the discriminator
    # iteration and the lazy R1 penalty are single autograd nodes over DiscriminatorTrainer's taped forward / backward.  Under
    # torch.no_grad() the same code returns values only.
    def trainer(self, **kw):
        """The GeneratorTrainer that owns this model's E1 / E2 / G (and, through it, D's DiscriminatorTrainer): created on
        first use, kept on the model.  Its constructor rebinds every parameter into a flat buffer per network."""
        from .train_g import GeneratorTrainer
        return GeneratorTrainer.for_model(self, **kw)

    def compute_image_discriminator_losses(self, real, rec, mix, cyc=None):
        from .autograd import DLossesFn
        lam = self.opt.lambda_GAN
        if lam == 0.0:
            return {}
        assert cyc is None, "the reference never passes cyc (ppst_model.py:133-138)"
        tr = self.trainer().d_trainer
        names = ["D_real", "D_rec"] + (["D_mix"] if mix is not None else [])
        vals = DLossesFn.apply(tr.anchor, tr, real, rec, mix, float(lam))
        return dict(zip(names, vals))

    def compute_discriminator_losses(self, real, mask=None):
        from .train import d_step_images
        if hasattr(self, "num_discriminator_iters"):
            self.num_discriminator_iters.add_(1)
        with torch.no_grad():                       # ppst_model.py:106-131: rec / mix come from the frozen E1 / E2 / G
            rec, mix, sp, gl = d_step_images(self, real, getattr(self.opt, "lambda_StyleCon", 1.0), want_codes=True)
        return self.compute_image_discriminator_losses(real, rec, mix), {}, sp, gl

    def compute_R1_loss(self, real):
        from .autograd import R1Fn
        lam = float(getattr(self.opt, "lambda_R1", 10.0))
        if lam <= 0.0:
            return {"D_R1": 0.0}
        tr = self.trainer().d_trainer
        return {"D_R1": R1Fn.apply(tr.anchor, tr, real, lam)}

    def compute_generator_losses(self, real, sp_ma, gl_ma, mask):
        """ppst_model.py:161-235 (sp_ma / gl_ma are unused there too).  Differentiable: losses carry grad_fn when gradients are
        enabled; also performs the NCE queue updates."""
        return self.trainer().compute_generator_losses(real, mask)

    # DistributedDataParallel's constructor broadcasts rank 0's parameters and buffers (models/__init__.py:88): without it the
    # NCE queues -- torch.randn per process (networks/rscl.py:24-31) -- and any unseeded initialisation differ between ranks.
    def sync_from_rank0(self):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return self
        with torch.no_grad():
            tr = self.__dict__.get("_trainer")
            if tr is not None:                     # parameters live in flat buffers: one broadcast per network
                for f in list(tr.fp.values()) + ([tr.d_trainer] if tr.d_trainer is not None else []):
                    dist.broadcast(f.flat, 0)
            else:
                for p in self.parameters():
                    dist.broadcast(p.data, 0)
            for b in self.buffers():
                dist.broadcast(b, 0)
        self._weights_changed()
        return self

    def get_parameters_for_mode(self, mode):
        m = {"generator": "G", "contentencoder": "E1", "colorencoder": "E2", "discriminator": "D"}[mode]
        return list(getattr(self, m).parameters()) if hasattr(self, m) else []


def create_model(opt=None, state_dict=None, seed=0, with_D=False, device="cuda", with_nce=False):
    """models.create_model (models/__init__.py:57-72) without the DDP wrapper: inference
    is collective-free."""
    m = PPSTModel(opt, with_D=with_D, with_nce=with_nce)
    if state_dict is None:
        state_dict = weights.make_state_dict(seed, with_D=with_D, with_nce=with_nce)
    m.load_weights(state_dict)
    return m.to(device)
