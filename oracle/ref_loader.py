"""TEST INFRASTRUCTURE ONLY -- loader that imports the *reference* Python
(/root/reference) under import shims so it runs on CPU in the build container.

Used only by oracle/gen_golden.py (fixture generation) and by CPU tests that
check the oracle restatement against the reference when /root/reference is
present.  Nothing here is imported by the product path (ppst_amd/), and the
reference itself never travels to the GPU box: only the fixtures it produced
(tests/golden/*.npz) do.

The shims are exactly the ones SURVEY.md Appendix B lists; no reference file is
edited or copied:
  1. synthetic ``util`` package populated from util/util.py only, with stub
     torchvision modules (util/__init__.py needs dominate/func_timeout).
  2. util.is_custom_kernel_supported -> False (the reference parses
     torch.version.cuda, which is None here: util/util.py:439-443), so the
     reference takes its own pure-PyTorch fallbacks upfirdn2d_native
     (stylegan2_op/upfirdn2d.py:162-222) and F.leaky_relu (fused_act.py:93-96).
  3. Module.cuda / Tensor.cuda -> identity (encoder_con.py:30 etc.).
  4. models.networks.rscl <- /root/reference/networks/rscl.py (ppst_model.py:11).
  5. stub lpips.LPIPS.
  6. DDP wrapper skipped: PPSTModel.forward(command=...) is called directly.
"""
import importlib.util
import os
import sys
import types
from argparse import Namespace

import torch

REF = os.environ.get("PPST_REFERENCE_DIR", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF, "models", "networks"))


def default_opt(**over):
    """Defaults of every flag that shapes the hot path."""
    o = dict(
        spatial_code_ch=256, global_code_ch=2048, num_classes=0,
        netE_num_downsampling_sp=3, netE_num_downsampling_gl=2,
        netE_nc_steepness=2.0, netE_scale_capacity=1.0,
        netE2_num_downsampling_gl1=3, netE2_num_downsampling_gl2=0,
        netE2_nc_steepness=2.0, netE2_scale_capacity=1.0,
        netG_num_base_resnet_layers=4, netG_use_noise=True,
        netG_scale_capacity=1.0, netG_resnet_ch=256,
        netD_scale_capacity=1.0, use_antialias=True,
        crop_size=512, load_size=512, match_kernel=1, training_stage=2,
        lambda_R1=10.0, lambda_L1=3.0, lambda_GAN=1.0, lambda_StyleCon=1.0,
        lambda_Maskwarp=10.0, lambda_Cycwarp=0.0, lambda_triplet=0.0,
        lambda_hist=0.0, lambda_patch_R1=0.0, nce_T=0.07, num_patches=128,
        nce_includes_all_negatives_from_minibatch=True,
        netE1="StyleGAN2Resnet", netE2="StyleGAN2Resnet",
        netG="StyleGAN2Resnet", netD="StyleGAN2",
        num_gpus=0, local_rank=1, isTrain=True, continue_train=False,
        checkpoints_dir="/nonexistent", name="oracle", pretrained_name=None,
        resume_iter="latest", batch_size=2, lr=1e-3, beta1=0.0, beta2=0.99,
        R1_once_every=16,
    )
    o.update(over)
    return Namespace(**o)


_loaded = {}


def load_reference():
    """Import the reference under shims; returns a namespace of its modules."""
    if _loaded:
        return _loaded["ns"]
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REF)
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)

    # (1) stub torchvision / lpips
    for name in ("torchvision", "torchvision.transforms", "torchvision.models"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    tv = sys.modules["torchvision"]
    tvt = sys.modules["torchvision.transforms"]
    tv.transforms = tvt
    tv.models = sys.modules["torchvision.models"]

    class _Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class _ToTensor:
        def __call__(self, x):
            import numpy as np
            a = np.asarray(x)
            return torch.from_numpy(a).permute(2, 0, 1).float() / 255.0

    tvt.Compose = _Compose
    tvt.ToTensor = _ToTensor

    lp = types.ModuleType("lpips")

    class LPIPS(torch.nn.Module):
        def __init__(self, net="alex"):
            super().__init__()

        def forward(self, a, b):
            return (a - b).abs().mean()

    lp.LPIPS = LPIPS
    sys.modules["lpips"] = lp

    # (1b) synthetic util package from util/util.py only
    upkg = types.ModuleType("util")
    upkg.__path__ = [os.path.join(REF, "util")]
    sys.modules["util"] = upkg
    spec = importlib.util.spec_from_file_location(
        "util.util", os.path.join(REF, "util", "util.py"))
    uu = importlib.util.module_from_spec(spec)
    sys.modules["util.util"] = uu
    spec.loader.exec_module(uu)
    for k, v in vars(uu).items():
        if not k.startswith("__"):
            setattr(upkg, k, v)
    # (2) force the reference's own native fallbacks
    uu.is_custom_kernel_supported = lambda: False
    upkg.is_custom_kernel_supported = lambda: False
    upkg.util = uu

    # (3) .cuda() -> identity
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.cuda = lambda self, *a, **k: self

    import models.networks  # noqa: F401  (reference package)
    # (4) rscl lives at top-level networks/ in the snapshot
    spec = importlib.util.spec_from_file_location(
        "models.networks.rscl", os.path.join(REF, "networks", "rscl.py"))
    rscl = importlib.util.module_from_spec(spec)
    sys.modules["models.networks.rscl"] = rscl
    spec.loader.exec_module(rscl)

    import models.ppst_model as ppst_model
    import models.networks.stylegan2_layers as layers
    import models.networks.generator as generator
    import models.networks.encoder_con as encoder_con
    import models.networks.encoder_col as encoder_col
    import models.networks.discriminator as discriminator
    import models.networks.stylegan2_op as stylegan2_op
    import models.networks.loss as loss

    ns = Namespace(ppst_model=ppst_model, layers=layers, generator=generator,
                   encoder_con=encoder_con, encoder_col=encoder_col,
                   discriminator=discriminator, stylegan2_op=stylegan2_op,
                   util=upkg, loss=loss, rscl=rscl)
    _loaded["ns"] = ns
    return ns


def build_reference_model(opt=None, seed=0):
    ns = load_reference()
    opt = opt or default_opt()
    torch.manual_seed(seed)
    m = ns.ppst_model.PPSTModel(opt)
    m.initialize()
    m.eval()
    return m
