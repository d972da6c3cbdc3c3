"""Tuning aid (GPU): same-process A/B of the training generator's fusion switches (ppst_amd.train_g.TRAIN_FUSE) on the
bench's train step (one D + one G iteration, batch 2).  python tests/train_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppst_amd import train_g, weights as W
from ppst_amd.ppst_model import Options, create_model
from ppst_amd.train_g import PPSTOptimizer


def main():
    dev = torch.device("cuda", 0)
    sd = W.make_state_dict(0, bias_std=0.1, noise_weight=0.1)
    model = create_model(Options(training_stage=2, lambda_Cycwarp=0.0), state_dict=sd, with_D=True, with_nce=True, device=dev)
    model.noise = "random"
    NB = int(os.environ.get("AB_BATCH", "2"))
    real = W.synthetic_images(40, NB).to(dev)
    g = torch.Generator().manual_seed(7)
    lab = torch.randint(0, 3, (NB, 32, 32), generator=g).repeat_interleave(16, 1).repeat_interleave(16, 2)
    mask = torch.nn.functional.one_hot(lab, 3).permute(0, 3, 1, 2).float().contiguous().to(dev)
    opt = PPSTOptimizer(model)
    data = {"real_A": real, "mask_A": mask}

    def run(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            opt.train_one_step(data, 0); opt.train_one_step(data, 0)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    run(2)
    from ppst_amd import autograd as A, ops, train
    if len(sys.argv) > 1 and sys.argv[1] == "r5":
        # round 5: the backward fusions one at a time (same process, three rounds): the discriminator's fan-in add on the input-gradient
        # conv + skip-branch scale on its consumers, StyledConv's gate on the norm backward, the linear backward in 2-3 launches
        def setall(on):
            train.TRAIN_FUSE.update(d_fanin=on, d_skip_scale=on)
            train_g.TRAIN_FUSE["gate"] = on
            train_g.TRAIN_FUSE["gmp_multi"] = on
            A.FUSE_LINEAR["value"] = on
        for rep in range(3):
            for name, fn in (("all off", lambda: setall(False)),
                             ("D fan-in + skip scale", lambda: (setall(False), train.TRAIN_FUSE.update(d_fanin=True, d_skip_scale=True))),
                             ("StyledConv gate", lambda: (setall(False), train_g.TRAIN_FUSE.update(gate=True))),
                             ("linear backward", lambda: (setall(False), A.FUSE_LINEAR.update(value=True))),
                             ("multi-head GAP/GMP", lambda: (setall(False), train_g.TRAIN_FUSE.update(gmp_multi=True))),
                             ("all on", lambda: setall(True))):
                fn()
                run(1)
                print(rep, "%-24s %.2f ms / step" % (name, run(4)), flush=True)
        return
    for rep in range(3):
        for cfg in ({"merge": False, "res_up2": False}, {"merge": True, "res_up2": False}, {"merge": True, "res_up2": True}):
            train_g.TRAIN_FUSE.update(cfg)
            opt.gen.relearn_overlap() if hasattr(opt.gen, "relearn_overlap") else None
            run(1)
            print(rep, cfg, "%.2f ms / step" % run(4), flush=True)


if __name__ == "__main__":
    main()
