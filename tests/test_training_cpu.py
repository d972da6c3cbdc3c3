"""CPU: host logic of the training loop (ppst_amd/training.py) and the overlapped gradient all-reduce."""
import json
import os
import subprocess
import sys
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_iteration_counter_matches_reference_schedule(golden_dir, tmp_path):
    """Image-count schedule (save / evaluate / print / done) against the reference's own IterationCounter
    (tests/golden/iter_counter.json, oracle/gen_golden.py:gen_iter_counter)."""
    from ppst_amd.training import IterationCounter
    for ci, case in enumerate(json.load(open(os.path.join(golden_dir, "iter_counter.json")))):
        opt = Namespace(checkpoints_dir=str(tmp_path), name="c%d" % ci, dataset_mode="celebamask", isTrain=True, continue_train=False,
                        resume_iter="latest", pretrained_name=None, **case["opt"])
        ic = IterationCounter(opt)
        for steps, sv, ev, pr, done in case["rows"]:
            assert (ic.steps_so_far, ic.needs_saving(), ic.needs_evaluation(), ic.needs_printing(), ic.completed_training()) == \
                (steps, sv, ev, pr, done)
            if done:
                break
            ic.record_one_iteration()
    # resume: iter.txt written at a save point is picked up by --continue_train
    opt = Namespace(checkpoints_dir=str(tmp_path), name="c0", dataset_mode="celebamask", isTrain=True, continue_train=True,
                    resume_iter="latest", pretrained_name=None, batch_size=2, save_freq=50, evaluation_freq=30, print_freq=8,
                    total_nimgs=200)
    assert IterationCounter(opt).steps_so_far == 150      # the count at the last save point (every 50 images)
    opt.resume_iter = "3k"
    assert IterationCounter(opt).steps_so_far == 3000


def test_dataset_epoch_order_is_a_rank_partition(tmp_path):
    """DistributedSampler-style split: every sample of an epoch on exactly one rank, same permutation on all ranks,
    a different one per epoch."""
    from ppst_amd.training import CelebAMaskDataset
    os.makedirs(tmp_path / "images"); os.makedirs(tmp_path / "labels")
    for i in range(7):
        (tmp_path / "images" / ("%03d.png" % i)).write_bytes(b"")
    dss = [CelebAMaskDataset(str(tmp_path), rank=r, world=3, device="cpu") for r in range(3)]
    seen = sorted(i for d in dss for i in d._order)
    assert seen == list(range(7))
    e0 = [list(d._order) for d in dss]
    for d in dss:
        d.epoch = 1
    assert [d._epoch_order() for d in dss] != e0


def test_optimizer_state_round_trip(tmp_path):
    """Adam moments, step counts and the D / G alternation state survive save -> load (the reference restarts from zero)."""
    from ppst_amd.ppst_model import Options, PPSTModel
    from ppst_amd.train_g import PPSTOptimizer
    from ppst_amd.training import load_optimizer_state, save_optimizer_state
    m = PPSTModel(Options(), with_D=True, with_nce=True)
    opt = PPSTOptimizer(m)
    for f in opt.gen.fp.values():
        f.m.normal_(); f.v.uniform_(); f.step_count = 7
    opt.dis.m.normal_(); opt.dis.step_count = 9; opt.dis.iter_counter = 33; opt.train_mode_counter = 1
    path = save_optimizer_state(opt, str(tmp_path / "o.pth"))
    m2 = PPSTModel(Options(), with_D=True, with_nce=True)
    opt2 = load_optimizer_state(PPSTOptimizer(m2), path)
    for k in opt.gen.fp:
        assert torch.equal(opt.gen.fp[k].m, opt2.gen.fp[k].m) and torch.equal(opt.gen.fp[k].v, opt2.gen.fp[k].v)
        assert opt2.gen.fp[k].step_count == 7
    assert torch.equal(opt.dis.m, opt2.dis.m) and opt2.dis.step_count == 9 and opt2.dis.iter_counter == 33 and opt2.train_mode_counter == 1
    # the flat buffers own parameters and gradients
    assert all(f.owns_parameters() for f in opt2.gen.fp.values())


WORKER_OVERLAP = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from torch import nn
from ppst_amd.train_g import FlatParams, GeneratorTrainer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)

class Net(nn.Module):
    def __init__(self, n):
        super().__init__()
        torch.manual_seed(n)
        self.a = nn.Parameter(torch.randn(n, n)); self.b = nn.Parameter(torch.randn(n)); self.unused = nn.Parameter(torch.randn(3))
        self._flat, self._cache = {}, {}

tr = object.__new__(GeneratorTrainer)
tr.world = world
tr.fp = {"G": FlatParams(Net(5), 1e-3, 0.0, 0.99), "E2": FlatParams(Net(4), 1e-3, 0.0, 0.99), "E1": FlatParams(Net(3), 1e-3, 0.0, 0.99)}
def loss_fn(seed):
    torch.manual_seed(seed)
    tot = 0
    for f in tr.fp.values():
        x = torch.randn(f.net.a.shape[0])
        tot = tot + ((f.net.a @ x + f.net.b) ** 2).sum() + (f.net.a * f.net.a).sum()      # a is used twice
    return tot
for step in range(2):                       # twice: the hook counters must re-arm
    tr.zero_grad()
    tr._install_overlap_hooks(); tr._pending = {}
    for k in tr._done_count: tr._done_count[k] = 0
    loss_fn(100 * step + rank).backward()
    launched = sorted(tr._pending)          # 'unused' never gets a gradient -> nothing completes during backward
    tr.all_reduce()
    got = {k: f.grad.clone() for k, f in tr.fp.items()}
    # expected: average over ranks of the single-rank gradients (autograd.grad: no AccumulateGrad, the hooks stay quiet)
    exp = {}
    for r in range(world):
        ps = [p for f in tr.fp.values() for p in (f.net.a, f.net.b)]
        gs = torch.autograd.grad(loss_fn(100 * step + r), ps)
        for p, g in zip(ps, gs):
            exp[id(p)] = exp.get(id(p), 0) + g / world
    for k, f in tr.fp.items():
        for n_, p in zip(f.names, f.params):
            off, sz = f.offsets[n_]
            mine = got[k][off:off + sz].view_as(p)
            if n_ == "unused":
                assert float(mine.abs().max()) == 0.0
            else:
                assert torch.allclose(mine, exp[id(p)], atol=1e-5), (step, k, n_)
if rank == 0:
    print("OK")
dist.destroy_process_group()
'''


def test_two_rank_gloo_overlapped_gradient_all_reduce(tmp_path):
    """The generator trainer's data-parallel path, world_size 2 on gloo: per-network flat all-reduce launched from the
    post-accumulate hooks (or after backward for networks with gradient-less parameters), averaged, re-armed per step."""
    script = tmp_path / "wo.py"
    script.write_text(WORKER_OVERLAP)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert b"OK" in outs[0][0]
