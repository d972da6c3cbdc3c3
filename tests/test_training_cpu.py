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
for step in range(3):                       # the counters must re-arm; step 0 LEARNS the contribution count per network
    tr.zero_grad()
    tr.begin_backward_overlap()
    loss_fn(100 * step + rank).backward()
    launched = sorted(tr._pending)          # from step 1 on every network's all-reduce starts inside backward() -- although
    assert launched == ([] if step == 0 else ["E1", "E2", "G"]), (step, launched)   # 'unused' never gets a gradient
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
# a changed graph (one contribution fewer) must be refused, not silently reduced early
tr.zero_grad(); tr.begin_backward_overlap()
f = tr.fp["G"]
(f.net.a.sum() * 1.0).backward()
try:
    tr.all_reduce()
    raise SystemExit("a changed contribution count went unnoticed")
except RuntimeError as e:
    assert "relearn_overlap" in str(e)
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


WORKER_RANKS = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
from ppst_amd import ops
from ppst_amd.ppst_model import Options, PPSTModel, RsclQueues

# ---- (1) RsclQueues.enqueue_all == the reference's 24 sequential dequeue_and_enqueue calls (ppst_model.py:214-219 ->
# networks/rscl.py:67-90: each call all_gathers its [1, 2048] key and writes ranks 0..world-1 at the queue pointer)
torch.manual_seed(5)                                   # same initial queues on both ranks
qa, qb = RsclQueues(Options()), RsclQueues(Options())
qb.load_state_dict(qa.state_dict())
for li in range(4):                                    # pointers off zero, and one queue about to wrap
    getattr(qa, "queue_ptr_A%d" % li)[0] = getattr(qb, "queue_ptr_A%d" % li)[0] = (0, 6, 124, 64)[li]
torch.manual_seed(100 + rank)                          # the keys differ per rank
pending = [(torch.randn(6, 2048), li) for li in range(4)]
for keys, li in pending:                               # the reference's order: six calls per layer, one key each
    for i in range(6):
        qa.dequeue_and_enqueue(keys[i:i + 1], li)
qb.enqueue_all(pending)
for li in range(4):
    assert torch.equal(getattr(qa, "queue_data_A%d" % li), getattr(qb, "queue_data_A%d" % li)), li
    assert int(getattr(qa, "queue_ptr_A%d" % li)) == int(getattr(qb, "queue_ptr_A%d" % li)) == ((0, 6, 124, 64)[li] + 12) % 128
# every rank holds the same queue afterwards, and it contains the OTHER rank's keys too
got = [torch.empty_like(qb.queue_data_A0) for _ in range(world)]
dist.all_gather(got, qb.queue_data_A0)
assert torch.equal(got[0], got[1])
assert torch.equal(qb.queue_data_A0[:, rank:12:world].t(), pending[0][0])

# ---- (2) PPSTModel.sync_from_rank0 = the parameter / buffer broadcast of DistributedDataParallel's constructor
# (models/__init__.py:88): per-process torch.randn NCE queues and unseeded parameters must end up rank 0's
torch.manual_seed(1000 + rank)
m = PPSTModel(Options(), with_D=True, with_nce=True)
for p in m.parameters():
    p.data.normal_()
sig = lambda: torch.stack([t.double().sum() for t in list(m.parameters()) + [b.double() for b in m.buffers()]])
before = [torch.empty_like(sig()) for _ in range(world)]
dist.all_gather(before, sig())
assert not torch.equal(before[0], before[1])
tr = m.trainer() if rank == 0 else None                # mixed: rank 0 already flattened, rank 1 not yet -- both layouts broadcast
if rank == 1:
    m.trainer()
m.sync_from_rank0()
after = [torch.empty_like(sig()) for _ in range(world)]
dist.all_gather(after, sig())
assert torch.equal(after[0], after[1]) and torch.equal(after[0], before[0])
assert all(f.owns_parameters() for f in m.trainer().fp.values()) and m.trainer().d_trainer.owns_parameters()

# ---- (3) the deferred discriminator step: async all-reduce now, average + Adam in front of the next use of D
def adam_cpu(p, g, m_, v, lr, b1, b2, eps, step):       # torch.optim.Adam's update (the HIP kernel needs a GPU)
    m_.mul_(b1).add_(g, alpha=1 - b1); v.mul_(b2).addcmul_(g, g, value=1 - b2)
    p.sub_(lr * (m_ / (1 - b1 ** step)) / ((v / (1 - b2 ** step)).sqrt() + eps))
ops.adam_step_ = adam_cpu
d = m.trainer().d_trainer
d.world = world
w0 = d.flat.clone()
d.zero_grad()
d.grad.fill_(float(rank + 1))                          # rank gradients 1 and 2 -> average 1.5
d.step_deferred()
assert d._pending is not None and torch.equal(d.flat, w0)          # nothing applied yet
d.finish_pending()
assert d._pending is None and d.step_count == 1
assert torch.allclose(d.grad, torch.full_like(d.grad, 1.5))
exp = w0.clone(); adam_cpu(exp, torch.full_like(w0, 1.5), torch.zeros_like(w0), torch.zeros_like(w0), d.lr, d.b1, d.b2, d.eps, 1)
assert torch.allclose(d.flat, exp)
d.grad.fill_(float(rank)); d.step_deferred()
d.zero_grad()                                          # the next zero_grad settles an owed step first, then clears
assert d._pending is None and d.step_count == 2 and float(d.grad.abs().max()) == 0.0
# the entry points that zero / reduce / step on their own settle an owed step first too (ADVICE r3: losses_and_grads used to zero
# the buffer the all-reduce was still reading; all_reduce / adam used to run past a pending handle)
d.d_forward = lambda *a, **k: ({}, None)               # (the real forward / backward need the GPU: only the bookkeeping is under test)
d.d_backward = lambda st, gouts=None: None
w2 = d.flat.clone(); m2, v2 = d.m.clone(), d.v.clone()
d.grad.fill_(float(rank + 3)); d.step_deferred()       # gradients 3 and 4 -> the owed step must use 3.5
d.losses_and_grads(None, None, None)
assert d._pending is None and d.step_count == 3 and float(d.grad.abs().max()) == 0.0
exp = w2.clone(); adam_cpu(exp, torch.full_like(w2, 3.5), m2, v2, d.lr, d.b1, d.b2, d.eps, 3)
assert torch.allclose(d.flat, exp), float((d.flat - exp).abs().max())
d.r1_forward = lambda real, lam: (torch.zeros(2), None)
d.r1_backward = lambda st, g: None
d.grad.fill_(1.0); d.step_deferred()
d.r1_losses_and_grads(None)
assert d._pending is None and d.step_count == 4 and float(d.grad.abs().max()) == 0.0
d.grad.fill_(float(rank)); d.step_deferred()
d.all_reduce()                                         # settles (step 5), then averages what is in the buffer now
assert d._pending is None and d.step_count == 5
d.grad.fill_(2.0); d.step_deferred()
d.adam()                                               # the owed step (6) first, then this one (7)
assert d._pending is None and d.step_count == 7
if rank == 0:
    print("OK")
dist.destroy_process_group()
'''


def test_two_rank_gloo_nce_enqueue_order_broadcast_and_deferred_d_step(tmp_path):
    """world_size 2 on gloo: (1) the fused NCE enqueue writes exactly what the reference's 24 per-key all_gather calls
    write, incl. wrap-around; (2) sync_from_rank0 makes parameters / buffers rank 0's (DDP's constructor broadcast);
    (3) DiscriminatorTrainer.step_deferred / finish_pending average the gradient and apply Adam once, before the next use."""
    script = tmp_path / "wr.py"
    script.write_text(WORKER_RANKS)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29623")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert b"OK" in outs[0][0]


def test_scalar_cache_invalidated_by_weight_load():
    """FlatParams.scalar() caches the NoiseInjection weights on the host; a load_state_dict after the first generator step
    writes through the parameter views (their version counters, not the flat buffer's): load_weights() must invalidate."""
    from ppst_amd.ppst_model import Options, PPSTModel
    from ppst_amd import weights as W
    m = PPSTModel(Options(), with_D=False, with_nce=False)
    m.load_weights(W.make_state_dict(1, with_D=False, with_nce=False, noise_weight=0.25))
    from ppst_amd.train_g import FlatParams
    f = FlatParams(m.G, 1e-3, 0.0, 0.99)
    m.__dict__["_trainer"] = type("T", (), {"fp": {"G": f}, "d_trainer": None,
                                             "invalidate": lambda self: [x.invalidate() for x in self.fp.values()]})()
    name = "HeadResnetBlock0.conv1.noise.weight"
    assert abs(f.scalar(name) - 0.25) < 1e-7
    m.load_weights(W.make_state_dict(2, with_D=False, with_nce=False, noise_weight=0.5))
    assert abs(f.scalar(name) - 0.5) < 1e-7 and f.owns_parameters()
