"""The training loop around the HIP train step: what /root/reference/train.py:24-60 does
with util/iter_counter.py (image-count schedule, iter.txt resume), data/CelebAMask_dataset.py (image + label map ->
{'real_A', 'mask_A' one-hot}) and optimizers/ppst_optimizer.py -- plus optimiser-state checkpointing, which the
reference lacks (it restarts Adam from zero moments on resume).

Host logic only; every arithmetic step is in ppst_amd/train.py / train_g.py (HIP kernels).  Out of scope as in the
survey: visdom / HTML visualiser, the argparse option system, the tmux launcher.
"""
import os
import random
import time

import numpy as np
import torch

from . import glue


class IterationCounter:
    """util/iter_counter.py:7-92: progress is counted in IMAGES; saving / evaluation / printing fire when the image
    count crosses a multiple of their period within one batch; ``iter.txt`` holds the count at the last save."""

    def __init__(self, opt):
        self.opt = opt
        self.iter_record_path = os.path.join(opt.checkpoints_dir, opt.name, "iter.txt")
        self.batch_size = opt.batch_size * (2 if "unaligned" in getattr(opt, "dataset_mode", "") else 1)
        self.time_measurements = {}
        self.steps_so_far = 0
        resume = getattr(opt, "resume_iter", "latest")
        cont = getattr(opt, "isTrain", True) and getattr(opt, "continue_train", False)
        if cont and resume == "latest" and getattr(opt, "pretrained_name", None) is None:
            try:
                self.steps_so_far = int(np.loadtxt(self.iter_record_path, delimiter=",", dtype=int))
                print("Resuming from iteration %d" % self.steps_so_far)
            except Exception:
                print("Could not load iteration record at %s. Starting from beginning." % self.iter_record_path)
        elif cont and str(resume).replace("k", "").isnumeric():
            steps = int(str(resume).replace("k", ""))
            self.steps_so_far = steps * 1000 if "k" in str(resume) else steps

    def record_one_iteration(self, write=True):
        """``write``: only rank 0 writes iter.txt (the reference calls this inside its ``local_rank == 0`` block, train.py:33-56;
        here every rank advances its count, one writes) -- through a temporary file, so a reader never sees a truncated record."""
        if write and self.needs_saving():
            os.makedirs(os.path.dirname(self.iter_record_path), exist_ok=True)
            tmp = self.iter_record_path + ".tmp%d" % os.getpid()
            np.savetxt(tmp, [self.steps_so_far], delimiter=",", fmt="%d")
            os.replace(tmp, self.iter_record_path)
        self.steps_so_far += self.batch_size

    def needs_saving(self):
        return (self.steps_so_far % self.opt.save_freq) < self.batch_size

    def needs_evaluation(self):
        return self.steps_so_far >= self.opt.evaluation_freq and (self.steps_so_far % self.opt.evaluation_freq) < self.batch_size

    def needs_printing(self):
        return (self.steps_so_far % self.opt.print_freq) < self.batch_size

    def completed_training(self):
        return self.steps_so_far >= self.opt.total_nimgs

    class _Timer:
        def __init__(self, name, parent):
            self.name, self.parent = name, parent

        def __enter__(self):
            self.t0 = time.time()

        def __exit__(self, *exc):
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            dt = (time.time() - self.t0) / self.parent.batch_size
            tm = self.parent.time_measurements
            tm[self.name] = dt if self.name not in tm else tm[self.name] * 0.98 + dt * 0.02     # EMA 0.98 (:84-88)

    def time_measurement(self, name):
        return IterationCounter._Timer(name, self)


class MetricTracker:
    """util/metric_tracker.py:9-23: exponential moving average (0.98) of the reported losses."""

    def __init__(self):
        self.metrics = {}

    def update_metrics(self, d, smoothe=True):
        for k, v in d.items():
            v = float(v)
            self.metrics[k] = self.metrics[k] * 0.98 + v * 0.02 if (smoothe and k in self.metrics) else v

    def current_metrics(self):
        return dict(self.metrics)


class CelebAMaskDataset:
    """data/CelebAMask_dataset.py:20-60 + data/__init__.py's per-rank sharding: <dataroot>/images/*.jpg|png and
    <dataroot>/labels/<same stem>.png (integer label map {0, 1, 2}, g_mask.py's aggregation).  Images are decoded on the
    host, resized and normalised ON THE DEVICE with the Pillow-exact kernels (ppst_amd/imageio.py); label maps are turned into
    one-hot masks by the exact glue kernel (:54-60).  ``next()`` yields batches forever (ConfigurableDataLoader,
    data/__init__.py:131-149), shuffled per epoch with a DistributedSampler-style split: sample k of an epoch's permutation
    belongs to rank k mod world; ``batch_size`` is the PER-RANK batch (the reference's ``opt.batch_size / opt.num_gpus``,
    data/__init__.py:116).

    The defaults are the launcher's TRAINING transform (preprocess="resize" + RandomHorizontalFlip, CelebA_launcher.py:17-18,
    base_dataset.py:130-132); the evaluators' transform (short side scaled, no flip) is passed explicitly at its call sites.
    ``preprocess``: "resize" = the launcher's training transform (square ``load_size``, CelebA_launcher.py:17-18);
    "scale_shortside" = the evaluators' transform.  ``flip``: RandomHorizontalFlip as in the reference's training transform
    (``no_flip`` is not set there).  Two stated deviations: (1) the reference builds the image and the label transform
    separately, so its two RandomHorizontalFlips draw independently and image / mask can end up mirrored against each other
    (base_dataset.py:130-132, CelebAMask_dataset.py:17-18); here ONE draw flips both.  (2) the reference resizes the 'L' label
    image with its BICUBIC transform and then compares ``mask_np == i``; at the dataset's native 512 that is the identity, off
    size it invents label values -- here labels are resized NEAREST.  The next batch is decoded on a worker thread while the
    current step runs (the reference: DataLoader workers)."""

    def __init__(self, dataroot, size=512, batch_size=2, rank=0, world=1, device="cuda", seed=0, preprocess="resize",
                 flip=True, prefetch=True):
        assert preprocess in ("resize", "scale_shortside")
        self.size, self.batch_size, self.rank, self.world, self.device = size, batch_size, rank, world, device
        self.preprocess, self.flip, self.prefetch = preprocess, flip, prefetch
        img_dir, lab_dir = os.path.join(dataroot, "images"), os.path.join(dataroot, "labels")
        exts = (".jpg", ".jpeg", ".png")
        names = sorted(f for f in os.listdir(img_dir) if f.lower().endswith(exts))
        self.pairs = [(os.path.join(img_dir, f), os.path.join(lab_dir, os.path.splitext(f)[0] + ".png")) for f in names]
        if not self.pairs:
            raise RuntimeError("no images under %s" % img_dir)
        self.epoch, self.pos, self.rng = 0, 0, random.Random(seed)
        self._order = self._epoch_order()
        self._next = None

    def __len__(self):
        return len(self.pairs)

    def _epoch_order(self):
        g = random.Random(1000003 * self.epoch + 17)        # same permutation on every rank (DistributedSampler.set_epoch)
        order = list(range(len(self.pairs)))
        g.shuffle(order)
        return order[self.rank::self.world] or order[:1]

    def _decode(self, idx):
        """host side: file -> (uint8 HWC image, PIL 'L' label), optionally mirrored together."""
        from PIL import Image
        ip, lp = self.pairs[idx]
        try:
            img = np.asarray(Image.open(ip).convert("RGB"))
            lab = Image.open(lp).convert("L")
        except OSError as err:                                   # CelebAMask_dataset.py:33-38: retry a random index
            print(err)
            return self._decode(self.rng.randrange(len(self.pairs)))
        if self.flip and self.rng.random() < 0.5:
            img = np.ascontiguousarray(img[:, ::-1])
            lab = lab.transpose(Image.FLIP_LEFT_RIGHT)
        return img, lab

    def _to_device(self, img, lab):
        from PIL import Image
        from . import imageio
        fn = imageio.preprocess_resize if self.preprocess == "resize" else imageio.preprocess
        x = fn(torch.from_numpy(img[None]).to(self.device), self.size)[0]
        H, W = x.shape[1], x.shape[2]
        lab = torch.from_numpy(np.asarray(lab.resize((W, H), Image.NEAREST)).astype(np.int64))
        return x, lab

    def _load(self, idx):
        return self._to_device(*self._decode(idx))

    def _decode_batch(self):
        out = []
        while len(out) < self.batch_size:
            if self.pos >= len(self._order):
                self.epoch, self.pos = self.epoch + 1, 0
                self._order = self._epoch_order()
            out.append(self._decode(self._order[self.pos]))
            self.pos += 1
        return out

    def __iter__(self):
        return self

    def __next__(self):
        if self.prefetch:
            import threading
            if self._next is None:
                cur = self._decode_batch()
            else:
                th, box = self._next
                th.join()
                if "err" in box:
                    raise box["err"]
                cur = box["batch"]
            box = {}

            def work():
                try:
                    box["batch"] = self._decode_batch()
                except Exception as e:       # surfaced by the consumer's next call
                    box["err"] = e
            th = threading.Thread(target=work, daemon=True)
            th.start()
            self._next = (th, box)
        else:
            cur = self._decode_batch()
        xs, labs = zip(*(self._to_device(img, lab) for img, lab in cur))
        labels = torch.stack(labs).to(self.device)
        return {"real_A": torch.stack(xs).contiguous(), "mask_A": glue.one_hot_mask(labels)}


class SyntheticMaskDataset:
    """Stand-in with the same interface (no dataset ships with the repo): smooth synthetic portraits + block label maps."""

    def __init__(self, size=512, batch_size=2, rank=0, device="cuda", seed=0):
        self.size, self.batch_size, self.rank, self.device, self.k = size, batch_size, rank, device, seed

    def __iter__(self):
        return self

    def __next__(self):
        from . import weights as W
        self.k += 1
        real = W.synthetic_images(1000 * self.rank + self.k, self.batch_size, self.size).to(self.device)
        g = torch.Generator().manual_seed(7919 * self.rank + self.k)
        s = self.size // 16
        lab = torch.randint(0, 3, (self.batch_size, s, s), generator=g).repeat_interleave(16, 1).repeat_interleave(16, 2)
        return {"real_A": real, "mask_A": glue.one_hot_mask(lab.to(self.device))}


# ---------------------------------------------------------------------------------------------- optimiser state
def save_optimizer_state(optimizer, path):
    """Adam moments + step counts of all four networks and the D / G alternation state.  The reference saves the model
    only (base_model.py:33-41) and resumes with zero moments; this makes a resumed run continue the same trajectory."""
    st = {"train_mode_counter": optimizer.train_mode_counter}
    for k, f in optimizer.gen.fp.items():
        st[k] = {"m": f.m.detach().cpu(), "v": f.v.detach().cpu(), "step": f.step_count}
    d = optimizer.dis
    if d is not None:
        st["D"] = {"m": d.m.detach().cpu(), "v": d.v.detach().cpu(), "step": d.step_count, "iter_counter": d.iter_counter}
    torch.save(st, path)
    return path


def load_optimizer_state(optimizer, path):
    st = torch.load(path, map_location="cpu", weights_only=True)        # nested dicts of tensors / ints only: nothing is executed
    optimizer.train_mode_counter = int(st["train_mode_counter"])
    for k, f in optimizer.gen.fp.items():
        if f.m.numel() != st[k]["m"].numel():
            raise ValueError("optimizer state of %s has %d entries, the network has %d" % (k, st[k]["m"].numel(), f.m.numel()))
        f.m.copy_(st[k]["m"]); f.v.copy_(st[k]["v"]); f.step_count = int(st[k]["step"])
    if optimizer.dis is not None and "D" in st:
        d = optimizer.dis
        d.m.copy_(st["D"]["m"]); d.v.copy_(st["D"]["v"])
        d.step_count, d.iter_counter = int(st["D"]["step"]), int(st["D"]["iter_counter"])
    return optimizer


def train_loop(opt, model, dataset, optimizer, iter_counter=None, log=print, max_iterations=None):
    """train.py:24-60: while not done: batch -> train_one_step (D / G alternation) -> EMA metrics -> rank 0 prints / saves
    (model checkpoint in the reference's layout + optimiser state + iter.txt).  Returns the metric tracker."""
    iter_counter = iter_counter or IterationCounter(opt)
    tracker = MetricTracker()
    rank0 = getattr(opt, "local_rank", 0) == 0
    n = 0
    while not iter_counter.completed_training():
        with iter_counter.time_measurement("data"):
            cur = next(dataset)
        with iter_counter.time_measurement("train"):
            losses = optimizer.train_one_step(cur, iter_counter.steps_so_far)
            tracker.update_metrics(losses, smoothe=True)
        if rank0:
            if iter_counter.needs_printing():
                tm = iter_counter.time_measurements
                log("(iters: %d) %s | %s" % (iter_counter.steps_so_far, " ".join("%s: %.3f" % kv for kv in tm.items()),
                                              " ".join("%s: %.3f" % kv for kv in sorted(tracker.current_metrics().items()))))
            if iter_counter.needs_saving():
                save_all(opt, optimizer, iter_counter.steps_so_far)
        n += 1
        if iter_counter.completed_training() or (max_iterations is not None and n >= max_iterations):
            break
        iter_counter.record_one_iteration(write=rank0)
    if rank0:
        save_all(opt, optimizer, iter_counter.steps_so_far)
        log("Training finished.")
    return tracker


def save_all(opt, optimizer, steps):
    path = optimizer.save(steps)
    save_optimizer_state(optimizer, os.path.join(os.path.dirname(path), "latest_optimizer.pth"))
    return path
