"""Gate tape: a verification aid for the backward pass (tests/gstep_diag.py:compare_gstep_replay).

The generator iteration is piecewise linear in many places -- leaky-ReLU / ReLU / PReLU slopes, the arg-max of the global max
pooling, the sign of the L1 loss.  A pre-activation that lies within the rounding error of 0 takes the other branch in a run
with differently rounded convolutions, and changes that element's gradient by O(1): the production convs (bf16 hi + lo
operands) therefore differ from an exact-fp32 run by (a) operand rounding, ~2^-17 per product, and (b) a handful of flipped
gates whose effect the cancelling gradient sums amplify.  To tell the two apart, the backward pass can RECORD every gate
decision of one run and REPLAY them in another:

    gates.start("record");  <backward of the exact-fp32 run>;   tape = gates.stop()
    gates.start("replay", tape);  <backward of the production run>;  flips = gates.stop()

In replay mode a gate site uses the recorded decision instead of its own and counts how many of its own decisions differ
(``flips``: list of (site, elements, flipped)).  What then remains between the two runs' gradients is operand rounding only.
Sites are visited in backward order, which is deterministic for a fixed graph; a mismatch of site name or shape raises.

MODE None (always, outside that test): every function returns its argument unchanged -- no copies, no extra launches.
"""
import torch

MODE = {"value": None}
_TAPE, _POS, _FLIPS = [], {"i": 0}, []


def start(mode, tape=None):
    assert mode in ("record", "replay")
    MODE["value"] = mode
    _TAPE.clear(); _FLIPS.clear()
    _POS["i"] = 0
    if mode == "replay":
        _TAPE.extend(tape)


def stop():
    mode = MODE["value"]
    MODE["value"] = None
    if mode == "record":
        out = list(_TAPE)
        _TAPE.clear()
        return out
    if _POS["i"] != len(_TAPE):
        raise RuntimeError("gate replay consumed %d of %d recorded sites" % (_POS["i"], len(_TAPE)))
    _TAPE.clear()
    return list(_FLIPS)


def _next(site, shape):
    i = _POS["i"]
    if i >= len(_TAPE):
        raise RuntimeError("gate replay ran past the recorded tape at site %s" % site)
    name, data = _TAPE[i]
    _POS["i"] = i + 1
    ref = data[0] if isinstance(data, tuple) else data
    if name != site or tuple(ref.shape) != tuple(shape):
        raise RuntimeError("gate replay out of step: site %s %s, tape has %s %s" % (site, tuple(shape), name, tuple(ref.shape)))
    return data


def sign_gate(own, site, ge=False):
    """``own``: the tensor whose sign picks the branch (x > 0; ``ge``: x >= 0).  Returns the tensor to hand to the kernel as its
    gate reference: ``own`` itself, or (replay) +-1 from the recorded decisions."""
    if MODE["value"] is None:
        return own
    mask = (own >= 0) if ge else (own > 0)
    if MODE["value"] == "record":
        _TAPE.append((site, mask))
        return own
    rec = _next(site, own.shape)
    _FLIPS.append((site, own.numel(), int((rec != mask).sum())))
    one = torch.ones((), device=own.device, dtype=torch.float32)
    return torch.where(rec, one, -one).contiguous()


def values(site, *tensors):
    """Tensors that only steer a routing decision (the global max pool's arg-max: (x, max); the L1 loss: prediction - target
    sign): recorded as they are, replayed in place of the run's own."""
    if MODE["value"] is None:
        return tensors
    if MODE["value"] == "record":
        _TAPE.append((site, tuple(t.detach().clone() for t in tensors)))
        return tensors
    rec = _next(site, tensors[0].shape)
    return rec
