"""A 64-bit linear congruential generator shared by make_goldens.py (reference side) and the tests (our side), so that
weights and inputs of the network fixtures need not be stored: only the reference's OUTPUT is.  Pure integer arithmetic
(Knuth's MMIX constants) -> identical values on every machine and numpy version."""
import numpy as np

_A, _C, _M = 6364136223846793005, 1442695040888963407, 1 << 64


def _lcg_uniform_loop(n: int, seed: int) -> np.ndarray:
    """The definition: one state per value (kept as the check of the vectorised form below)."""
    out = np.empty(n, dtype=np.float32)
    s = (seed * 2654435761 + 12345) % _M
    for i in range(n):
        s = (_A * s + _C) % _M
        out[i] = np.float32(((s >> 40) - (1 << 23)) / float(1 << 23))
    return out


def lcg_uniform(n: int, seed: int) -> np.ndarray:
    """n float32 values in [-1, 1): the top 24 bits of successive LCG states.  The same values as `_lcg_uniform_loop`, from
    block jumps: the first B states one by one, every later block = A^B * (previous block) + C * (A^B - 1) / (A - 1) in
    uint64 arithmetic, which wraps modulo 2^64 exactly like the definition (the neck's 77 M parameters take seconds, not hours)."""
    B = 65536
    states = np.empty(max(n, 1), dtype=np.uint64)
    s = (seed * 2654435761 + 12345) % _M
    ab, cb = 1, 0                         # after k steps: state -> ab * state + cb
    for i in range(min(B, n)):
        s = (_A * s + _C) % _M
        states[i] = s
        ab, cb = (_A * ab) % _M, (_A * cb + _C) % _M
    if n > B:
        ab_, cb_ = np.uint64(ab), np.uint64(cb)
        with np.errstate(over="ignore"):
            for lo in range(B, n, B):
                m = min(B, n - lo)
                states[lo:lo + m] = states[lo - B:lo - B + m] * ab_ + cb_
    top = (states[:n] >> np.uint64(40)).astype(np.int64) - (1 << 23)
    return (top.astype(np.float64) / float(1 << 23)).astype(np.float32)


def lcg_fill_state(module, seed: int) -> None:
    """Deterministic parameters / BatchNorm statistics for a torch module, visited in sorted key order:
    convolution weights ~ U(-1,1)/sqrt(fan_in), BN weight in [0.75,1.25], BN bias / running_mean in [-0.1,0.1],
    running_var in [0.5,1.5]."""
    import torch
    sd = module.state_dict()
    for k, key in enumerate(sorted(sd)):
        t = sd[key]
        if key.endswith("num_batches_tracked"):
            continue
        u = torch.from_numpy(lcg_uniform(t.numel(), seed * 1000 + k)).reshape(t.shape)
        if t.dim() > 1:
            v = u / float(t[0].numel()) ** 0.5
        elif key.endswith("running_var"):
            v = 1.0 + 0.5 * u
        elif key.endswith("running_mean") or key.endswith("bias"):
            v = 0.1 * u
        else:                      # BatchNorm weight
            v = 1.0 + 0.25 * u
        t.copy_(v)
