"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the HIP kernels' counter-hash dropout mask
(pixelrec_amd/csrc/pxr_common.h: pxr_fmix32 / pxr_hash32 / pxr_drop_threshold / pxr_keep), so that tests can
inject the SAME keep-masks into the CPU oracle and check training-mode (dropout on) parity exactly.
The reference's nn.Dropout draws from ATen's Philox stream, which no other implementation can reproduce
(SURVEY.md §7 hard part 5); equivalence with it is distributional (keep-rate test in tests/)."""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _fmix32(h):
    h = h.astype(np.uint64)
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


def hash32(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    idx = np.asarray(idx, dtype=np.uint64)
    lo, hi = idx & _M32, idx >> np.uint64(32)
    h = _fmix32(lo ^ np.uint64(seed & 0xFFFFFFFF))
    h = (h + hi * np.uint64(0x9E3779B1) + np.uint64((stream * 0x85EBCA77) & 0xFFFFFFFF)
         + np.uint64((seed >> 32) & 0xFFFFFFFF)) & _M32
    return _fmix32(h).astype(np.uint32)


def drop_threshold(p: float) -> int:
    t = float(np.float32(p)) * 4294967296.0
    if t <= 0.0:
        return 0
    if t >= 4294967295.0:
        return 4294967295
    return int(t)


def keep_mask(seed: int, stream: int, shape, p: float) -> np.ndarray:
    """Boolean keep mask for a tensor of `shape` whose flat element index is the hash counter."""
    n = int(np.prod(shape))
    return (hash32(seed, stream, np.arange(n, dtype=np.uint64)) >= np.uint32(drop_threshold(p))).reshape(shape)


def sasrec_masks(seed: int, B: int, L: int, D: int, H: int, n_layers: int, p_hidden: float, p_attn: float):
    """The masks of one training step in the layout oracle.sasrec_oracle expects (stream ids as in
    pixelrec_amd/model/sasrec.py: 0 input; 1+3i attention probs; 2+3i attention output; 3+3i FFN output)."""
    import torch

    t = lambda a: torch.from_numpy(a)
    drop = {"input": t(keep_mask(seed, 0, (B, L, D), p_hidden))}
    for i in range(n_layers):
        drop[(i, "attn")] = t(keep_mask(seed, 1 + 3 * i, (B, H, L, L), p_attn))
        drop[(i, "attn_out")] = t(keep_mask(seed, 2 + 3 * i, (B, L, D), p_hidden))
        drop[(i, "ffn_out")] = t(keep_mask(seed, 3 + 3 * i, (B, L, D), p_hidden))
    return drop
