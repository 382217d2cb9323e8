"""
TEST INFRASTRUCTURE (checker only; nothing under coot-videotext_amd/ imports this).

numpy restatement of the HIP library's counter-based dropout generator
(coot-videotext_amd/csrc/common.h: mix32, drop_key, drop_hash, drop_scales_key; attention.hip: attn_drop;
api.hip: mkdrop and the SITE_* table) and of the element index every dropout site of a network call uses.

Why it exists: the reference draws its masks from torch's Philox stream (nn.Dropout), the library from a hash of
(seed, site, element) — so train-mode parity is only checkable with the SAME masks on both sides.  oracle/gen_golden.py
replaces every nn.Dropout of the unmodified reference by a module that multiplies with the mask this file computes for that
site (nntrainer/models/transformer_legacy.py:418,435,487,553,592-598, nntrainer/models/poolers.py:139-143,171-196), and the
GPU tests run the library with the same seed (tests/test_gpu_train_parity.py).

Layouts (one network call = up to two SEGMENTS of sequences, api.hip: Segs):
  * token rows: padded   row = sum_{s' < s} N_s' L_s' + n L_s + l
                packed   row = cu[n_global] + l   (valid tokens only; padded positions take no mask: value 1)
  * element-wise sites (post-LN, FF1, FF2, pool FC1, pool FC2): idx = row * ld + col
  * attention probabilities: hash input (((n H + h) Lq + q) * ((Lk + 1) >> 1) + (k >> 1)) mod 2^32, n segment-local,
    the second segment under seed + 0x9E3779B97F4A7C15; packed rows: Lq = Lk = the sequence's own length
  * pooling weights (dropout3): idx = (n L + l) D + c segment-local (packed: the global row), segment s under seed + 977 s
"""
import numpy as np

U32 = np.uint64(0xFFFFFFFF)
SITE_ATTN, SITE_POSTLN, SITE_FF1, SITE_FF2, SITE_POOL1, SITE_POOL2, SITE_POOL3 = 1, 2, 3, 4, 5, 6, 7
SITE_BASE_CTX = 16 * 8      # api.hip: context layers use site_base 16 (8 + i)
SITE_BASE_POOL = 16 * 15
SEG2_DELTA = 0x9E3779B97F4A7C15
M64 = (1 << 64) - 1


def _u32(x):
    return np.asarray(x, dtype=np.uint64) & U32


def mix32(x):
    x = _u32(x)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & U32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & U32
    x ^= x >> np.uint64(16)
    return x


def drop_key(seed: int, site: int) -> np.uint64:
    """common.h: drop_key(seed, site)."""
    seed &= M64
    lo, hi = seed & 0xFFFFFFFF, seed >> 32
    inner = mix32((hi + site * 0x9E3779B9) & 0xFFFFFFFF)
    return mix32(np.uint64(lo) ^ inner)


def _umul24(a, c):
    return ((a & np.uint64(0xFFFFFF)) * np.uint64(c & 0xFFFFFF)) & U32


def drop_hash(t):
    """common.h: drop_hash — the per-pair mixer on v_mad_u32_u24."""
    t = _u32(t)
    t = (_umul24(t, 0xD2B54B) + (t >> np.uint64(16))) & U32
    t ^= t >> np.uint64(13)
    t = (_umul24(t, 0x95A53D) + (t >> np.uint64(11))) & U32
    t ^= t >> np.uint64(16)
    return t


def quantise_p(p: float):
    """api.hip: mkdrop — returns (16-bit threshold, 1 / keep-probability as the fp32 the kernels multiply with)."""
    t = float(np.float32(p)) * 4294967296.0
    thr = 4294967295 if t >= 4294967295.0 else int(t)
    thr = max(thr, 65536)
    t16 = thr >> 16
    return t16, np.float32(1.0 / (1.0 - t16 / 65536.0))


def scales_from_index(key, idx, p):
    """keep-scale (0 or 1/keep) of the elements with 64-bit indices idx (common.h: drop_scale_key)."""
    t16, inv_keep = quantise_p(p)
    idx = np.asarray(idx, dtype=np.uint64)
    h = drop_hash(((idx >> np.uint64(1)) & U32) ^ np.uint64(key))
    u = np.where(idx & np.uint64(1), h >> np.uint64(16), h & np.uint64(0xFFFF))
    return np.where(u >= np.uint64(t16), inv_keep, np.float32(0)).astype(np.float32)


def scales_attn(key, row32, k, lk_half, p):
    """attention.hip: attn_drop — row32 = (n H + h) Lq + q (32-bit), k = key index."""
    t16, inv_keep = quantise_p(p)
    t = (np.asarray(row32, np.uint64) * np.asarray(lk_half, np.uint64) + (np.asarray(k, np.uint64) >> np.uint64(1))) & U32
    h = drop_hash(t ^ np.uint64(key))
    u = np.where(np.asarray(k, np.uint64) & np.uint64(1), h >> np.uint64(16), h & np.uint64(0xFFFF))
    return np.where(u >= np.uint64(t16), inv_keep, np.float32(0)).astype(np.float32)


class CallLayout:
    """Where the sequences of one reference module call sit in the library's token matrix.

    seg        0 / 1: first / second segment of the network call (e.g. videos, then clips through the same local network)
    N, L       sequences and padded length of THIS segment
    row0       first token row of the segment in the padded layout (= N_0 L_0 for the second segment)
    lens       valid lengths [N] (only needed for packed rows)
    cu         None (padded layout) or the row start of every sequence of this segment in the packed token matrix [N]
    """

    def __init__(self, seg, N, L, row0=0, lens=None, cu=None):
        self.seg, self.N, self.L, self.row0 = seg, N, L, row0
        self.lens = None if lens is None else np.asarray(lens, np.int64)
        self.cu = None if cu is None else np.asarray(cu, np.int64)

    def rows(self):
        """[N, L] global token row of (n, l) and validity (packed rows: positions >= len have no row)."""
        n, l = np.arange(self.N)[:, None], np.arange(self.L)[None, :]
        if self.cu is None:
            return self.row0 + n * self.L + l, np.ones((self.N, self.L), bool)
        return self.cu[:, None] + l, l < self.lens[:, None]


def mask_rows(seed, site, lay: CallLayout, ld, p, cols=None):
    """[N, L, ld] scales of an element-wise site on token rows (post-LN, FF1, FF2; also the query rows of a context layer
    with L = 1)."""
    rows, valid = lay.rows()
    cols = np.arange(ld) if cols is None else cols
    idx = rows[:, :, None].astype(np.uint64) * np.uint64(ld) + cols[None, None, :].astype(np.uint64)
    m = scales_from_index(drop_key(seed, site), idx, p)
    return np.where(valid[:, :, None], m, np.float32(1))


def mask_heads(seed, site, lay: CallLayout, heads, dh, p, pool3=False):
    """[N, heads, L, dh] scales of the GenPool sites (poolers.py:171-196: tensors are [batch, heads, seq, d]); the library's
    column is h * dh + e of a row of width heads * dh.  pool3: the softmax-weight dropout, indexed segment-locally under
    seed + 977 * segment (pool.hip)."""
    ld = heads * dh
    if pool3:
        seed = (seed + 977 * lay.seg) & M64
        if lay.cu is None:
            lay = CallLayout(lay.seg, lay.N, lay.L, 0)
    m = mask_rows(seed, site, lay, ld, p)                 # [N, L, heads * dh]
    return np.ascontiguousarray(m.reshape(lay.N, lay.L, heads, dh).transpose(0, 2, 1, 3))


def mask_attention(seed, site, lay: CallLayout, H, Lq, Lk, p, n0=0):
    """[N, H, Lq, Lk] scales of the attention probabilities.  Self-attention: Lq = Lk = lay.L; context block: Lq = 1.
    n0: index of the call's first sequence within its launch (0: sequences are numbered per segment)."""
    seed = (seed + (SEG2_DELTA if lay.seg else 0)) & M64
    key = drop_key(seed, site)
    n = (n0 + np.arange(lay.N))[:, None, None, None].astype(np.uint64)
    h = np.arange(H)[None, :, None, None].astype(np.uint64)
    q = np.arange(Lq)[None, None, :, None].astype(np.uint64)
    k = np.arange(Lk)[None, None, None, :].astype(np.uint64)
    if lay.cu is None:
        row32 = ((n * np.uint64(H) + h) * np.uint64(Lq) + q) & U32
        return scales_attn(key, row32, np.broadcast_to(k, (lay.N, H, Lq, Lk)), (Lk + 1) >> 1, p)
    ln = lay.lens[:, None, None, None].astype(np.uint64)  # packed rows: the sequence's own length is the kernel's L
    row32 = ((n * np.uint64(H) + h) * ln + q) & U32
    m = scales_attn(key, row32, np.broadcast_to(k, (lay.N, H, Lq, Lk)), (ln + np.uint64(1)) >> np.uint64(1), p)
    valid = (q < ln) & (k < ln)
    return np.where(valid, m, np.float32(1))


def step_net_seeds(step_seed: int):
    """api_step.hip (coot_train_step / coot_step_forward): the per-network seeds a step derives from its seed argument —
    video local, video global, text local, text global."""
    return [(step_seed + 0) & M64, (step_seed + 11) & M64, (step_seed + 1000 + 22) & M64, (step_seed + 1000 + 33) & M64]
