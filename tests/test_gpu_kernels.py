"""Kernel-level parity: each HIP kernel (through the C ABI) vs numpy on the same seeded inputs."""
import ctypes as C
import math

import numpy as np
import pytest

from tests.helpers import cosine_flat, from_bf16_bits, rel_err, to_bf16_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import coot_videotext_amd as cva
    assert torch.cuda.is_available()
    lib = cva.lib.load()
    return torch, cva, lib


def _dev_bf16(torch, x):
    return torch.from_numpy(to_bf16_bits(x).view(np.int16)).cuda()


def _sp(torch):
    return torch.cuda.current_stream().cuda_stream


def test_probe_tr16(env):
    """ds_read_b64_tr_b16 semantics the TN GEMM relies on: with lane p of each 16-lane group pointing at
    row (4g + p/4), cols (p%4)*4.., lane i receives column i of the 4-row block: out[l][j] = tile[4g+j][l%16]."""
    torch, cva, lib = env
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    cva.lib.check(lib.coot_probe_tr16(out.data_ptr(), _sp(torch)))
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.int64).reshape(64, 4)
    exp = np.zeros((64, 4), dtype=np.int64)
    for l in range(64):
        for j in range(4):
            exp[l, j] = (4 * (l >> 4) + j) * 64 + (l & 15)
    if not (got == exp).all():
        print("tr16 mapping (lane: 4 values as (row, col)):")
        for l in range(64):
            print(l, [(int(v) // 64, int(v) % 64) for v in got[l]])
    assert (got == exp).all()


@pytest.mark.parametrize("M,N,K,act,use_res,out_f32", [
    (128, 128, 64, 0, False, True), (300, 200, 72, 1, True, False), (1000, 384, 384, 0, True, False),
    (4096, 1152, 384, 0, False, False), (77, 768, 384, 1, False, True), (2560, 384, 2048, 1, False, False)])
def test_gemm_nt(env, M, N, K, act, use_res, out_f32):
    torch, cva, lib = env
    rs = np.random.RandomState(M + N + K)
    X = rs.randn(M, K).astype(np.float32)
    W = (rs.randn(N, K) / math.sqrt(K)).astype(np.float32)
    bias = rs.randn(N).astype(np.float32)
    R = rs.randn(M, N).astype(np.float32)
    Xq, Wq, Rq = from_bf16_bits(to_bf16_bits(X)), from_bf16_bits(to_bf16_bits(W)), from_bf16_bits(to_bf16_bits(R))
    ref = Xq.astype(np.float64) @ Wq.astype(np.float64).T + bias
    if act == 1:
        from oracle.coot_oracle import gelu
        ref = gelu(ref)
    if use_res:
        ref = ref + Rq
    dX, dW, dR = _dev_bf16(torch, X), _dev_bf16(torch, W), _dev_bf16(torch, R)
    dbias = torch.from_numpy(bias).cuda()
    out = torch.zeros(M, N, dtype=torch.float32 if out_f32 else torch.int16, device="cuda")
    cva.lib.check(lib.coot_gemm_nt(dX.data_ptr(), K, dW.data_ptr(), K, M, N, K, dbias.data_ptr(), act,
                                   dR.data_ptr() if use_res else None, N, out.data_ptr(), N, int(out_f32), _sp(torch)), "gemm_nt")
    torch.cuda.synchronize()
    got = out.cpu().numpy() if out_f32 else from_bf16_bits(out.cpu().numpy().view(np.uint16))
    err = rel_err(got, ref)
    print(f"gemm_nt {M}x{N}x{K} act={act} rel_err={err:.2e}")
    assert err < (2e-5 if out_f32 else 6e-3), err


@pytest.mark.parametrize("mode,use_ws", [(0, True), (0, False), (1, True)])
@pytest.mark.parametrize("T,Mo,No", [(64, 128, 128), (1000, 384, 384), (5000, 384, 2048), (333, 200, 72)])
def test_gemm_tn(env, mode, use_ws, T, Mo, No):
    torch, cva, lib = env
    rs = np.random.RandomState(T + Mo + No)
    A = rs.randn(T, Mo).astype(np.float32)
    B = rs.randn(T, No).astype(np.float32)
    Aq, Bq = from_bf16_bits(to_bf16_bits(A)), from_bf16_bits(to_bf16_bits(B))
    ref = Aq.astype(np.float64).T @ Bq.astype(np.float64)
    C0 = rs.randn(Mo, No).astype(np.float32)
    dA, dB = _dev_bf16(torch, A), _dev_bf16(torch, B)
    dC = torch.from_numpy(C0.copy()).cuda()
    cva.lib.check(lib.coot_set_option(b"tn_mode", mode))
    try:
        ws = torch.empty(lib.coot_gemm_tn_workspace_bytes(T, Mo, No) if use_ws else 0, dtype=torch.uint8, device="cuda")
        cva.lib.check(lib.coot_gemm_tn(dA.data_ptr(), Mo, dB.data_ptr(), No, T, Mo, No, dC.data_ptr(), No,
                                       ws.data_ptr() if use_ws else None, ws.numel(), _sp(torch)), "gemm_tn")
        torch.cuda.synchronize()
    finally:
        lib.coot_set_option(b"tn_mode", 0)
    got = dC.cpu().numpy() - C0
    err = rel_err(got, ref)
    print(f"gemm_tn mode={mode} ws={use_ws} T={T} {Mo}x{No} rel_err={err:.2e}")
    assert err < 2e-5, err


@pytest.mark.parametrize("dma,xcd", [(0, 5), (0, 7), (1, 5), (1, 7)])
@pytest.mark.parametrize("T", [6000, 1237, 90])
def test_gemm_tn_batch(env, dma, xcd, T):
    """The batched weight-gradient launch of a backward pass (coot_gemm_tn_batch): wide 384-row tiles register-staged or fed by
    LDS-DMA, linear or XCD-grouped workgroup order, against float64 products of the bf16-rounded operands.  Problems: a wide
    one written (overwrite), a grouped one with 192-column groups (half-empty column tile), a grouped one sharing A, a 3-slab
    one accumulated onto C, a narrow one (128 x 128 tiles in the same flush), one with a bias-gradient column sum."""
    torch, cva, lib = env
    from coot_videotext_amd.lib import TnProblem
    rs = np.random.RandomState(T + dma)
    keep, probs, checks = [], [], []

    def add(Aw, Bw, Mo, No, groups=1, zA=0, zB=0, overwrite=0, colsum=False):
        A = rs.randn(T, Aw).astype(np.float32)
        B = rs.randn(T, Bw).astype(np.float32)
        Aq, Bq = from_bf16_bits(to_bf16_bits(A)).astype(np.float64), from_bf16_bits(to_bf16_bits(B)).astype(np.float64)
        C0 = rs.randn(groups, Mo, No).astype(np.float32)
        dA, dB, dC = _dev_bf16(torch, A), _dev_bf16(torch, B), torch.from_numpy(C0.copy()).cuda()
        dcs = torch.zeros(Mo, device="cuda") if colsum else None
        keep.extend([dA, dB, dC, dcs])
        probs.append(TnProblem(dA.data_ptr(), Aw, dB.data_ptr(), Bw, T, Mo, No, dC.data_ptr(), No, dcs.data_ptr() if colsum else None,
                               overwrite, groups, zA, zB, Mo * No))
        ref = np.stack([Aq[:, z * zA:z * zA + Mo].T @ Bq[:, z * zB:z * zB + No] for z in range(groups)])
        checks.append((dC, C0 * (0 if overwrite else 1), ref, dcs, Aq[:, :Mo].sum(0)))

    add(384, 640, 384, 640, overwrite=1)
    add(768, 384, 384, 192, groups=2, zA=384, zB=192)
    add(384, 768, 384, 384, groups=2, zA=0, zB=384)
    add(1152, 384, 1152, 384)
    add(200, 72, 200, 72)
    add(384, 384, 384, 384, colsum=True)
    add(1152, 384, 1152, 384, colsum=True)
    arr = (TnProblem * len(probs))(*probs)
    ws = torch.empty(8 * sum(p.Mo * p.No * p.groups for p in probs), device="cuda")
    dma0, xcd0 = cva.lib.get_option("tn_dma"), cva.lib.get_option("xcd_order")
    cva.lib.check(lib.coot_set_option(b"tn_dma", dma))
    cva.lib.check(lib.coot_set_option(b"xcd_order", xcd))
    try:
        cva.lib.check(lib.coot_gemm_tn_batch(arr, len(probs), ws.data_ptr(), ws.numel() * 4, None, _sp(torch)), "gemm_tn_batch")
        torch.cuda.synchronize()
    finally:
        lib.coot_set_option(b"tn_dma", dma0)
        lib.coot_set_option(b"xcd_order", xcd0)
    for i, (dC, C0, ref, dcs, csref) in enumerate(checks):
        err = rel_err(dC.cpu().numpy() - C0, ref)
        print(f"gemm_tn_batch dma={dma} xcd={xcd} T={T} problem {i} rel_err={err:.2e}")
        assert err < 2e-5, (i, err)
        if dcs is not None:
            assert rel_err(dcs.cpu().numpy(), csref) < 2e-5


@pytest.mark.parametrize("R,D", [(37, 384), (100, 2048), (9, 1536), (5, 4096), (64, 64), (7, 2816)])
def test_ln_fwd(env, R, D):
    torch, cva, lib = env
    from oracle.coot_oracle import ln_coot
    rs = np.random.RandomState(R + D)
    x = (rs.randn(R, D) * 2 + 0.5).astype(np.float32)
    x[0] = 0  # zero padding row: output == bias
    g, b = (1 + 0.1 * rs.randn(D)).astype(np.float32), (0.1 * rs.randn(D)).astype(np.float32)
    ref = ln_coot(x.astype(np.float64), g, b)
    y = torch.zeros(R, D, dtype=torch.float32, device="cuda")
    dx, dg, db = torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda(), torch.from_numpy(b).cuda()  # keep alive
    cva.lib.check(lib.coot_ln_fwd(dx.data_ptr(), R, D, dg.data_ptr(), db.data_ptr(), None, y.data_ptr(), _sp(torch)), "ln_fwd")
    torch.cuda.synchronize()
    err = rel_err(y.cpu().numpy(), ref)
    assert err < 1e-5, err


@pytest.mark.parametrize("Nseq,L,H,dh", [(3, 7, 4, 16), (5, 80, 8, 48), (2, 150, 8, 48), (4, 64, 2, 64), (3, 33, 2, 32)])
def test_attn_fwd(env, Nseq, L, H, dh):
    torch, cva, lib = env
    rs = np.random.RandomState(Nseq * 1000 + L)
    D = H * dh
    qkv = rs.randn(Nseq, L, 3 * D).astype(np.float32)
    lens = rs.randint(1, L + 1, size=Nseq).astype(np.int64)
    lens[0] = L
    q = from_bf16_bits(to_bf16_bits(qkv)).astype(np.float64)
    Q, K, V = [q[..., i * D:(i + 1) * D].reshape(Nseq, L, H, dh).transpose(0, 2, 1, 3) for i in range(3)]
    S = Q @ K.transpose(0, 1, 3, 2) / math.sqrt(dh)
    valid = np.arange(L)[None, :] < lens[:, None]
    S = np.where(valid[:, None, None, :], S, -32752.0)
    A = np.exp(S - S.max(-1, keepdims=True))
    A /= A.sum(-1, keepdims=True)
    ref = (A @ V).transpose(0, 2, 1, 3).reshape(Nseq, L, D)
    lse_ref = (np.log(np.exp(S - S.max(-1, keepdims=True)).sum(-1)) + S.max(-1)).transpose(0, 2, 1)
    dq = _dev_bf16(torch, qkv.reshape(Nseq * L, 3 * D))
    out = torch.zeros(Nseq * L, D, dtype=torch.int16, device="cuda")
    lse = torch.zeros(Nseq * L, H, dtype=torch.float32, device="cuda")
    dlens = torch.from_numpy(lens).cuda()
    cva.lib.check(lib.coot_attn_fwd(dq.data_ptr(), Nseq, L, H, dh, dlens.data_ptr(), out.data_ptr(),
                                    lse.data_ptr(), _sp(torch)), "attn_fwd")
    torch.cuda.synchronize()
    got = from_bf16_bits(out.cpu().numpy().view(np.uint16)).reshape(Nseq, L, D)
    err = rel_err(got, ref)
    lerr = np.abs(lse.cpu().numpy().reshape(Nseq, L, H) - lse_ref).max()
    print(f"attn_fwd N={Nseq} L={L} H={H} dh={dh} rel_err={err:.2e} lse_err={lerr:.2e}")
    assert err < 1.5e-2 and cosine_flat(got, ref) > 0.9999, err
    assert lerr < 1e-3
