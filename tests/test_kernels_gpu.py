"""HIP kernels (through the C ABI) vs plain PyTorch fp32 CPU references of the same op."""
import math

import numpy as np

import pytest
import torch
from torch.nn import functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


CONV_CASES = [
    # B, H, W, Cin, Cout, k, s, p
    (2, 9, 11, 8, 70, 3, 1, 1),
    (1, 16, 20, 64, 64, 1, 1, 0),
    (2, 15, 20, 32, 130, 3, 2, 1),
    (1, 23, 31, 4, 64, 7, 2, 3),
    (2, 7, 5, 3, 17, 3, 1, 1),        # scalar K path (Cin % 4 != 0)
    (1, 15, 20, 300, 128, 3, 1, 1),   # Cin = 300 (corr volume)
    (3, 30, 40, 256, 256, 3, 1, 1),   # 128x128 tile path
    (1, 60, 80, 128, 512, 1, 1, 0),
    (2, 12, 16, 512, 128, 1, 2, 0),   # strided 1x1 shortcut
    # >= 192 tiles of 128x128 and Cin % 64 == 0: the LDS-DMA (global_load_lds) kernel in bf16
    (4, 60, 80, 128, 256, 3, 1, 1),
    (5, 61, 79, 64, 256, 1, 1, 0),    # M tail (M % 128 != 0), K = 64 (single K-tile)
    (6, 64, 80, 256, 64, 1, 1, 0),    # BN = 64 variant
    (4, 62, 82, 128, 192, 3, 2, 1),   # stride 2 + N tail inside a 128-wide tile
    (8, 60, 80, 64, 64, 3, 1, 1),     # BN = 64, 3x3
    (3, 15, 20, 512, 128, 3, 1, 1),   # few tiles, K = 4608: 64x64 tiles with two tiles of prefetch (the DMA kernel with 128x64 tiles when forced: see the channel-major test), M tail
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv2d(device, case, dtype):
    from nopesac_amd import ops
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 10000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    scale = 1 + 0.1 * torch.randn(Cout, generator=g)
    bias = 0.1 * torch.randn(Cout, generator=g)
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(x, w, None, s, p) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    res = torch.randn(ref.shape, generator=g)
    if dtype == torch.bfloat16:
        res = res.bfloat16().float()
    ref = F.leaky_relu(ref + res, 0.01)
    y = ops.conv2d(_nhwc(x).to(device, dtype), w.permute(0, 2, 3, 1).contiguous().to(device, dtype), scale.to(device),
                   bias.to(device), _nhwc(res).to(device, dtype), stride=s, pad=p, act=ops.ACT_LEAKY)
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    assert y.dtype == dtype
    assert _rel(y.float().permute(0, 3, 1, 2), ref) < tol


def test_conv2d_bf16_in_f32_out_and_slices(device):
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 6, 7, 48, generator=g).bfloat16()
    w = (torch.randn(40, 1, 1, 16, generator=g) / 4).bfloat16()
    xs = x.to(device)[..., 16:32]                       # channel slice of a wider buffer
    out = torch.zeros(2, 6, 7, 100, device=device)
    y = ops.conv2d(xs, w.to(device), out=out[..., 50:90], out_dtype=torch.float32, act=ops.ACT_RELU)
    ref = F.relu(torch.einsum("bhwc,nc->bhwn", x[..., 16:32].float(), w.view(40, 16).float()))
    assert _rel(out[..., 50:90], ref) < 1e-5
    assert float(out[..., :50].abs().max()) == 0 and float(out[..., 90:].abs().max()) == 0


def test_conv2d_batched_weights(device):
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(6)
    x = torch.randn(3, 10, 12, 256, generator=g)
    w = torch.randn(3, 50, 1, 1, 256, generator=g) / 16
    y = ops.conv2d(x.to(device), w.to(device), batched_weights=True, act=ops.ACT_SIGMOID)
    ref = torch.sigmoid(torch.einsum("bhwc,bnc->bhwn", x, w.view(3, 50, 256)))
    assert _rel(y, ref) < 1e-5


@pytest.mark.parametrize("case", [(3, 31, 29, 128, 128, 3, 2, 1),     # stride 2, odd sizes, M tail, tiles that straddle images
                                  (2, 8, 10, 128, 64, 1, 1, 0),       # one K-tile (prologue + tail only)
                                  (2, 8, 10, 256, 70, 1, 1, 0),       # two K-tiles (no steady-state iteration), N tail
                                  (1, 8, 10, 384, 64, 1, 1, 0),       # three (one iteration, clamped second request)
                                  (1, 15, 20, 256, 128, 3, 1, 1),     # 18 K-tiles, one pair's res5-sized 3x3
                                  (2, 12, 16, 512, 128, 1, 2, 0),     # strided 1x1
                                  (2, 15, 20, 320, 128, 3, 1, 1),     # K = 2880 = 22.5 tiles: zero-filled K tail (pose-net branch conv0)
                                  (1, 9, 7, 200, 64, 1, 1, 0)])       # K = 200: one full tile + a 72-wide tail
def test_conv2d_small_grid_two_tile_prefetch(device, case, monkeypatch):
    """conv_igemm.hip, bf16 / K % 128 == 0 / <= 2048 tiles of 64x64: two tiles of branch-free buffer loads in flight (PF2).  The
    input is a channel slice of a wider buffer (x_cstride > Cin) and the output goes into one."""
    from nopesac_amd import ops
    monkeypatch.setenv("NOPESAC_CONV_FORCE", "t64")               # read per call by the C side: the 64x64-tile generic kernel
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 10000)
    xw = torch.randn(B, H, W, Cin + 72, generator=g).bfloat16()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).bfloat16()
    scale, bias = 1 + 0.1 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g)
    x = xw[..., 40:40 + Cin]
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), None, s, p) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    ref = F.relu(ref).permute(0, 2, 3, 1)
    out = torch.full((B, ref.shape[1], ref.shape[2], Cout + 24), 7.0, device=device, dtype=torch.bfloat16)
    ops.conv2d(xw.to(device)[..., 40:40 + Cin], w.permute(0, 2, 3, 1).contiguous().to(device), scale.to(device), bias.to(device),
               stride=s, pad=p, act=ops.ACT_RELU, out=out[..., 8:8 + Cout])
    assert _rel(out[..., 8:8 + Cout].float(), ref) < 1e-2
    assert float((out[..., :8].float() - 7).abs().max()) == 0 and float((out[..., 8 + Cout:].float() - 7).abs().max()) == 0


def test_conv2d_batched_weights_bf16_two_tile_prefetch(device):
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(16)
    x = torch.randn(3, 10, 12, 256, generator=g).bfloat16()
    w = (torch.randn(3, 50, 1, 1, 256, generator=g) / 16).bfloat16()
    y = ops.conv2d(x.to(device), w.to(device), batched_weights=True, act=ops.ACT_SIGMOID)
    ref = torch.sigmoid(torch.einsum("bhwc,bnc->bhwn", x.float(), w.view(3, 50, 256).float()))
    assert _rel(y.float(), ref) < 1e-2


@pytest.mark.parametrize("K,N", [(3, 256), (8, 1024), (50, 128), (1280, 1024), (768, 256)])
def test_linear(device, K, N):
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(K * 7 + N)
    x = torch.randn(37, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    y = ops.linear(x.to(device), w.to(device), b.to(device), act=ops.ACT_RELU)
    assert _rel(y, F.relu(F.linear(x, w, b))) < 2e-5


def test_linear_concat_buffers(device):
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(9)
    buf = torch.randn(2, 50, 1280, generator=g)
    w = torch.randn(256, 1024, generator=g) / 32
    dbuf = buf.to(device)
    y = ops.linear(dbuf[..., :1024], w.to(device), out=dbuf[..., 1024:])
    ref = F.linear(buf[..., :1024], w)
    assert _rel(dbuf[..., 1024:], ref) < 2e-5 and _rel(dbuf[..., :1024], buf[..., :1024]) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pool_and_upsample(device, dtype):
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(10)
    x = torch.randn(2, 24, 13, 17, generator=g).to(dtype).float()
    xd = _nhwc(x).to(device, dtype)
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert _rel(ops.maxpool(xd, 3, 2, 1).float().permute(0, 3, 1, 2), F.max_pool2d(x, 3, 2, 1)) < tol
    assert _rel(ops.maxpool(xd, 2, 2, 0).float().permute(0, 3, 1, 2), F.max_pool2d(x, 2, 2)) < tol
    up = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    add = torch.randn(up.shape, generator=g).to(dtype).float()
    y = ops.upsample2x_bilinear(xd, _nhwc(add).to(device, dtype), act=ops.ACT_RELU)
    assert _rel(y.float().permute(0, 3, 1, 2), F.relu(up) + add) < max(tol, 2e-6)
    lat = torch.randn(2, 24, 26, 34, generator=g).to(dtype).float()
    y = ops.upsample2x_nearest_add(xd, _nhwc(lat).to(device, dtype))
    assert _rel(y.float().permute(0, 3, 1, 2), lat + F.interpolate(x, scale_factor=2, mode="nearest")) < tol


def test_preprocess(device):
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randint(0, 256, (2, 3, 20, 24), generator=g).float()
    mean, std = torch.tensor([123.675, 116.28, 103.53]), torch.tensor([58.395, 57.12, 57.375])
    y = ops.preprocess(x.to(device), mean.to(device), std.to(device), 4, torch.float32)
    ref = (x - mean.view(1, 3, 1, 1)) / std.view(1, 3, 1, 1)
    assert _rel(y[..., :3].permute(0, 3, 1, 2), ref) < 1e-6 and float(y[..., 3].abs().max()) == 0


def test_norms_and_softmax(device):
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 128, 15, 20, generator=g) * 3 + 1
    ga, be = torch.randn(128, generator=g), torch.randn(128, generator=g)
    y = ops.groupnorm(_nhwc(x).to(device), ga.to(device), be.to(device), 32, 1e-5, act=ops.ACT_RELU)
    assert _rel(y.permute(0, 3, 1, 2), F.relu(F.group_norm(x, 32, ga, be, 1e-5))) < 1e-5
    t = torch.randn(77, 256, generator=g) * 2
    r = torch.randn(77, 256, generator=g)
    pos = torch.randn(11, 256, generator=g)
    ga, be = torch.randn(256, generator=g), torch.randn(256, generator=g)
    y, y2 = ops.layernorm(t.to(device), ga.to(device), be.to(device), res=r.to(device), addend=pos.to(device))
    ref = F.layer_norm(t + r, (256,), ga, be, 1e-5)
    assert _rel(y, ref) < 1e-5
    assert _rel(y2, ref + pos.repeat(7, 1)) < 1e-5
    s = torch.randn(45, 300, generator=g) * 4
    assert _rel(ops.softmax_rows(s.to(device)), F.softmax(s, -1)) < 1e-5
    assert _rel(ops.add_rows(t.to(device), pos.to(device)), t + pos.repeat(7, 1)) == 0
    q = torch.randn(9, 4, generator=g)
    ref = F.normalize(q, dim=-1)
    ref = torch.where(ref[:, :1] < 0, -ref, ref)
    assert _rel(ops.normalize_rows(q.to(device), True), ref) < 1e-6
    m = torch.randn(2, 15 * 20, 8, generator=g)
    ref = m.view(2, 15, 20, 8).transpose(1, 2).reshape(2, 300, 8)
    assert _rel(ops.transpose_hw_rows(m.to(device), 15, 20), ref) == 0


@pytest.mark.parametrize("mfma", [False, True])
@pytest.mark.parametrize("B,Lq,Lk", [(2, 300, 300), (3, 50, 300), (4, 50, 50), (2, 7, 5), (1, 130, 512)])
def test_attention(device, B, Lq, Lk, mfma):
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(B * 100 + Lq)
    qk = torch.randn(B * Lq, 512, generator=g)          # q in cols 0..255 of a wider buffer
    k = torch.randn(B * Lk, 256, generator=g)
    v = torch.randn(B * Lk, 256, generator=g)
    qlen = torch.tensor([Lq - (i % 3) for i in range(B)], dtype=torch.int32)
    klen = torch.tensor([Lk - (i % 2) * 2 for i in range(B)], dtype=torch.int32)
    dq = qk.to(device)
    o = ops.attention(dq[:, :256], k.to(device), v.to(device), B, Lq, Lk, 8, 32 ** -0.5, qlen.to(device), klen.to(device),
                      mfma_bf16=mfma)
    ref = torch.zeros(B * Lq, 256)
    for b in range(B):
        nq_, nk_ = int(qlen[b]), int(klen[b])
        qq = qk[b * Lq: b * Lq + nq_, :256].view(nq_, 8, 32)
        kk = k[b * Lk: b * Lk + nk_].view(nk_, 8, 32)
        vv = v[b * Lk: b * Lk + nk_].view(nk_, 8, 32)
        a = torch.softmax(torch.einsum("lhd,shd->hls", qq, kk) * 32 ** -0.5, -1)
        ref[b * Lq: b * Lq + nq_] = torch.einsum("hls,shd->lhd", a, vv).reshape(nq_, 256)
    assert _rel(o, ref) < (2e-2 if mfma else 1e-5)


@pytest.mark.parametrize("B,H,W", [(2, 480, 640), (1, 64, 96), (3, 70, 90)])
def test_stem_fused(device, B, H, W):
    """Fused bf16 stem (conv7x7/s2 + BN + ReLU + maxpool) vs the PyTorch fp32 ops on bf16-rounded operands."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, 3, H, W, generator=g).bfloat16().float()
    w = (torch.randn(64, 3, 7, 7, generator=g) / math.sqrt(147)).bfloat16().float()
    scale, bias = 1 + 0.1 * torch.randn(64, generator=g), 0.1 * torch.randn(64, generator=g)
    ref = F.max_pool2d(F.relu(F.conv2d(x, w, None, 2, 3) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)), 3, 2, 1)
    x4 = torch.zeros(B, H, W, 4)
    x4[..., :3] = x.permute(0, 2, 3, 1)
    w8 = torch.zeros(64, 7, 8, 4)
    w8[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    y = ops.stem_fused(x4.to(device, torch.bfloat16), w8.reshape(64, 224).to(device, torch.bfloat16), scale.to(device), bias.to(device))
    assert y.shape == (B, ref.shape[2], ref.shape[3], 64)
    assert _rel(y.float().permute(0, 3, 1, 2), ref) < 1e-2


@pytest.mark.parametrize("V,H,W,nq,seed", [(3, 48, 64, 50, 0), (2, 37, 53, 50, 1), (2, 480, 640, 128, 2), (1, 5, 3, 50, 3), (2, 120, 97, 6, 4)])
def test_rle_from_winner_map(device, V, H, W, nq, seed):
    """RLE kernels + device string compressor vs the numpy COCO restatement on blocky synthetic winner maps
    (ragged n_kept, a fallback view, sizes with H*W not a multiple of 16, ids up to 127); seed 4 = per-pixel noise: thousands of
    short runs per mask (many 256-run chunks per workgroup, negative run differences)."""
    from nopesac_amd import rle
    from oracle import rle_oracle as R
    g = torch.Generator().manual_seed(seed)
    bh, bw = max(H // 6, 1), max(W // 5, 1)
    coarse = torch.randint(0, nq, (V, (H + bh - 1) // bh, (W + bw - 1) // bw), generator=g)
    ids = coarse.repeat_interleave(bh, 1).repeat_interleave(bw, 2)[:, :H, :W]
    if seed == 4:
        ids = torch.randint(0, nq, (V, H, W), generator=g)
    passed = torch.rand(V, H, W, generator=g) < 0.8
    winner = (ids | (passed.long() << 7)).to(torch.uint8)
    n_kept = torch.tensor([min(nq, 1 + 7 * v) for v in range(V)], dtype=torch.int32)
    kept = torch.full((V, nq), -1, dtype=torch.int32)
    for v in range(V):
        present = torch.unique(ids[v])
        extra = torch.tensor([q for q in range(nq) if q not in set(present.tolist())], dtype=torch.long)
        pool = torch.cat([present, extra])[: int(n_kept[v])]
        kept[v, : int(n_kept[v])] = pool.sort().values.int()
    flags = torch.zeros(V, dtype=torch.int32)
    flags[V - 1] = 2                                                   # last view: fallback mask (no probability test)
    out = rle.encode_views(winner.to(device), kept.to(device), n_kept.to(device), flags.to(device))
    for v in range(V):
        assert len(out[v]) == int(n_kept[v])
        for p in range(int(n_kept[v])):
            m = ids[v] == int(kept[v, p])
            if not (int(flags[v]) & 2):
                m = m & passed[v]
            ref = R.encode(m.numpy())
            assert out[v][p]["segmentation"] == {"size": [H, W], "counts": ref["counts"]}, (v, p)
            assert out[v][p]["bbox"] == R.to_bbox(ref).tolist()
    # the enqueue-now / slice-later form (rle.PendingRLE, what forward_device + package use): identical strings and boxes whether the
    # bytes fit the pinned head copy, need the second copy (host_cap small) or overflow the device buffer (cap tiny -> encode_views)
    total = sum(len(r["segmentation"]["counts"]) for row in out for r in row)
    for cap, host_cap in ((64 << 20, 6 << 20), (64 << 20, max(total // 2, 1)), (max(total - 1, 1), 1 << 20)):
        pend = rle.PendingRLE(winner.to(device), kept.to(device), n_kept.to(device), flags.to(device), cap=cap, host_cap=host_cap)
        torch.cuda.synchronize()
        assert pend.finish(n_kept.tolist()) == out, (cap, host_cap)


@pytest.mark.parametrize("V,H,W,nq", [(3, 48, 64, 50), (2, 37, 53, 50), (2, 480, 640, 128), (1, 5, 3, 50)])
def test_decode_masks_all_views_in_one_launch(device, V, H, W, nq):
    """nopesac_decode_masks (dense `pred_plane_masks` of every kept plane of every view, one launch) against the per-view formula
    (siamese_planeTR.py:685, :743 in the fallback case): ragged n_kept, a fallback view, H*W not a multiple of 16, ids up to 127."""
    from nopesac_amd import ops
    from nopesac_amd.modeling.meta_arch import decode_masks
    g = torch.Generator().manual_seed(V * 100 + H)
    winner = torch.randint(0, 256, (V, H, W), generator=g).to(torch.uint8)
    winner = ((winner & 0x80) | (torch.randint(0, nq, (V, H, W), generator=g).to(torch.uint8))).to(device)
    n_kept = torch.tensor([min(nq, 1 + 9 * v) for v in range(V)], dtype=torch.int32)
    kept = torch.full((V, nq), -1, dtype=torch.int32)
    for v in range(V):
        kept[v, : int(n_kept[v])] = torch.randperm(nq, generator=g)[: int(n_kept[v])].sort().values.int()
    flags = torch.zeros(V, dtype=torch.int32)
    flags[V - 1] = 2
    total = int(n_kept.sum())
    m = ops.decode_masks(winner, kept.to(device), n_kept.to(device), flags.to(device), total)
    assert m.dtype == torch.bool and m.shape == (total, H, W)
    off = 0
    for v in range(V):
        n = int(n_kept[v])
        ref = decode_masks(winner[v], kept[v, :n].to(device), bool(int(flags[v]) & 2))
        assert torch.equal(m[off:off + n], ref), v
        off += n


@pytest.mark.parametrize("C,proj,CN,stride,M_odd", [
    (64, False, 64, 1, False), (64, False, 128, 1, True), (64, True, 64, 1, False), (64, False, 0, 1, False), (64, True, 0, 1, True),
    (128, False, 128, 1, True), (128, False, 256, 1, False), (128, True, 128, 2, True), (128, True, 0, 2, False), (128, False, 0, 1, False),
    (128, False, 128, 1, False), (128, True, 128, 2, False), (128, False, 256, 1, True), (64, True, 64, 1, True),     # full 128-pixel tiles -> rt4 / rt8 forms; ragged -> 32/64-pixel forms
    (256, False, 256, 1, True), (256, False, 512, 1, False), (256, True, 256, 2, True), (256, True, 0, 2, False), (256, False, 0, 1, True)])
def test_bottleneck_tail(device, C, proj, CN, stride, M_odd):
    """Fused conv3 + shortcut + ReLU (+ next conv1) vs the per-layer bf16 kernels: identity blocks bit-exact, projection
    blocks within bf16 rounding (the fused kernel does not round the shortcut to bf16 in between)."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(11 + CN + proj + C)
    B, H, W = (3, 9, 13) if M_odd else (2, 16, 24)          # 351 pixels: a ragged last tile
    C4, C2 = 4 * C, (64 if C == 64 else 2 * C)
    bf = lambda t: t.to(device, torch.bfloat16).contiguous()
    f = lambda t: t.to(device).contiguous()
    b = bf(torch.randn(B, H, W, C, generator=g))
    w3 = bf(torch.randn(C4, 1, 1, C, generator=g) / C ** 0.5)
    s3, b3 = f(1 + 0.1 * torch.randn(C4, generator=g)), f(0.1 * torch.randn(C4, generator=g))
    kw, sc = {}, None
    if proj:
        H2, W2 = (H - 1) * stride + 1 + (stride - 1), (W - 1) * stride + 1      # odd/even source sizes both occur
        x = bf(torch.randn(B, H2, W2, C2, generator=g))
        wsc = bf(torch.randn(C4, 1, 1, C2, generator=g) / C2 ** 0.5)
        ssc, bsc = f(1 + 0.1 * torch.randn(C4, generator=g)), f(0.1 * torch.randn(C4, generator=g))
        kw.update(x2=x, wsc=ops.mfma_fragment_major(wsc.view(C4, C2)), ssc=ssc, bsc=bsc, stride=stride)
        sc = ops.conv2d(x, wsc, ssc, bsc, stride=stride)
        assert sc.shape == (B, H, W, C4)
    else:
        sc = bf(torch.randn(B, H, W, C4, generator=g))
        kw.update(residual=sc)
    y_ref = ops.conv2d(b, w3, s3, b3, sc, act=ops.ACT_RELU)
    if CN:
        w1 = bf(torch.randn(CN, 1, 1, C4, generator=g) / C4 ** 0.5)
        s1, b1 = f(1 + 0.1 * torch.randn(CN, generator=g)), f(0.1 * torch.randn(CN, generator=g))
        kw.update(w1=ops.mfma_fragment_major(w1.view(CN, C4)), s1=s1, b1=b1)
    y, o = ops.bottleneck_tail(b, ops.mfma_fragment_major(w3.view(C4, C)), s3, b3, **kw)
    if proj:
        assert _rel(y.float(), y_ref.float()) < 1e-2
    else:
        assert torch.equal(y, y_ref)
    if CN:
        o_ref = ops.conv2d(y, w1, s1, b1, act=ops.ACT_RELU)      # from the fused y: isolates phase 2
        assert torch.equal(o, o_ref)
    else:
        assert o is None


def test_backbone_fused_tail_matches_unfused(device, sd50):
    """bf16 backbone with the fused bottleneck tails / stem vs the same model with per-layer kernels."""
    from tests.util import make_model
    model = make_model(device, dtype="bfloat16")
    x = torch.randn(2, 96, 128, 4, device=device).bfloat16()
    x[..., 3] = 0
    bb = model.backbone
    fused = bb(x)
    bb.fused_tail = False
    plain = bb(x)
    bb.fused_tail = True
    for k in plain:
        assert _rel(fused[k].float(), plain[k].float()) < 2e-2, k


def test_layernorm_ex_and_attention_bf16io(device):
    """bf16-output LayerNorm and bf16 I/O attention are the f32 kernels' results rounded once (same statistics)."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(300, 256, generator=g).to(device)
    gamma, beta = (1 + 0.1 * torch.randn(256, generator=g)).to(device), (0.1 * torch.randn(256, generator=g)).to(device)
    pos = torch.randn(100, 256, generator=g).to(device)
    y, y2 = ops.layernorm(x, gamma, beta, addend=pos)
    r = ops.layernorm_ex(x, gamma, beta, addend=pos, want=("y", "y16", "y2", "y2_16"))
    assert torch.equal(r["y"], y) and torch.equal(r["y2"], y2)
    assert torch.equal(r["y16"], y.bfloat16()) and torch.equal(r["y2_16"], y2.bfloat16())
    r = ops.layernorm_ex(x, gamma, beta, want=("y16",))
    assert set(r) == {"y16"} and torch.equal(r["y16"], y.bfloat16())
    B, Lq, Lk, H = 3, 50, 300, 8
    q = torch.randn(B * Lq, 256, generator=g).to(device).bfloat16()
    kv = torch.randn(B * Lk, 1536, generator=g).to(device).bfloat16()          # column slices of a wide buffer
    k, v = kv[:, 256:512], kv[:, 768:1024]
    o16 = ops.attention(q, k, v, B, Lq, Lk, H, 0.25, mfma_bf16=True)           # scale 2^-2: the q scaling is exact in bf16
    o32 = ops.attention(q.float(), k.float().contiguous(), v.float().contiguous(), B, Lq, Lk, H, 0.25, mfma_bf16=True)
    assert o16.dtype == torch.bfloat16 and torch.equal(o16, o32.bfloat16())
    # 300 query rows: the ten-wave build (one workgroup per (batch, head)), ragged lengths
    B, Lq, Lk = 3, 300, 300
    q = torch.randn(B * Lq, 256, generator=g).to(device).bfloat16()
    kv = torch.randn(B * Lk, 1536, generator=g).to(device).bfloat16()
    k, v = kv[:, 256:512], kv[:, 768:1024]
    ql, kl = torch.tensor([300, 257, 31], dtype=torch.int32, device=device), torch.tensor([300, 299, 33], dtype=torch.int32, device=device)
    o16 = ops.attention(q, k, v, B, Lq, Lk, H, 0.25, ql, kl, mfma_bf16=True)
    o32 = ops.attention(q.float(), k.float().contiguous(), v.float().contiguous(), B, Lq, Lk, H, 0.25, ql, kl, mfma_bf16=True)
    for b in range(B):
        n = int(ql[b])
        assert torch.equal(o16[b * Lq: b * Lq + n], o32[b * Lq: b * Lq + n].bfloat16()), b


@pytest.mark.parametrize("M", [19200 // 8, 333])
def test_encoder_tail(device, M):
    """Fused out-proj + LN1 + FFN + LN2 (csrc/enc_tail.hip) vs the same chain through the per-op bf16-mode kernels
    (differences: LayerNorm summation order, hence an occasional 1-ulp flip of a bf16 intermediate)."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(M)
    rn = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(device)
    attn, src, pos = rn(M, 256).bfloat16(), rn(M, 256), rn(300, 256)
    wo, w1, w2 = rn(256, 256, k=1 / 16).bfloat16(), rn(1024, 256, k=1 / 16).bfloat16(), rn(256, 1024, k=1 / 32).bfloat16()
    bo, b1, b2 = rn(256, k=0.1), rn(1024, k=0.1), rn(256, k=0.1)
    g1, be1, g2, be2 = 1 + rn(256, k=0.1), rn(256, k=0.1), 1 + rn(256, k=0.1), rn(256, k=0.1)
    W = {"wo": ops.mfma_fragment_major(wo), "w1": ops.mfma_fragment_major(w1), "w2": ops.mfma_fragment_major(w2),
         "bo": bo, "b1": b1, "b2": b2, "g1": g1, "be1": be1, "g2": g2, "be2": be2}
    out = ops.encoder_tail(attn, src, W, pos=pos)
    s = ops.linear(attn, wo, bo, residual=src, out_dtype=torch.float32)
    r = ops.layernorm_ex(s, g1, be1, want=("y", "y16"))
    h = ops.linear(r["y16"], w1, b1, act=ops.ACT_RELU, out_dtype=torch.bfloat16)
    s2 = ops.linear(h, w2, b2, residual=r["y"], out_dtype=torch.float32)
    ref = ops.layernorm_ex(s2, g2, be2, addend=pos, want=("y", "y16", "y2_16"))
    assert _rel(out["y"], ref["y"]) < 5e-3
    assert _rel(out["y16"].float(), ref["y16"].float()) < 1e-2 and _rel(out["ypos16"].float(), ref["y2_16"].float()) < 1e-2
    assert float((out["y"] - ref["y"]).abs().mean()) < 2e-4 * float(ref["y"].abs().mean() + 1)
    # ... and against an INDEPENDENT fp32 PyTorch reference of the same op (bf16 weights / attention input as given, everything
    # else in f32 on the CPU): the only differences left are the bf16 roundings of the LN1 output and of the FFN hidden tile
    a, sr = attn.float().cpu(), src.cpu()
    s_ = a @ wo.float().cpu().T + bo.cpu() + sr
    y1 = F.layer_norm(s_, (256,), g1.cpu(), be1.cpu(), 1e-5)
    h_ = F.relu(y1 @ w1.float().cpu().T + b1.cpu())
    y2 = F.layer_norm(h_ @ w2.float().cpu().T + b2.cpu() + y1, (256,), g2.cpu(), be2.cpu(), 1e-5)
    assert _rel(out["y"].cpu(), y2) < 1.5e-2, _rel(out["y"].cpu(), y2)
    assert float((out["y"].cpu() - y2).abs().mean()) < 2e-3 * float(y2.abs().mean() + 1)


@pytest.mark.parametrize("H,W,OH,OW", [(968, 1296, 480, 640), (480, 640, 480, 640), (37, 53, 48, 64), (1000, 700, 480, 640)])
def test_resize_bilinear_u8(device, H, W, OH, OW):
    """ScanNet input resize (cv2 INTER_LINEAR, 8-bit fixed point) vs the numpy restatement: bit-exact."""
    from nopesac_amd import ops
    from oracle.resize_oracle import resize_bilinear_u8
    rng = np.random.default_rng(H + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    out = ops.resize_bilinear_u8(torch.from_numpy(img).to(device), OH, OW).cpu().numpy()
    assert np.array_equal(out, resize_bilinear_u8(img, OH, OW))


def test_scannet_mapper_resizes_on_gpu(device, tmp_path):
    from PIL import Image
    from nopesac_amd import data
    from nopesac_amd.config import get_cfg
    from oracle.resize_oracle import resize_bilinear_u8
    from tests.util import ROOT
    import os
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "inference_scannet.yaml"))
    rng = np.random.default_rng(3)
    entry = {}
    arrs = []
    for v in "01":
        arr = rng.integers(0, 256, (968, 1296, 3), dtype=np.uint8)
        Image.fromarray(arr).save(tmp_path / f"{v}.png")
        arrs.append(arr)
        entry[v] = {"file_name": str(tmp_path / f"{v}.png"), "image_id": f"0-{v}"}
    out = data.PairMapper(cfg, "scannet_test")(entry)
    for v in "01":
        ref = resize_bilinear_u8(arrs[int(v)], 480, 640).transpose(2, 0, 1).astype("float32")
        assert out[v]["image"].shape == (3, 480, 640) and np.array_equal(out[v]["image"].numpy(), ref)


@pytest.mark.parametrize("case", [(2, 30, 40, 256, 256, 3, 1, 1), (3, 31, 29, 128, 128, 3, 2, 1), (2, 24, 32, 512, 256, 1, 1, 0),
                                  (1, 15, 20, 64, 384, 3, 1, 1), (2, 9, 7, 64, 128, 1, 1, 0)])
@pytest.mark.parametrize("nstage", [3, 32, 3 + 256, 32 + 256])
def test_conv2d_bfrag(device, case, nstage):
    """'A through LDS, B from L2' conv kernel (fragment-major weights) vs F.conv2d on bf16-rounded operands: M tails, stride 2,
    1x1 and 3x3, several output-channel tiles, K loops shorter than the ring; + 256 = channel-major K order (round 4: the taps of a
    channel slice back to back), whose f32 output must agree with the tap-major order to summation-order accuracy."""
    from nopesac_amd import _lib, ops
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case) + nstage)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).bfloat16().float()
    scale, bias = 1 + 0.1 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, None, s, p) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    res = torch.randn(ref.shape, generator=g).bfloat16().float()
    ref = F.relu(ref + res)
    xd, rd = _nhwc(x).to(device, torch.bfloat16), _nhwc(res).to(device, torch.bfloat16)
    wd = w.permute(0, 2, 3, 1).contiguous().to(device, torch.bfloat16)
    y = torch.empty(B, ref.shape[2], ref.shape[3], Cout, device=device, dtype=torch.bfloat16)
    sd, bd = scale.to(device), bias.to(device)                     # keep the device copies alive across the raw-pointer call
    rc = _lib.load().nopesac_conv2d_nhwc_bfrag(xd.data_ptr(), ops._frag_weights(wd).data_ptr(), sd.data_ptr(),
                                                bd.data_ptr(), rd.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, k, k, s, p,
                                                Cin, Cout, Cout, ops.ACT_RELU, 1, nstage, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc == 0
    assert _rel(y.float().permute(0, 3, 1, 2), ref) < 1.5e-2
    if nstage >= 256:
        outs = []
        for v in (nstage, nstage - 256):
            y32 = torch.full((B, ref.shape[2], ref.shape[3], Cout), float("nan"), device=device)
            rc = _lib.load().nopesac_conv2d_nhwc_bfrag(xd.data_ptr(), ops._frag_weights(wd).data_ptr(), sd.data_ptr(), bd.data_ptr(), None,
                                                        y32.data_ptr(), B, H, W, Cin, Cout, k, k, s, p, Cin, Cout, 0, ops.ACT_NONE, 0, v,
                                                        torch.cuda.current_stream().cuda_stream)
            assert rc == 0
            outs.append(y32)
        torch.cuda.synchronize()
        assert _rel(outs[0], outs[1]) < 2e-5
        if k == 1:
            assert torch.equal(outs[0], outs[1])                 # one tap: both orders are the same walk


@pytest.mark.parametrize("case", [(4, 60, 80, 128, 256, 3, 1, 1), (3, 15, 20, 512, 128, 3, 1, 1), (2, 15, 20, 2048, 128, 3, 1, 1)])
def test_conv2d_lds_dma_kernel_channel_major_k_order(device, case, monkeypatch):
    """conv_igemm_glds_kernel walks K channel-major on stride-1 KxK layers (round 4: the taps of a 64-channel slice back to back, so
    the re-read pixels stay in L2).  Same products, another summation order: the f32 output must agree with the tap-major walk
    (NOPESAC_GLDS_KMAJOR=0) to summation-order accuracy, and with F.conv2d on the bf16-rounded operands."""
    from nopesac_amd import ops
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).bfloat16()
    xd, wd = _nhwc(x.float()).to(device, torch.bfloat16), w.float().permute(0, 2, 3, 1).contiguous().to(device, torch.bfloat16)
    monkeypatch.setenv("NOPESAC_CONV_FORCE", "glds")            # (small grids with a long K default to the 64x64-tile kernel since round 4)
    y_cm = ops.conv2d(xd, wd, stride=s, pad=p, out_dtype=torch.float32)
    monkeypatch.setenv("NOPESAC_GLDS_KMAJOR", "0")
    y_tm = ops.conv2d(xd, wd, stride=s, pad=p, out_dtype=torch.float32)
    monkeypatch.delenv("NOPESAC_GLDS_KMAJOR")
    torch.cuda.synchronize()
    ref = F.conv2d(x.float(), w.float(), None, s, p)
    assert _rel(y_cm, y_tm) < 2e-5 and not torch.equal(y_cm, y_tm)          # (not equal: proves the two orders were both exercised)
    assert _rel(y_cm.permute(0, 3, 1, 2), ref) < 2e-5


@pytest.mark.parametrize("B", [1, 5])
def test_posenet_branch_tail_matches_the_per_layer_convs(device, B):
    """nopesac_posenet_branch_tail_bf16 (round 4: layers 1..5 of both pose-net branches in one launch, activations in LDS) against
    five ops.conv2d launches per branch with the same bf16 weights, folded BatchNorm and LeakyReLU: same rounding points (bf16
    activations between the layers, f32 out of the last one), only the f32 summation order differs."""
    from nopesac_amd import ops
    from nopesac_amd.modeling.params import ConvW
    g = torch.Generator().manual_seed(40 + B)
    convs = [[ConvW((torch.randn(128, 128, 3, 3, generator=g) / math.sqrt(9 * 128 / 2)).to(device), (1 + 0.1 * torch.randn(128, generator=g)).to(device),
                    (0.1 * torch.randn(128, generator=g)).to(device)) for _ in range(5)] for _ in range(2)]
    xs = [torch.randn(B, 15, 20, 128, generator=g).to(device, torch.bfloat16) for _ in range(2)]
    packed = ops.PoseBranchTail(convs[0], convs[1])
    yt, yr = ops.posenet_branch_tail(xs[0], xs[1], packed)
    for br, y in ((0, yt), (1, yr)):
        t = xs[br]
        for i, c in enumerate(convs[br]):
            t = ops.conv2d(t, c.w(torch.bfloat16), c.scale, c.bias, stride=2 if i % 2 == 0 else 1, pad=1, act=ops.ACT_LEAKY,
                           out_dtype=torch.float32 if i == 4 else None)
        assert y.shape == t.shape == (B, 2, 3, 128)
        assert _rel(y, t) < 1e-2, br
    yt2, yr2 = ops.posenet_branch_tail(xs[0], xs[1], packed)
    assert torch.equal(yt, yt2) and torch.equal(yr, yr2)


@pytest.mark.parametrize("M,last", [(3200, False), (3200, True), (77, False)])
def test_decoder_tail(device, M, last):
    """Pre-norm decoder tail (enc_tail kernel, pre_norm = 1) vs the same chain through the per-op bf16-mode kernels."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(M + last)
    rn = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(device)
    attn, tgt, qpos = rn(M, 256).bfloat16(), rn(M, 256), rn(50, 256)
    wo, w1, w2 = rn(256, 256, k=1 / 16).bfloat16(), rn(1024, 256, k=1 / 16).bfloat16(), rn(256, 1024, k=1 / 32).bfloat16()
    bo, b1, b2 = rn(256, k=0.1), rn(1024, k=0.1), rn(256, k=0.1)
    g3, be3, gn, ben = 1 + rn(256, k=0.1), rn(256, k=0.1), 1 + rn(256, k=0.1), rn(256, k=0.1)
    W = {"wo": ops.mfma_fragment_major(wo), "w1": ops.mfma_fragment_major(w1), "w2": ops.mfma_fragment_major(w2),
         "bo": bo, "b1": b1, "b2": b2, "g3": g3, "be3": be3, "gn": gn, "ben": ben}
    out = ops.decoder_tail(attn, tgt, W, pos=qpos, want=("yn",) if last else ("y", "y16", "ypos16"))
    s = ops.linear(attn, wo, bo, residual=tgt, out_dtype=torch.float32)
    t16 = ops.layernorm_ex(s, g3, be3, want=("y16",))["y16"]
    h = ops.linear(t16, w1, b1, act=ops.ACT_RELU, out_dtype=torch.bfloat16)
    u = ops.linear(h, w2, b2, residual=s, out_dtype=torch.float32)
    ref = ops.layernorm_ex(u, gn, ben, addend=qpos, want=("y", "y16", "y2_16"))
    # independent fp32 PyTorch reference (pre-norm decoder tail: transformer.py:293-322 after the cross-attention)
    a, tg = attn.float().cpu(), tgt.cpu()
    s_ = a @ wo.float().cpu().T + bo.cpu() + tg
    u_ = s_ + F.relu(F.layer_norm(s_, (256,), g3.cpu(), be3.cpu(), 1e-5) @ w1.float().cpu().T + b1.cpu()) @ w2.float().cpu().T + b2.cpu()
    n_ = F.layer_norm(u_, (256,), gn.cpu(), ben.cpu(), 1e-5)
    if last:
        assert set(out) == {"yn"} and _rel(out["yn"], ref["y"]) < 5e-3
        assert _rel(out["yn"].cpu(), n_) < 1.5e-2
    else:
        assert _rel(out["y"], u) < 5e-3
        assert _rel(out["y"].cpu(), u_) < 1.5e-2
        assert _rel(out["y16"].float(), ref["y16"].float()) < 1e-2 and _rel(out["ypos16"].float(), ref["y2_16"].float()) < 1e-2


@pytest.mark.parametrize("B,H,W", [(2, 120, 160), (3, 37, 45), (1, 16, 16), (1, 5, 70)])
def test_conv3x3_c64(device, B, H, W):
    """Halo-tile 3x3 64 -> 64 conv vs the generic bf16 conv kernel (same MFMA K order: bit-exact) and F.conv2d."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(B, 64, H, W, generator=g).bfloat16()
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).bfloat16()
    scale, bias = (1 + 0.1 * torch.randn(64, generator=g)).to(device), (0.1 * torch.randn(64, generator=g)).to(device)
    xd, wd = _nhwc(x.float()).to(device, torch.bfloat16), w.float().permute(0, 2, 3, 1).contiguous().to(device, torch.bfloat16)
    y = ops.conv3x3_c64(xd, wd, scale, bias)
    ref = ops.conv2d(xd, wd, scale, bias, stride=1, pad=1, act=ops.ACT_RELU)
    assert torch.equal(y, ref)
    ref32 = F.relu(F.conv2d(x.float(), w.float(), None, 1, 1) * scale.cpu().view(1, -1, 1, 1) + bias.cpu().view(1, -1, 1, 1))
    assert _rel(y.float().permute(0, 3, 1, 2).cpu(), ref32) < 1.5e-2


@pytest.mark.parametrize("dtype,B,H,W,C", [(torch.float32, 3, 60, 80, 128), (torch.bfloat16, 2, 30, 40, 128), (torch.bfloat16, 1, 7, 9, 256)])
def test_groupnorm_split_path(device, dtype, B, H, W, C):
    """GroupNorm (split statistics + per-channel affine apply) vs F.group_norm, incl. a mean far from zero."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(H)
    x = torch.randn(B, C, H, W, generator=g) * 2 + 5
    ga, be = 1 + 0.2 * torch.randn(C, generator=g), torch.randn(C, generator=g)
    xd = _nhwc(x).to(device, dtype)
    y = ops.groupnorm(xd, ga.to(device), be.to(device), 32, 1e-5, act=ops.ACT_NONE)
    ref = F.group_norm(xd.float().cpu().permute(0, 3, 1, 2), 32, ga, be, 1e-5)
    assert y.dtype == dtype and _rel(y.float().cpu().permute(0, 3, 1, 2), ref) < (2e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("case", [(2, 30, 40, 256, 256), (1, 60, 80, 128, 128), (2, 17, 21, 64, 128), (1, 16, 16, 512, 384)])
@pytest.mark.parametrize("tile", [0, 1])
def test_conv3x3_halo(device, case, tile):
    """Halo-tile 3x3 conv (input channels walked in chunks of 64, all nine taps read from the LDS halo) vs the generic bf16 kernel
    and F.conv2d; partial tiles at the image border, several output-channel blocks."""
    from nopesac_amd import _lib, ops
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case) + tile)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).bfloat16()
    scale, bias = (1 + 0.1 * torch.randn(Cout, generator=g)).to(device), (0.1 * torch.randn(Cout, generator=g)).to(device)
    xd, wd = _nhwc(x.float()).to(device, torch.bfloat16), w.float().permute(0, 2, 3, 1).contiguous().to(device, torch.bfloat16)
    y = torch.empty(B, H, W, Cout, device=device, dtype=torch.bfloat16)
    wf = ops._frag_weights(wd)
    rc = _lib.load().nopesac_conv3x3_halo_bf16(xd.data_ptr(), wf.data_ptr(), scale.data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout,
                                                ops.ACT_LEAKY, tile, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc == 0
    ref = ops.conv2d(xd, wd, scale, bias, stride=1, pad=1, act=ops.ACT_LEAKY)
    assert _rel(y.float(), ref.float()) < 1e-2
    ref32 = F.leaky_relu(F.conv2d(x.float(), w.float(), None, 1, 1) * scale.cpu().view(1, -1, 1, 1) + bias.cpu().view(1, -1, 1, 1), 0.01)
    assert _rel(y.float().permute(0, 3, 1, 2).cpu(), ref32) < 1.5e-2


def _q8(t):
    """float -> e4m3fn with the kernels' saturating round-to-nearest-even."""
    return t.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn)


@pytest.mark.parametrize("case", [(2, 30, 40, 256, 256, 3, 1, 1), (3, 31, 29, 128, 128, 3, 2, 1), (2, 24, 32, 512, 256, 1, 1, 0),
                                  (1, 15, 20, 64, 384, 3, 1, 1), (2, 9, 7, 64, 128, 1, 1, 0), (2, 15, 20, 512, 512, 3, 2, 1)])
@pytest.mark.parametrize("out_dt", ["bf16", "f32", "fp8"])
def test_conv2d_fp8(device, case, out_dt):
    """fp8 (e4m3fn) conv on the K = 64 fp8 MFMA vs F.conv2d on the de-quantised operands (the products of two e4m3 values are exact
    in f32, so only the accumulation order differs): operand layouts, M tails, stride 2, both K-tile variants, all output types."""
    from nopesac_amd import ops
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x8 = _q8(torch.randn(B, Cin, H, W, generator=g) * 2)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k) * (1 + torch.arange(Cout).view(-1, 1, 1, 1) % 5)
    w8f, wsc = ops.quantize_weights_fp8(w.permute(0, 2, 3, 1).contiguous())
    wdq = (w / wsc.view(-1, 1, 1, 1)).clamp(-448, 448).to(torch.float8_e4m3fn).float() * wsc.view(-1, 1, 1, 1)
    assert _rel(wdq, w) < 0.07                                        # 3 mantissa bits
    bn_s, bias = 1 + 0.1 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g)
    ref = F.relu(F.conv2d(x8.float(), wdq, None, s, p) * bn_s.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1))
    dt = {"bf16": torch.bfloat16, "f32": torch.float32, "fp8": torch.float8_e4m3fn}[out_dt]
    for variant in ([3, 32] if Cin % 128 == 0 else [32]):
        y = ops.conv2d_fp8(_nhwc(x8.view(torch.uint8)).view(torch.float8_e4m3fn).to(device), w8f.to(device), (bn_s * wsc).to(device),
                           bias.to(device), ksize=k, stride=s, pad=p, act=ops.ACT_RELU, out_dtype=dt, variant=variant)
        torch.cuda.synchronize()
        got = y.float().permute(0, 3, 1, 2).cpu()
        if out_dt == "fp8":                                           # same rounding as torch's cast; allow one-ulp flips at ties
            want = _q8(ref).float()
            assert (got != want).float().mean() < 2e-3 and _rel(got, want) < 0.07
        else:
            assert _rel(got, ref) < (1e-2 if out_dt == "bf16" else 1e-4), variant


def test_conv2d_generic_and_tail_fp8_outputs(device):
    """The producers of the fp8 conv inputs: nopesac_conv2d_nhwc with out_dt = FP8, and the bottleneck tail's next-conv1 output
    written as fp8 - both must equal the bf16 result rounded to e4m3fn."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 20, 24, 64, generator=g).to(device, torch.bfloat16)
    w = (torch.randn(64, 1, 1, 64, generator=g) / 8).to(device, torch.bfloat16)
    sc, bi = (1 + 0.1 * torch.randn(64, generator=g)).to(device) * 150, torch.randn(64, generator=g).to(device)
    y16 = ops.conv2d(x, w, sc, bi, act=ops.ACT_RELU)
    y8 = ops.conv2d(x, w, sc, bi, act=ops.ACT_RELU, out_dtype=torch.float8_e4m3fn)
    # the kernel rounds the f32 result, the reference the bf16-rounded one: identical except for rare double-rounding flips
    want = _q8(y16)
    assert y8.dtype == torch.float8_e4m3fn and (y8.float() != want.float()).float().mean() < 0.02
    assert _rel(y8.float(), want.float().to(device)) < 0.08 and float(y8.float().max()) <= 448.0 and float(y16.float().max()) > 448.0   # saturates
    # tail: C=64, C4=256, CN=64 identity block
    b = torch.randn(2, 20, 24, 64, generator=g).to(device, torch.bfloat16)
    res = torch.randn(2, 20, 24, 256, generator=g).to(device, torch.bfloat16)
    w3 = ops.mfma_fragment_major((torch.randn(256, 64, generator=g) / 8).to(device, torch.bfloat16))
    w1 = ops.mfma_fragment_major((torch.randn(64, 256, generator=g) / 16).to(device, torch.bfloat16))
    s3, b3 = torch.ones(256, device=device), torch.zeros(256, device=device)
    s1, b1 = torch.full((64,), 30.0, device=device), torch.zeros(64, device=device)
    ya, oa = ops.bottleneck_tail(b, w3, s3, b3, residual=res, w1=w1, s1=s1, b1=b1)
    yb, ob = ops.bottleneck_tail(b, w3, s3, b3, residual=res, w1=w1, s1=s1, b1=b1, o_fp8=True)
    assert torch.equal(ya, yb) and ob.dtype == torch.float8_e4m3fn
    assert torch.equal(ob.float(), _q8(oa).float().to(device))         # converted from the same bf16 staging values: bit-identical


@pytest.mark.parametrize("H,W", [(100, 172), (100, 170), (480, 640)])
def test_stem_fused_raw_equals_preprocess_plus_stem(device, H, W):
    """The raw-input stem (f32 NCHW images, normalisation while staging) is bit-identical to preprocess + fused stem, including
    image borders, heights / widths that are not multiples of the tile, and the real 480 x 640 size."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(21)
    img = torch.randint(0, 256, (3, 3, H, W), generator=g).float().to(device)
    mean = torch.tensor([123.675, 116.28, 103.53], device=device)
    std = torch.tensor([58.395, 57.12, 57.375], device=device)
    w = (torch.randn(64, 224, generator=g) / 12).to(device, torch.bfloat16)
    sc, bi = (1 + 0.1 * torch.randn(64, generator=g)).to(device), (0.1 * torch.randn(64, generator=g)).to(device)
    a = ops.stem_fused(ops.preprocess(img, mean, std, 4, torch.bfloat16), w, sc, bi)
    b = ops.stem_fused_raw(img, mean, std, w, sc, bi)
    assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("H,W", [(100, 172), (480, 640)])
def test_stem_fused_raw_shifted_has_no_input_rounding(device, H, W):
    """Round 4: the raw-image stem with the normalisation folded into its weights / BN shift (the patch holds v - 128, exact in bf16
    for 8-bit pixels) against an f64 reference of normalise -> conv7x7/s2 -> BN -> ReLU -> maxpool on the SAME bf16-rounded folded
    weights: what is left is accumulation order + the output rounding (< 1/256 relative), at the image borders too (zero padding of
    the normalised image = raw pad value mean - 128).  The unfolded stem (normalised image rounded to bf16) must be further away from
    the exact-operand reference than the folded one."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(H)
    img = torch.randint(0, 256, (2, 3, H, W), generator=g).float()
    mean, std = torch.tensor([123.675, 116.28, 103.53]), torch.tensor([58.395, 57.12, 57.375])
    w = torch.randn(64, 7, 7, 3, generator=g) / 12
    sc, bi = 1 + 0.1 * torch.randn(64, generator=g), 0.1 * torch.randn(64, generator=g)
    pad3, w224, bsh = ops.fold_stem_normalisation(w, sc, bi, mean, std)
    y = ops.stem_fused_raw_shifted(img.to(device), pad3.to(device), w224.to(device), sc.to(device), bsh.to(device)).float().cpu()
    # reference with the rounded folded weights, everything else exact: conv over v - 128 with the raw pad value, f64
    wf = w224.view(64, 7, 8, 4)[:, :, :7, :3].double().permute(0, 3, 1, 2)
    xs = F.pad(img.double() - 128.0, (3, 3, 3, 3))
    pb = pad3.to(torch.bfloat16).double()                      # the kernel stores the pad value as bf16 too
    for c in range(3):
        xs[:, c, :3, :] = pb[c]; xs[:, c, -3:, :] = pb[c]; xs[:, c, :, :3] = pb[c]; xs[:, c, :, -3:] = pb[c]
    ref = F.max_pool2d(F.relu(F.conv2d(xs, wf, None, 2, 0) * sc.double().view(1, -1, 1, 1) + bsh.double().view(1, -1, 1, 1)), 3, 2, 1)
    assert _rel(y.permute(0, 3, 1, 2).double(), ref) < 3e-3                    # bf16 output rounding only
    # exact-operand reference (f64 weights, exact normalisation) vs folded and vs unfolded stem
    exact = F.max_pool2d(F.relu(F.conv2d((img.double() - mean.double().view(1, 3, 1, 1)) / std.double().view(1, 3, 1, 1),
                                         w.double().permute(0, 3, 1, 2), None, 2, 3) * sc.double().view(1, -1, 1, 1)
                                + bi.double().view(1, -1, 1, 1)), 3, 2, 1)
    w8 = torch.zeros(64, 7, 8, 4)
    w8[:, :, :7, :3] = w
    old = ops.stem_fused_raw(img.to(device), mean.to(device), std.to(device), w8.reshape(64, 224).to(device, torch.bfloat16), sc.to(device),
                             bi.to(device)).float().cpu()
    # (the maximum error is the bf16 OUTPUT rounding in both; the root-mean-square error shows the operand rounding that is gone)
    rms = lambda a: float(((a.permute(0, 3, 1, 2).double() - exact) ** 2).mean().sqrt())
    e_new, e_old = rms(y), rms(old)
    assert e_new < e_old and _rel(y.permute(0, 3, 1, 2).double(), exact) < 6e-3, (e_new, e_old)


@pytest.mark.parametrize("case", [(2, 30, 40, 256, 256, 3, 1, 1),      # 10 tiles, 36 K-tiles (even)
                                  (3, 31, 29, 128, 256, 3, 2, 1),      # stride 2, M tail (720 rows = 2.8 tiles), 18 K-tiles
                                  (2, 24, 32, 512, 512, 1, 1, 0),      # 1x1, two channel tiles, 8 K-tiles
                                  (1, 15, 20, 64, 256, 3, 1, 1),       # 9 K-tiles (odd), Cin = 64: every K-tile is another tap
                                  (2, 9, 7, 64, 256, 1, 1, 0),         # ONE K-tile, 126 rows (half-empty tile)
                                  (1, 17, 16, 128, 256, 1, 1, 0),      # two K-tiles
                                  (1, 12, 20, 192, 256, 1, 1, 0),      # three K-tiles
                                  (5, 60, 80, 64, 512, 3, 1, 1)])      # 188 tiles in two channel columns
@pytest.mark.parametrize("variant", [0, 32, 64, 0 | (3 << 8), 32 | (1 << 8)])     # + 32: channel-major K order, + 64: generic epilogue, 3 / 1 persistent workgroups
def test_conv2d_p8(device, case, variant):
    """256x256-tile phase-interleaved conv kernel (csrc/conv_p8.hip) vs F.conv2d on bf16-rounded operands, f32 output (the only
    rounding left is the accumulation order): K loops of 1, 2, 3, odd and even tile counts (every prologue / tail branch of the
    counted-vmcnt schedule), M tails, stride 2, padding taps, two channel tiles; repeated launches must agree bit for bit (a
    DMA / LDS race would show up as run-to-run differences)."""
    from nopesac_amd import _lib, ops
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).bfloat16().float()
    scale, bias = 1 + 0.1 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, None, s, p) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    res = torch.randn(ref.shape, generator=g)
    ref = F.relu(ref + res)
    xd, rd = _nhwc(x).to(device, torch.bfloat16), _nhwc(res).to(device)
    wd = w.permute(0, 2, 3, 1).contiguous().to(device, torch.bfloat16)
    sd, bd = scale.to(device), bias.to(device)
    outs = []
    for rep in range(3):
        y = torch.full((B, ref.shape[2], ref.shape[3], Cout), float("nan"), device=device)
        rc = _lib.load().nopesac_conv2d_nhwc_p8(xd.data_ptr(), wd.data_ptr(), sd.data_ptr(), bd.data_ptr(), rd.data_ptr(), y.data_ptr(),
                                                 B, H, W, Cin, Cout, k, k, s, p, Cin, Cout, Cout, ops.ACT_RELU, 0, variant,
                                                 torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        outs.append(y)
    torch.cuda.synchronize()
    assert _rel(outs[0].permute(0, 3, 1, 2), ref) < 2e-5
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("case", [(2, 30, 40, 256, 256, 3, 1, 1),      # 10 tiles x 36 K-tiles over 256 workgroups: every tile cut 26 ways
                                  (3, 31, 29, 128, 256, 3, 2, 1),      # stride 2, M tail, 18 K-tiles
                                  (2, 24, 32, 512, 512, 1, 1, 0),      # 1x1, two channel tiles, 8 K-tiles
                                  (1, 15, 20, 64, 256, 3, 1, 1),       # 9 K-tiles (odd), Cin = 64
                                  (2, 9, 7, 64, 256, 1, 1, 0),         # ONE K-tile: a single workgroup, nothing to cut
                                  (1, 12, 20, 192, 256, 1, 1, 0),      # three K-tiles
                                  (5, 60, 80, 64, 512, 3, 1, 1),       # 188 tiles x 9 K-tiles: 6.6 units per workgroup
                                  (16, 30, 40, 256, 256, 3, 1, 1)])    # 75 tiles x 36: 10.5 units per workgroup, a tile has 3-4 contributors
@pytest.mark.parametrize("variant", [0, 32, 32 | 64, 32 | (5 << 8), 0 | (19 << 8), 32 | (64 << 8)])     # + (n << 8): n persistent workgroups
def test_conv2d_p8_stream_k(device, case, variant):
    """Stream-K form of the 256x256-tile kernel (conv_igemm_p8_kernel<.., SK>): against F.conv2d (f32 output: only the summation order
    differs), against the plain kernel, and bit for bit against ITSELF over repeated launches - which workgroup arrives last on a tile
    changes from launch to launch, the K-ordered reduction must not.  Workgroup counts 5 / 19 / 64 / 256 give whole tiles + head / tail
    pieces, ranges inside one tile, and tiles with many contributors; the arrival counters must be zero again after every launch."""
    from nopesac_amd import _lib, ops
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).bfloat16().float()
    scale, bias = 1 + 0.1 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, None, s, p) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    res = torch.randn(ref.shape, generator=g)
    ref = F.relu(ref + res)
    xd, rd = _nhwc(x).to(device, torch.bfloat16), _nhwc(res).to(device)
    wd = w.permute(0, 2, 3, 1).contiguous().to(device, torch.bfloat16)
    sd, bd = scale.to(device), bias.to(device)
    ws = ops.p8_sk_workspace(device)
    outs = []
    for rep in range(4):
        y = torch.full((B, ref.shape[2], ref.shape[3], Cout), float("nan"), device=device)
        args = (xd.data_ptr(), wd.data_ptr(), sd.data_ptr(), bd.data_ptr(), rd.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, k, k, s, p, Cin, Cout,
                Cout, ops.ACT_RELU, 0, variant)
        if rep == 3:
            rc = _lib.load().nopesac_conv2d_nhwc_p8(*args, torch.cuda.current_stream().cuda_stream)
        else:
            rc = _lib.load().nopesac_conv2d_nhwc_p8_sk(*args, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        outs.append(y)
    torch.cuda.synchronize()
    assert int(ws[:16384].view(torch.int32).abs().sum()) == 0
    assert _rel(outs[0].permute(0, 3, 1, 2), ref) < 2e-5
    assert _rel(outs[0], outs[3]) < 2e-5
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("act,residual", [("ACT_RELU", False), ("ACT_NONE", False), ("ACT_LEAKY", False), ("ACT_RELU", True)])
def test_conv2d_p8_stream_k_specialised_epilogues(device, act, residual):
    """bf16-output epilogue builds (EPI 1-4) of the stream-K kernel against the plain kernel: both round the same f32 sums up to the
    K-split's summation order; next to a second stream that keeps the chip busy (arrival order and residency then vary)."""
    from nopesac_amd import _lib, ops
    B, H, W, Cin, Cout, k = 16, 30, 40, 256, 512, 3
    g = torch.Generator().manual_seed(13)
    x = torch.randn(B, H, W, Cin, generator=g).to(device, torch.bfloat16)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / math.sqrt(Cin * k * k)).to(device, torch.bfloat16)
    sc, bi = (1 + 0.1 * torch.randn(Cout, generator=g)).to(device), (0.1 * torch.randn(Cout, generator=g)).to(device)
    r = torch.randn(B, H, W, Cout, generator=g).to(device, torch.bfloat16) if residual else None
    ws = ops.p8_sk_workspace(device)
    side = torch.cuda.Stream(device=device)
    big = torch.randn(4096, 4096, device=device)
    outs = []
    for rep in range(5):
        y = torch.zeros(B, H, W, Cout, device=device, dtype=torch.bfloat16)
        args = (x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), r.data_ptr() if residual else None, y.data_ptr(), B, H, W, Cin, Cout,
                k, k, 1, 1, Cin, Cout, Cout if residual else 0, getattr(ops, act), 1, 32)
        if rep in (1, 2):
            with torch.cuda.stream(side):
                for _ in range(3):
                    big = big * 1.0001 + 0.5
        if rep == 4:
            rc = _lib.load().nopesac_conv2d_nhwc_p8(*args, torch.cuda.current_stream().cuda_stream)
        else:
            rc = _lib.load().nopesac_conv2d_nhwc_p8_sk(*args, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        outs.append(y)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[3])
    assert _rel(outs[0].float(), outs[4].float()) < 4e-3 and torch.isfinite(outs[0].float()).all()
    assert (outs[0].float() - outs[4].float()).abs().max() <= 2.0 ** -6 * max(1.0, float(outs[4].float().abs().max()))


@pytest.mark.parametrize("case", [(2, 30, 40, 128, 128, 3, 1, 1),      # 10 tiles, 18 K-tiles (a multiple of the ring depth)
                                  (3, 31, 29, 128, 128, 3, 2, 1),      # stride 2, M tail (720 rows = 2.8 tiles)
                                  (2, 24, 32, 512, 256, 1, 1, 0),      # 1x1, two channel tiles, 8 K-tiles (3 + 3 + 2)
                                  (1, 15, 20, 64, 128, 3, 1, 1),       # 9 K-tiles, Cin = 64: every K-tile is another tap
                                  (2, 9, 7, 64, 128, 1, 1, 0),         # ONE K-tile, 126 rows (half-empty tile)
                                  (1, 17, 16, 128, 128, 1, 1, 0),      # two K-tiles
                                  (1, 12, 20, 192, 128, 1, 1, 0),      # three K-tiles
                                  (1, 12, 20, 256, 128, 1, 1, 0),      # four K-tiles
                                  (1, 12, 20, 320, 384, 1, 1, 0),      # five K-tiles, three channel tiles
                                  (5, 60, 80, 64, 256, 3, 1, 1),       # 188 tiles in two channel columns
                                  (1, 15, 20, 2048, 128, 3, 1, 1)])    # pose-net layer_3's K: 288 K-tiles
@pytest.mark.parametrize("variant", [0, 32, 0 | (3 << 8), 32 | (1 << 8)])     # + 32: channel-major K order; 3 / 1 persistent workgroups
@pytest.mark.parametrize("act", ["ACT_RELU", "ACT_NONE", "ACT_LEAKY"])
def test_conv2d_p8n(device, case, variant, act):
    """256x128-tile kernel (csrc/conv_p8n.hip) vs F.conv2d on bf16-rounded operands (bf16 output: one rounding) and bit for bit vs the
    library's generic bf16 kernel family is NOT expected (another summation order) - so: within bf16 rounding of the f64-accurate result,
    and repeated launches identical (a DMA / LDS race of the three-deep ring would show up as run-to-run differences).  K loops of
    1, 2, 3, 4, 5, 8, 9, 18, 288 K-tiles walk every prologue / tail branch of the counted-vmcnt schedule."""
    from nopesac_amd import _lib, ops
    B, H, W, Cin, Cout, k, s, p = case
    if variant >> 8 and B * H * W * Cin * Cout * k * k > 3e10:
        pytest.skip("one workgroup on the long-K case: covered by the uncapped variants")
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).bfloat16().float()
    scale, bias = 1 + 0.1 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), None, s, p) * scale.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1)
    ref = {"ACT_RELU": F.relu, "ACT_NONE": lambda t: t, "ACT_LEAKY": lambda t: F.leaky_relu(t, 0.01)}[act](ref)
    xd = _nhwc(x).to(device, torch.bfloat16)
    wd = w.permute(0, 2, 3, 1).contiguous().to(device, torch.bfloat16)
    sd, bd = scale.to(device), bias.to(device)
    outs = []
    for rep in range(3):
        y = torch.full((B, ref.shape[2], ref.shape[3], Cout), float("nan"), device=device, dtype=torch.bfloat16)
        rc = _lib.load().nopesac_conv2d_nhwc_p8n(xd.data_ptr(), wd.data_ptr(), sd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, k, k,
                                                  s, p, Cin, Cout, getattr(ops, act), variant, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        outs.append(y)
    torch.cuda.synchronize()
    got = outs[0].float().permute(0, 3, 1, 2).double().cpu()
    err = (got - ref).abs()
    assert torch.isfinite(got).all()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-4 * (1 + math.sqrt(Cin * k * k) * 0.01)).all()), float(err.max())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("case", [(2, 15, 20, 2048, 128, 3, 1, 1, 3, 0),     # layer_3's shape at two images: 3 tiles x 3 slices of 11 / 10 / 11 channel groups
                                  (4, 15, 20, 2048, 128, 3, 1, 1, 8, 0),     # 5 tiles (M tail: 1200 rows) x 8 slices of 4 groups
                                  (1, 15, 20, 512, 128, 3, 1, 1, 5, 0),      # 8 groups in 5 slices: 1 / 2 / 1 / 2 / 2 groups (9 and 18 K-tiles per unit)
                                  (1, 9, 7, 128, 128, 1, 1, 0, 2, 0),        # 1x1: ONE K-tile per unit, 63 rows
                                  (3, 16, 16, 256, 256, 3, 2, 1, 4, 0),      # stride 2, two channel tiles
                                  (6, 15, 20, 1024, 128, 3, 1, 1, 4, 3),     # 8 tiles x 4 slices on THREE persistent workgroups: units walked with a stride, next unit's DMAs behind the partial stores
                                  (2, 12, 20, 320, 384, 1, 1, 0, 5, 0)])     # five groups, five slices, three channel tiles
@pytest.mark.parametrize("act", ["ACT_RELU", "ACT_NONE", "ACT_LEAKY"])
def test_conv2d_p8n_split_k(device, case, act):
    """Round 6: split-K work units of the 256x128-tile kernel (K slices = runs of 64-channel groups, f32 partial tiles in a workspace, a
    second launch sums them in fixed order and applies the epilogue) vs F.conv2d in float64 on the bf16-rounded operands: within one bf16
    rounding; run-to-run identical (fixed summation order; a race between a unit's partial stores and the next unit's DMAs would show);
    the output may be a channel slice of a wider buffer; a workspace that is too small is an argument error."""
    from nopesac_amd import _lib, ops
    B, H, W, Cin, Cout, k, s, p, splits, cap = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).bfloat16().float()
    scale, bias = 1 + 0.1 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), None, s, p) * scale.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1)
    ref = {"ACT_RELU": F.relu, "ACT_NONE": lambda t: t, "ACT_LEAKY": lambda t: F.leaky_relu(t, 0.01)}[act](ref)
    xd = _nhwc(x).to(device, torch.bfloat16)
    wd = w.permute(0, 2, 3, 1).contiguous().to(device, torch.bfloat16)
    sd, bd = scale.to(device), bias.to(device)
    OH, OW = ref.shape[2], ref.shape[3]
    ws = torch.full((splits * B * OH * OW * Cout,), float("nan"), device=device)
    L, st = _lib.load(), torch.cuda.current_stream().cuda_stream
    outs = []
    for rep in range(3):
        wide = torch.full((B, OH, OW, Cout + 16), float("nan"), device=device, dtype=torch.bfloat16)
        y = wide[..., 8:8 + Cout]
        rc = L.nopesac_conv2d_nhwc_p8n_splitk(xd.data_ptr(), wd.data_ptr(), sd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, k, k, s, p,
                                              Cin, Cout + 16, getattr(ops, act), 32 | (cap << 8), splits, ws.data_ptr(), ws.numel() * 4, st)
        assert rc == 0
        outs.append(wide)
    torch.cuda.synchronize()
    assert bool(torch.isnan(outs[0][..., :8].float()).all()) and bool(torch.isnan(outs[0][..., 8 + Cout:].float()).all())      # nothing written next to the slice
    got = outs[0][..., 8:8 + Cout].float().permute(0, 3, 1, 2).double().cpu()
    err = (got - ref).abs()
    assert torch.isfinite(got).all()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-4 * (1 + math.sqrt(Cin * k * k) * 0.01)).all()), float(err.max())
    assert torch.equal(outs[0][..., 8:8 + Cout], outs[1][..., 8:8 + Cout]) and torch.equal(outs[0][..., 8:8 + Cout], outs[2][..., 8:8 + Cout])
    y = torch.empty(B, OH, OW, Cout, device=device, dtype=torch.bfloat16)
    args = (xd.data_ptr(), wd.data_ptr(), sd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, k, k, s, p, Cin, Cout, getattr(ops, act))
    assert L.nopesac_conv2d_nhwc_p8n_splitk(*args, 32, splits, ws.data_ptr(), ws.numel() * 4 - 4, st) != 0        # workspace too small
    assert L.nopesac_conv2d_nhwc_p8n_splitk(*args, 0, splits, ws.data_ptr(), ws.numel() * 4, st) != 0             # tap-major K order
    assert L.nopesac_conv2d_nhwc_p8n_splitk(*args, 32, 1, ws.data_ptr(), ws.numel() * 4, st) != 0                  # one slice is not a split
    assert L.nopesac_conv2d_nhwc_p8n_splitk(*args, 32, Cin // 64 + 1, ws.data_ptr(), ws.numel() * 4, st) != 0     # more slices than channel groups


def test_conv2d_p8n_split_k_through_the_tuner_route(device, monkeypatch):
    """ops.conv2d routed to the split-K configuration (the pose net's first conv at four images: no scale / bias / activation)."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(8)
    B, H, W, Cin, Cout = 4, 15, 20, 2048, 128
    x = torch.randn(B, H, W, Cin, generator=g).to(device, torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / math.sqrt(9 * Cin)).to(device, torch.bfloat16)
    ref = ops.conv2d(x, w, None, None, pad=1)
    monkeypatch.setattr(ops.TUNER, "measuring", True)
    monkeypatch.setattr(ops.TUNER, "choose", lambda key, launch, extra=(): ops.CFG_P8N_SPLIT if ops.CFG_P8N_SPLIT in extra else 0)
    y = ops.conv2d(x, w, None, None, pad=1)
    assert ops.LAST_CONV_CFG[0] == ops.CFG_P8N_SPLIT
    assert _rel(y.float(), ref.float()) < 1e-2


def test_conv2d_p8n_through_the_tuner_route(device, monkeypatch):
    """ops.conv2d routed to the 256x128-tile configuration: strided output view (a channel slice of a wider buffer), no scale."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(6)
    B, H, W, Cin, Cout = 2, 60, 80, 128, 128
    x = torch.randn(B, H, W, Cin, generator=g).to(device, torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / math.sqrt(9 * Cin)).to(device, torch.bfloat16)
    bi = (0.1 * torch.randn(Cout, generator=g)).to(device)
    ref = ops.conv2d(x, w, None, bi, pad=1, act=ops.ACT_RELU)
    monkeypatch.setattr(ops.TUNER, "measuring", True)
    monkeypatch.setattr(ops.TUNER, "choose", lambda key, launch, extra=(): ops.CFG_P8N if ops.CFG_P8N in extra else 0)
    wide = torch.zeros(B, H, W, 2 * Cout, device=device, dtype=torch.bfloat16)
    y = ops.conv2d(x, w, None, bi, pad=1, act=ops.ACT_RELU, out=wide[..., Cout:])
    assert ops.LAST_CONV_CFG[0] == ops.CFG_P8N
    assert _rel(y.float(), ref.float()) < 1e-2 and float(wide[..., :Cout].abs().max()) == 0.0


def test_conv2d_p8_through_the_tuner_route(device, monkeypatch):
    """ops.conv2d routed to the p8 configuration (as the autotuner would): bf16 output, no residual."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(5)
    B, H, W, Cin, Cout = 2, 60, 80, 256, 256
    x = torch.randn(B, H, W, Cin, generator=g).to(device, torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / math.sqrt(9 * Cin)).to(device, torch.bfloat16)
    sc, bi = (1 + 0.1 * torch.randn(Cout, generator=g)).to(device), (0.1 * torch.randn(Cout, generator=g)).to(device)
    ref = ops.conv2d(x, w, sc, bi, pad=1, act=ops.ACT_RELU)
    monkeypatch.setattr(ops.TUNER, "measuring", True)
    monkeypatch.setattr(ops.TUNER, "choose", lambda key, launch, extra=(): ops.CFG_P8 if ops.CFG_P8 in extra else 0)
    y = ops.conv2d(x, w, sc, bi, pad=1, act=ops.ACT_RELU)
    assert ops.LAST_CONV_CFG[0] == ops.CFG_P8
    assert _rel(y.float(), ref.float()) < 1e-2


@pytest.mark.parametrize("act", ["ACT_RELU", "ACT_NONE", "ACT_LEAKY", "ACT_SIGMOID"])
def test_conv2d_p8_specialised_epilogues_match_the_generic_build(device, act):
    """The no-residual / bf16-output epilogue builds (activation fixed at compile time: 20 KB of code instead of 60 KB) must give
    bit for bit what the generic build (variant + 64) gives; several tiles per persistent workgroup (grid capped at 4)."""
    from nopesac_amd import _lib, ops
    B, H, W, Cin, Cout, k = 3, 40, 48, 128, 512, 3
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, H, W, Cin, generator=g).to(device, torch.bfloat16)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / math.sqrt(Cin * k * k)).to(device, torch.bfloat16)
    sc, bi = (1 + 0.1 * torch.randn(Cout, generator=g)).to(device), (0.1 * torch.randn(Cout, generator=g)).to(device)
    outs = []
    for variant in (0, 64, 0 | (4 << 8), 32 | 64 | (4 << 8)):
        y = torch.zeros(B, H, W, Cout, device=device, dtype=torch.bfloat16)
        rc = _lib.load().nopesac_conv2d_nhwc_p8(x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), None, y.data_ptr(), B, H, W, Cin, Cout,
                                                 k, k, 1, 1, Cin, Cout, 0, getattr(ops, act), 1, variant, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        outs.append(y)
    torch.cuda.synchronize()
    ref = ops.conv2d(x, w, sc, bi, pad=1, act=getattr(ops, act))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert _rel(outs[3].float(), outs[0].float()) < 1e-2        # channel-major K order: another summation order
    assert _rel(outs[0].float(), ref.float()) < 1e-2


@pytest.mark.parametrize("case", [(3, 30, 40, 256, 1024, 1, 1, 0, 0),     # res4's expand conv: four K-tiles, 14.06 row tiles x 4 channel tiles
                                  (2, 9, 7, 64, 256, 1, 1, 0, 0),          # one K-tile, 126 rows: half of the only tile is out of bounds
                                  (2, 15, 20, 512, 512, 3, 1, 1, 0),       # 3x3, 72 K-tiles
                                  (5, 23, 31, 128, 512, 1, 1, 0, 64)])     # M tail + y / residual as channel slices of wider buffers
def test_conv2d_p8_residual_relu_epilogue_matches_the_generic_build(device, case):
    """EPI 4 (bf16 residual + ReLU, bf16 output; round 4: residual rows of pass q+1 requested inside pass q, buffer-descriptor bounds
    instead of row tests) must give bit for bit what the generic epilogue build (variant + 64) gives: whole grids, grids capped at
    3 / 1 persistent workgroups (several tiles per workgroup: the residual prefetch crosses the next tile's DMAs), M tails, strided
    output / residual views; bytes outside the output view stay untouched."""
    from nopesac_amd import _lib, ops
    B, H, W, Cin, Cout, k, s, p, extra = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, H, W, Cin, generator=g).to(device, torch.bfloat16)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / math.sqrt(Cin * k * k)).to(device, torch.bfloat16)
    sc, bi = (1 + 0.1 * torch.randn(Cout, generator=g)).to(device), (0.1 * torch.randn(Cout, generator=g)).to(device)
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    cs = Cout + extra
    res = torch.randn(B, OH, OW, cs, generator=g).to(device, torch.bfloat16)
    outs = []
    for variant in (0, 64, 0 | (3 << 8), 32 | (1 << 8), 32 | 64 | (1 << 8)):
        y = torch.full((B, OH, OW, cs), 7.0, device=device, dtype=torch.bfloat16)
        rc = _lib.load().nopesac_conv2d_nhwc_p8(x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), res.data_ptr(), y.data_ptr(), B, H, W, Cin,
                                                 Cout, k, k, s, p, Cin, cs, cs, ops.ACT_RELU, 1, variant, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        outs.append(y)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(outs[3], outs[4])                                    # channel-major K order: specialised vs generic
    if extra:
        assert bool((outs[0][..., Cout:] == 7.0).all())
    ref = F.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), None, s, p) * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1)
                 + res[..., :Cout].float().permute(0, 3, 1, 2))
    assert _rel(outs[0][..., :Cout].float().permute(0, 3, 1, 2), ref) < 1e-2
    assert _rel(outs[3][..., :Cout].float().permute(0, 3, 1, 2), ref) < 1e-2


def test_conv2d_p8_bf16_epilogue_into_a_channel_slice_with_a_row_tail(device):
    """EPI 1 (bf16 staging, stores through a buffer descriptor that ends with the last row: round 4) writing a channel slice of a
    wider buffer, M = 2.46 tiles: the other channels and nothing beyond the last row may be touched."""
    from nopesac_amd import _lib, ops
    B, H, W, Cin, Cout, pad_c = 1, 21, 30, 64, 256, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H, W, Cin, generator=g).to(device, torch.bfloat16)
    w = (torch.randn(Cout, 1, 1, Cin, generator=g) / 8).to(device, torch.bfloat16)
    sc, bi = (1 + 0.1 * torch.randn(Cout, generator=g)).to(device), (0.1 * torch.randn(Cout, generator=g)).to(device)
    buf = torch.full((B * H * W + 300, Cout + pad_c), 7.0, device=device, dtype=torch.bfloat16)      # 300 guard rows behind the tensor
    y = buf[:B * H * W].view(B, H, W, Cout + pad_c)
    rc = _lib.load().nopesac_conv2d_nhwc_p8(x.data_ptr(), w.data_ptr(), sc.data_ptr(), bi.data_ptr(), None, y.data_ptr(), B, H, W, Cin, Cout,
                                             1, 1, 1, 0, Cin, Cout + pad_c, 0, ops.ACT_RELU, 1, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    ref = ops.conv2d(x, w, sc, bi, act=ops.ACT_RELU)
    assert _rel(y[..., :Cout].float(), ref.float()) < 1e-2
    assert bool((y[..., Cout:] == 7.0).all()) and bool((buf[B * H * W:] == 7.0).all())


@pytest.mark.parametrize("nq", [50, 63, 7])
def test_sinkhorn_four_wave_kernel_matches_the_1024_thread_kernel(device, nq, monkeypatch):
    """matcher_sinkhorn_w4_kernel (round 4: four waves, lane = row / column, per-wave partial log-sum-exp merged through one LDS
    exchange, hardware exp / log) against the 1024-thread kernel it replaces for nq <= 63 (library expf / logf, shuffle reductions):
    log scores within 2e-5 absolute after 200 iterations, identical assignments; ragged plane counts incl. 0 and nq."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(nq)
    B = 9
    dots = (2.0 * torch.randn(B, nq, nq, generator=g)).to(device)
    p1, p2 = torch.randn(B, nq, 3, generator=g).to(device), torch.randn(B, nq, 3, generator=g).to(device)
    cam = torch.randn(B, 7, generator=g)
    cam[:, 3:] = torch.nn.functional.normalize(cam[:, 3:], dim=1)
    cam = cam.to(device)
    n1 = torch.tensor([nq, 0, 1, nq, nq // 2, 3, nq - 1, 1, nq // 3][:B], dtype=torch.int32, device=device)
    n2 = torch.tensor([nq, 5, 1, 2, nq, nq // 2, nq, 0, nq // 3 + 1][:B], dtype=torch.int32, device=device)
    bin_score = torch.tensor([0.7], device=device)
    args = (dots, p1, p2, cam, n1, n2, bin_score, 4.0, 8.0, 200, 0.2)
    ls4, A4 = ops.matcher_sinkhorn(*args)
    monkeypatch.setenv("NOPESAC_SINKHORN_NO_W4", "1")
    ls, A = ops.matcher_sinkhorn(*args)
    monkeypatch.delenv("NOPESAC_SINKHORN_NO_W4")
    valid = ls > -1e29
    assert torch.equal(valid, ls4 > -1e29)
    assert float((ls4[valid] - ls[valid]).abs().max()) < 2e-5
    assert torch.equal(A4, A)
    ls4b, A4b = ops.matcher_sinkhorn(*args)
    assert torch.equal(ls4, ls4b) and torch.equal(A4, A4b)


@pytest.mark.parametrize("nq", [64, 67, 68, 100, 104, 127, 128])
def test_sinkhorn_row_group_kernel_matches_the_1024_thread_kernel(device, nq, monkeypatch):
    """matcher_sinkhorn_wg_kernel (round 6: rows / columns in groups of 64 lanes, 8 or 12 waves, one exchange + one barrier per phase) for
    nq + 1 > 64 - BASELINE configs[2] (nq = 64) and configs[4] (nq = 128) - against the 1024-thread kernel it replaces: log scores within
    2e-5 absolute after 200 iterations, identical assignments, bit-identical run to run; plane counts incl. 0, 1, 63 / 64 / 65 (a row
    group that is exactly full / one row into the next) and nq; every template instance (KW 17 / 26 / 32, three row groups at nq = 128)."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(nq)
    B = 12
    dots = (2.0 * torch.randn(B, nq, nq, generator=g)).to(device)
    p1, p2 = torch.randn(B, nq, 3, generator=g).to(device), torch.randn(B, nq, 3, generator=g).to(device)
    cam = torch.randn(B, 7, generator=g)
    cam[:, 3:] = torch.nn.functional.normalize(cam[:, 3:], dim=1)
    cam = cam.to(device)
    n1 = torch.tensor([nq, 0, 1, nq, nq // 2, 3, nq - 1, 1, 63, 64, min(65, nq), nq], dtype=torch.int32, device=device)
    n2 = torch.tensor([nq, 5, 1, 2, nq, nq // 2, nq, 0, 64, 63, 3, min(65, nq)], dtype=torch.int32, device=device)
    bin_score = torch.tensor([0.7], device=device)
    args = (dots, p1, p2, cam, n1, n2, bin_score, 4.0, 8.0, 200, 0.2)
    lsg, Ag = ops.matcher_sinkhorn(*args)
    monkeypatch.setenv("NOPESAC_SINKHORN_NO_WG", "1")
    ls, A = ops.matcher_sinkhorn(*args)
    monkeypatch.delenv("NOPESAC_SINKHORN_NO_WG")
    valid = ls > -1e29
    assert torch.equal(valid, lsg > -1e29)
    assert float((lsg[valid] - ls[valid]).abs().max()) < 2e-5
    assert torch.equal(Ag, A)
    lsb, Ab = ops.matcher_sinkhorn(*args)
    assert torch.equal(lsg, lsb) and torch.equal(Ag, Ab)


MLP_CHAIN_CASES = {
    # name: (rows, x_width, bcast_width, rows_per, [(N, act, tapped)])
    "geo_encoder+proj": (100, 8, 0, 1, [(1024, "ACT_RELU", False), (1024, "ACT_RELU", False), (1024, "ACT_RELU", False), (256, "ACT_NONE", True)]),
    "bcast_prefix+taps": (75, 256, 256, 7, [(512, "ACT_RELU", False), (512, "ACT_RELU", False), (256, "ACT_RELU", True), (4, "ACT_NONE", True)]),
    "score_proj_k50": (33, 50, 0, 1, [(128, "ACT_RELU", False), (128, "ACT_RELU", True), (64, "ACT_NONE", True)]),
    "widest_1280_in_9_layers": (1600, 1280, 0, 1, [(1024, "ACT_RELU", False), (1024, "ACT_RELU", False), (1024, "ACT_NONE", True),
                                                    (512, "ACT_RELU", False), (512, "ACT_RELU", False), (512, "ACT_RELU", False),
                                                    (512, "ACT_RELU", False), (512, "ACT_LEAKY", False), (256, "ACT_NONE", True)]),
    "one_layer_n3": (32, 256, 0, 1, [(3, "ACT_NONE", True)]),
    "odd_widths_sigmoid": (65, 19, 5, 3, [(96, "ACT_SIGMOID", True), (40, "ACT_NONE", True)]),
}


@pytest.mark.parametrize("name", sorted(MLP_CHAIN_CASES))
def test_mlp_chain_matches_per_layer_launches_and_fp32(device, name):
    """nopesac_mlp_chain_bf16 (whole MLP stack in one launch, activations in LDS) against (i) one ops.linear launch per layer with
    the same bf16 weights - same rounding points, only the f32 summation order differs - and (ii) a plain fp32 torch reference
    with the bf16 operand rounding applied explicitly.  Tapped intermediate outputs, the broadcast input prefix, row tails,
    widths that are not multiples of 4 / 32 / 128, outputs written into column slices of wider buffers."""
    from nopesac_amd import ops
    rows, kx, kb, rows_per, spec = MLP_CHAIN_CASES[name]
    g = torch.Generator().manual_seed(len(name) * 7 + rows)
    x = torch.randn(rows, kx + 3, generator=g).to(device)[:, :kx] if kx % 4 else torch.randn(rows, kx, generator=g).to(device)
    xb = torch.randn(-(-rows // rows_per), kb, generator=g).to(device) if kb else None
    layers, acts, outs, ws = [], [], [], []
    k = kx + kb
    for (n, act, tap) in spec:
        w = (torch.randn(n, k, generator=g) * (1.6 / math.sqrt(k))).to(device)
        b = (0.2 * torch.randn(n, generator=g)).to(device)
        ws.append((w, b))
        layers.append(ops.MlpLayer(w, b))
        acts.append(getattr(ops, act))
        if tap:          # a column slice of a wider NaN-filled buffer: nothing outside the slice may be touched
            buf = torch.full((rows, n + 8), float("nan"), device=device)
            outs.append(buf[:, 4:4 + n] if n % 4 == 0 else buf[:, 1:1 + n])
        else:
            outs.append(None)
        k = n
    y = ops.mlp_chain(x, layers, acts, outs, x_bcast=xb, rows_per=rows_per)
    torch.cuda.synchronize()
    # references
    full = x if xb is None else torch.cat([xb.repeat_interleave(rows_per, 0)[:rows], x], 1)
    a_lin, a_ref = full.contiguous(), full.double()
    act_fn = {"ACT_RELU": torch.relu, "ACT_NONE": lambda t: t, "ACT_LEAKY": lambda t: F.leaky_relu(t, 0.01), "ACT_SIGMOID": torch.sigmoid}
    for i, ((w, b), (n, act, tap)) in enumerate(zip(ws, spec)):
        a_lin = ops.linear(a_lin, w.to(torch.bfloat16).contiguous(), b, act=getattr(ops, act))
        a_ref = act_fn[act](a_ref.float().bfloat16().double() @ w.bfloat16().double().t() + b.double())
        if tap:
            o = outs[i]
            assert torch.isfinite(o).all(), (name, i)
            bound = 2e-3 + 6e-4 * i          # bf16 roundings of nearly equal f32 values flip and propagate: grows with depth
            assert _rel(o, a_lin) < bound, (name, i, _rel(o, a_lin))
            assert _rel(o, a_ref) < bound, (name, i, _rel(o, a_ref))
            buf = o._base if o._base is not None else o
            mask = torch.ones_like(buf, dtype=torch.bool)
            c0 = 4 if n % 4 == 0 else 1
            mask[:, c0:c0 + n] = False
            assert torch.isnan(buf[mask]).all(), "wrote outside its column slice"
    assert y is outs[-1]


def test_mlp_chain_rejects_bad_chains(device):
    from nopesac_amd import _lib, ops
    x = torch.randn(8, 256, device=device)
    l1, l2 = ops.MlpLayer(torch.randn(128, 256, device=device), None), ops.MlpLayer(torch.randn(64, 100, device=device), None)
    with pytest.raises(_lib.HipKernelError):                       # K of layer 2 is not the width of layer 1
        ops.mlp_chain(x, [l1, l2], [0, 0], [None, torch.empty(8, 64, device=device)])
    with pytest.raises(ops.OpsArgumentError):                      # no output for the last layer is an argument error before the launch
        ops.mlp_chain(x, [l1], [0], [torch.empty(8, 100, device=device)])
    with pytest.raises(_lib.HipKernelError):
        ops.mlp_chain(x, [l1], [0], [None])


@pytest.mark.parametrize("B,nq,K", [(3, 50, 32), (2, 64, 64), (2, 128, 128), (1, 50, 1), (4, 128, 77), (3, 100, 64)])
def test_force_k_select_matches_the_torch_formulation(device, B, nq, K):
    """The benchmark-only K control as one launch vs the torch formulation the oracle uses (topk -> sorted indices -> gathers):
    identical feats / n_kept, including rows >= K zeroed and a larger leading dimension of the inputs (2B views)."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + K)
    logits = torch.randn(2 * B, nq, 2, generator=g).to(device)
    qf = torch.randn(2 * B, nq, 256, generator=g).to(device)
    perm = torch.stack([torch.randperm(K, generator=g) for _ in range(B)]).to(device)
    noise = (0.01 * torch.randn(B, K, 256, generator=g)).to(device)
    feats, n_kept = ops.force_k_select(logits, qf, perm, noise, B, K)
    score = logits[:B, :, 0] - logits[:B, :, 1]
    idx = torch.topk(score, K, dim=1).indices.sort(dim=1).values
    f1 = torch.gather(qf[:B], 1, idx.unsqueeze(-1).expand(B, K, 256))
    f2 = torch.gather(f1, 1, perm.unsqueeze(-1).expand(B, K, 256)) + noise
    want = torch.zeros(2 * B, nq, 256, device=device)
    want[:B, :K], want[B:, :K] = f1, f2
    assert torch.equal(feats, want)
    assert n_kept.tolist() == [K] * (2 * B)


def test_score_maps_reproducible_next_to_mfma_kernels(device):
    """nopesac_ransac_score_maps on constant inputs while ANOTHER stream loops the res3 bottleneck tail (an MFMA kernel that leaves room
    for co-resident waves): every launch must reproduce the result of an idle GPU.  With packed-f32 VALU instructions in the kernel
    (the SLP vectoriser's default) 60 % of the launches had lanes 48-63 of some waves wrong (profiles/r3_packed_fp32_hazard.txt)."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(3)
    rn = lambda *s: torch.randn(*s, generator=g).to(device)
    B, nq = 32, 50
    geo, rot_raw, trans_raw = rn(B, nq, 6), rn(B, nq, 4), rn(B, nq, 3)
    init_rot, init_trans = F.normalize(rn(B, 4), dim=-1), rn(B, 3)
    m = torch.full((B,), 32, device=device, dtype=torch.int32)
    victim = lambda: ops.ransac_score_maps(geo, rot_raw, trans_raw, init_rot, init_trans, m, diagnostics=False)
    ref = victim()
    torch.cuda.synchronize()
    bf = lambda *s: (torch.randn(*s, device=device) * 0.1).bfloat16()
    xb, res = bf(64, 60, 80, 128), bf(64, 60, 80, 512)
    w3, w1 = ops.mfma_fragment_major(bf(512, 128)), ops.mfma_fragment_major(bf(128, 512))
    s512, b512, s128, b128 = torch.ones(512, device=device), torch.zeros(512, device=device), torch.ones(128, device=device), torch.zeros(128, device=device)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    bad = 0
    for it in range(12):
        with torch.cuda.stream(sb):
            for _ in range(2):
                ops.bottleneck_tail(xb, w3, s512, b512, residual=res, w1=w1, s1=s128, b1=b128)
        with torch.cuda.stream(sa):
            outs = [victim() for _ in range(8)]
        torch.cuda.synchronize()
        bad += sum(1 for o in outs if not (torch.equal(o["normal_score"], ref["normal_score"]) and torch.equal(o["param_score"], ref["param_score"])))
    assert bad == 0, "%d of 96 launches differ from the idle-GPU result" % bad


def test_mask_head_shared_taps_equal_the_per_pixel_form(device):
    """Round 5: the bilinear phase of the fused mask head loads the eight taps of four consecutive output pixels once (half the tap loads);
    the result - probabilities and the optional p1 map - must equal the one-pixel-per-item form bit for bit, image borders included
    (first / last rows and columns: clamped taps), for several images and both plane paddings."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(21)
    for B, H, W, nq in ((2, 120, 160, 50), (1, 16, 32, 50), (1, 8, 64, 128)):
        c1 = (0.5 * torch.randn(B, H, W, 256, generator=g)).to(device, torch.bfloat16)
        t1 = (0.5 * torch.randn(B, H // 2, W // 2, 256, generator=g)).to(device, torch.bfloat16)
        wl = ops.mfma_fragment_major((torch.randn(256, 256, generator=g) / 16).to(device, torch.bfloat16))
        sc, bi = (1 + 0.1 * torch.randn(256, generator=g)).to(device), (0.1 * torch.randn(256, generator=g)).to(device)
        mw, mb = (torch.randn(B, nq, 256, generator=g) / 16).to(device), torch.randn(B, nq, generator=g).to(device)
        a, pa = ops.mask_head(c1, t1, wl, sc, bi, mw, mb, want_p1=True)
        b, pb = ops.mask_head(c1, t1, wl, sc, bi, mw, mb, want_p1=True, taps1=True)
        torch.cuda.synchronize()
        assert torch.equal(pa, pb) and torch.equal(a, b), (B, H, W, nq)


def test_mask_head_persistent_pipeline_equals_the_two_workgroup_form(device):
    """Round 5: the persistent, software-pipelined mask head (one workgroup per CU walking its XCD's run of tiles; c1 by LDS-DMA one tile
    ahead, taps prefetched under the lateral GEMM, descriptor-checked stores; off by default - it measured slower) must give the bits of
    the two-workgroups-per-CU kernel:
    fewer tiles than CUs, many tiles per CU (several images: the per-image mask operands change inside a workgroup's walk), tile counts
    that do not divide by 8 XCDs, both plane paddings, nq that leaves the last store vectors of a tile out of range, logits and p1."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(22)
    for B, H, W, nq, sig in ((1, 16, 32, 50, True), (3, 120, 160, 50, True), (5, 24, 48, 20, False), (2, 48, 64, 128, True), (9, 40, 32, 66, True),
                             (16, 120, 160, 50, True)):
        c1 = (0.5 * torch.randn(B, H, W, 256, generator=g)).to(device, torch.bfloat16)
        t1 = (0.5 * torch.randn(B, H // 2, W // 2, 256, generator=g)).to(device, torch.bfloat16)
        wl = ops.mfma_fragment_major((torch.randn(256, 256, generator=g) / 16).to(device, torch.bfloat16))
        sc, bi = (1 + 0.1 * torch.randn(256, generator=g)).to(device), (0.1 * torch.randn(256, generator=g)).to(device)
        mw, mb = (torch.randn(B, nq, 256, generator=g) / 16).to(device), torch.randn(B, nq, generator=g).to(device)
        for want_p1 in (False, True):
            a = ops.mask_head(c1, t1, wl, sc, bi, mw, mb, sigmoid=sig, want_p1=want_p1, pipe=True)
            b = ops.mask_head(c1, t1, wl, sc, bi, mw, mb, sigmoid=sig, want_p1=want_p1)
            torch.cuda.synchronize()
            if want_p1:
                assert torch.equal(a[1], b[1]), ("p1", B, H, W, nq)
                a, b = a[0], b[0]
            assert torch.equal(a, b), (B, H, W, nq, want_p1, (a != b).sum().item())
        for _ in range(3):                                            # back to back on one stream: a launch must not see its predecessor's LDS / tiles
            a2 = ops.mask_head(c1, t1, wl, sc, bi, mw, mb, sigmoid=sig, pipe=True)
        torch.cuda.synchronize()
        assert torch.equal(a2, b), (B, H, W, nq, "repeat")


def test_stream_set_places_streams_by_hardware_queue(device):
    """streams.StreamSet: the probe (a tiny kernel behind another stream's spin kernel) sorts candidate streams into hardware-queue
    classes; the batch streams it hands out are pairwise on DIFFERENT queues (as long as there are queues left), side stream i shares
    the queue of batch stream (i + side_shift) % n."""
    from nopesac_amd.streams import StreamSet, shares_queue
    for shift in (0, 1):
        ss = StreamSet(4, device, side_shift=shift)
        d = ss.describe()
        assert 2 <= d["queue_classes"] <= 16 and sum(d["class_sizes"]) == len(ss.pool)
        n_distinct = min(4, d["queue_classes"])
        assert len(set(d["batch_stream_class"][:n_distinct])) == n_distinct
        scratch = torch.zeros(4, device=device, dtype=torch.int64)
        torch.cuda.synchronize()
        for i in range(n_distinct):
            for j in range(i + 1, n_distinct):
                assert not shares_queue(ss.mains[i], ss.mains[j], scratch), (i, j)
        if d["queue_classes"] == 4:                                  # as many slots as queues: side i behind batch (i + shift) % 4
            for i in range(4):
                assert d["side_stream_class"][i] == d["batch_stream_class"][(i + shift) % 4]
                assert shares_queue(ss.mains[(i + shift) % 4], ss.sides[i], scratch)
    two = StreamSet(2, device)                                        # fewer slots than queues: the side streams get the idle queues
    d = two.describe()
    if d["queue_classes"] >= 4:
        assert len(set(d["batch_stream_class"] + d["side_stream_class"])) == 4


def _stream_set_under_load_worker(idx, q):
    import torch
    from nopesac_amd import runner
    from nopesac_amd.streams import StreamSet
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    load = torch.cuda.Stream(device=dev)
    a = torch.randn(2048, 2048, device=dev)
    stop = False
    with torch.cuda.stream(load):                       # this process's own background load; the OTHER processes are the disturbance
        for _ in range(200):
            a = (a @ a).clamp_(-1, 1)
    ss = StreamSet(4, dev)
    d = ss.describe()
    loop = runner.InflightLoop(4, 2, dev, 1, side_shift=0, gather_every=3)

    def device_step(slot):
        t = torch.full((2, 3), float(slot), device=dev)
        qv = torch.nn.functional.normalize(torch.ones(2, 4, device=dev), dim=-1)
        k = torch.full((2,), 5, dtype=torch.int32, device=dev)
        return None, runner.metric_rows(t, qv, k, k, k, 0)

    for i in range(10):
        loop.step(i, device_step)
    loop.barrier()
    rows = loop.last_step_rows()
    q.put((idx, d, len({id(s) for s in ss.mains}), float(rows[0, 0]), int(loop.collectives)))


def test_stream_set_under_load_from_other_processes(device):
    """Round-5 hardening (unmeasured on an 8-GPU node): three processes share THE ONE GPU - each keeps it busy and builds its StreamSet
    at the same time, so the queue probe (a timing observation) is disturbed.  Every process must end with four distinct batch streams:
    either the probe still found >= 4 queue classes, or it says `probe_inconclusive` and hands out the streams as the runtime placed
    them - never a set that forces several batches onto one observed class; the in-flight loop (gather every 3 steps) then runs."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stream_set_under_load_worker, args=(i, q)) for i in range(3)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for idx, d, distinct, first, collectives in got:
        assert distinct == 4, d
        assert d["probe_inconclusive"] or d["queue_classes"] >= 4, d
        if not d["probe_inconclusive"]:
            assert len(set(d["batch_stream_class"])) == 4, d
        assert first == 1.0 and collectives == 4          # 10 steps: 3 + 3 + 3 + the flushed last one (slot 9 % 4 = 1)


@pytest.mark.parametrize("rows,D,pad_to,dt", [(7, 300, 304, torch.bfloat16), (5, 300, 304, torch.float32), (9, 64, 64, torch.bfloat16),
                                              (3, 129, 192, torch.bfloat16), (4, 300, 0, torch.float32)])
def test_softmax_rows_padded_output(device, rows, D, pad_to, dt):
    """nopesac_softmax_rows_pad: the plain kernel's f32 values, rounded to the requested type, in rows of max(D, pad_to) elements
    with zeros behind column D (what camera_head.pixel_pose_net used to build with a zero fill and a strided cast copy)."""
    from nopesac_amd import ops
    torch.manual_seed(rows * D)
    x = torch.randn(rows, D, device=device) * 4
    ref = ops.softmax_rows(x)
    y = ops.softmax_rows(x, out_dtype=dt, pad_to=pad_to)
    ld = max(D, pad_to)
    assert y.shape == (rows, ld) and y.dtype == dt
    assert torch.equal(y[:, :D], ref.to(dt))
    assert ld == D or float(y[:, D:].float().abs().max()) == 0.0
    assert float((ref - torch.softmax(x, -1)).abs().max()) < 1e-6


def test_add_rows_bf16(device):
    """nopesac_add_rows_bf16 = (a.to(bf16), (a + b broadcast over row blocks).to(bf16)) exactly."""
    from nopesac_amd import ops
    torch.manual_seed(3)
    a = torch.randn(6 * 300, 256, device=device)
    b = torch.randn(300, 256, device=device)
    a16, ab16 = ops.add_rows_bf16(a, b)
    assert torch.equal(a16, a.to(torch.bfloat16))
    assert torch.equal(ab16, (a.view(6, 300, 256) + b).view(-1, 256).to(torch.bfloat16))


@pytest.mark.parametrize("B,nq", [(3, 50), (2, 64), (2, 100), (1, 128), (2, 6)])
def test_mask_operands_from_the_folded_embeddings(device, B, nq):
    """nopesac_mask_operands (the mask head's per-image weights in MFMA fragment order + bias, zero-padded planes) equals the torch
    formulation ops.mask_head used before (zero fill, strided cast copy, permuting copy)."""
    from nopesac_amd import _lib, ops
    torch.manual_seed(B * 1000 + nq)
    fold = torch.randn(B * nq, 264, device=device)
    nqp = 64 if nq <= 64 else 128
    mw = torch.empty(B, nqp // 32, 16, 2, 32, 8, device=device, dtype=torch.bfloat16)
    mb = torch.empty(B, nqp, device=device, dtype=torch.float32)
    _lib.check(_lib.load().nopesac_mask_operands(fold.data_ptr(), 264, mw.data_ptr(), mb.data_ptr(), B, nq, nqp, ops._stream()), "nopesac_mask_operands")
    ref_w = torch.zeros(B, nqp, 256, device=device, dtype=torch.bfloat16)
    ref_w[:, :nq] = fold[:, :256].view(B, nq, 256)
    ref_w = ref_w.view(B, nqp // 32, 32, 16, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous()
    ref_b = torch.zeros(B, nqp, device=device)
    ref_b[:, :nq] = fold[:, 256].view(B, nq)
    assert torch.equal(mw, ref_w) and torch.equal(mb, ref_b)


@pytest.mark.parametrize("M", [19200 // 8, 333])
def test_transformer_tail_with_chained_projections(device, M):
    """nopesac_transformer_tail_bf16: (a) its tail equals the encoder / decoder tail entries bit for bit, (b) the chained projections
    equal ops.linear on the tail's bf16 outputs (same bf16 operands, same MFMA k order: bit for bit), (c) skip_ffn = the decoder's
    self-attention half (out-proj + residual, norm, cross-q projection) against the per-op kernels."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(M + 7)
    rn = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(device)
    bf = torch.bfloat16
    attn, src, pos = rn(M, 256).bfloat16(), rn(M, 256), rn(300, 256)
    wo, w1, w2 = rn(256, 256, k=1 / 16).bfloat16(), rn(1024, 256, k=1 / 16).bfloat16(), rn(256, 1024, k=1 / 32).bfloat16()
    bo, b1, b2 = rn(256, k=0.1), rn(1024, k=0.1), rn(256, k=0.1)
    ga, bea, gb, beb = 1 + rn(256, k=0.1), rn(256, k=0.1), 1 + rn(256, k=0.1), rn(256, k=0.1)
    wqk, bqk, wv, bv = rn(512, 256, k=1 / 16).bfloat16(), rn(512, k=0.1), rn(256, 256, k=1 / 16).bfloat16(), rn(256, k=0.1)
    fm = ops.mfma_fragment_major
    W = {"wo": fm(wo), "w1": fm(w1), "w2": fm(w2), "bo": bo, "b1": b1, "b2": b2, "ga": ga, "bea": bea, "gb": gb, "beb": beb}
    # (a) + (b), encoder form
    ref = ops.encoder_tail(attn, src, dict(W, g1=ga, be1=bea, g2=gb, be2=beb), pos=pos)
    out = ops.transformer_tail(attn, src, W, pre_norm=False, pos=pos, want=("y", "y16", "ypos16"),
                               proj_pos=(fm(wqk), bqk, 512), proj=(fm(wv), bv, 256))
    assert torch.equal(out["y"], ref["y"]) and torch.equal(out["y16"], ref["y16"]) and torch.equal(out["ypos16"], ref["ypos16"])
    qk = ops.linear(ref["ypos16"], wqk, bqk, out_dtype=bf)
    v = ops.linear(ref["y16"], wv, bv, out_dtype=bf)
    assert _rel(out["proj_pos"].float(), qk.float()) < 1e-2 and _rel(out["proj"].float(), v.float()) < 1e-2
    assert float((out["proj_pos"].float() - qk.float()).abs().mean()) < 1e-3 * float(qk.float().abs().mean() + 1)
    # decoder form (pre-norm), projections only (no y16 / ypos16 written)
    ref = ops.decoder_tail(attn, src, dict(W, g3=ga, be3=bea, gn=gb, ben=beb), pos=pos, want=("y", "y16", "ypos16", "yn"))
    out = ops.transformer_tail(attn, src, W, pre_norm=True, pos=pos, want=("y", "yn"), proj_pos=(fm(wqk), bqk, 512), proj=(fm(wv), None, 256))
    assert torch.equal(out["y"], ref["y"]) and torch.equal(out["yn"], ref["yn"])
    assert _rel(out["proj_pos"].float(), ops.linear(ref["ypos16"], wqk, bqk, out_dtype=bf).float()) < 1e-2
    assert _rel(out["proj"].float(), ops.linear(ref["y16"], wv, None, out_dtype=bf).float()) < 1e-2
    # (c) the self-attention half: s = src + out_proj(attn), n = LN(s), q = (n + pos) Wq
    out = ops.transformer_tail(attn, src, {"wo": fm(wo), "bo": bo, "ga": ga, "bea": bea}, pre_norm=True, skip_ffn=True, pos=pos,
                               want=("y",), proj_pos=(fm(wv), bv, 256))
    s = ops.linear(attn, wo, bo, residual=src, out_dtype=torch.float32)
    r = ops.layernorm_ex(s, ga, bea, addend=pos, want=("y2_16",))
    q = ops.linear(r["y2_16"], wv, bv, out_dtype=bf)
    assert _rel(out["y"], s) < 1e-5
    assert _rel(out["proj_pos"].float(), q.float()) < 1e-2
    assert float((out["proj_pos"].float() - q.float()).abs().mean()) < 2e-3 * float(q.float().abs().mean() + 1)


@pytest.mark.parametrize("M", [300, 77, 2048])
def test_transformer_tail_next_weight_prefetch_changes_nothing(device, M):
    """nopesac_transformer_tail_bf16_pf: the extra workgroups that read the NEXT tail's matrices into L2 (one pair per call: few row
    tiles) leave every output bit for bit what the plain launch writes; tensors whose size is no multiple of the prefetch chunk and a
    list longer than the entry's 8 slots are accepted."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(M)
    rn = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(device)
    attn, src, pos = rn(M, 256).bfloat16(), rn(M, 256), rn(M, 256)
    fm = ops.mfma_fragment_major
    W = {"wo": fm(rn(256, 256, k=1 / 16).bfloat16()), "w1": fm(rn(1024, 256, k=1 / 16).bfloat16()), "w2": fm(rn(256, 1024, k=1 / 32).bfloat16()),
         "bo": rn(256, k=0.1), "b1": rn(1024, k=0.1), "b2": rn(256, k=0.1), "ga": 1 + rn(256, k=0.1), "bea": rn(256, k=0.1),
         "gb": 1 + rn(256, k=0.1), "beb": rn(256, k=0.1)}
    pp, pj = (fm(rn(512, 256, k=1 / 16).bfloat16()), rn(512, k=0.1), 512), (fm(rn(256, 256, k=1 / 16).bfloat16()), None, 256)
    kw = dict(pre_norm=False, pos=pos, want=("y", "y16", "ypos16"), proj_pos=pp, proj=pj)
    ref = ops.transformer_tail(attn, src, W, **kw)
    nxt = [rn(1024, 256).bfloat16(), rn(256, 1024).bfloat16(), None, rn(4099).bfloat16(), rn(3), rn(256, 256).bfloat16()] + \
          [rn(512, 256).bfloat16() for _ in range(5)]
    for wg in (1, 10, 64):
        out = ops.transformer_tail(attn, src, W, prefetch=(nxt, wg), **kw)
        assert all(torch.equal(out[k], ref[k]) for k in ref)
    torch.cuda.synchronize()


def test_metric_rows_kernel_equals_the_torch_formulation(device):
    """runner.metric_rows on the GPU (nopesac_metric_rows, one launch) = the torch formulation the CPU / gloo tests use."""
    from nopesac_amd import runner
    torch.manual_seed(5)
    B = 37
    t, q = torch.randn(B, 3, device=device), torch.randn(B, 4, device=device)
    n1, n2, m = (torch.randint(0, 50, (B,), device=device, dtype=torch.int32) for _ in range(3))
    te, re = torch.rand(B, device=device), torch.rand(B, device=device)
    nf = torch.tensor([3], device=device, dtype=torch.int32)
    for kw in ({}, {"t_err": te, "r_err": re}, {"nonfinite": nf}, {"t_err": te, "r_err": re, "nonfinite": nf}):
        got = runner.metric_rows(t, q, n1, n2, m, 96, **kw)
        ref = runner.metric_rows(t.cpu(), q.cpu(), n1.cpu(), n2.cpu(), m.cpu(), 96, **{k: v.cpu() for k, v in kw.items()})
        assert got.is_cuda and torch.equal(got.cpu(), ref)


def test_mlp_chain_parallel_stacks_over_the_same_rows(device):
    """NOPESAC_MLP_RESTART: stacks that all read the chain input run in ONE launch and give exactly what one launch per stack gives
    (the plane head's embedding -> fold chain plus its prob / param / center heads; odd and even restart positions, a ragged row tile)."""
    from nopesac_amd import ops
    from nopesac_amd.modeling.params import ConvW
    from nopesac_amd.modeling.plane_head import run_mlp, run_stacks
    torch.manual_seed(11)
    rows, bf = 50 * 7 + 3, torch.bfloat16
    x = torch.randn(rows, 256, device=device)
    mk = lambda n, k: ConvW((torch.randn(n, k, device=device) / k ** 0.5), None, torch.randn(n, device=device) * 0.1)
    emb = [mk(256, 256), mk(256, 256), mk(256, 256)]
    fold = [mk(264, 256)]
    prob = [mk(2, 256)]
    param = [mk(256, 256), mk(256, 256), mk(3, 256)]
    center = [mk(256, 256), mk(2, 256)]
    r = run_stacks(x, [(emb, ops.ACT_NONE, None), (fold, ops.ACT_NONE, True), (prob, ops.ACT_NONE, True), (param, ops.ACT_NONE, True),
                       (center, ops.ACT_SIGMOID, True)], bf, parallel=[False, False, True, True, True])
    ref_fold = run_stacks(x, [(emb, ops.ACT_NONE, None), (fold, ops.ACT_NONE, True)], bf)[1]
    assert r[0] is None and torch.equal(r[1], ref_fold)
    assert torch.equal(r[2], run_mlp(x, prob, gd=bf))
    assert torch.equal(r[3], run_mlp(x, param, gd=bf))
    assert torch.equal(r[4], run_mlp(x, center, final_act=ops.ACT_SIGMOID, gd=bf))
    # every stack parallel (restart on an odd layer index too)
    r = run_stacks(x, [(prob, ops.ACT_NONE, True), (param, ops.ACT_NONE, True), (center, ops.ACT_SIGMOID, True)], bf, parallel=True)
    assert torch.equal(r[0], run_mlp(x, prob, gd=bf)) and torch.equal(r[1], run_mlp(x, param, gd=bf))
    assert torch.equal(r[2], run_mlp(x, center, final_act=ops.ACT_SIGMOID, gd=bf))


@pytest.mark.parametrize("M", [2048, 19200 // 4 + 17])
def test_encoder_tail_64_token_kernel_equals_the_32_token_one(device, M, monkeypatch):
    """enc_tail64_kernel (64 tokens per workgroup, K = 128 weight steps, hidden tile in two halves) computes the same sums in the same
    order as enc_tail_kernel: bit-identical outputs and chained projections (ragged last tile included)."""
    monkeypatch.setenv("NOPESAC_ENC_TAIL_64", "1")
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(M)
    rn = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(device)
    attn, src, pos = rn(M, 256).bfloat16(), rn(M, 256), rn(300, 256)
    fm = ops.mfma_fragment_major
    W = {"wo": fm(rn(256, 256, k=1 / 16).bfloat16()), "w1": fm(rn(1024, 256, k=1 / 16).bfloat16()), "w2": fm(rn(256, 1024, k=1 / 32).bfloat16()),
         "bo": rn(256, k=0.1), "b1": rn(1024, k=0.1), "b2": rn(256, k=0.1), "ga": 1 + rn(256, k=0.1), "bea": rn(256, k=0.1),
         "gb": 1 + rn(256, k=0.1), "beb": rn(256, k=0.1)}
    pp = (fm(rn(512, 256, k=1 / 16).bfloat16()), rn(512, k=0.1), 512)
    pj = (fm(rn(256, 256, k=1 / 16).bfloat16()), None, 256)
    run = lambda: ops.transformer_tail(attn, src, W, pre_norm=False, pos=pos, want=("y", "y16", "ypos16"), proj_pos=pp, proj=pj)
    new = run()
    monkeypatch.setenv("NOPESAC_ENC_TAIL_32", "1")
    old = run()
    torch.cuda.synchronize()
    for k in ("y", "y16", "ypos16", "proj_pos", "proj"):
        assert torch.equal(new[k], old[k]), k


@pytest.mark.parametrize("M", [2048, 19200 // 4 + 17, 19200 // 2 + 97, 2048 + 1])
@pytest.mark.parametrize("outputs", ["all", "y_and_projections", "no_projections"])
@pytest.mark.parametrize("rows", [3, 4])
def test_encoder_tail_128_token_kernel(device, M, outputs, rows, monkeypatch):
    """enc_tail128_kernel (round 6: 96 / 128 tokens per workgroup, every weight fragment feeds three / four MFMAs, linear2 accumulating on
    top of y1 + b2, hidden tile in four quarters, every output staged through LDS and stored as whole rows) against the 32-token kernel
    on the same inputs.  Same operands and the same bf16 rounding points; some f32 sums run in a different order ((y1 + b2) + sum_k, fused
    multiply-adds in the LayerNorms), so the comparison is a tolerance, and the ragged last tile must leave the rows behind M untouched."""
    from nopesac_amd import ops
    monkeypatch.setenv("NOPESAC_ENC_TAIL_ROWS", str(rows))
    g = torch.Generator().manual_seed(M)
    rn = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(device)
    attn, src, pos = rn(M, 256).bfloat16(), rn(M, 256), rn(300, 256)
    fm = ops.mfma_fragment_major
    W = {"wo": fm(rn(256, 256, k=1 / 16).bfloat16()), "w1": fm(rn(1024, 256, k=1 / 16).bfloat16()), "w2": fm(rn(256, 1024, k=1 / 32).bfloat16()),
         "bo": rn(256, k=0.1), "b1": rn(1024, k=0.1), "b2": rn(256, k=0.1), "ga": 1 + rn(256, k=0.1), "bea": rn(256, k=0.1),
         "gb": 1 + rn(256, k=0.1), "beb": rn(256, k=0.1)}
    pp = (fm(rn(512, 256, k=1 / 16).bfloat16()), rn(512, k=0.1), 512)
    pj = (fm(rn(256, 256, k=1 / 16).bfloat16()), None, 256)
    kw = {"all": dict(want=("y", "y16", "ypos16"), proj_pos=pp, proj=pj), "y_and_projections": dict(want=("y",), proj_pos=pp, proj=pj),
          "no_projections": dict(want=("y", "y16", "ypos16"))}[outputs]
    run = lambda: ops.transformer_tail(attn, src, W, pre_norm=False, pos=pos, **kw)
    new = run()
    again = run()
    monkeypatch.setenv("NOPESAC_ENC_TAIL_32", "1")
    old = run()
    torch.cuda.synchronize()
    assert set(new) == set(old)
    for k in new:
        a, b = new[k].float(), old[k].float()
        assert a.shape == b.shape and torch.equal(new[k], again[k]), k            # (deterministic run to run)
        # a 1-ulp flip of a bf16 intermediate (LN1 output, hidden tile) moves an output by ~1e-3 of its scale: the bound is statistical
        # (relative l2 distance, mean absolute difference) plus a cap on the largest single difference
        d = (a - b).abs()
        scale = float(b.abs().mean()) + 1e-6
        assert _rel(a, b) < (5e-3 if new[k].dtype == torch.float32 else 2.0 ** -6), (k, _rel(a, b))          # (bf16: two ulps of the largest value)
        assert float(d.mean()) < 3e-4 * scale and float(d.max()) < 0.05 * max(1.0, float(b.abs().max())), (k, float(d.mean()) / scale, float(d.max()))
    monkeypatch.delenv("NOPESAC_ENC_TAIL_32")
    if M % 128 and outputs == "all":                                               # rows behind M: never written (raw C ABI, padded buffers)
        from nopesac_amd.ops import _L, _p, _stream
        pad = {"y": torch.full((M + 128, 256), 7.0, device=device), "y16": torch.full((M + 128, 256), 7.0, device=device).bfloat16(),
               "ypos16": torch.full((M + 128, 256), 7.0, device=device).bfloat16(), "pp": torch.full((M + 128, 512), 7.0, device=device).bfloat16(),
               "pj": torch.full((M + 128, 256), 7.0, device=device).bfloat16()}
        rc = _L().nopesac_transformer_tail_bf16_pf(
            _p(attn), _p(src), _p(W["wo"]), _p(W["bo"]), _p(W["ga"]), _p(W["bea"]), _p(W["w1"]), _p(W["b1"]), _p(W["w2"]), _p(W["b2"]), _p(W["gb"]),
            _p(W["beb"]), _p(pos), pos.shape[0], _p(pad["y"]), _p(pad["y16"]), _p(pad["ypos16"]), None, 0, 0, _p(pp[0]), _p(pp[1]), _p(pad["pp"]), 512,
            _p(pj[0]), None, _p(pad["pj"]), 256, M, None, None, 0, 0, _stream())
        assert rc == 0
        torch.cuda.synchronize()
        for k, t in pad.items():
            assert bool((t[M:].float() == 7.0).all()), k
        assert torch.equal(pad["y"][:M], new["y"]) and torch.equal(pad["pp"][:M], new["proj_pos"])


def test_host_fetch_gather_kernel(device):
    """ops.HostFetch through nopesac_gather_bytes: unaligned views, empty tensors, more than 32 segments (two launches), a segment whose
    valid length only the device knows, a caller-provided pinned buffer, snapshot views."""
    from nopesac_amd import ops
    g = torch.Generator().manual_seed(3)
    base = torch.randint(0, 255, (5000,), generator=g, dtype=torch.uint8).to(device)
    tensors = {"a": torch.randn(7, 3, generator=g).to(device), "odd": base[3:1000], "empty": torch.zeros(0, 4, device=device),
               "i64": torch.arange(11, device=device), "b": torch.rand(5, generator=g).to(device) > 0.5,
               "f64": torch.randn(4, 4, generator=g, dtype=torch.float64).to(device), "host": torch.arange(3)}
    for i in range(40):
        tensors["t%d" % i] = torch.full((i + 1,), float(i), device=device, dtype=torch.bfloat16)
    big = torch.randint(0, 255, (300000,), generator=g, dtype=torch.uint8).to(device)
    tensors["dyn"] = big
    n_valid = torch.tensor([123457], device=device, dtype=torch.int64)
    f = ops.HostFetch(tensors, dynamic={"dyn": n_valid})
    h = f.wait().views()
    for k, t in tensors.items():
        if k == "dyn":
            assert torch.equal(h[k][:123457], t.cpu()[:123457])
        else:
            assert h[k].dtype == t.dtype and h[k].shape == t.shape and torch.equal(h[k], t.cpu()), k
    # caller-provided buffer + snapshots: a second fetch into the same buffer must not change views already handed out
    buf = torch.empty(f.host_bytes() + 64, dtype=torch.uint8, pin_memory=True)
    f1 = ops.HostFetch({"x": tensors["a"]}, private_views=True, host=buf)
    v1 = f1.wait().views()["x"]
    f2 = ops.HostFetch({"x": tensors["a"] * 2}, private_views=True, host=buf)
    v2 = f2.wait().views()["x"]
    assert torch.equal(v1, tensors["a"].cpu()) and torch.equal(v2, (tensors["a"] * 2).cpu())
    with pytest.raises(Exception):
        ops.HostFetch({"x": tensors["a"]}, host=torch.empty(4, dtype=torch.uint8, pin_memory=True))
