"""Each HIP kernel, called through the C ABI, against a plain fp32 torch computation on the CPU
of the same op on the same bf16-rounded inputs.  Tolerance: bf16 output rounding (2^-8 relative)
on top of fp32 accumulation, checked as max abs error <= 1e-2 * max|ref| (stated per test)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

HW = 196


def dev():
    return torch.device('cuda:0')


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale)


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def to_slots(x_nchw):
    """[S, C, 14, 14] fp32 -> bf16 [S, 196, C] device tensor (slot layout)."""
    s, c = x_nchw.shape[:2]
    return x_nchw.reshape(s, c, HW).permute(0, 2, 1).contiguous().to(torch.bfloat16).to(dev())


def from_slots(t, c):
    s = t.shape[0]
    return t.to(torch.float32).cpu().reshape(s, HW, c).permute(0, 2, 1).reshape(s, c, 14, 14)


def tiles_tensor(rows):
    return torch.tensor(rows, dtype=torch.int32).reshape(-1).to(dev())


def close(got, ref, tol=1e-2):
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= tol * scale + 1e-6, 'max err %g vs scale %g' % (err, scale)


def test_conv3x3_bias_res_relu_grouped():
    from vqa_amd import ops
    C, G, S = 128, 3, 7
    x = bf(rnd((S, C, 14, 14), 1))
    res = bf(rnd((S, C, 14, 14), 2))
    w = bf(rnd((G, C, C, 3, 3), 3, 0.05))
    b = rnd((G, C), 4, 0.1)
    groups = [0, 2, 1, 1, 0, 2, 2]
    arena = torch.cat([to_slots(x), to_slots(res), torch.zeros(S, HW, C, dtype=torch.bfloat16, device=dev())])
    tiles = tiles_tensor([[i, -1, 2 * S + i, -1, S + i, -1, -1, groups[i]] for i in range(S)])
    wp = ops.pack_conv_weight(w.to(dev()), G, C, C, 9, C * C * 9, C * 9, 9)
    ops.conv_tiles(arena, arena, wp, tiles, S, taps=9, k_chunks=1, n_out=C, src_slot_stride=HW * C,
                   dst_slot_stride=HW * C, src_pix_stride=C, dst_pix_stride=C, w_group_stride=9 * C * C,
                   bias=b.to(dev()), res=arena, relu=True)
    torch.cuda.synchronize()
    got = from_slots(arena[2 * S:], C)
    ref = torch.stack([F.relu(F.conv2d(x[i:i + 1], w[groups[i]], b[groups[i]], padding=1)[0] + res[i]) for i in range(S)])
    close(got, ref)


def test_conv3x3_multichunk_strided_src_wide_out():
    """stem-like: 256 input channels read from a wider channels-last buffer, 256 outputs."""
    from vqa_amd import ops
    Cin, Cout, S = 256, 256, 3
    x = bf(rnd((S, Cin, 14, 14), 5))
    w = bf(rnd((1, Cout, Cin, 3, 3), 6, 0.03))
    b = rnd((1, Cout), 7, 0.1)
    src = to_slots(x)
    dst = torch.zeros(S, HW, Cout, dtype=torch.bfloat16, device=dev())
    tiles = tiles_tensor([[i, -1, i, -1, -1, -1, -1, 0] for i in range(S)])
    wp = ops.pack_conv_weight(w.to(dev()), 1, Cout, Cin, 9, 0, Cin * 9, 9)
    ops.conv_tiles(src, dst, wp, tiles, S, taps=9, k_chunks=2, n_out=Cout, src_slot_stride=HW * Cin,
                   dst_slot_stride=HW * Cout, src_pix_stride=Cin, dst_pix_stride=Cout,
                   w_group_stride=9 * Cin * Cout, bias=b.to(dev()), relu=True)
    torch.cuda.synchronize()
    ref = F.relu(F.conv2d(x, w[0], b[0], padding=1))
    close(from_slots(dst, Cout), ref)


def test_conv1x1_pair_sources():
    """ConcatBlock.proj: 1x1 conv over the channel concatenation of two maps, no cat copy."""
    from vqa_amd import ops
    C, S, G = 128, 4, 2
    x0, x1 = bf(rnd((S, C, 14, 14), 8)), bf(rnd((S, C, 14, 14), 9))
    w = bf(rnd((G, C, 2 * C, 1, 1), 10, 0.06))
    b = rnd((G, C), 11, 0.1)
    arena = torch.cat([to_slots(x0), to_slots(x1), torch.zeros(S, HW, C, dtype=torch.bfloat16, device=dev())])
    tiles = tiles_tensor([[i, S + i, 2 * S + i, -1, -1, -1, -1, i % G] for i in range(S)])
    wp = ops.pack_conv_weight(w.to(dev()), G, C, 2 * C, 1, 2 * C * C, 2 * C, 1)
    ops.conv_tiles(arena, arena, wp, tiles, S, taps=1, k_chunks=2, n_out=C, src_slot_stride=HW * C,
                   dst_slot_stride=HW * C, src_pix_stride=C, dst_pix_stride=C, w_group_stride=2 * C * C,
                   bias=b.to(dev()), pair_chunks=1, relu=True)
    torch.cuda.synchronize()
    ref = torch.stack([F.relu(F.conv2d(torch.cat([x0[i:i + 1], x1[i:i + 1]], 1), w[i % G], b[i % G]))[0] for i in range(S)])
    close(from_slots(arena[2 * S:], C), ref)


def test_conv_dgrad_with_mask_and_residual():
    """Input gradient of a 3x3 conv = the same kernel on flipped/transposed weights; epilogue
    (acc + res) * (mask > 0)."""
    from vqa_amd import ops
    C, S = 128, 3
    dy = bf(rnd((S, C, 14, 14), 12))
    res = bf(rnd((S, C, 14, 14), 13))
    act = bf(F.relu(rnd((S, C, 14, 14), 14)))
    w = bf(rnd((1, C, C, 3, 3), 15, 0.05))
    arena = torch.cat([to_slots(dy), to_slots(res), to_slots(act), torch.zeros(S, HW, C, dtype=torch.bfloat16, device=dev())])
    tiles = tiles_tensor([[i, -1, 3 * S + i, -1, S + i, 2 * S + i, -1, 0] for i in range(S)])
    # dgrad weights: n = original input channel, k = original output channel, taps flipped
    wd = ops.pack_conv_weight(w.to(dev()), 1, C, C, 9, 0, 9, C * 9, flip=True)
    ops.conv_tiles(arena, arena, wd, tiles, S, taps=9, k_chunks=1, n_out=C, src_slot_stride=HW * C,
                   dst_slot_stride=HW * C, src_pix_stride=C, dst_pix_stride=C, w_group_stride=9 * C * C,
                   res=arena, mask=arena)
    torch.cuda.synchronize()
    ref = (F.conv_transpose2d(dy, w[0], padding=1) + res) * (act > 0)
    close(from_slots(arena[3 * S:], C), ref)


@pytest.mark.parametrize('backward', [False, True])
def test_resblock_fused_matches_two_launches_and_torch(backward):
    """vqa_resblock_tiles (both 3x3 convolutions of a residual body in one workgroup) gives the same
    bits as two vqa_conv_tiles launches, and both match torch to bf16 rounding.  The backward
    form masks the intermediate map by a stored activation and the result by another."""
    from vqa_amd import ops
    C, G, S = 128, 3, 9
    x = bf(rnd((S, C, 14, 14), 41))
    w1 = bf(rnd((G, C, C, 3, 3), 42, 0.04))
    w2 = bf(rnd((G, C, C, 3, 3), 43, 0.04))
    b1, b2 = rnd((G, C), 44, 0.1), rnd((G, C), 45, 0.1)
    hmask = bf(F.relu(rnd((S, C, 14, 14), 46)))
    omask = bf(F.relu(rnd((S, C, 14, 14), 47)))
    groups = [0, 2, 1, 1, 0, 2, 2, 0, 1]
    zeros = torch.zeros(S, HW, C, dtype=torch.bfloat16, device=dev())
    # slots: x | hmask | omask | mid(fused) | out(fused) | mid(two) | out(two)
    arena = torch.cat([to_slots(x), to_slots(hmask), to_slots(omask), zeros, zeros, zeros, zeros])
    if backward:
        p1 = ops.pack_conv_weight(w1.to(dev()), G, C, C, 9, C * C * 9, 9, C * 9, flip=True)
        p2 = ops.pack_conv_weight(w2.to(dev()), G, C, C, 9, C * C * 9, 9, C * 9, flip=True)
        bias1 = bias2 = None
        m1 = lambda i: S + i
        m0 = lambda i: (2 * S + i) if i % 3 else -1       # some tiles without a final mask
    else:
        p1 = ops.pack_conv_weight(w1.to(dev()), G, C, C, 9, C * C * 9, C * 9, 9)
        p2 = ops.pack_conv_weight(w2.to(dev()), G, C, C, 9, C * C * 9, C * 9, 9)
        bias1, bias2 = b1.to(dev()), b2.to(dev())
        m1 = m0 = lambda i: -1
    fused = tiles_tensor([[i, -1, 4 * S + i, 3 * S + i, i, m0(i), m1(i), groups[i], -1, -1, -1, -1] for i in range(S)])
    ops.resblock_tiles(arena, p1, p2, fused, S, bias_first=bias1, bias_second=bias2, relu=not backward)
    first = tiles_tensor([[i, -1, 5 * S + i, -1, -1, m1(i), -1, groups[i]] for i in range(S)])
    second = tiles_tensor([[5 * S + i, -1, 6 * S + i, -1, i, m0(i), -1, groups[i]] for i in range(S)])
    geo = dict(taps=9, k_chunks=1, n_out=C, src_slot_stride=HW * C, dst_slot_stride=HW * C,
               src_pix_stride=C, dst_pix_stride=C, w_group_stride=9 * C * C, relu=not backward)
    ops.conv_tiles(arena, arena, p1, first, S, bias=bias1, mask=arena, **geo)
    ops.conv_tiles(arena, arena, p2, second, S, bias=bias2, res=arena, mask=arena, **geo)
    torch.cuda.synchronize()
    assert torch.equal(arena[3 * S:4 * S], arena[5 * S:6 * S])
    assert torch.equal(arena[4 * S:5 * S], arena[6 * S:7 * S])
    mid_ref, out_ref = [], []
    for i in range(S):
        g = groups[i]
        if backward:
            mid = bf(F.conv_transpose2d(x[i:i + 1], w1[g], padding=1)[0] * (hmask[i] > 0))
            out = F.conv_transpose2d(mid[None], w2[g], padding=1)[0] + x[i]
            if m0(i) >= 0:
                out = out * (omask[i] > 0)
        else:
            mid = bf(F.relu(F.conv2d(x[i:i + 1], w1[g], b1[g], padding=1)[0]))
            out = F.relu(F.conv2d(mid[None], w2[g], b2[g], padding=1)[0] + x[i])
        mid_ref.append(mid)
        out_ref.append(out)
    close(from_slots(arena[3 * S:4 * S], C), torch.stack(mid_ref))
    close(from_slots(arena[4 * S:5 * S], C), torch.stack(out_ref), tol=2e-2)


@pytest.mark.parametrize('backward', [False, True])
def test_resblock_two_image_form_equals_one_image_form(backward):
    """A launch of at most one workgroup per CU runs the two-image form of the fused kernel (second halo
    image, residual read from LDS, conflict-free epilogue writes), a larger launch the one-image form: the
    same nodes must come out bit for bit the same either way, a node's level size being an accident of the
    batch.  300 tiles = 6 distinct nodes repeated, against the same 6 nodes alone."""
    from vqa_amd import ops
    C, G, S, R = 128, 2, 6, 50
    x = bf(rnd((S, C, 14, 14), 71))
    w1 = bf(rnd((G, C, C, 3, 3), 72, 0.04))
    w2 = bf(rnd((G, C, C, 3, 3), 73, 0.04))
    b1, b2 = rnd((G, C), 74, 0.1), rnd((G, C), 75, 0.1)
    hmask, omask = bf(F.relu(rnd((S, C, 14, 14), 76))), bf(F.relu(rnd((S, C, 14, 14), 77)))
    zeros = lambda k: torch.zeros(k, HW, C, dtype=torch.bfloat16, device=dev())
    # slots: x | hmask | omask | mid, out of the small launch | mid, out of every repetition of the large one
    arena = torch.cat([to_slots(x), to_slots(hmask), to_slots(omask), zeros(2 * S), zeros(2 * S * R)])
    flip = dict(flip=True) if backward else {}
    order = (9, C * 9) if backward else (C * 9, 9)
    p1 = ops.pack_conv_weight(w1.to(dev()), G, C, C, 9, C * C * 9, *order, **flip)
    p2 = ops.pack_conv_weight(w2.to(dev()), G, C, C, 9, C * C * 9, *order, **flip)
    kw = dict(relu=False) if backward else dict(bias_first=b1.to(dev()), bias_second=b2.to(dev()), relu=True)
    m1 = (lambda i: S + i) if backward else (lambda i: -1)
    m0 = (lambda i: 2 * S + i if i % 2 else -1) if backward else (lambda i: -1)
    row = lambda i, mid, out: [i, -1, out, mid, i, m0(i), m1(i), i % G, -1, -1, -1, -1]
    small = tiles_tensor([row(i, 3 * S + i, 4 * S + i) for i in range(S)])
    ops.resblock_tiles(arena, p1, p2, small, S, **kw)
    base = 5 * S
    large = tiles_tensor([row(i, base + (2 * r) * S + i, base + (2 * r + 1) * S + i) for r in range(R) for i in range(S)])
    ops.resblock_tiles(arena, p1, p2, large, S * R, **kw)
    torch.cuda.synchronize()
    assert arena[4 * S:5 * S].float().abs().max() > 0
    for r in range(R):
        assert torch.equal(arena[base + 2 * r * S:base + (2 * r + 1) * S], arena[3 * S:4 * S]), r
        assert torch.equal(arena[base + (2 * r + 1) * S:base + (2 * r + 2) * S], arena[4 * S:5 * S]), r


def test_resblock_fused_with_projection_pre_and_post():
    """Binary nodes: the fused kernel also runs the 1x1 projection of the two children in front of the
    node's body (forward) and, backward, each child's half of the transposed projection in front of
    the child's body; compared bit for bit with the separate launches."""
    from vqa_amd import ops
    C, G, S = 128, 2, 5
    x0, x1 = bf(F.relu(rnd((S, C, 14, 14), 51))), bf(F.relu(rnd((S, C, 14, 14), 52)))
    w1 = bf(rnd((G, C, C, 3, 3), 53, 0.04))
    w2 = bf(rnd((G, C, C, 3, 3), 54, 0.04))
    wp = bf(rnd((G, C, 2 * C, 1, 1), 55, 0.06))
    b1, b2, bp = rnd((G, C), 56, 0.1), rnd((G, C), 57, 0.1), rnd((G, C), 58, 0.1)
    groups = [0, 1, 1, 0, 1]
    zeros = lambda k: torch.zeros(k * S, HW, C, dtype=torch.bfloat16, device=dev())
    d = dev()
    geo = dict(k_chunks=1, n_out=C, src_slot_stride=HW * C, dst_slot_stride=HW * C, src_pix_stride=C,
               dst_pix_stride=C)
    # ---- forward.  slots: x0 | x1 | P H Y (fused) | P H Y (separate)
    arena = torch.cat([to_slots(x0), to_slots(x1), zeros(6)])
    p1 = ops.pack_conv_weight(w1.to(d), G, C, C, 9, C * C * 9, C * 9, 9)
    p2 = ops.pack_conv_weight(w2.to(d), G, C, C, 9, C * C * 9, C * 9, 9)
    pp = ops.pack_conv_weight(wp.to(d), G, C, 2 * C, 1, 2 * C * C, 2 * C, 1)
    fused = tiles_tensor([[i, S + i, 4 * S + i, 3 * S + i, 2 * S + i, -1, -1, groups[i], -1, -1, -1, -1] for i in range(S)])
    ops.resblock_tiles(arena, p1, p2, fused, S, bias_first=b1.to(d), bias_second=b2.to(d), relu=True,
                       w_pre=pp, bias_pre=bp.to(d), num_pre=S)
    t_proj = tiles_tensor([[i, S + i, 5 * S + i, -1, -1, -1, -1, groups[i]] for i in range(S)])
    t_c1 = tiles_tensor([[5 * S + i, -1, 6 * S + i, -1, -1, -1, -1, groups[i]] for i in range(S)])
    t_c2 = tiles_tensor([[6 * S + i, -1, 7 * S + i, -1, 5 * S + i, -1, -1, groups[i]] for i in range(S)])
    ops.conv_tiles(arena, arena, pp, t_proj, S, taps=1, w_group_stride=2 * C * C, bias=bp.to(d), pair_chunks=1,
                   relu=True, **dict(geo, k_chunks=2))
    ops.conv_tiles(arena, arena, p1, t_c1, S, taps=9, w_group_stride=9 * C * C, bias=b1.to(d), relu=True, **geo)
    ops.conv_tiles(arena, arena, p2, t_c2, S, taps=9, w_group_stride=9 * C * C, bias=b2.to(d), res=arena, relu=True, **geo)
    torch.cuda.synchronize()
    assert torch.equal(arena[2 * S:5 * S], arena[5 * S:8 * S])
    P = from_slots(arena[2 * S:3 * S], C)
    ref_p = torch.stack([F.relu(F.conv2d(torch.cat([x0[i:i + 1], x1[i:i + 1]], 1), wp[groups[i]], bp[groups[i]]))[0] for i in range(S)])
    close(P, ref_p)
    assert from_slots(arena[4 * S:5 * S], C).abs().max() > 0
    # ---- backward.  A binary node's tile ends with dP; each of its two inputs starts its own tile with
    # its half of the transposed projection of that dP, masked by its own output (its dZ), and runs its
    # body on it.  slots: dP | Y0 Y1 (the inputs' outputs) | H0 H1 | X0 X1 (masks of the inputs' bodies) |
    # fused: dZ0 dZ1 dH0 dH1 dX0 dX1 | separate: dZ0 dZ1 dH0 dH1 dX0 dX1
    dpar = bf(rnd((S, C, 14, 14), 59))
    hm = [bf(F.relu(rnd((S, C, 14, 14), 60 + c))) for c in range(2)]
    xm = [bf(F.relu(rnd((S, C, 14, 14), 62 + c))) for c in range(2)]
    arena = torch.cat([to_slots(dpar), to_slots(x0), to_slots(x1), to_slots(hm[0]), to_slots(hm[1]),
                       to_slots(xm[0]), to_slots(xm[1]), zeros(12)])
    d1 = ops.pack_conv_weight(w1.to(d), G, C, C, 9, C * C * 9, 9, C * 9, flip=True)
    d2 = ops.pack_conv_weight(w2.to(d), G, C, C, 9, C * C * 9, 9, C * 9, flip=True)
    dp = ops.pack_conv_weight(wp.to(d), G, 2 * C, C, 1, 2 * C * C, 1, 2 * C)
    own = lambda i, c: (i + c) % G          # the inputs' own tokens differ from their parent's
    base = 7 * S
    fused = tiles_tensor([[i, -1, base + (4 + c) * S + i, base + (2 + c) * S + i, base + c * S + i, (5 + c) * S + i,
                           (3 + c) * S + i, own(i, c), 2 * groups[i] + c, (1 + c) * S + i, 0, 0]
                          for c in range(2) for i in range(S)])
    ops.resblock_tiles(arena, d2, d1, fused, 2 * S, w_pre=dp, num_pre=S)
    sep = base + 6 * S
    t_dp = tiles_tensor([[i, -1, sep + i, sep + S + i, -1, S + i, 2 * S + i, groups[i]] for i in range(S)])
    t_g2 = tiles_tensor([[sep + c * S + i, -1, sep + (2 + c) * S + i, -1, -1, (3 + c) * S + i, -1, own(i, c)]
                         for c in range(2) for i in range(S)])
    t_g1 = tiles_tensor([[sep + (2 + c) * S + i, -1, sep + (4 + c) * S + i, -1, sep + c * S + i, (5 + c) * S + i, -1, own(i, c)]
                         for c in range(2) for i in range(S)])
    ops.conv_tiles(arena, arena, dp, t_dp, S, taps=1, w_group_stride=2 * C * C, mask=arena, pair_out=1,
                   **dict(geo, n_out=2 * C))
    ops.conv_tiles(arena, arena, d2, t_g2, 2 * S, taps=9, w_group_stride=9 * C * C, mask=arena, **geo)
    ops.conv_tiles(arena, arena, d1, t_g1, 2 * S, taps=9, w_group_stride=9 * C * C, res=arena, mask=arena, **geo)
    torch.cuda.synchronize()
    assert torch.equal(arena[base:base + 6 * S], arena[sep:sep + 6 * S])
    full = F.conv_transpose2d(dpar, wp[0]) if G == 1 else torch.stack(
        [F.conv_transpose2d(dpar[i:i + 1], wp[groups[i]])[0] for i in range(S)])
    close(from_slots(arena[base:base + S], C), full[:, :C] * (x0 > 0))
    close(from_slots(arena[base + S:base + 2 * S], C), full[:, C:] * (x1 > 0))
    assert from_slots(arena[base + 4 * S:base + 6 * S], C).abs().max() > 0
    # a destination nobody wants (the scene inputs' launch has one per tile at most): left alone
    arena[sep:sep + 2 * S] = 7
    t_half = tiles_tensor([[i, -1, -1 if i % 2 else sep + i, sep + S + i if i % 2 else -1, -1, S + i, 2 * S + i, groups[i]]
                           for i in range(S)])
    ops.conv_tiles(arena, arena, dp, t_half, S, taps=1, w_group_stride=2 * C * C, mask=arena, pair_out=1,
                   **dict(geo, n_out=2 * C))
    torch.cuda.synchronize()
    for i in range(S):
        keep, written = (sep + i, sep + S + i) if i % 2 else (sep + S + i, sep + i)
        assert (arena[keep] == 7).all()
        assert torch.equal(arena[written], arena[base + (written - sep)])


def test_conv1x1_dgrad_pair_out():
    """proj input gradient: 256 outputs split over two destination maps with separate masks."""
    from vqa_amd import ops
    C, S = 128, 3
    dp = bf(rnd((S, C, 14, 14), 16))
    a0, a1 = bf(F.relu(rnd((S, C, 14, 14), 17))), bf(F.relu(rnd((S, C, 14, 14), 18)))
    w = bf(rnd((1, C, 2 * C, 1, 1), 19, 0.06))
    z = torch.zeros(S, HW, C, dtype=torch.bfloat16, device=dev())
    arena = torch.cat([to_slots(dp), to_slots(a0), to_slots(a1), z, z.clone()])
    tiles = tiles_tensor([[i, -1, 3 * S + i, 4 * S + i, -1, S + i, 2 * S + i, 0] for i in range(S)])
    wd = ops.pack_conv_weight(w.to(dev()), 1, 2 * C, C, 1, 0, 1, 2 * C)
    ops.conv_tiles(arena, arena, wd, tiles, S, taps=1, k_chunks=1, n_out=2 * C, src_slot_stride=HW * C,
                   dst_slot_stride=HW * C, src_pix_stride=C, dst_pix_stride=C, w_group_stride=2 * C * C,
                   mask=arena, pair_out=1)
    torch.cuda.synchronize()
    full = F.conv_transpose2d(dp, w[0])
    close(from_slots(arena[3 * S:4 * S], C), full[:, :C] * (a0 > 0))
    close(from_slots(arena[4 * S:], C), full[:, C:] * (a1 > 0))


def _wgrad_ref(x, dy, w_shape, padding):
    w = torch.zeros(w_shape, requires_grad=True)
    b = torch.zeros(w_shape[0], requires_grad=True)
    F.conv2d(x, w, b, padding=padding).backward(dy)
    return w.grad, b.grad


def test_wgrad3x3_grouped_accumulates_over_nodes():
    from vqa_amd import ops
    C, S, G = 128, 9, 3
    x = bf(rnd((S, C, 14, 14), 20))
    dy = bf(rnd((S, C, 14, 14), 21))
    arena = torch.cat([to_slots(x), to_slots(dy)])
    # group 0: nodes 0..3 (two items), group 2: nodes 4..8; group 1 unused
    items = tiles_tensor([[0, 0, 3, 0], [0, 3, 1, 0], [2, 4, 5, 0]])
    src_slots = tiles_tensor(list(range(S)))
    dy_slots = tiles_tensor([S + i for i in range(S)])
    dw = torch.zeros(G, 9, C, C, device=dev())
    db = torch.zeros(G, C, device=dev())
    ops.wgrad_tiles(arena, arena, (items, 0), 3, (src_slots, 0), (dy_slots, 0), dw, db, taps=9, k_chunks=1,
                    n_out=C, src_slot_stride=HW * C, dy_slot_stride=HW * C, src_pix_stride=C,
                    dy_pix_stride=C, dw_group_stride=9 * C * C)
    out = torch.zeros(G, C, C, 3, 3, device=dev())
    ops.unpack_conv_grad(dw, out, G, C, C, 9, C * C * 9)
    torch.cuda.synchronize()
    out, db = out.cpu(), db.cpu()
    for g, sl in ((0, slice(0, 4)), (2, slice(4, 9))):
        rw, rb = _wgrad_ref(x[sl], dy[sl], (C, C, 3, 3), 1)
        close(out[g], rw, 5e-3)
        close(db[g], rb, 5e-3)
    assert out[1].abs().max() == 0 and db[1].abs().max() == 0
    # partial-block mode + reduction: same sums (fp32 addition order differs), group 1 written as zeros,
    # and the same bits on every run
    part = torch.empty(3 * 2 * 9 * 8192, device=dev())
    outs = []
    for _ in range(2):
        o2 = torch.full((G, C, C, 3, 3), 7.0, device=dev())
        part.fill_(float('nan'))
        ops.wgrad_tiles(arena, arena, (items, 0), 3, (src_slots, 0), (dy_slots, 0), None, None, taps=9, k_chunks=1,
                        n_out=C, src_slot_stride=HW * C, dy_slot_stride=HW * C, src_pix_stride=C,
                        dy_pix_stride=C, dw_group_stride=9 * C * C, partial=part)
        ops.wgrad_reduce(part, (items, 0), 3, G, C, 1, 9, o2, C * C * 9)
        torch.cuda.synchronize()
        outs.append(o2.cpu())
    assert torch.equal(outs[0], outs[1])
    close(outs[0], out, 1e-5)
    assert outs[0][1].abs().max() == 0
    # the bias sums as per-item rows behind the partial blocks, added per group by the reduction:
    # the atomics' sums, group 1 written as zeros, db untouched by the first launch
    part = torch.full((sum(ops.wgrad_blocks(3, C, 1, 9)) * 8192,), float('nan'), device=dev())
    db2 = torch.full((G, C), 7.0, device=dev())
    ops.wgrad_tiles(arena, arena, (items, 0), 3, (src_slots, 0), (dy_slots, 0), None, db2, taps=9, k_chunks=1,
                    n_out=C, src_slot_stride=HW * C, dy_slot_stride=HW * C, src_pix_stride=C,
                    dy_pix_stride=C, dw_group_stride=9 * C * C, partial=part, bias_rows=True)
    torch.cuda.synchronize()
    assert (db2 == 7.0).all()
    o3 = torch.empty(G, C, C, 3, 3, device=dev())
    ops.wgrad_reduce(part, (items, 0), 3, G, C, 1, 9, o3, C * C * 9, db=db2)
    torch.cuda.synchronize()
    assert torch.equal(o3.cpu(), outs[0])
    close(db2.cpu(), db, 1e-5)
    assert db2[1].abs().max() == 0


def test_wgrad1x1_pair_sources():
    from vqa_amd import ops
    C, S = 128, 5
    x0, x1 = bf(rnd((S, C, 14, 14), 22)), bf(rnd((S, C, 14, 14), 23))
    dy = bf(rnd((S, C, 14, 14), 24))
    arena = torch.cat([to_slots(x0), to_slots(x1), to_slots(dy)])
    items = tiles_tensor([[0, 0, S, 0]])
    dw = torch.zeros(1, 1, 2 * C, C, device=dev())
    db = torch.zeros(1, C, device=dev())
    ops.wgrad_tiles(arena, arena, (items, 0), 1, (tiles_tensor(list(range(S))), 0),
                    (tiles_tensor([2 * S + i for i in range(S)]), 0), dw, db, taps=1, k_chunks=2, n_out=C,
                    src_slot_stride=HW * C, dy_slot_stride=HW * C, src_pix_stride=C, dy_pix_stride=C,
                    dw_group_stride=2 * C * C, src1_slots=(tiles_tensor([S + i for i in range(S)]), 0),
                    pair_chunks=1)
    out = torch.zeros(1, C, 2 * C, 1, 1, device=dev())
    ops.unpack_conv_grad(dw, out, 1, C, 2 * C, 1, 2 * C * C)
    torch.cuda.synchronize()
    rw, rb = _wgrad_ref(torch.cat([x0, x1], 1), dy, (C, 2 * C, 1, 1), 0)
    close(out[0].cpu(), rw, 5e-3)
    close(db[0].cpu(), rb, 5e-3)


def _padded(t, rows, cols):
    out = torch.zeros(rows, cols, dtype=torch.bfloat16)
    out[:t.shape[0], :t.shape[1]] = t.to(torch.bfloat16)
    return out.to(dev())


@pytest.mark.parametrize('M,N,K,split', [(128, 128, 64, 1), (200, 136, 320, 1), (70, 300, 1024, 4), (5, 32, 128, 2)])
def test_gemm_nt(M, N, K, split):
    from vqa_amd import ops
    a, b = bf(rnd((M, K), 30)), bf(rnd((N, K), 31))
    Mp, Np = (M + 127) // 128 * 128, (N + 127) // 128 * 128
    ldc = (N + 3) // 4 * 4
    c = torch.full((split, M, ldc), 7.0, device=dev())
    ops.gemm(_padded(a, Mp, K), _padded(b, Np, K), c, M, N, K, lda=K, ldb=K, ldc=ldc, split_k=split)
    torch.cuda.synchronize()
    close(c.sum(0).cpu()[:, :N], a @ b.t(), 2e-3)


@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (100, 264, 192)])
def test_gemm_nn_bf16_mask(M, N, K):
    from vqa_amd import ops
    a, b = bf(rnd((M, K), 32)), bf(rnd((K, N), 33))
    mk = bf(F.relu(rnd((M, N), 34)))
    Mp, Np = (M + 127) // 128 * 128, (N + 127) // 128 * 128
    c = torch.zeros(M, Np, dtype=torch.bfloat16, device=dev())
    ops.gemm(_padded(a, Mp, K), _padded(b, K, Np), c, M, N, K, lda=K, ldb=Np, ldc=Np, b_kmajor=True,
             out_bf16=True, mask=_padded(mk, M, Np))
    torch.cuda.synchronize()
    close(c.to(torch.float32).cpu()[:, :N], (a @ b) * (mk > 0))


@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (40, 392, 256)])
def test_gemm_tn(M, N, K):
    from vqa_amd import ops
    a, b = bf(rnd((K, M), 35)), bf(rnd((K, N), 36))
    Mp, Np = (M + 127) // 128 * 128, (N + 127) // 128 * 128
    c = torch.zeros(M, N, device=dev())
    ops.gemm(_padded(a, K, Mp), _padded(b, K, Np), c, M, N, K, lda=Mp, ldb=Np, ldc=N, a_kmajor=True, b_kmajor=True)
    torch.cuda.synchronize()
    close(c.cpu(), a.t() @ b, 2e-3)


def test_gemm_large_tiles_all_layouts():
    """Problems with M, N multiples of 256 and at least 128 tiles take the 256 x 256-tile kernel:
    the four operand layouts, split-k fp32 slabs, and the masked bf16 result."""
    from vqa_amd import ops
    M, N, K = 512, 16384, 192
    a, b = bf(rnd((M, K), 50)) * 0.25, bf(rnd((N, K), 51)) * 0.25
    ref = a @ b.t()
    ad, bd = a.to(torch.bfloat16).to(dev()), b.to(torch.bfloat16).to(dev())
    at, bt = ad.t().contiguous(), bd.t().contiguous()
    for akm, bkm in ((False, False), (False, True), (True, True), (True, False)):
        c = torch.full((M, N), 7.0, device=dev())
        ops.gemm(at if akm else ad, bt if bkm else bd, c, M, N, K, lda=M if akm else K, ldb=N if bkm else K,
                 ldc=N, a_kmajor=akm, b_kmajor=bkm)
        torch.cuda.synchronize()
        close(c.cpu(), ref, 2e-3)
    c = torch.full((3, M, N), 7.0, device=dev())
    ops.gemm(ad, bd, c, M, N, K, lda=K, ldb=K, ldc=N, split_k=3)
    torch.cuda.synchronize()
    close(c.sum(0).cpu(), ref, 2e-3)
    mk = bf(F.relu(rnd((M, N), 52)))
    cb = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
    ops.gemm(ad, bt, cb, M, N, K, lda=K, ldb=N, ldc=N, b_kmajor=True, out_bf16=True,
             mask=mk.to(torch.bfloat16).to(dev()))
    torch.cuda.synchronize()
    close(cb.to(torch.float32).cpu(), ref * (mk > 0))


@pytest.mark.parametrize('K', [64, 256, 320, 1024])
def test_gemm_large_tiles_k_major_b_any_k_depth(K):
    """The shapes of the first linear layer's backward GEMMs on the 256 x 256-tile kernel (A k-contiguous,
    B k-major, M and N multiples of 256), whose LDS fragment reads carry their own counted waits while
    the next stage's LDS-DMA is in flight: one K step (no next stage), four, five (odd: the buffers
    swap an odd number of times), sixteen (the layer's).  fp32 result, masked bf16 result, and operands
    inside wider arrays."""
    from vqa_amd import ops
    M, N = 512, 16384
    a, b = bf(rnd((M, K), 70)) * 0.25, bf(rnd((K, N), 71)) * 0.25
    ref = a @ b
    ad, bd = a.to(torch.bfloat16).to(dev()), b.to(torch.bfloat16).to(dev())
    c = torch.full((M, N), 7.0, device=dev())
    ops.gemm(ad, bd, c, M, N, K, lda=K, ldb=N, ldc=N, b_kmajor=True)
    torch.cuda.synchronize()
    close(c.cpu(), ref, 2e-3)
    mk = bf(F.relu(rnd((M, N), 72)))
    cb = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
    ops.gemm(ad, bd, cb, M, N, K, lda=K, ldb=N, ldc=N, b_kmajor=True, out_bf16=True, mask=mk.to(torch.bfloat16).to(dev()))
    torch.cuda.synchronize()
    close(cb.to(torch.float32).cpu(), ref * (mk > 0))
    # leading dimensions larger than the rows (operands inside wider arrays)
    aw = torch.zeros(M, K + 64, dtype=torch.bfloat16, device=dev()); aw[:, :K] = ad
    bw = torch.zeros(K, N + 256, dtype=torch.bfloat16, device=dev()); bw[:, :N] = bd
    cw = torch.full((M, N + 8), 7.0, device=dev())
    ops.gemm(aw, bw, cw, M, N, K, lda=K + 64, ldb=N + 256, ldc=N + 8, b_kmajor=True)
    torch.cuda.synchronize()
    close(cw.cpu()[:, :N], ref, 2e-3)
    assert (cw[:, N:] == 7.0).all()


def test_layout_helpers():
    from vqa_amd import ops
    n, cin, cpad = 3, 100, 128
    x = rnd((n, cin, 14, 14), 40)
    out = torch.full((n, HW, cpad), 9.0, dtype=torch.bfloat16, device=dev())
    ops.nchw_to_nhwc(x.to(dev()), out, n, cin, cpad)
    back = torch.zeros(n, cpad, HW, device=dev())
    ops.nhwc_to_nchw(out, back, n, cpad)
    torch.cuda.synchronize()
    ref = torch.zeros(n, cpad, HW)
    ref[:, :cin] = bf(x).reshape(n, cin, HW)
    assert torch.equal(out.to(torch.float32).cpu(), ref.permute(0, 2, 1))
    assert torch.equal(back.cpu(), ref)
    # FC weight permutation and its inverse on gradients
    f, p = 5, 64
    w = rnd((f, p * HW), 41)
    wn = torch.zeros(128, HW * p, dtype=torch.bfloat16, device=dev())
    ops.fc_weight_to_nhwc(w.to(dev()), wn, f, 128, p)
    g = torch.zeros(f, p * HW, device=dev())
    gin = rnd((f, HW * p), 42)
    ops.fc_grad_from_nhwc(gin.to(dev()), g, f, p)
    torch.cuda.synchronize()
    assert torch.equal(wn[:f].to(torch.float32).cpu(), bf(w).reshape(f, p, HW).permute(0, 2, 1).reshape(f, -1))
    assert torch.equal(g.cpu(), gin.reshape(f, HW, p).permute(0, 2, 1).reshape(f, -1))
    # bf16 activations, channels-last -> the reference's flatten order
    a = rnd((7, HW, 192), 45).to(torch.bfloat16)
    at = torch.zeros(7, 192, HW, dtype=torch.bfloat16, device=dev())
    ops.transpose_hw_c(a.to(dev()), at, 7, 192)
    torch.cuda.synchronize()
    assert torch.equal(at.cpu(), a.permute(0, 2, 1).contiguous())


def test_bias_act_colsum_castpad():
    from vqa_amd import ops
    rows, cols = 37, 100
    acc = rnd((3, rows, 104), 43)
    bias = rnd((cols,), 44)
    out = torch.zeros(rows, 128, dtype=torch.bfloat16, device=dev())
    ops.bias_act(acc.to(dev()), bias.to(dev()), out, rows, cols, 104, 128, True, slabs=3, slab_stride=rows * 104)
    outf = torch.zeros(rows, cols, device=dev())
    ops.bias_act(acc.to(dev()), bias.to(dev()), outf, rows, cols, 104, cols, False, slabs=3, slab_stride=rows * 104)
    cs = torch.zeros(cols, device=dev())
    ops.col_sum(outf, cs, rows, cols, cols)
    cb = torch.zeros(cols, device=dev())
    ops.col_sum(out, cb, rows, cols, 128)
    cp = torch.full((128, 128), 3.0, dtype=torch.bfloat16, device=dev())
    ops.cast_pad(outf, cp, rows, cols, 128, 128)
    torch.cuda.synchronize()
    ref = acc[:, :, :cols].sum(0) + bias
    close(out[:, :cols].to(torch.float32).cpu(), F.relu(ref))
    close(outf.cpu(), ref, 1e-5)
    close(cs.cpu(), ref.sum(0), 1e-4)
    close(cb.cpu(), bf(F.relu(ref)).sum(0), 1e-2)
    want = torch.zeros(128, 128)
    want[:rows, :cols] = bf(outf.cpu())
    assert torch.equal(cp.to(torch.float32).cpu(), want)


def test_slot_sum_mask():
    from vqa_amd import ops
    C, S = 128, 6
    x = bf(rnd((S, HW, C), 45))
    mk = bf(F.relu(rnd((2, HW, C), 46)))
    arena = torch.cat([x, mk, torch.zeros(2, HW, C)]).to(torch.bfloat16).to(dev())
    blob = tiles_tensor([0, 2, 6] + [0, 3, 1, 2, 4, 5] + [S + 2, S + 3] + [S, S + 1])
    ops.slot_sum_mask(arena, blob, 0, 3, 9, 11, 2, HW * C)
    torch.cuda.synchronize()
    got = arena[S + 2:].to(torch.float32).cpu()
    close(got[0], (x[0] + x[3]) * (mk[0] > 0))
    close(got[1], (x[1] + x[2] + x[4] + x[5]) * (mk[1] > 0))


@pytest.mark.parametrize('M,N,K', [(37, 70, 50), (128, 64, 256), (5, 200, 1000)])
def test_gemm_f32_all_layouts_ragged_sizes_split_and_mask(M, N, K):
    """vqa_gemm_f32: exact fp32 products and sums (v_mfma_f32_16x16x4_f32) against float64 on the
    same fp32 inputs, 1e-5; sizes that are multiples of nothing, the four operand layouts, split-k
    slabs and the masked form."""
    from vqa_amd import ops
    a, b = rnd((M, K), 60), rnd((N, K), 61)
    ref = (a.double() @ b.double().t()).float()
    ad, bd = a.to(dev()), b.to(dev())
    at, bt = ad.t().contiguous(), bd.t().contiguous()
    for akm, bkm in ((False, False), (False, True), (True, True), (True, False)):
        c = torch.full((M, N), 7.0, device=dev())
        ops.gemm_f32(at if akm else ad, bt if bkm else bd, c, M, N, K, lda=M if akm else K, ldb=N if bkm else K, ldc=N,
                     a_kmajor=akm, b_kmajor=bkm)
        torch.cuda.synchronize()
        close(c.cpu(), ref, 1e-5)
    c = torch.full((3, M, N), 7.0, device=dev())
    ops.gemm_f32(ad, bd, c, M, N, K, lda=K, ldb=K, ldc=N, split_k=3)
    torch.cuda.synchronize()
    close(c.sum(0).cpu(), ref, 1e-5)
    mk = F.relu(rnd((M, N), 62))
    c = torch.full((M, N), 7.0, device=dev())
    ops.gemm_f32(ad, bt, c, M, N, K, lda=K, ldb=N, ldc=N, b_kmajor=True, mask=mk.to(dev()))
    torch.cuda.synchronize()
    close(c.cpu(), ref * (mk > 0), 1e-5)


def test_transpose_f32_batched_with_padded_rows():
    from vqa_amd import ops
    n, rows, cols, ld = 3, 50, 196, 64
    x = rnd((n, rows, cols), 63)
    out = torch.full((n, cols, ld), -1.0, device=dev())
    ops.transpose_f32(x.to(dev()), out, n, rows, cols, ld_out=ld, out_stride=cols * ld)
    torch.cuda.synchronize()
    assert torch.equal(out[:, :, :rows].cpu(), x.permute(0, 2, 1))
    assert float(out[:, :, rows:].max()) == -1.0           # the padding is left alone


def test_wgrad1x1_wide_output_bias_rows():
    """A 1x1 weight gradient with 1024 output channels (a wide classifier projection): the bias sums
    stored per item and added by the reduction's bias workgroup (four output channels per thread)
    equal the atomically accumulated ones and torch's."""
    from vqa_amd import ops
    C, P, S = 128, 1024, 6
    x = bf(rnd((S, C, 14, 14), 30))
    dy = bf(rnd((S, P, 14, 14), 31))
    src, dyt = to_slots(x), to_slots(dy)
    items = tiles_tensor([[0, 0, 4, 0], [0, 4, 2, 0]])
    slots = tiles_tensor(list(range(S)))
    kw = dict(taps=1, k_chunks=1, n_out=P, src_slot_stride=HW * C, dy_slot_stride=HW * P, src_pix_stride=C,
              dy_pix_stride=P, dw_group_stride=C * P)
    dw = torch.zeros(1, 1, C, P, device=dev())
    db = torch.zeros(1, P, device=dev())
    ops.wgrad_tiles(src, dyt, (items, 0), 2, (slots, 0), (slots, 0), dw, db, **kw)
    part = torch.full((sum(ops.wgrad_blocks(2, P, 1, 1)) * 8192,), float('nan'), device=dev())
    db2 = torch.full((1, P), 7.0, device=dev())
    out = torch.empty(P, C, 1, 1, device=dev())
    ops.wgrad_tiles(src, dyt, (items, 0), 2, (slots, 0), (slots, 0), None, db2, partial=part, bias_rows=True, **kw)
    ops.wgrad_reduce(part, (items, 0), 2, 1, P, 1, 1, out, 0, db=db2)
    torch.cuda.synchronize()
    rw, rb = _wgrad_ref(x, dy, (P, C, 1, 1), 0)
    close(db2.cpu()[0], rb, 5e-3)
    close(db2.cpu(), db.cpu(), 1e-5)
    close(out.cpu(), rw, 5e-3)


@pytest.mark.parametrize('Cin,S,with_copy', [(1024, 5, True), (128, 3, True), (256, 2, False)])
def test_stem_from_nchw_equals_layout_pass_plus_convolution(Cin, S, with_copy):
    """vqa_stem_nchw (the layout change inside the convolution kernel) against vqa_nchw_to_nhwc followed by
    vqa_conv_tiles: the same bits in the output map and in the bf16 channels-last copy, and fp32 torch to the
    kernels' usual tolerance."""
    from vqa_amd import ops
    C = 128
    x = torch.relu(rnd((S, Cin, 14, 14), 31))
    w = bf(rnd((1, C, Cin, 3, 3), 32, 0.02))
    b = rnd((1, C), 33, 0.1)
    xd = x.to(dev())
    wp = ops.pack_conv_weight(w.to(dev()), 1, C, Cin, 9, 0, Cin * 9, 9)
    # reference path: layout pass, then the grouped convolution over the channels-last copy
    xn = torch.empty(S * HW * Cin, dtype=torch.bfloat16, device=dev())
    ops.nchw_to_nhwc(xd, xn, S, Cin, Cin)
    first = 2                                    # destination slots 2..: slots in front stay untouched
    ref_dst = torch.zeros(first + S, HW, C, dtype=torch.bfloat16, device=dev())
    tiles = tiles_tensor([[i, -1, first + i, -1, -1, -1, -1, 0] for i in range(S)])
    ops.conv_tiles(xn, ref_dst, wp, tiles, S, taps=9, k_chunks=Cin // 128, n_out=C, src_slot_stride=HW * Cin,
                   dst_slot_stride=HW * C, src_pix_stride=Cin, dst_pix_stride=C, w_group_stride=0, bias=b.to(dev()), relu=True)
    dst = torch.zeros(first + S, HW, C, dtype=torch.bfloat16, device=dev())
    copy = torch.full((S * HW * Cin,), 7.0, dtype=torch.bfloat16, device=dev()) if with_copy else None
    for _ in range(2):
        ops.stem_nchw(xd, copy, dst, wp, b.to(dev()), S, Cin, HW * C, first, C, relu=True)
    torch.cuda.synchronize()
    assert torch.equal(dst.view(torch.int16), ref_dst.view(torch.int16))
    if with_copy:
        assert torch.equal(copy.view(torch.int16), xn.view(torch.int16))
    close(from_slots(dst[first:], C), F.relu(F.conv2d(bf(x), w[0], b[0], padding=1)))
