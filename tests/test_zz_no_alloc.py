"""vqa_wgrad_tiles_noalloc (partial blocks in the library's static device array, no allocation)
against the existing pair vqa_wgrad_tiles(partial workspace) + vqa_wgrad_reduce, through the C ABI.

Same kernels on the same inputs, so the weight gradients must be bit-identical.  Neither call here
asks for rows of bias sums behind the partial blocks (the training path does, and sums them in item
order), so on both sides the bias gradient takes the kernel's other form, fp32 atomics whose order
varies: 1e-6.  Two calls in a row, the second on
a smaller input (the array keeps the first call's contents), then a single-tap call, then a
request beyond the array's size, which must be refused with a message."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HW, C = 196, 128


def _case(seed, n_nodes, groups, chunk):
    """Random maps plus a work-item list: nodes sorted by weight group, cut into items of <= chunk."""
    g = torch.Generator(device='cuda').manual_seed(seed)
    arena = torch.randn(2 * n_nodes, HW, C, device='cuda', generator=g).to(torch.bfloat16)
    rng = np.random.default_rng(seed)
    grp = np.sort(rng.integers(0, groups, size=n_nodes))
    grp[0], grp[-1] = 0, groups - 1
    items = []
    first = 0
    for gi in range(groups):
        cnt = int((grp == gi).sum())
        while cnt > 0:
            take = min(cnt, chunk)
            items.append((gi, first, take, 0))
            first += take
            cnt -= take
    perm = rng.permutation(n_nodes).astype(np.int32)
    src_slots, dy_slots = perm, (n_nodes + rng.permutation(n_nodes)).astype(np.int32)
    blob = np.concatenate([np.array(items, dtype=np.int32).reshape(-1), src_slots, dy_slots])
    return arena, torch.from_numpy(blob).cuda(), len(items), 4 * len(items), 4 * len(items) + n_nodes


def _both(seed, n_nodes, groups, chunk, taps):
    from vqa_amd import ops
    arena, blob, n_items, off_s, off_d = _case(seed, n_nodes, groups, chunk)
    geo = dict(k_chunks=1, n_out=C, src_slot_stride=HW * C, dy_slot_stride=HW * C, src_pix_stride=C,
               dy_pix_stride=C, dw_group_stride=taps * C * C, total_nodes=n_nodes)
    part = torch.empty(n_items * 2 * taps * 8192, dtype=torch.float32, device='cuda')
    ref = torch.full((groups, C, C, taps), float('nan'), device='cuda')
    db_ref = torch.zeros(groups, C, device='cuda')
    ops.wgrad_tiles(arena, arena, (blob, 0), n_items, (blob, off_s), (blob, off_d), None, db_ref, taps=taps,
                    partial=part, **geo)
    ops.wgrad_reduce(part, (blob, 0), n_items, groups, C, 1, taps, ref, C * C * taps)
    out = torch.full((groups, C, C, taps), float('nan'), device='cuda')
    db = torch.zeros(groups, C, device='cuda')
    ops.wgrad_tiles_noalloc(arena, arena, (blob, 0), n_items, (blob, off_s), (blob, off_d), db, out, taps=taps,
                            groups=groups, out_g_stride=C * C * taps, **geo)
    torch.cuda.synchronize()
    assert torch.isfinite(ref).all() and float(ref.abs().max()) > 0
    assert torch.equal(out, ref)
    assert torch.allclose(db, db_ref, rtol=1e-6, atol=1e-4)
    return n_items


def test_noalloc_weight_gradient_matches_the_workspace_form():
    assert _both(1, 40, 3, 8, 9) >= 5
    _both(2, 7, 2, 8, 9)            # smaller: the array still holds the first call's blocks
    _both(3, 33, 4, 16, 1)          # single tap (the 1x1 projection's shape)
    _both(1, 40, 3, 8, 9)


def test_noalloc_refuses_more_than_its_array_holds():
    from vqa_amd import _lib, ops
    arena, blob, n_items, off_s, off_d = _case(5, 64, 2, 1)          # 64 items x 2 x 9 = 1152 blocks > 1024
    out = torch.zeros(2, C, C, 9, device='cuda')
    with pytest.raises(_lib.VqaError, match='static scratch holds 1024'):
        ops.wgrad_tiles_noalloc(arena, arena, (blob, 0), n_items, (blob, off_s), (blob, off_d), None, out, taps=9,
                                groups=2, out_g_stride=C * C * 9, k_chunks=1, n_out=C, src_slot_stride=HW * C,
                                dy_slot_stride=HW * C, src_pix_stride=C, dy_pix_stride=C, dw_group_stride=9 * C * C,
                                total_nodes=64)
