"""Host logic of the fused gradient reduce (no GPU): how the plans' registered partial reduces are carved into ReduceGrads regions, and the
slab count the bf16-storage weight gradients ask for."""
import torch

from pulse_amd import kernels as K


def test_parts_are_carved_out_of_their_base_ranges():
    cs0, hs, cs1 = torch.zeros(64, 2048), torch.zeros(16, 1100), torch.zeros(64, 1024)
    # flat layout of a [1024, 512] actor / critic pair: W1 | b1 | W2 | b2 | heads
    base = [(0, 1966080 + 2048, 8), (1968128, 1048576 + 1024 + 72000 + 144, 8)]
    parts = [(1968128 + 1048576 + 1024, 72144, 16, hs), (1966080, 2048, 64, cs0), (1968128 + 1048576, 1024, 64, cs1)]
    regions, fused = K.carve_reduce_regions(base, parts)
    assert fused
    assert [r[:3] for r in regions] == [(0, 1966080, 8), (1966080, 2048, 64), (1968128, 1048576, 8), (3016704, 1024, 64), (3017728, 72144, 16)]
    assert regions[1][4] is cs0 and regions[1][5] == 2048 and regions[4][4] is hs and regions[4][5] == 1100
    assert all(len(r) == 4 for r in (regions[0], regions[2]))
    # the regions tile the flat buffer without gaps
    assert regions[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(regions, regions[1:]))
    assert regions[-1][0] + regions[-1][1] == base[-1][0] + base[-1][1]


def test_fallbacks_keep_the_plain_ranges():
    t = torch.zeros(8, 1024)
    base = [(0, 4096, 4), (4096, 4096, 8)]
    plain = [(0, 4096, 4, 0.0), (4096, 4096, 8, 0.0)]
    assert K.carve_reduce_regions(base, []) == (plain, False)                                     # nothing registered
    assert K.carve_reduce_regions(base, [(3584, 1024, 8, t)]) == (plain, False)                   # straddles two base ranges
    assert K.carve_reduce_regions(base, [(1022, 1024, 8, t)]) == (plain, False)                   # offset not a multiple of 4
    assert K.carve_reduce_regions(base, [(0, 1024, 8, torch.zeros(8, 1026))]) == (plain, False)   # row stride not a multiple of 4
    assert K.carve_reduce_regions(base, [(0, 1024, 8, t), (512, 1024, 8, t)]) == (plain, False)   # overlapping parts
    assert K.carve_reduce_regions(base, [(9000, 1024, 8, t)]) == (plain, False)                   # outside every base range
    many = [(512 * i, 256, 8, t) for i in range(5)]
    assert K.carve_reduce_regions(base, many)[1] is False                                         # 11 regions > 8
    assert K.carve_reduce_regions(base, many, max_regions=16)[1] is True


def test_bf16_weight_gradient_slab_counts():
    # layer-1-sized outputs: the 256 x 256 tiling fills the chip within 8 slabs (32 tiles x 8); smaller ones keep the narrow tiling's count
    assert K.dw_split_b16(2048, 960, 1, 8) == 8 and K.dw_split_b16(1024, 1960, 1, 8) == 8
    assert K.dw_split_b16(512, 1024, 2, 8) == 8             # 16 wide tiles x 8 = 128 workgroups: narrow (32 tiles x 8)
    assert K.dw_split_b16(2048, 2048, 1, 8) == 4             # 64 wide tiles: 4 slabs fill 256 CUs
    assert K.dw_split_b16(8192, 8192, 1, 8) == 1
    assert K.dw_split_b16(64, 64, 1, 8) == 8
