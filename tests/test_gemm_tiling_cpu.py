"""Host-side logic of the two-tiling x3 GEMM (no GPU): the planner's cost model / slab-count choice (pulse_amd/kernels.py: x3_tile_costs,
dw_split_x3 -- the python mirror of gemm_f32.hip: x3_wide_tile) and the per-layer slab regions of the actor / critic gradient reduce
(pulse_amd/learning/network.py: _slab_regions).  Reference of the op being planned: the weight gradients of nn.Linear in
phc/learning/network_builder.py:105-124 (autograd's mm over the batch dimension)."""
import pytest

from pulse_amd import kernels as K


def test_costs_are_rounds_of_the_two_tilings():
    # 16384 x 2048: 2048 narrow tiles = 4 rounds of 512 (2 units each); 512 wide tiles = 2 rounds
    narrow, wide = K.x3_tile_costs(16384, 2048, 1, k=934)
    assert narrow == 8.0
    assert 6.0 < wide < 7.0 and wide < narrow                        # the layer-1 forward of cfg2 takes the wide tile
    # a rollout-sized launch: 128 wide tiles would leave half the chip idle for a whole round
    narrow, wide = K.x3_tile_costs(4096, 2048, 1, k=934)
    assert narrow == 2.0 and wide > narrow
    # outputs of at most 128 rows or columns never take the wide tile
    assert K.x3_tile_costs(128, 4096, 8)[1] == float("inf") and K.x3_tile_costs(16384, 69, 2)[1] == float("inf")
    # a wide round costs 3.2-3.4 narrow-tile units whatever the reduction length (plain epilogue: both tilings carry a fixed cost per round)
    for k in (16, 64, 934, 4096, 16384):
        assert 3.1 < K.x3_tile_costs(256, 256, 1, k=k)[1] < 3.5


@pytest.mark.parametrize("m,n,batch,max_split,k,expect", [
    (2048, 960, 1, 8, 16384, 8),      # cfg2 layer 1: 32 wide tiles x 8 slabs fill the chip (4 narrow slabs before round 5)
    (512, 1024, 2, 16, 16384, 16),    # cfg2 layer 2 pair: 16 wide tiles x 16 slabs
    (512, 1024, 2, 8, 16384, 8),      # ... with only 8 slabs available it stays on 8 narrow ones
    (69, 512, 2, 8, 16384, 8),        # heads: never wide
])
def test_slab_counts_of_the_cfg2_weight_gradients(monkeypatch, m, n, batch, max_split, k, expect):
    monkeypatch.setattr(K, "F32_MODE", "x3")
    monkeypatch.delenv("PULSE_X3_WIDE", raising=False)
    assert K.dw_split_x3(m, n, batch, max_split, K=k) == expect


def test_slab_count_is_a_power_of_two_fraction_and_falls_back(monkeypatch):
    monkeypatch.setattr(K, "F32_MODE", "x3")
    monkeypatch.delenv("PULSE_X3_WIDE", raising=False)
    for m, n, b in ((2048, 3096, 1), (1024, 1536, 1), (300, 300, 1), (3096, 392, 1), (64, 64, 3)):
        for ms in (1, 2, 8, 16, 32):
            s = K.dw_split_x3(m, n, b, ms, K=16384)
            assert 1 <= s <= ms and ms % s == 0 and (ms // s) & ((ms // s) - 1) == 0
    # the fp32-MFMA arithmetic and a launcher held to the 128 x 128 tile keep the round-4 rule (smallest count that still gives 512 narrow
    # workgroups).  The planner asks the LIBRARY which tiling is in effect (pulse_gemm_x3_mode: gemm option 4 of this thread, or PULSE_X3_WIDE
    # as the library read it once), so planner and launcher cannot disagree.
    K.gemm_set_option(4, 1)
    try:
        assert K._lib.load().pulse_gemm_x3_mode() == 1
        assert K.dw_split_x3(2048, 960, 1, 8) == K.dw_split(16 * 8, 8) == 4
    finally:
        K.gemm_set_option(4, 0)
    assert K._lib.load().pulse_gemm_x3_mode() == 0
    monkeypatch.setattr(K, "F32_MODE", "mfma32")
    assert K.dw_split_x3(2048, 960, 1, 8) == 4


def test_per_layer_slab_regions_cover_the_flat_gradient_once():
    from pulse_amd.learning.network import A2CNetwork
    net = object.__new__(A2CNetwork)                                  # layout arithmetic only: no device, no parameters
    net.units, net.actions_num, net.in_dim, net.in_pitch = [1024, 512], 69, 934, 960
    net.a_pitch = 72
    net._build_layout()
    ws = {"layer_slabs": [8, 16]}
    regions = net._slab_regions(ws)
    assert [r[2] for r in regions] == [8, 16, 1]                      # layer 1, layer 2, heads (their sums come from the plan's own partial rows)
    pos = 0
    for off, cnt, ns in regions:
        assert off == pos and cnt > 0
        pos += cnt
    assert pos == net.n_flat
    assert regions[0][1] == net.w_off[1] and regions[1][0] == net.w_off[1] and regions[2][0] == net.wh_off
    # the carved form (heads' partial rows registered by the plan) still tiles the buffer and keeps each layer's own count
    import torch
    hs = torch.zeros(32, net.n_flat - net.wh_off)
    carved, fused = K.carve_reduce_regions(regions, [(net.wh_off, net.n_flat - net.wh_off, 32, hs)])
    assert fused and [r[2] for r in carved] == [8, 16, 32] and sum(r[1] for r in carved) == net.n_flat


def test_param_book_reduce_regions_cover_the_flat_buffer():
    """[r6] ParamBook.reduce_regions: per-parameter slab counts, zeros over untouched ranges, adjacent equal counts merged, tail padding covered."""
    from pulse_amd.learning.graph import ParamBook
    book = ParamBook("cpu", split_k=8)
    book.add("a.w", 8, 10)           # pitch 12 -> 96 floats
    book.add("a.b", 1, 8)            # 8
    book.add("b.w", 4, 6)            # pitch 8 -> 32
    book.add("b.b", 1, 4)            # 4
    book.add("c.w", 2, 3)            # pitch 4 -> 8
    book.add("c.b", 1, 2)            # pitch 4 -> 4   (total 152; n_flat = 152)
    book.add("sigma", 1, 5)          # pitch 8 -> 8  (no launch registered: split_k)
    book.finalize(trainable=False)
    book.note_slab_layout("a.w", 4, 256); book.note_slab_layout("a.b", 4, 256)
    book.note_slab_layout("b.w", 4, 256); book.note_slab_layout("b.b", 4, 256)
    book.note_slab_layout("c.w", 1, 256); book.note_slab_layout("c.b", 1, 256)
    regs = book.reduce_regions(())
    assert regs == [(0, 140, 4), (140, 12, 1), (152, 8, 8)]
    regs = book.reduce_regions([(104, 140)])                      # sub-network b untouched
    assert regs == [(0, 104, 4), (104, 36, 0), (140, 12, 1), (152, 8, 8)]
    assert sum(c for _, c, _ in regs) == book.n_flat and all(o % 4 == 0 and c % 4 == 0 for o, c, _ in regs)
    import pytest
    with pytest.raises(RuntimeError):
        book.note_slab_layout("a.w", 8, 128)                      # another plan writing a different slab count: refused
