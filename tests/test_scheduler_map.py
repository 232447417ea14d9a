"""The forward's workgroup -> tile map (csrc/fa_fwd_kernel.h: tile_of_wg, host side csrc/fa_fwd_api.hip) restated in Python: every
tile is visited exactly once, padding workgroups fall outside, and the XCDs (workgroup id & 7) get equal shares for any head
count -- whole (batch, kv head) units as far as they deal evenly, the remaining heads by m_block of the GQA group."""
import pytest


def host(b, h_k, h_ratio, num_m_blocks):
    tiles = num_m_blocks * h_k * h_ratio * b
    bk_units = b * h_k
    per_kvh = h_ratio * num_m_blocks
    whole_units = bk_units // 8 * 8 if bk_units >= 16 else 0
    whole_slots = whole_units // 8 * per_kvh
    rem_units = (tiles - whole_slots * 8 + h_ratio - 1) // h_ratio
    grid = 8 * (whole_slots + (rem_units + 7) // 8 * h_ratio)
    return tiles, per_kvh, whole_slots, grid


def tile_of_wg(wg, unit_tiles, whole_slots, h_ratio):
    xcd, slot = wg & 7, wg >> 3
    if slot < whole_slots:
        return ((slot // unit_tiles) * 8 + xcd) * unit_tiles + slot % unit_tiles
    s2 = slot - whole_slots
    return whole_slots * 8 + ((s2 // h_ratio) * 8 + xcd) * h_ratio + s2 % h_ratio


@pytest.mark.parametrize("b,h_k,h_ratio,m", [(1, 1, 1, 1), (2, 10, 1, 64), (2, 21, 1, 32), (1, 12, 1, 7), (4, 16, 1, 32), (8, 8, 4, 5),
                                             (3, 7, 2, 9), (1, 40, 1, 3), (5, 5, 8, 2), (2, 8, 1, 64), (1, 17, 3, 4)])
def test_every_tile_once_and_xcds_balanced(b, h_k, h_ratio, m):
    tiles, unit_tiles, whole_slots, grid = host(b, h_k, h_ratio, m)
    assert grid % 8 == 0
    seen, per_xcd = [], [0] * 8
    for wg in range(grid):
        t = tile_of_wg(wg, unit_tiles, whole_slots, h_ratio)
        assert t >= 0
        if t < tiles:
            seen.append(t)
            per_xcd[wg & 7] += 1
    assert sorted(seen) == list(range(tiles))
    # an XCD's share differs from another's by at most one GQA group's worth of tiles
    assert max(per_xcd) - min(per_xcd) <= h_ratio
    # the tiles of one (batch, kv head, m_block) GQA group always share an XCD
    xcd_of = {}
    for wg in range(grid):
        t = tile_of_wg(wg, unit_tiles, whole_slots, h_ratio)
        if t < tiles:
            xcd_of[t] = wg & 7
    for t0 in range(0, tiles, h_ratio):
        assert len({xcd_of[t] for t in range(t0, t0 + h_ratio)}) == 1


@pytest.mark.parametrize("units,blocks", [(1, 1), (20, 64), (42, 32), (12, 7), (64, 32), (17, 5), (40, 3), (16, 9)])
def test_backward_block_map(units, blocks):
    """decode_block() of the backward kernels (csrc/fa_bwd_kernel.h) with unit_grid() (csrc/fa_bwd_api.hip)."""
    tiles = units * blocks
    ws = units // 8 * blocks if units >= 16 else 0
    grid = 8 * (ws + (tiles - ws * 8 + 7) // 8)
    seen, per_xcd = [], [0] * 8
    for wg in range(grid):
        xcd, slot = wg & 7, wg >> 3
        t = ((slot // blocks) * 8 + xcd) * blocks + slot % blocks if slot < ws else slot * 8 + xcd
        if t < tiles:
            seen.append(t)
            per_xcd[xcd] += 1
    assert sorted(seen) == list(range(tiles))
    assert max(per_xcd) - min(per_xcd) <= 1


@pytest.mark.parametrize("b,h_k,h_ratio,m,cus", [(4, 16, 1, 32, 256), (32, 16, 1, 2, 256), (8, 16, 1, 8, 256), (2, 21, 1, 32, 256),
                                                 (3, 7, 2, 9, 64), (8, 8, 4, 5, 256), (1, 40, 1, 3, 104), (4, 16, 1, 64, 256)])
def test_persistent_chains_cover_every_tile_once(b, h_k, h_ratio, m, cus):
    """The persistent form of the 256-row kernel (fwd_kernel_w64<.., PERSIST>, role of hopper/tile_scheduler.hpp:140-214): one
    workgroup per CU, round t of the CU with index k inside its XCD takes slot cpx t + k of the XCD's slot list, odd rounds in
    reverse.  Every valid tile is taken by exactly one workgroup, a chain stays on one XCD, a chain has at most 64 rounds
    whenever the host lets the form run, and under a causal mask (cost ~ m_block + 1) the zig-zag keeps the chains of an XCD
    within two items' worth of each other."""
    tiles, unit_tiles, whole_slots, grid = host(b, h_k, h_ratio, m)
    wgs = min(grid, cus & ~7)
    cpx = wgs >> 3
    rounds = (grid + wgs - 1) // wgs
    seen, cost = [], {}
    for wg0 in range(wgs):
        k, xcd = wg0 >> 3, wg0 & 7
        for t in range(rounds):
            slot = cpx * t + (cpx - 1 - k if t & 1 else k)
            wg = slot * 8 + xcd
            if cpx * t * 8 >= grid or wg >= grid:
                continue
            tile = tile_of_wg(wg, unit_tiles, whole_slots, h_ratio)
            if tile >= tiles:
                continue
            seen.append(tile)
            m_block = m - 1 - (tile % unit_tiles) // h_ratio
            cost[wg0] = cost.get(wg0, 0) + m_block + 1
    assert sorted(seen) == list(range(tiles))
    if grid <= 64 * wgs:
        assert rounds <= 64
    if rounds >= 2 and rounds % 2 == 0 and whole_slots * 8 == grid:
        for xcd in range(8):
            c = [cost.get(k * 8 + xcd, 0) for k in range(cpx)]
            assert max(c) - min(c) <= 2 * m, (xcd, c)
