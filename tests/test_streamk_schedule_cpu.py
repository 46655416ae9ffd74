"""The stream-K work split of gemm256sk_kernel (devias_amd/csrc/gemm.hip), restated in Python and checked as a PARTITION: for every XCD group
size, workgroup count and K depth the items of all workgroups cover every (tile, K-tile) exactly once, every split tile has exactly one head
fragment [0, e) and one tail fragment [e, nk) owned by consecutive workgroups (the tail's owner has the higher index: a waiter only waits for a
lower one), a workgroup publishes at most one partial and consumes at most one, and all workgroups do the same number of K-iterations +-1.
(The kernel itself is checked bitwise against the data-parallel kernels on the GPU: tests/test_kernels_gpu.py.)"""
import itertools

import pytest


def items_of(j, W, cnt, nk):
    """items of workgroup j of one XCD group, in processing order: (tile, kb, ke, kind); mirrors the kernel's table construction"""
    sk = W + cnt % W if cnt >= 2 * W else cnt
    skb = cnt - sk
    dp_rounds = skb // W
    S0, S1 = j * sk * nk // W, (j + 1) * sk * nk // W
    b, tile_t = S0 % nk, S0 // nk
    e, tile_h = S1 % nk, S1 // nk
    first_full = (S0 + nk - 1) // nk
    nfull = S1 // nk - first_full
    has_head, has_tail = int(e > 0), int(b > 0)
    n_items = has_head + nfull + dp_rounds + has_tail
    pos_tail = has_head + (1 if nfull + dp_rounds > 0 else 0)
    out = []
    for n in range(n_items):
        if n < has_head:
            out.append((skb + tile_h, 0, e, 1))
        elif has_tail and n == pos_tail:
            out.append((skb + tile_t, b, nk, 2))
        else:
            w = n - has_head - (1 if has_tail and n > pos_tail else 0)
            t = skb + first_full + w if w < nfull else j + W * (w - nfull)
            out.append((t, 0, nk, 0))
    return out


@pytest.mark.parametrize("W", [1, 4, 32])
@pytest.mark.parametrize("nk", [1, 2, 3, 12, 48])
def test_streamk_items_partition_the_work(W, nk):
    for cnt in itertools.chain(range(W, 4 * W + 3), (73, 74, 220, 294) if W == 32 else ()):
        cover = {}
        iters = []
        heads, tails = {}, {}
        for j in range(W):
            its = items_of(j, W, cnt, nk)
            assert its and its[0][3] != 2, (W, cnt, nk, j)                     # a tail fragment is never a workgroup's first item
            assert sum(1 for it in its if it[3] == 1) <= 1 and sum(1 for it in its if it[3] == 2) <= 1
            if any(it[3] == 1 for it in its):
                assert its[0][3] == 1                                          # the published fragment is the first thing a workgroup does
            iters.append(sum(ke - kb for _, kb, ke, _ in its))
            for t, kb, ke, kind in its:
                assert 0 <= t < cnt and 0 <= kb < ke <= nk
                for k in range(kb, ke):
                    assert (t, k) not in cover, (W, cnt, nk, j, t, k)
                    cover[(t, k)] = j
                if kind == 1:
                    heads[t] = (j, ke)
                if kind == 2:
                    tails[t] = (j, kb)
        assert len(cover) == cnt * nk                                          # everything, exactly once
        assert max(iters) - min(iters) <= 1
        assert set(heads) == set(tails)
        for t, (jh, e) in heads.items():
            jt, b = tails[t]
            assert jt == jh + 1 and b == e                                     # the chain continues where it stopped, in the next workgroup
