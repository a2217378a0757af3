"""Stress of the fence-free "last workgroup finishes" reductions (csrc/osq_device.h: partials published with agent-scope
stores, drained, sharded arrival tickets, partials read back with agent-scope loads) and of the token selection's
rendezvous word: about 10^5 launches of observe_flat, lsq_backward and observe_tokens (token_minmax + token_select) over
two streams with changing sizes -- hence changing grids and ticket shard counts -- every single result compared with a
stock torch reduction on the same stream.  A stale partial, a lost ticket or a counter left non-zero shows up as one
wrong statistic among thousands."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_last_block_reductions_under_load(sum_tier):
    """Both tiers of the LSQ+ backward's sums: the order-free kernel (float64 partials, one ticket) at 1e-5 of the float64
    torch sum, and the default reference-order kernel (aten_order.h: published chunk sums, ticket, serial upper levels) --
    autograd's TWO fp32 sums (sum g_in, sum -g_mul) cancel, so its bound is relative to their magnitudes, which a stale
    chunk sum or a lost ticket still exceeds by orders of magnitude."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    strict = sum_tier == "reference-order"
    from outlier_suppression_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(123)
    big = torch.randn(1 << 22, device=dev)
    big[torch.randint(0, big.numel(), (4096,), device=dev)] *= 50.0
    gy_all = torch.randn(1 << 22, device=dev)
    sizes = [int(v) for v in torch.randint(1000, 1 << 22, (64,), generator=gen)] + [256, 1024, 4096, 1 << 20, (1 << 22) - 3]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    lens = [torch.randint(1, 65, (16,), generator=gen).to(dev) for _ in range(8)]
    torch.cuda.synchronize()
    rounds = 150 if strict else 450    # x 2 streams x 69 sizes x (1 + 1/3 + 2/5) launches of this library: 1.08e5, each beside its stock reductions
    bad = [torch.zeros(3, dtype=torch.int64, device=dev) for _ in streams]
    s1 = torch.tensor([0.731], device=dev)
    z1 = torch.tensor([29.0], device=dev)
    launches = 0
    for r in range(rounds):
        for si, st in enumerate(streams):
            with torch.cuda.stream(st):
                for k, n in enumerate(sizes):
                    off = ((r * 977 + k * 131 + si * 17) % (big.numel() - n)) & ~3       # 16-byte aligned start
                    x = big[off:off + n]
                    # 1. flat observer (running min / max from the untouched state)
                    mn = torch.full((), float("inf"), device=dev)
                    mx = torch.full((), float("-inf"), device=dev)
                    ops.observe_flat(x, ops.UPDATE_RUNNING, 0, mn, mx, 0, 63, False)
                    bad[si][0] += ((mn != x.min()) | (mx != x.max())).long()
                    launches += 1
                    if (r + k) % 3 == 0:
                        # 2. LSQ+ backward: dx exact, the zero-point sum against a float64 torch sum
                        m = min(n, 1 << 18)
                        xs, gy = x[:m], gy_all[off:off + m]
                        dx, ds, dz = ops.lsq_backward_per_tensor(xs, gy, s1, z1, 0, 63, ops.PARAM_LSQPLUS, 1.0)
                        xi = torch.round(xs / s1) + z1
                        inside = (xi >= 0) & (xi <= 63)
                        g_mul = gy * s1
                        ref_dz = (torch.where(inside, g_mul, torch.zeros_like(gy)) - g_mul).double().sum()
                        bad[si][1] += (dx != torch.where(inside, g_mul, torch.zeros_like(gy)) / s1).long().sum()
                        bound = 1e-5 * (g_mul.abs().double().sum() if strict else ref_dz.abs()) + 1e-6
                        bad[si][1] += ((dz.double().sum() - ref_dz).abs() > bound).long()
                        launches += 1
                    if (r + k) % 5 == 0:
                        # 3. masked token observer: per-token extrema + two-workgroup selection with its rendezvous word
                        B, T, H = 16, 64, 64
                        o2 = (off % (big.numel() - B * T * H)) & ~3
                        t = big[o2:o2 + B * T * H].view(B, T, H)
                        L = lens[(r + k) % len(lens)]
                        mn2 = torch.full((), float("inf"), device=dev)
                        mx2 = torch.full((), float("-inf"), device=dev)
                        ops.observe_tokens(t, 1, L, False, 1.0, ops.UPDATE_RUNNING, 0, mn2, mx2, 0, 63, False)
                        valid = torch.arange(T, device=dev)[None, :] < L[:, None]
                        v = t[valid]
                        bad[si][2] += ((mn2 != v.min()) | (mx2 != v.max())).long()
                        launches += 2
    torch.cuda.synchronize()
    total = (bad[0] + bad[1]).cpu().tolist()
    assert launches >= (30000 if strict else 100000)
    assert total == [0, 0, 0], total
