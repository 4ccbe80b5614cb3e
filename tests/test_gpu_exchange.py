"""nnue_dp_factor_pack / nnue_dp_factor_unpack (include/nnue_hip.h): the all-gathered chunks of the factor exchange
rebuild the global byte map, sink and d_ft exactly, and the small gradients come out as the ranks' sum in rank order
(the gradient of the mean loss over the global batch, train.py:359-366).  Ranks are simulated on one GPU by packing each
rank's chunk into its row of the receive buffer.  ``-m gpu``."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("world,b,p,l1,f", [(1, 8, 968, 256, 800), (2, 16, 968, 64, 800), (3, 5, 36, 8, 36), (2, 12, 4096, 128, 4096),
                                            (4, 7, 132, 12, 100), (8, 4, 65536, 64, 65536)])
def test_pack_gather_unpack_round_trip(world, b, p, l1, f):
    from nnue_hip import lib
    lib.load()
    gen = torch.Generator().manual_seed(world * 1000 + p)
    head, count = 20, 20 + f * l1 + l1 + 52
    direct = min(f - 1, p)
    tail_lo = head + direct * l1
    tail = count - tail_lo
    ranks = [lib.FactorExchange(world, r, b, p, f, l1, head=head, tail_lo=tail_lo, tail=tail, device=DEV) for r in range(world)]
    maps, dfts, sinks, grads = [], [], [], []
    for r, fx in enumerate(ranks):
        bits = (torch.rand(b, p, generator=gen) < 0.43).to(torch.uint8)
        if r == 0:
            bits[0].fill_(1)
            bits[-1].zero_()
        g = torch.randn(count, generator=gen)
        fm = lib.FeatureMatrix(bits.to(DEV), torch.zeros(b, dtype=torch.int32, device=DEV), fx.sink, torch.empty(16, dtype=torch.uint8, device=DEV), p, f)
        fx.d_ft.copy_(torch.randn(b, l1, generator=gen))
        fx.sink.copy_(torch.rand(b, generator=gen) * 9)
        g_dev = g.to(DEV)
        fx.pack(fm, g_dev)
        maps.append(bits), dfts.append(fx.d_ft.cpu().clone()), sinks.append(fx.sink.cpu().clone()), grads.append(g)
    recv = ranks[0]
    for r in range(1, world):  # what the all-gather does
        recv.chunks[r].copy_(ranks[r].chunks[r])
    flat = grads[0].to(DEV).clone()
    before = flat.clone()
    recv.unpack(flat)
    torch.cuda.synchronize()
    assert torch.equal(recv.g_fm.bits.cpu(), torch.cat(maps))
    assert torch.equal(recv.g_dft.cpu(), torch.cat(dfts)) and torch.equal(recv.g_fm.sink.cpu(), torch.cat(sinks))
    want = torch.zeros(count)
    for g in grads:  # rank order, from zero: the kernel's own order
        want = want + g
    got = flat.cpu()
    assert torch.equal(got[:head], want[:head]) and torch.equal(got[tail_lo:], want[tail_lo:])
    assert torch.equal(got[head:tail_lo], before.cpu()[head:tail_lo])  # the product's rows are not touched


def test_argument_errors():
    from nnue_hip import lib
    L = lib.load()
    assert L.nnue_dp_factor_chunk_bytes(128, 65536, 1024, 170000) == 128 * 1024 * 4 + 512 + 680000 + 128 * 8192
    assert L.nnue_dp_factor_offset(3, 128, 65536, 1024, 170000) == 128 * 1024 * 4 + 512 + 680000
    assert L.nnue_dp_factor_chunk_bytes(0, 4, 4, 0) == 0 and L.nnue_dp_factor_offset(7, 4, 4, 4, 0) == -1
    x = torch.zeros(64, dtype=torch.uint8, device=DEV)
    g = torch.zeros(64, device=DEV)
    assert L.nnue_dp_factor_pack(x.data_ptr(), g.data_ptr(), 0, 0, 0, 2, 6, 4, x.data_ptr(), None) == -1  # P not a multiple of 4
    assert L.nnue_dp_factor_unpack(x.data_ptr(), 0, 2, 8, 4, 0, 0, 0, x.data_ptr(), g.data_ptr(), g.data_ptr(), g.data_ptr(), None) == -1
