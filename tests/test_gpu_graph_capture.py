"""The default dispatch of a large batch is asynchronous and capturable: the spread sample's verdict stays
on the device, both routes are enqueued and the one whose turn it is not returns at once.  One captured graph is replayed on a
spread batch (bucketed route) and on a windowed one (direct route) in the same buffers; both must answer like the direct kernel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rank_and_select_in_default_mode_capture_into_a_graph(gpu):
    import torch
    dev = torch.device("cuda:0")
    n_bits = 448 * (1 << 22) + 12345  # 2^22 rank lines: large enough for the automatic choice to consider the passes
    g = torch.Generator(device=dev).manual_seed(5)
    words = torch.randint(-2**63, 2**63 - 1, ((n_bits + 63) // 64,), device=dev, dtype=torch.int64, generator=g)
    bv = gpu.bit_vector(words, n_bits)
    del words
    nq = 9_000_000  # >= 2 x lines
    idx = torch.randint(0, n_bits + 1, (nq,), device=dev, dtype=torch.int64, generator=g)
    out = torch.empty_like(idx)
    ones = bv.ones()
    sel = torch.randint(1, ones + 1, (nq,), device=dev, dtype=torch.int64, generator=g)
    sout = torch.empty_like(sel)
    gpu.set_option("rank_sorted", -1)
    gpu.set_option("select_sorted", -1)
    # warm-up outside the capture: the handle's scratch gets its size, the select plan is built
    bv.rank(idx, 1, out)
    bv.select(sel, 1, sout)
    torch.cuda.synchronize()
    # the warm-up took the bucketed route (a spread batch)
    gpu.set_option("trace_phases", 1)
    bv.rank(idx, 1, out)
    assert gpu.last_phases().get("select") == 0, "a spread batch did not take the passes"
    gpu.set_option("trace_phases", 0)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        bv.rank(idx, 1, out)
        bv.select(sel, 1, sout)

    def direct():
        gpu.set_option("rank_sorted", 0)
        gpu.set_option("select_sorted", 0)
        try:
            return bv.rank(idx, 1).clone(), bv.select(sel, 1).clone()
        finally:
            gpu.set_option("rank_sorted", -1)
            gpu.set_option("select_sorted", -1)

    # 1: a fresh spread batch in the captured buffers
    idx.copy_(torch.randint(0, n_bits + 1, (nq,), device=dev, dtype=torch.int64, generator=g))
    sel.copy_(torch.randint(1, ones + 1, (nq,), device=dev, dtype=torch.int64, generator=g))
    out.fill_(-7)
    sout.fill_(-7)
    graph.replay()
    torch.cuda.synchronize()
    want_r, want_s = direct()
    assert torch.equal(out, want_r) and torch.equal(sout, want_s)
    # 2: a windowed batch: the same graph now answers through the direct kernels
    idx.copy_(n_bits // 3 + torch.randint(0, 1 << 16, (nq,), device=dev, dtype=torch.int64, generator=g))
    sel.copy_(ones // 3 + torch.randint(0, 1 << 15, (nq,), device=dev, dtype=torch.int64, generator=g))
    out.fill_(-7)
    sout.fill_(-7)
    graph.replay()
    torch.cuda.synchronize()
    want_r, want_s = direct()
    assert torch.equal(out, want_r) and torch.equal(sout, want_s)
    # and outside a graph the windowed batch reports no passes
    gpu.set_option("trace_phases", 1)
    bv.rank(idx, 1, out)
    assert not gpu.last_phases(), "a windowed batch went through the passes"
    gpu.set_option("trace_phases", 0)
    bv.release_scratch()
