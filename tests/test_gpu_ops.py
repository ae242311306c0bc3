"""Single HIP operators (through the C ABI) against plain fp32 torch on the CPU.
Tolerance: rtol 1e-4 with an absolute floor scaled to the tensor (tests/util.py)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from tests.util import check_close

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def check_fp32_accuracy(got, fp32_ref, exact, what):
    """The kernels that run their products on the bf16 matrix instruction over operands cut into three bf16 pieces
    claim fp32 accuracy.  Held to the float64 result `exact`, `got` must be as close as plain fp32 arithmetic
    (`fp32_ref`: torch on the CPU, same operands) gets: largest error relative to the tensor's largest magnitude at
    most 1.1 times that of the fp32 reference, plus 5e-8 (a tenth of one fp32 rounding)."""
    exact = exact.double().cpu()
    scale = max(float(exact.abs().max()), 1e-30)
    err_gpu = float((got.detach().double().cpu().reshape(exact.shape) - exact).abs().max()) / scale
    err_f32 = float((fp32_ref.detach().double().cpu().reshape(exact.shape) - exact).abs().max()) / scale
    print(f'{what}: max error vs float64 / max magnitude: GPU {err_gpu:.2e}, fp32 torch {err_f32:.2e} ({err_gpu / max(err_f32, 1e-30):.2f}x)')
    assert err_gpu <= 1.1 * err_f32 + 5e-8, (what, err_gpu, err_f32)


@pytest.mark.parametrize('M,N,K,relu', [
    (2964, 300, 200, False), (2964, 300, 300, True), (128, 400, 1200, False), (128, 112, 400, False),
    (1, 5, 7, True), (67, 42, 16, False), (3000, 4, 656, False), (1408, 302, 200, False), (130, 66, 130, True),
    # aligned operands whose reduction range is shorter than one 32-wide K-step in one of the three products (the
    # prefetch then runs past the range from its first tile on): 1, 5, 6 and 20 rows, 4 and 8 input features
    (1, 8, 8, False), (6, 400, 1200, False), (5, 300, 300, True), (20, 64, 4, False), (33, 128, 36, True), (64, 4, 4, False),
])
def test_linear_forward_backward(M, N, K, relu):
    from ps_vae_amd import ops
    x, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3, scale=0.1)
    dy = _rand(M, N, seed=4)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.linear(xr, wr, br)
    yr = torch.relu(yr) if relu else yr
    yr.backward(dy)
    xg, wg, bg = (t.to(dev()).requires_grad_() for t in (x, w, b))
    y = ops.linear(xg, wg, bg, relu=relu)
    y.backward(dy.to(dev()))
    check_close(y, yr, 'y')
    check_close(xg.grad, xr.grad, 'dx')
    check_close(wg.grad, wr.grad, 'dw')
    check_close(bg.grad, br.grad, 'db')


def test_linear_strided_mask_accumulate_and_segments():
    """The C entry points directly: leading dimensions, ReLU-mask epilogue, accumulation, summed segments."""
    from ps_vae_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    M, N, K, t = 333, 300, 300, 4
    dy_all = _rand(t, M, N, seed=5)
    x_all = _rand(t, M, K, seed=6)
    w = _rand(N, K, seed=7, scale=0.05)
    mask = _rand(M, K, seed=8)
    prev = _rand(M, 2 * K, seed=9)
    # dx written into a [M, 2K] buffer at column offset K with mask and accumulate
    d_dy, d_w, d_mask, d_prev = dy_all[0].to(dev()), w.to(dev()), mask.to(dev()), prev.to(dev())
    _lib.check(L.psvae_linear_backward_input(d_dy.data_ptr(), N, d_w.data_ptr(), d_prev.data_ptr() + 4 * K, 2 * K,
                                             M, N, K, d_mask.data_ptr(), K, 1, st))
    want = prev.clone()
    want[:, K:] += (dy_all[0] @ w) * (mask > 0)
    check_close(d_prev, want, 'dx strided/masked/accumulated')
    # dw summed over t segments + bias gradient
    d_dyall, d_xall = dy_all.to(dev()), x_all.to(dev())
    dw = torch.empty(N, K, device=dev())
    db = torch.empty(N, device=dev())
    nbytes = L.psvae_linear_scratch_bytes(M, N, K, t)
    sc = torch.empty(nbytes, dtype=torch.uint8, device=dev())
    _lib.check(L.psvae_linear_backward_weight(d_dyall.data_ptr(), N, M * N, d_xall.data_ptr(), K, M * K, t, dw.data_ptr(),
                                              db.data_ptr(), M, N, K, sc.data_ptr(), nbytes, st))
    check_close(dw, sum(dy_all[i].t() @ x_all[i] for i in range(t)), 'dw over segments')
    check_close(db, dy_all.sum((0, 1)), 'db over segments')


@pytest.mark.parametrize('bias', [True, False])
@pytest.mark.parametrize('scratch_frac', [1.0, 0.3, 0.0])
def test_weight_gradient_with_any_scratch_size(bias, scratch_frac):
    """Weight gradient at tile-aligned sizes, with and without the bias column, with the scratch buffer the
    size query asks for, a smaller one (a smaller split must be chosen) and none (no split)."""
    from ps_vae_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    M, N, K = 5000, 128, 192
    dy, x = _rand(M, N, seed=21), _rand(M, K, seed=22)
    d_dy, d_x = dy.to(dev()), x.to(dev())
    dw = torch.full((N, K), float('nan'), device=dev())
    db = torch.full((N,), float('nan'), device=dev())
    nbytes = int(L.psvae_linear_scratch_bytes(M, N, K, 1) * scratch_frac) // 16 * 16
    sc = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev())
    _lib.check(L.psvae_linear_backward_weight(d_dy.data_ptr(), N, 0, d_x.data_ptr(), K, 0, 1, dw.data_ptr(),
                                              db.data_ptr() if bias else None, M, N, K,
                                              sc.data_ptr() if nbytes else None, nbytes, st))
    check_close(dw, dy.t() @ x, 'dw', rtol=2e-4, atol_scale=2e-5)
    if bias:
        check_close(db, dy.sum(0), 'db', rtol=2e-4, atol_scale=2e-5)


def _random_graph(n, E, seed):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, E), generator=g)
    et = torch.randint(1, 4, (E,), generator=g)
    return ei, et


@pytest.mark.parametrize('n,E,H', [(500, 1100, 300), (64, 0, 20), (1, 3, 8), (2000, 9000, 512),
                                   # outputs beyond 16 MB: the kernels write them with non-temporal stores
                                   (16001, 40000, 300), (14000, 60000, 320), (24000, 0, 200), (70000, 90000, 64)])
def test_gine_aggregate_forward_backward(n, E, H):
    from ps_vae_amd import ops
    ei, et = _random_graph(n, E, seed=n)
    x, ew, eb = _rand(n, H, seed=1), _rand(H, 4, seed=2), _rand(H, seed=3, scale=0.1)
    dout = _rand(n, H, seed=4)
    xr, ewr, ebr = x.clone().requires_grad_(), ew.clone().requires_grad_(), eb.clone().requires_grad_()
    e = F.linear(F.one_hot(et, 4).float(), ewr, ebr)
    msg = torch.relu(xr.index_select(0, ei[0]) + e)
    ref = torch.zeros_like(xr).index_add_(0, ei[1], msg) + 1.25 * xr
    ref.backward(dout)
    xg, ewg, ebg = (t.to(dev()).requires_grad_() for t in (x, ew, eb))
    csr = ops.Csr(ei.to(dev()), et.to(dev()), n)
    out = ops.GineAggregateFn.apply(xg, ewg, ebg, csr, 1.25)
    out.backward(dout.to(dev()))
    check_close(out, ref, 'aggregate')
    check_close(xg.grad, xr.grad, 'dx')
    check_close(ewg.grad, ewr.grad if E else torch.zeros_like(ew), 'd edge_w')
    check_close(ebg.grad, ebr.grad if E else torch.zeros_like(eb), 'd edge_b')


def test_gru_forward_stamped_is_the_same_recurrence():
    """The instrumented multi-CU forward kernel writes the same bits as psvae_gru_forward and four rising clock values
    per time step; shapes outside its one kernel form are refused."""
    from ps_vae_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, T, R = 100, 7, 200
    gi, h0 = _rand(B, T, 3 * R, seed=1).to(dev()), _rand(B, R, seed=2).to(dev())
    whh, bhh = _rand(3 * R, R, seed=3, scale=R ** -0.5).to(dev()), _rand(3 * R, seed=4).to(dev())
    nx = L.psvae_gru_exchange_bytes(B, T, R)
    exch = torch.zeros(nx // 8 + 1, dtype=torch.int64, device=dev())
    outs = []
    for stamped in (False, True):
        hseq, hprev = torch.empty(B, T, R, device=dev()), torch.empty(B, T, R, device=dev())
        gates = torch.empty(B, T, 4 * R, device=dev())
        stamps = torch.zeros(T, 4, dtype=torch.int64, device=dev())
        args = (gi.data_ptr(), h0.data_ptr(), whh.data_ptr(), bhh.data_ptr(), hseq.data_ptr(), hprev.data_ptr(), gates.data_ptr(),
                B, T, R, exch.data_ptr(), nx)
        if stamped:
            _lib.check(L.psvae_gru_forward_stamped(*args, stamps.data_ptr(), st))
        else:
            _lib.check(L.psvae_gru_forward(*args, st))
        torch.cuda.synchronize()
        outs.append((hseq, hprev, gates))
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)
    flat = stamps.cpu().reshape(-1)
    assert bool((flat[1:] >= flat[:-1]).all()) and int(flat[-1] - flat[0]) > 0
    assert L.psvae_gru_forward_stamped(*args[:9], 256, exch.data_ptr(), nx, stamps.data_ptr(), st) == -2      # PSVAE_E_SHAPE


def test_segment_pool_forward_backward():
    from ps_vae_amd import ops
    lens = torch.tensor([3, 1, 7, 38, 12, 1])
    off = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)])
    n, W = int(off[-1]), 1200
    gid = torch.repeat_interleave(torch.arange(len(lens)), lens)
    x = _rand(n, W, seed=1)
    dp = _rand(len(lens), W, seed=2)
    xr = x.clone().requires_grad_()
    ref = torch.zeros(len(lens), W).index_add_(0, gid, xr)
    ref.backward(dp)
    xg = x.to(dev()).requires_grad_()
    out = ops.SegmentPoolFn.apply(xg, off.to(torch.int32).to(dev()), gid.to(torch.int32).to(dev()))
    out.backward(dp.to(dev()))
    check_close(out, ref, 'pooled')
    check_close(xg.grad, xr.grad, 'dx')


@pytest.mark.parametrize('multi_cu', [False, True])
@pytest.mark.parametrize('B,T,R', [(128, 11, 200), (5, 3, 16), (33, 1, 20), (16, 7, 64), (300, 9, 200), (47, 13, 256),
                                   (700, 4, 24),
                                   # batches whose sequence tiles outnumber the resident workgroup groups: a workgroup
                                   # steps 2 or 4 tiles together (last group ragged), wide rows take 2 at most;
                                   # beyond 768 sequences the library takes the single-CU kernels by itself
                                   (600, 6, 200), (750, 4, 200), (330, 4, 256), (1024, 3, 200)])
def test_gru_forward_backward(B, T, R, multi_cu):
    from ps_vae_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    nx = L.psvae_gru_exchange_bytes(B, T, R) if multi_cu else 0
    # a dirty exchange buffer: the library must not rely on its previous contents
    exch = torch.full((max(nx, 16) // 4,), 7, dtype=torch.int32, device=dev())
    xp = exch.data_ptr() if multi_cu else None
    gi, h0 = _rand(B, T, 3 * R, seed=1), _rand(B, R, seed=2)
    whh, bhh = _rand(3 * R, R, seed=3, scale=R ** -0.5), _rand(3 * R, seed=4, scale=0.1)
    dh = _rand(B, T, R, seed=5)
    gir, h0r, whr, bhr = (t.clone().requires_grad_() for t in (gi, h0, whh, bhh))
    h, outs = h0r, []
    for t in range(T):
        gh = F.linear(h, whr, bhr)
        r = torch.sigmoid(gir[:, t, :R] + gh[:, :R])
        u = torch.sigmoid(gir[:, t, R:2 * R] + gh[:, R:2 * R])
        c = torch.tanh(gir[:, t, 2 * R:] + r * gh[:, 2 * R:])
        h = (1 - u) * c + u * h
        outs.append(h)
    ref = torch.stack(outs, 1)
    ref.backward(dh)
    d = dev()
    g_gi, g_h0, g_whh, g_bhh, g_dh = (t.to(d).contiguous() for t in (gi, h0, whh, bhh, dh))
    hseq, hprev = torch.empty(B * T, R, device=d), torch.empty(B * T, R, device=d)
    gates = torch.empty(B * T, 4 * R, device=d)
    _lib.check(L.psvae_gru_forward(g_gi.data_ptr(), g_h0.data_ptr(), g_whh.data_ptr(), g_bhh.data_ptr(), hseq.data_ptr(),
                                   hprev.data_ptr(), gates.data_ptr(), B, T, R, xp, nx, st))
    check_close(hseq.view(B, T, R), ref, 'gru output')
    dgi, dgh, dh0 = torch.empty(B * T, 3 * R, device=d), torch.empty(B * T, 3 * R, device=d), torch.empty(B, R, device=d)
    _lib.check(L.psvae_gru_backward(g_dh.data_ptr(), hprev.data_ptr(), gates.data_ptr(), g_whh.data_ptr(), dgi.data_ptr(),
                                    dgh.data_ptr(), dh0.data_ptr(), B, T, R, xp, nx, st))
    if multi_cu and (B + 15) // 16 <= 48:      # (larger batches run the single-CU kernels, which leave the buffer alone)
        assert int(exch[0]) == 0, 'a workgroup of the multi-CU recurrence gave up waiting'
    check_close(dgi.view(B, T, 3 * R), gir.grad, 'd gi')
    check_close(dh0, h0r.grad, 'd h0')
    check_close(dgh.view(B * T, 3 * R).t().cpu() @ hprev.cpu(), whr.grad, 'd whh (from dgh, hprev)')
    check_close(dgh.sum(0), bhr.grad, 'd bhh')


def test_softmax_ce_rows():
    from ps_vae_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, Lseq, V2, pad = 9, 6, 302, 300
    T = Lseq - 1
    logits = _rand(B * T, 304, seed=1, scale=4.0)
    g = torch.Generator().manual_seed(2)
    pieces = torch.randint(0, V2, (B, Lseq), generator=g)
    pieces[2, 3:] = pad
    pieces[5, 1:] = pad
    d = dev()
    row, lse = torch.empty(B * T, device=d), torch.empty(B * T, device=d)
    lg, pc = logits.to(d), pieces.to(torch.int32).to(d)
    _lib.check(L.psvae_softmax_ce_forward(lg.data_ptr(), 304, pc.data_ptr(), Lseq, 1, T, B * T, V2, pad, row.data_ptr(),
                                          lse.data_ptr(), st))
    gold = pieces[:, 1:].reshape(-1)
    want = F.cross_entropy(logits[:, :V2], gold, ignore_index=pad, reduction='none')
    check_close(row, want, 'row losses')
    check_close(lse, torch.logsumexp(logits[:, :V2], -1), 'lse')


def test_clip_adam_matches_torch():
    from ps_vae_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    n = 100003
    p0, g0 = _rand(n, seed=1), _rand(n, seed=2, scale=0.2)   # norm ~ 36 > 10 -> clipping active
    pr = p0.clone().requires_grad_()
    opt = torch.optim.Adam([pr], lr=1e-3)
    d = dev()
    p, m, v = p0.to(d), torch.zeros(n, device=d), torch.zeros(n, device=d)
    part = torch.zeros(L.psvae_adam_partials(n), device=d)
    norm = torch.zeros(1, device=d)
    for step in range(1, 4):
        gr = g0 * step
        pr.grad = gr.clone()
        want_norm = torch.nn.utils.clip_grad_norm_([pr], 10.0)
        opt.step()
        g = gr.to(d)
        _lib.check(L.psvae_clip_adam(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, part.data_ptr(), 10.0, 1e-3,
                                     0.9, 0.999, 1e-8, 1 - 0.9 ** step, 1 - 0.999 ** step, norm.data_ptr(), st))
        assert abs(float(norm) - float(want_norm)) <= 1e-5 * float(want_norm)
        check_close(p, pr.detach(), f'params after step {step}', rtol=1e-6, atol_scale=1e-6)
        # the updates themselves (parameters minus start; ~ step * lr each): a constant formed in fp32, such as
        # 1.f - 0.999f, scales all of them by some 1e-5 and shows in a loss curve after one step
        upd, want_upd = (p.cpu() - p0).double(), (pr.detach() - p0).double()
        assert float((upd - want_upd).abs().max()) <= 2e-6 * float(want_upd.abs().max()) + 6e-7, step      # (+ fp32 spacing of p)
        assert abs(float(upd.abs().sum()) / float(want_upd.abs().sum()) - 1.0) <= 2e-6, step


@pytest.mark.parametrize('n,E,H,save', [(2964, 6400, 300, True), (95, 160, 300, False), (33, 0, 64, True), (1, 0, 16, True),
                                        (700, 5000, 512, True), (5000, 9000, 328, False), (40, 400, 20, True),
                                        (24700, 52000, 300, True)])     # (enough row tiles for 64 nodes per workgroup)
def test_gine_layer_fused_forward(n, E, H, save):
    """One GINE application in one launch (aggregation -> Linear + ReLU -> Linear) against plain fp32 torch; the
    copies of agg and hid it leaves for backward must equal the intermediate results; graphs with nodes of many
    bonds (more than the four fetched per round, more than the eight staged per node), with none, ragged last
    row tile."""
    from ps_vae_amd import _lib, ops
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    ei, et = _random_graph(n, E, seed=n + 1)
    x, ew, eb = _rand(n, H, seed=1), _rand(H, 4, seed=2), _rand(H, seed=3, scale=0.1)
    w1, b1 = _rand(H, H, seed=5, scale=H ** -0.5), _rand(H, seed=6, scale=0.1)
    w2, b2 = _rand(H, H, seed=7, scale=H ** -0.5), _rand(H, seed=8, scale=0.1)
    e = F.linear(F.one_hot(et, 4).float(), ew, eb)
    agg_ref = torch.zeros_like(x).index_add_(0, ei[1], torch.relu(x.index_select(0, ei[0]) + e)) + 1.5 * x
    hid_ref = torch.relu(F.linear(agg_ref, w1, b1))
    y_ref = F.linear(hid_ref, w2, b2)
    csr = ops.Csr(ei.to(dev()), et.to(dev()), n)
    d = [t.to(dev()).contiguous() for t in (x, ew, eb, w1, b1, w2, b2)]
    ldy = H + 8
    y = torch.full((n, ldy), float('nan'), device=dev())
    agg = torch.full((n, H), float('nan'), device=dev()) if save else None
    hid = torch.full((n, H), float('nan'), device=dev()) if save else None
    nbytes = L.psvae_gine_layer_scratch_bytes(H)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev())
    _lib.check(L.psvae_gine_layer_forward(d[0].data_ptr(), H, csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.etype.data_ptr(),
                                          d[1].data_ptr(), d[2].data_ptr(), 1.5, d[3].data_ptr(), d[4].data_ptr(),
                                          d[5].data_ptr(), d[6].data_ptr(), agg.data_ptr() if save else None,
                                          hid.data_ptr() if save else None, y.data_ptr(), ldy, n, H, scratch.data_ptr(), nbytes, st))
    check_close(y[:, :H], y_ref, 'fused layer output')
    assert bool(torch.isnan(y[:, H:]).all()), 'columns past H of the output rows must stay untouched'
    # fp32 accuracy: the products run on the bf16 matrix instruction over operands cut into three bf16 pieces; held to
    # the float64 result, the output must be as close as plain fp32 arithmetic (torch on the CPU) gets, not merely
    # within the comparison tolerance above
    x6, w16, b16, w26, b26 = (t.double() for t in (x, w1, b1, w2, b2))
    e6 = F.linear(F.one_hot(et, 4).double(), ew.double(), eb.double())
    agg6 = torch.zeros_like(x6).index_add_(0, ei[1], torch.relu(x6.index_select(0, ei[0]) + e6)) + 1.5 * x6
    y64 = F.linear(torch.relu(F.linear(agg6, w16, b16)), w26, b26)
    scale = float(y64.abs().max())
    err_gpu = float((y[:, :H].double().cpu() - y64).abs().max()) / scale
    err_f32 = float((y_ref.double() - y64).abs().max()) / scale
    print(f'fused layer n={n} H={H}: max error vs float64 / max|y|: GPU {err_gpu:.2e}, fp32 torch {err_f32:.2e}')
    assert err_gpu <= 1.1 * err_f32 + 5e-8, (err_gpu, err_f32)
    if save:
        check_close(agg, agg_ref, 'saved aggregate')
        check_close(hid, hid_ref, 'saved hidden')
    # same aggregate as the stand-alone kernel, bit for bit (same sum order)
    if save:
        alone = ops.GineAggregateFn.apply(d[0], d[1], d[2], csr, 1.5)
        assert torch.equal(alone, agg)
    assert L.psvae_gine_layer_forward(d[0].data_ptr(), H, csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.etype.data_ptr(),
                                      d[1].data_ptr(), d[2].data_ptr(), 1.5, d[3].data_ptr(), d[4].data_ptr(), d[5].data_ptr(),
                                      d[6].data_ptr(), None, hid.data_ptr() if save else y.data_ptr(), y.data_ptr(), ldy, n, H,
                                      scratch.data_ptr(), nbytes, st) == -1      # agg and hid: both or neither
    assert L.psvae_gine_layer_forward(d[0].data_ptr(), H, csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.etype.data_ptr(),
                                      d[1].data_ptr(), d[2].data_ptr(), 1.5, d[3].data_ptr(), d[4].data_ptr(), d[5].data_ptr(),
                                      d[6].data_ptr(), None, None, y.data_ptr(), ldy, n, H, scratch.data_ptr(), nbytes - 16,
                                      st) == -4      # scratch too small


def _scorer_weights(P, P2):
    return (_rand(P2, P, seed=3, scale=P ** -0.5), _rand(P2, seed=4, scale=0.1), _rand(P2, P2, seed=5, scale=P2 ** -0.5),
            _rand(P2, seed=6, scale=0.1), _rand(P, P2, seed=7, scale=P2 ** -0.5), _rand(P, seed=8, scale=0.1),
            _rand(4, P, seed=9, scale=P ** -0.5), _rand(4, seed=10, scale=0.1))


@pytest.mark.parametrize('S,n,B', [(3000, 2964, 128), (17, 40, 3), (1, 5, 1), (4113, 600, 33), (16001, 11000, 480)])
def test_scorer_fused_forward(S, n, B):
    """The link scorer's forward pass in one launch (gather, three Linear layers, Linear(P, 4), cross-entropy
    rows) against plain fp32 torch, including every activation it leaves behind for backward."""
    from ps_vae_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    H, Z = 300, 56
    P, P2 = 2 * H + Z, (2 * H + Z) // 2
    g = torch.Generator().manual_seed(S)
    emb, z = _rand(n, H, seed=1), _rand(B, Z, seed=2)
    src, dst = torch.randint(0, n, (S,), generator=g), torch.randint(0, n, (S,), generator=g)
    gr, gold = torch.randint(0, B, (S,), generator=g), torch.randint(0, 4, (S,), generator=g)
    w0, b0, w1, b1, w2, b2, w3, b3 = _scorer_weights(P, P2)
    pin_ref = torch.cat([emb[src], emb[dst], z[gr]], -1)
    h1_ref = torch.relu(F.linear(pin_ref, w0, b0))
    h2_ref = torch.relu(F.linear(h1_ref, w1, b1))
    h3_ref = F.linear(h2_ref, w2, b2)
    lg_ref = F.linear(h3_ref, w3, b3)
    lse_ref = torch.logsumexp(lg_ref, -1)
    loss_ref = F.cross_entropy(lg_ref, gold, reduction='none')
    d = [t.to(dev()).contiguous() for t in (emb, z, w0, b0, w1, b1, w2, b2, w3, b3)]
    idx = [t.to(dev()).to(torch.int32).contiguous() for t in (src, dst, gr, gold)]
    out = [torch.full(shape, float('nan'), device=dev()) for shape in ((S, P), (S, P2), (S, P2), (S, P), (S, 4), (S,), (S,))]
    nbytes = L.psvae_scorer_scratch_bytes(H, Z)
    assert nbytes > 0 and L.psvae_scorer_scratch_bytes(512, 56) == 0
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev())
    _lib.check(L.psvae_scorer_forward(d[0].data_ptr(), H, d[1].data_ptr(), *[t.data_ptr() for t in idx],
                                      *[t.data_ptr() for t in d[2:]], *[t.data_ptr() for t in out], S, H, Z,
                                      scratch.data_ptr(), nbytes, st))
    for got, want, name in zip(out, (pin_ref, h1_ref, h2_ref, h3_ref, lg_ref, loss_ref, lse_ref),
                               ('pair_in', 'h1', 'h2', 'h3', 'logits', 'row_loss', 'lse')):
        check_close(got, want, name, atol_scale=2e-5)
    assert torch.equal(out[0].cpu(), pin_ref)           # the gather copies bits
    # fp32 accuracy of the three chained products on cut operands: h3 against its float64 value
    h1_64 = torch.relu(F.linear(pin_ref.double(), w0.double(), b0.double()))
    h2_64 = torch.relu(F.linear(h1_64, w1.double(), b1.double()))
    h3_64 = F.linear(h2_64, w2.double(), b2.double())
    check_fp32_accuracy(out[3], h3_ref, h3_64, f'scorer forward S={S}: h3')
    check_fp32_accuracy(out[1], h1_ref, h1_64, f'scorer forward S={S}: h1')


@pytest.mark.parametrize('S', [16001, 3000, 17, 1])
def test_scorer_fused_backward(S):
    """Backward of the link scorer's three Linear layers in ONE launch (`psvae_scorer_backward`) against plain fp32
    torch autograd on the same inputs, and its fp32 accuracy against float64.  16 001 pairs = 501 workgroups: the size
    from which the training step takes this kernel by itself."""
    from ps_vae_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    H, Z = 300, 56
    P, P2 = 2 * H + Z, (2 * H + Z) // 2
    w0, b0, w1, b1, w2, b2, _, _ = _scorer_weights(P, P2)
    pin = _rand(S, P, seed=11)
    dh3 = _rand(S, P, seed=12)

    h1 = torch.relu(F.linear(pin, w0, b0))
    h2 = torch.relu(F.linear(h1, w1, b1))
    # what autograd derives for the pre-activations of the two hidden layers and for the input
    x = pin.clone().requires_grad_()
    a1 = F.linear(x, w0, b0)
    a1.retain_grad()
    a2 = F.linear(torch.relu(a1), w1, b1)
    a2.retain_grad()
    F.linear(torch.relu(a2), w2, b2).backward(dh3)
    dh1_ref, dh2_ref, dpin_ref = a1.grad, a2.grad, x.grad
    # the float64 yardstick takes the SAME gates (the saved fp32 activations decide them, as in the kernel)
    dh2_64 = (dh3.double() @ w2.double()) * (h2 > 0)
    dh1_64 = (dh2_64 @ w1.double()) * (h1 > 0)
    dpin_64 = dh1_64 @ w0.double()
    d = [t.to(dev()).contiguous() for t in (dh3, h2, h1, w0, w1, w2)]
    out = [torch.full(shape, float('nan'), device=dev()) for shape in ((S, P2), (S, P2), (S, P))]
    nbytes = L.psvae_scorer_scratch_bytes(H, Z)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev())
    _lib.check(L.psvae_scorer_backward(*[t.data_ptr() for t in d], *[t.data_ptr() for t in out], S, H, Z,
                                       scratch.data_ptr(), nbytes, st), 'psvae_scorer_backward')
    for got, want, exact, name in zip(out, (dh2_ref, dh1_ref, dpin_ref), (dh2_64, dh1_64, dpin_64), ('d_h2', 'd_h1', 'd_pair_in')):
        check_close(got, want, name, atol_scale=2e-5)
        check_fp32_accuracy(got, want, exact, f'scorer backward S={S}: {name}')
    assert L.psvae_scorer_backward(*[t.data_ptr() for t in d], *[t.data_ptr() for t in out], S, H, Z, scratch.data_ptr(),
                                   nbytes - 16, st) == -4
    assert L.psvae_scorer_backward(*[t.data_ptr() for t in d], *[t.data_ptr() for t in out], S, 512, Z, scratch.data_ptr(),
                                   nbytes, st) == -2      # a size the one-launch kernels do not take


@pytest.mark.parametrize('n,H,rows', [(2964, 300, -1), (24700, 300, -1), (5928, 300, 64), (700, 300, 32), (95, 300, -1),
                                      (33, 64, -1), (1, 16, -1), (700, 512, -1), (5000, 328, -1), (40, 20, -1)])
def test_gine_layer_fused_backward(n, H, rows):
    """Backward of the two Linear layers of a GINE application in ONE launch (`psvae_gine_layer_backward`):
    d_hid = (g w2) * [hid > 0], d_agg = d_hid w1, against fp32 torch and, for accuracy, float64; 32- and 64-node
    workgroups as the library picks them by size and as option split_rows forces them."""
    from ps_vae_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    g, hid = _rand(n, H, seed=1), _rand(n, H, seed=2)
    w1, w2 = _rand(H, H, seed=5, scale=H ** -0.5), _rand(H, H, seed=7, scale=H ** -0.5)
    dh_ref = (g @ w2) * (hid > 0)
    da_ref = dh_ref @ w1
    dh_64 = (g.double() @ w2.double()) * (hid > 0)
    da_64 = dh_64 @ w1.double()
    d = [t.to(dev()).contiguous() for t in (g, hid, w1, w2)]
    dh = torch.full((n, H), float('nan'), device=dev())
    da = torch.full((n, H), float('nan'), device=dev())
    nbytes = L.psvae_gine_layer_scratch_bytes(H)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev())
    old = _lib.set_option('split_rows', rows)
    try:
        _lib.check(L.psvae_gine_layer_backward(*[t.data_ptr() for t in d], dh.data_ptr(), da.data_ptr(), n, H,
                                               scratch.data_ptr(), nbytes, st), 'psvae_gine_layer_backward')
    finally:
        _lib.set_option('split_rows', old)
    check_close(dh, dh_ref, 'd_hid')
    check_close(da, da_ref, 'd_agg')
    check_fp32_accuracy(dh, dh_ref, dh_64, f'GINE backward n={n} H={H}: d_hid')
    check_fp32_accuracy(da, da_ref, da_64, f'GINE backward n={n} H={H}: d_agg')
    assert L.psvae_gine_layer_backward(*[t.data_ptr() for t in d], dh.data_ptr(), None, n, H, scratch.data_ptr(), nbytes, st) == -1
    assert L.psvae_gine_layer_backward(*[t.data_ptr() for t in d], dh.data_ptr(), da.data_ptr(), n, H, scratch.data_ptr(),
                                       nbytes - 16, st) == -4


@pytest.mark.parametrize('nseg,M,N,K,bias', [(4, 5000, 300, 300, True), (1, 24000, 328, 656, True), (1, 17000, 656, 328, False),
                                             (2, 9000, 64, 200, True), (1, 16400, 600, 100, True), (1, 16384 + 7, 160, 160, True),
                                             # widths the 160 x 160 tiles fit: node_trans (600 x 150 + bias), a reduction
                                             # that ends inside a K-step, tiles that are full / hold only the bias column
                                             (1, 2968, 600, 150, True), (1, 7001, 300, 152, False), (1, 4999, 320, 319, True),
                                             (2, 1403, 160, 160, True), (3, 999, 480, 159, True)])
def test_weight_gradient_over_long_reductions(nseg, M, N, K, bias):
    """dw = sum over segments of dy^T x (and db = column sums of dy) where the reduction runs over tens of
    thousands of rows (split reductions with their slab sum; the 64 x 64 and the 160 x 160 tile kernels), against a
    float64 product."""
    from ps_vae_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device='cuda').manual_seed(M + N)
    dy = torch.randn(nseg, M, N, device=dev(), generator=g)
    x = torch.randn(nseg, M, K, device=dev(), generator=g)
    dw = torch.full((N, K), float('nan'), device=dev())
    db = torch.full((N,), float('nan'), device=dev()) if bias else None
    nbytes = L.psvae_linear_scratch_bytes(M, N, K, nseg)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev())
    seg_dy, seg_x = (M * N, M * K) if nseg > 1 else (0, 0)
    _lib.check(L.psvae_linear_backward_weight(dy.data_ptr(), N, seg_dy, x.data_ptr(), K, seg_x, nseg, dw.data_ptr(),
                                              db.data_ptr() if bias else None, M, N, K, scratch.data_ptr(), nbytes, st))
    exact = torch.einsum('smn,smk->nk', dy.double(), x.double())
    want = exact.float()
    check_close(dw, want, 'dw', atol_scale=2e-5)
    if bias:
        check_close(db, dy.double().sum((0, 1)).float(), 'db', atol_scale=2e-5)
    # fp32 accuracy of the cut-operand kernels (64-tile and 160-tile, with their split reductions): as close to the
    # float64 product as torch's fp32 product of the same operands on the CPU
    dy_c, x_c = dy.cpu(), x.cpu()
    fp32 = sum(dy_c[i].t() @ x_c[i] for i in range(nseg))
    check_fp32_accuracy(dw, fp32, exact, f'weight gradient {nseg}x{M}x{N}x{K}')
    # reproducible bit for bit, also with a scratch buffer that only holds a few slabs
    dw2 = torch.empty_like(dw)
    _lib.check(L.psvae_linear_backward_weight(dy.data_ptr(), N, seg_dy, x.data_ptr(), K, seg_x, nseg, dw2.data_ptr(),
                                              db.data_ptr() if bias else None, M, N, K, scratch.data_ptr(), nbytes, st))
    assert torch.equal(dw, dw2)
    small = min(nbytes, 3 * nseg * N * (K + 1) * 4 + 256)
    _lib.check(L.psvae_linear_backward_weight(dy.data_ptr(), N, seg_dy, x.data_ptr(), K, seg_x, nseg, dw2.data_ptr(),
                                              None, M, N, K, scratch.data_ptr(), small, st))
    check_close(dw2, want, 'dw with a small scratch buffer', atol_scale=2e-5)
