"""`psvae_linear_backward_weight_noalloc`: the weight gradient whose split-reduction slabs live in the library's
static device array instead of a caller's scratch buffer.  It must give the bits of
`psvae_linear_backward_weight` (same kernels, same split plan) on two calls in a row that share the array, the
second one smaller than the first, must refuse a problem whose slabs would not fit, and must allocate nothing."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(L, st, dy, x, nseg, M, N, K, with_bias):
    """(dw, db) through the scratch-argument entry point and through the no-allocation one."""
    from ps_vae_amd import _lib
    seg_dy, seg_x = (M * N, M * K) if nseg > 1 else (0, 0)
    nbytes = L.psvae_linear_scratch_bytes(M, N, K, nseg)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    out = []
    for noalloc in (False, True):
        dw = torch.full((N, K), float('nan'), device='cuda')
        db = torch.full((N,), float('nan'), device='cuda') if with_bias else None
        if noalloc:
            # device memory in use as the driver counts it (torch's own counters see only torch's allocator)
            torch.cuda.synchronize()
            before = torch.cuda.mem_get_info()[0], torch.cuda.memory_reserved()
            rc = L.psvae_linear_backward_weight_noalloc(dy.data_ptr(), N, seg_dy, x.data_ptr(), K, seg_x, nseg, dw.data_ptr(),
                                                        db.data_ptr() if with_bias else None, M, N, K, st)
            torch.cuda.synchronize()
            assert (torch.cuda.mem_get_info()[0], torch.cuda.memory_reserved()) == before
        else:
            rc = L.psvae_linear_backward_weight(dy.data_ptr(), N, seg_dy, x.data_ptr(), K, seg_x, nseg, dw.data_ptr(),
                                                db.data_ptr() if with_bias else None, M, N, K, scratch.data_ptr(), nbytes, st)
        _lib.check(rc)
        out.append((dw, db))
    return out


def test_no_alloc_weight_gradient_matches_the_scratch_variant():
    from ps_vae_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    assert L.psvae_noalloc_scratch_bytes() == 32 << 20
    # the library's init call looks the static array's device address up, so that even the first call below makes no
    # runtime call but its launches
    _lib.check(L.psvae_streams_init(st))
    g = torch.Generator(device='cuda').manual_seed(11)
    # first call: the step's GINE weight gradient (four summed segments, split reduction, bias column);
    # second call, right behind it on the same stream: a smaller problem that reuses the front of the array
    for nseg, M, N, K, with_bias in [(4, 2964, 300, 300, True), (1, 517, 56, 200, False), (1, 3000, 328, 656, True)]:
        dy = torch.randn(nseg, M, N, device='cuda', generator=g)
        x = torch.randn(nseg, M, K, device='cuda', generator=g)
        (dw_a, db_a), (dw_b, db_b) = _both(L, st, dy, x, nseg, M, N, K, with_bias)
        torch.cuda.synchronize()
        assert torch.equal(dw_a, dw_b), (nseg, M, N, K)
        if with_bias:
            assert torch.equal(db_a, db_b)
        want = torch.einsum('smn,smk->nk', dy.double(), x.double()).float()
        assert torch.allclose(dw_b, want, rtol=1e-4, atol=1e-4 * float(want.abs().max()))
    # a problem whose slabs exceed the 32 MiB array is refused with the error of its own, nothing is launched
    M, N, K = 64, 2048, 2048
    dy, x = torch.zeros(3, M, N, device="cuda"), torch.zeros(3, M, K, device="cuda")
    dw = torch.zeros(N, K, device='cuda')
    rc = L.psvae_linear_backward_weight_noalloc(dy.data_ptr(), N, M * N, x.data_ptr(), K, M * K, 3, dw.data_ptr(), None, M, N, K, st)
    assert rc == -6 and b'static array' in L.psvae_strerror(rc)
