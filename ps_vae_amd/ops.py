"""Single-operator autograd functions over the C ABI (the module-level API uses
these; the whole-step path in `model.py` bypasses them)."""
import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f'{what}: tensors must live on the GPU; ps_vae_amd has no CPU implementation')


def _f32c(t):
    return t.contiguous().float()


def _scratch(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


class LinearFn(torch.autograd.Function):
    """y = act(x w^T + b) over the last dimension (nn.Linear semantics)."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        _need_gpu(x, 'linear')
        L = _lib.lib()
        shape = x.shape
        x2 = _f32c(x).reshape(-1, shape[-1])
        w, bb = _f32c(w), (None if b is None else _f32c(b))
        M, K, N = x2.shape[0], x2.shape[1], w.shape[0]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        sc = _scratch(L.psvae_linear_scratch_bytes(M, N, K, 1), x.device)
        _lib.check(L.psvae_linear_forward(x2.data_ptr(), K, w.data_ptr(), None if bb is None else bb.data_ptr(),
                                          y.data_ptr(), N, M, N, K, int(relu), sc.data_ptr(), sc.numel(), _stream()),
                   'psvae_linear_forward')
        ctx.save_for_backward(x2, w, y if relu else None)
        ctx.has_bias, ctx.relu, ctx.shape = b is not None, bool(relu), shape
        return y.reshape(*shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x2, w, y = ctx.saved_tensors
        M, K, N = x2.shape[0], x2.shape[1], w.shape[0]
        dy2 = _f32c(dy).reshape(M, N)
        if ctx.relu:
            dy2 = dy2 * (y > 0)   # gradient gate of the fused ReLU
        dx = torch.empty(M, K, dtype=torch.float32, device=dy.device)
        _lib.check(L.psvae_linear_backward_input(dy2.data_ptr(), N, w.data_ptr(), dx.data_ptr(), K, M, N, K, None, 0, 0,
                                                 _stream()), 'psvae_linear_backward_input')
        dw = torch.empty_like(w)
        db = torch.empty(N, dtype=torch.float32, device=dy.device) if ctx.has_bias else None
        sc = _scratch(L.psvae_linear_scratch_bytes(M, N, K, 1), dy.device)
        _lib.check(L.psvae_linear_backward_weight(dy2.data_ptr(), N, 0, x2.data_ptr(), K, 0, 1, dw.data_ptr(),
                                                  None if db is None else db.data_ptr(), M, N, K, sc.data_ptr(), sc.numel(),
                                                  _stream()), 'psvae_linear_backward_weight')
        return dx.reshape(ctx.shape), dw, db, None


def linear(x, w, b=None, relu=False):
    return LinearFn.apply(x, w, b, relu)


class Csr:
    """By-target and by-source CSR of an edge list, built with index ops on the
    tensors' own device (stable sorts keep the edge order inside every row)."""

    def __init__(self, edge_index, edge_type, n):
        src, dst = edge_index[0].long(), edge_index[1].long()
        self.n, self.E = int(n), int(src.numel())
        dev = src.device

        def build(rows, cols):
            order = torch.sort(rows, stable=True).indices
            ptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
            if rows.numel():
                ptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0).to(torch.int32)
            return ptr, cols[order].to(torch.int32).contiguous(), edge_type[order].to(torch.int32).contiguous()
        self.rowptr, self.col, self.etype = build(dst, src)
        self.rowptr_t, self.col_t, self.etype_t = build(src, dst)


class GineAggregateFn(torch.autograd.Function):
    """out_i = (1+eps) x_i + sum_{j->i} relu(x_j + edge_w[:, type] + edge_b)."""

    @staticmethod
    def forward(ctx, x, edge_w, edge_b, csr, one_plus_eps):
        _need_gpu(x, 'gine_aggregate')
        L = _lib.lib()
        x, edge_w, edge_b = _f32c(x), _f32c(edge_w), _f32c(edge_b)
        n, H = x.shape
        out = torch.empty_like(x)
        _lib.check(L.psvae_gine_aggregate_forward(x.data_ptr(), H, csr.rowptr.data_ptr(), csr.col.data_ptr(),
                                                  csr.etype.data_ptr(), edge_w.data_ptr(), edge_b.data_ptr(),
                                                  float(one_plus_eps), out.data_ptr(), H, n, H, _stream()),
                   'psvae_gine_aggregate_forward')
        ctx.save_for_backward(x, edge_w, edge_b)
        ctx.csr, ctx.ope = csr, float(one_plus_eps)
        return out

    @staticmethod
    def backward(ctx, da):
        L = _lib.lib()
        x, edge_w, edge_b = ctx.saved_tensors
        csr, n, H = ctx.csr, x.shape[0], x.shape[1]
        da = _f32c(da)
        dx = torch.empty_like(x)
        dw, db = torch.zeros_like(edge_w), torch.zeros_like(edge_b)
        part = torch.empty(max(1, L.psvae_gine_edge_grad_scratch(n, H)), dtype=torch.float32, device=x.device)
        _lib.check(L.psvae_gine_aggregate_backward(da.data_ptr(), H, x.data_ptr(), H, csr.rowptr_t.data_ptr(),
                                                   csr.col_t.data_ptr(), csr.etype_t.data_ptr(), edge_w.data_ptr(),
                                                   edge_b.data_ptr(), ctx.ope, None, 0, None, dx.data_ptr(), H,
                                                   dw.data_ptr(), db.data_ptr(), part.data_ptr(), n, H, _stream()),
                   'psvae_gine_aggregate_backward')
        return dx, dw, db, None, None


class SegmentPoolFn(torch.autograd.Function):
    """pooled[b] = sum of the rows graph_off[b] .. graph_off[b+1] (rows grouped by graph)."""

    @staticmethod
    def forward(ctx, x, graph_off, node_graph):
        _need_gpu(x, 'segment_pool')
        L = _lib.lib()
        x = _f32c(x)
        n, W = x.shape
        B = graph_off.numel() - 1
        pooled = torch.empty(B, W, dtype=torch.float32, device=x.device)
        _lib.check(L.psvae_segment_pool_forward(x.data_ptr(), W, graph_off.data_ptr(), pooled.data_ptr(), W, B, W, _stream()),
                   'psvae_segment_pool_forward')
        ctx.save_for_backward(node_graph)
        ctx.n = n
        return pooled

    @staticmethod
    def backward(ctx, dp):
        L = _lib.lib()
        (node_graph,) = ctx.saved_tensors
        dp = _f32c(dp)
        W = dp.shape[1]
        dx = torch.empty(ctx.n, W, dtype=torch.float32, device=dp.device)
        _lib.check(L.psvae_gine_aggregate_backward(None, 0, None, 0, None, None, None, None, None, 1.0, dp.data_ptr(), W,
                                                   node_graph.data_ptr(), dx.data_ptr(), W, None, None, None, ctx.n, W,
                                                   _stream()),
                   'psvae_gine_aggregate_backward(pool broadcast)')
        return dx, None, None
