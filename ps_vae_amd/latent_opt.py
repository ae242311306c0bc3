"""Latent-space property optimisation (`direct_optimize`, `src/generate.py:67-89`, with `get_loss`
`:30-44`), for many latent vectors at once.

The reference runs Adam on ONE z through the property head until the loss has not improved for
`patience` steps.  Rows of a batch do not interact (the loss of a row depends on that row only and
Adam is element-wise), so here every row follows exactly the trajectory the reference would give
it: per-row best loss / patience counters, and a row that has stopped is frozen.  The property
head's products run on the HIP GEMM kernels (forward and input gradient); the element-wise Adam
update of z is plain device arithmetic."""
import torch

from . import ops


def _head(model, z):
    p = model.predictor
    h = ops.linear(z, p.mlp[0].weight, p.mlp[0].bias, relu=True)
    h = ops.linear(h, p.mlp[2].weight, p.mlp[2].bias, relu=True)
    return ops.linear(h, p.output.weight, p.output.bias)


def direct_optimize(model, z, target, lr=0.1, max_iter=100, patience=3, betas=(0.9, 0.999), adam_eps=1e-8):
    """`z` [n, latent]; `target`: one value per property in `PROPS` order, `None` = do not optimise that
    property (as in the reference).  Returns `(best_z [n, latent], best_loss [n], steps_run [n])`."""
    if not z.is_cuda:
        raise RuntimeError('direct_optimize runs on the GPU only (the property head has no CPU implementation here)')
    sel_props = list(model.config['selected_properties'])
    cols = [sel_props.index(i) for i, v in enumerate(target) if v is not None]     # columns of the head's output
    tar = torch.tensor([v for v in target if v is not None], dtype=torch.float32, device=z.device)
    if not cols:
        raise ValueError('no property selected by `target`')
    cols_t = torch.tensor(cols, dtype=torch.long, device=z.device)
    params = [q for q in model.predictor.parameters()]
    cur = z.detach().clone().float().contiguous()
    n = cur.shape[0]
    m = torch.zeros_like(cur)
    v = torch.zeros_like(cur)
    t = torch.zeros(n, device=z.device)                      # Adam step count per row
    best_loss = torch.full((n,), 100.0, device=z.device)     # the reference's initial best (generate.py:73)
    best_z = cur.clone()
    not_inc = torch.zeros(n, dtype=torch.long, device=z.device)
    active = torch.ones(n, dtype=torch.bool, device=z.device)
    steps = torch.zeros(n, dtype=torch.long, device=z.device)
    b1, b2 = betas
    for _ in range(max_iter):
        zz = cur.clone().requires_grad_(True)
        with torch.enable_grad():
            props = _head(model, zz)
            loss_rows = ((props.index_select(1, cols_t) - tar) ** 2).mean(dim=1)   # MSELoss of one row
            (g,) = torch.autograd.grad(loss_rows.sum(), zz, retain_graph=False)
        for q in params:
            q.grad = None
        a = active.unsqueeze(1)
        t = t + active.float()
        m = torch.where(a, b1 * m + (1 - b1) * g, m)
        v = torch.where(a, b2 * v + (1 - b2) * g * g, v)
        bc1 = 1 - b1 ** t.clamp(min=1)
        bc2 = 1 - b2 ** t.clamp(min=1)
        step = (lr / bc1).unsqueeze(1) * m / ((v.sqrt() / bc2.sqrt().unsqueeze(1)) + adam_eps)
        cur = torch.where(a, cur - step, cur)
        steps = steps + active.long()
        # bookkeeping as generate.py:85-91: the loss of THIS step belongs to the z AFTER the update
        loss = loss_rows.detach()
        better = active & (loss < best_loss)
        best_loss = torch.where(better, loss, best_loss)
        best_z = torch.where(better.unsqueeze(1), cur, best_z)
        not_inc = torch.where(better, torch.zeros_like(not_inc), not_inc + active.long())
        active = active & ~(not_inc > patience)
        if not bool(active.any()):
            break
    return best_z, best_loss, steps
