"""PSVAEModel: the reference's model surface (`src/pl_models/ps_vae_model.py`)
driving the fused HIP training step.

All parameters live in ONE flat fp32 buffer (each nn.Parameter is a view into
it); gradients live in a second flat buffer of the same layout.  The flat
buffers are what the fused optimiser updates and what data-parallel training
all-reduces, and their order groups parameters by the moment their gradient is
final during backward.  `state_dict()` / `load_state_dict()` keep the
reference's keys and shapes.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .data import ensure_native
from .modules import Encoder, Predictor, VAEPieceDecoder

# flat-buffer order: buckets in the order backward finishes them
FLAT_ORDER = [
    # bucket 0: piece branch (final once lane 1 has run the GRU's backward and its weight gradients)
    'decoder.to_vocab.weight', 'decoder.to_vocab.bias',
    'decoder.rnn.weight_ih_l0', 'decoder.rnn.weight_hh_l0', 'decoder.rnn.bias_ih_l0', 'decoder.rnn.bias_hh_l0',
    'decoder.latent_to_rnn_hidden.weight', 'decoder.latent_to_rnn_hidden.bias',
    # bucket 1: property head and link scorer (lane 2, beside the scorer's input-gradient chain)
    'predictor.mlp.0.weight', 'predictor.mlp.0.bias', 'predictor.mlp.2.weight', 'predictor.mlp.2.bias',
    'predictor.output.weight', 'predictor.output.bias',
    'decoder.edge_predictor.0.seq.0.weight', 'decoder.edge_predictor.0.seq.0.bias',
    'decoder.edge_predictor.0.seq.2.weight', 'decoder.edge_predictor.0.seq.2.bias',
    'decoder.edge_predictor.0.seq.4.weight', 'decoder.edge_predictor.0.seq.4.bias',
    'decoder.edge_predictor.1.weight', 'decoder.edge_predictor.1.bias',
    # bucket 2: latent head (mean | log-var kept adjacent), encoder.linear
    'decoder.W_mean.weight', 'decoder.W_log_var.weight', 'decoder.W_mean.bias', 'decoder.W_log_var.bias',
    'encoder.linear.weight', 'encoder.linear.bias',
    # bucket 3: both GINE stacks (node_trans of the two stacks kept adjacent), embedding tables
    'encoder.conv.nn.seq.0.weight', 'encoder.conv.nn.seq.0.bias', 'encoder.conv.nn.seq.2.weight', 'encoder.conv.nn.seq.2.bias',
    'decoder.graph_embedding.conv.nn.seq.0.weight', 'decoder.graph_embedding.conv.nn.seq.0.bias',
    'decoder.graph_embedding.conv.nn.seq.2.weight', 'decoder.graph_embedding.conv.nn.seq.2.bias',
    'encoder.edge_trans.weight', 'encoder.edge_trans.bias',
    'decoder.graph_embedding.edge_trans.weight', 'decoder.graph_embedding.edge_trans.bias',
    'encoder.node_trans.weight', 'decoder.graph_embedding.node_trans.weight',
    'encoder.node_trans.bias', 'decoder.graph_embedding.node_trans.bias',
    'decoder.atom_embedding.weight', 'decoder.piece_embedding.weight', 'decoder.pos_embedding.weight',
]
BUCKET_STARTS = ['decoder.to_vocab.weight', 'predictor.mlp.0.weight', 'decoder.W_mean.weight',
                 'encoder.conv.nn.seq.0.weight']
# never used by the training step (the decoder-side encoder's `linear`): no gradient, not optimised
UNTRAINED = ['decoder.graph_embedding.linear.weight', 'decoder.graph_embedding.linear.bias']


class _StepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, trigger, model, nb, eps, weights):
        L = _lib.lib()
        dims, params = model._dims, model._params_struct
        bs = nb.struct(eps, model._status_words(eps.device))
        nbytes = L.psvae_workspace_bytes(C.byref(dims), C.byref(bs))
        if nbytes == 0:
            raise RuntimeError('psvae_workspace_bytes rejected the model/batch sizes (hidden, latent and rnn sizes '
                               'must be multiples of 4; 4 bond types)')
        ws = torch.empty(nbytes, dtype=torch.uint8, device=eps.device)
        losses = torch.empty(6, dtype=torch.float32, device=eps.device)
        ctx.set_materialize_grads(False)
        stream = torch.cuda.current_stream().cuda_stream
        w3 = (C.c_float * 3)(*weights)
        _lib.check(L.psvae_step_forward(C.byref(dims), C.byref(params), C.byref(bs), ws.data_ptr(), nbytes,
                                        C.byref(w3), losses.data_ptr(), stream), 'psvae_step_forward')
        ctx.model, ctx.nb, ctx.eps, ctx.ws, ctx.nbytes, ctx.w3 = model, nb, eps, ws, nbytes, w3
        model._last_ws = (ws, nb)
        # the weighted total is handed out as an output of its own (same storage as losses[5]) so that
        # `total.backward()` needs no select-backward (a zero fill and a scatter of one number every step)
        return losses, losses[5]

    @staticmethod
    def backward(ctx, g, g_total):
        L = _lib.lib()
        model, nb = ctx.model, ctx.nb
        # upstream gradients of (pred, rec, piece, edge, kl, total) as one device vector
        if g is None:
            if model._g6 is None or model._g6.device != ctx.eps.device:
                model._g6 = torch.zeros(6, dtype=torch.float32, device=ctx.eps.device)
            g = model._g6
            if g_total is None:
                return None, None, None, None, None
            unit = model._unit
            if unit is not None and g_total.data_ptr() == unit[0].data_ptr():
                g = unit[1]            # `total.backward(model.unit_grad())`: the constant vector, no copy kernel
            else:
                g[5:6].copy_(g_total.reshape(1))
        else:
            g = g.contiguous()
            if g_total is not None:
                g = g.clone()
                g[5] += g_total
        grads_struct, fresh = model._grad_target()
        bs = nb.struct(ctx.eps, model._status_words(ctx.eps.device))
        stream = torch.cuda.current_stream().cuda_stream
        ev = model._early_events
        _lib.check(L.psvae_step_backward_signal(C.byref(model._dims), C.byref(model._params_struct), C.byref(bs),
                                                ctx.ws.data_ptr(), ctx.nbytes, C.byref(ctx.w3), g.data_ptr(),
                                                C.byref(grads_struct), stream,
                                                ev[0].cuda_event if ev and fresh else None, ev[1].cuda_event if ev and fresh else None),
                   'psvae_step_backward')
        model._grad_written(fresh)
        ctx.ws = None
        return None, None, None, None, None


class PSVAEModel(nn.Module):
    """`PSVAEModel(config, tokenizer)` with `.forward(batch, return_accu=False)`,
    `.weighted_loss`, `.cal_beta`, `.get_z`, `.sample_z`, `.predict_props`
    (`src/pl_models/ps_vae_model.py:18-74,130-170`).  `global_step` is advanced by
    the training loop (Lightning did it for the reference)."""

    def __init__(self, config, tokenizer):
        super().__init__()
        self.config = config
        self.tokenizer = tokenizer
        self.chem_vocab = tokenizer.chem_vocab
        self.predictor = Predictor(**config['predictor'])
        self.encoder = Encoder(**config['encoder'])
        self.decoder = VAEPieceDecoder(**config['vae_piece_decoder'])
        self.global_step = 0
        self.total_time = 0
        self._flat = self._gflat = self._gtmp = None
        self._grads_fresh = True
        self._trigger = None
        self._last_ws = None
        self._g6 = None
        self._unit = None
        self._early_events = None
        self._status = None
        self._ws_bytes, self._ws_keep = {}, None
        self.last_backward_signalled = False
        # epsilon of rsample: counter-based, keyed by (eps_seed, number of forward passes so far, global graph index)
        self.eps_seed = int(config.get('seed', 2021)) if hasattr(config, 'get') else 2021
        self.eps_draws = 0
        self.shard_rank = 0          # set by parallel.make_reducer: this process's place among the data-parallel ranks
        self._dims = self._make_dims()

    # ------------------------------------------------------------------ layout
    def _make_dims(self):
        dec, enc, cfg = self.decoder, self.encoder, self.config
        d = _lib.Dims()
        d.num_piece, d.num_atom, d.num_bond = self.tokenizer.num_piece_type(), self.tokenizer.num_atom_type(), enc.num_edge_type
        d.max_pos = dec.pos_embedding.num_embeddings
        d.d_atom, d.d_piece, d.d_pos = (dec.atom_embedding.embedding_dim, dec.piece_embedding.embedding_dim,
                                        dec.pos_embedding.embedding_dim)
        d.hidden, d.graph = enc.node_trans.out_features, enc.linear.out_features
        d.latent, d.rnn = dec.latent_dim, dec.rnn.hidden_size
        d.pred_hidden, d.n_props = self.predictor.output.in_features, self.predictor.output.out_features
        d.t_enc, d.t_dec = enc.t, dec.graph_embedding.t
        d.pad_idx = self.tokenizer.pad_idx()
        d.eps_enc, d.eps_dec = float(enc.conv.eps), float(dec.graph_embedding.conv.eps)
        if dec.graph_embedding.node_trans.out_features != d.hidden:
            raise ValueError('encoder and decoder node hidden sizes must match (as the reference CLI builds them)')
        return d

    def _flatten(self):
        """(Re)build the flat parameter / gradient buffers on the parameters' device."""
        named = dict(self.named_parameters())
        order = FLAT_ORDER + UNTRAINED
        assert sorted(order) == sorted(named.keys()), 'parameter set differs from the reference state-dict'
        dev = named[order[0]].device
        if dev.type == 'cuda':
            # the step's side streams are created (and checked to run beside the current stream) before anything
            # else this process may create later, e.g. a communicator's streams
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().psvae_streams_init(torch.cuda.current_stream(dev).cuda_stream), 'psvae_streams_init')
        offs, off = {}, 0
        for k in order:
            if k == UNTRAINED[0]:
                self._n_trained = off
            offs[k] = off
            off += (named[k].numel() + 3) // 4 * 4
        flat = torch.zeros(off, dtype=torch.float32, device=dev)
        # four spare floats behind the gradients: a data-parallel caller can let a few numbers (the next batch's
        # loss denominators) ride on the gradient message (`grad_message`)
        gflat = torch.zeros(self._n_trained + 4, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for k in order:
                p = named[k]
                view = flat[offs[k]:offs[k] + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                if k not in UNTRAINED:
                    p.grad = gflat[offs[k]:offs[k] + p.numel()].view(p.shape)
        self._flat, self._gflat, self._gtmp, self._offsets = flat, gflat, None, offs
        self._grads_fresh = True
        self._params_struct = self._struct_over(flat)
        self._grads_struct = self._struct_over(gflat)
        self._trigger = torch.zeros((), device=dev, requires_grad=True)
        self._dims = self._make_dims()
        self.bucket_bounds = [offs[k] for k in BUCKET_STARTS] + [self._n_trained]

    def _struct_over(self, buf):
        named = dict(self.named_parameters())
        s = _lib.Params()
        base = buf.data_ptr()
        for field, key in _lib.PARAM_KEYS.items():
            setattr(s, field, base + 4 * self._offsets[key])
            assert named[key].numel() > 0
        return s

    def _sync_flat(self):
        p0 = self.decoder.to_vocab.weight
        if (self._flat is None or p0.device != self._flat.device or
                p0.data_ptr() != self._flat.data_ptr() + 4 * self._offsets['decoder.to_vocab.weight']):
            self._flatten()
        elif p0.grad is None:
            named = dict(self.named_parameters())
            for k in FLAT_ORDER:
                p = named[k]
                p.grad = self._gflat[self._offsets[k]:self._offsets[k] + p.numel()].view(p.shape)
            self._gflat.zero_()
            self._grads_fresh = True

    @staticmethod
    def trained_names():
        """Names of the parameters the step trains, in the order of the flat buffers."""
        return list(FLAT_ORDER)

    def flat_buffers(self):
        """(parameters, gradients) as flat fp32 tensors over the trained range."""
        self._sync_flat()
        return self._flat[:self._n_trained], self._gflat[:self._n_trained]

    def grad_message(self):
        """(message, tail): the flat gradient buffer with its four spare floats, and a view of those four.  All-reducing
        `message` instead of the gradients alone carries four caller-chosen numbers along in the same collective."""
        self._sync_flat()
        return self._gflat, self._gflat[self._n_trained:]

    def early_gradient_events(self):
        """Two CUDA events that every following backward records while it is still running: the first once the
        gradients of the flat range `[0, bucket_bounds[1])` (piece branch) are final, the second once those of
        `[bucket_bounds[1], bucket_bounds[3])` (property head, link scorer, latent head, encoder.linear) are.
        A data-parallel caller waits for them on its communication stream and starts those ranges'
        all-reduces early (`parallel.OverlappedReducer`).  Only the first backward after `zero_grad` writes the
        flat buffer directly, so only that one signals (`last_backward_signalled`)."""
        self._sync_flat()
        if self._early_events is None:
            self._early_events = (torch.cuda.Event(), torch.cuda.Event())
            for e in self._early_events:
                e.record()          # creates the underlying event object
        return self._early_events

    def zero_grad(self, set_to_none=False):
        """Marks the flat gradient buffer as empty: the next backward overwrites it
        (no separate zero-fill pass is needed)."""
        self._grads_fresh = True

    def _grad_target(self):
        if self._grads_fresh:
            return self._grads_struct, True
        if self._gtmp is None:
            self._gtmp = torch.zeros_like(self._gflat)   # padding slots stay zero
            self._gtmp_struct = self._struct_over(self._gtmp)
        return self._gtmp_struct, False

    def _grad_written(self, fresh):
        self.last_backward_signalled = bool(fresh and self._early_events is not None)
        if not fresh:
            self._gflat.add_(self._gtmp)   # gradient accumulation across backward calls
        self._grads_fresh = False

    # ------------------------------------------------------------------ forward
    def _native(self, batch):
        nb = ensure_native(batch, self.tokenizer, tuple(self.config['selected_properties']),
                           self.decoder.pos_embedding.num_embeddings)
        dev = self._flat.device
        if nb.device != dev:
            nb = nb.to(dev)
            batch['_native'] = nb
        return nb

    def forward(self, batch, return_accu=False, eps=None):
        """Returns `(pred_loss, (rec, piece, edge, kl))`, or with `return_accu`
        `(pred_loss, ((rec, piece, edge, kl), (piece_accu, edge_accu)))`.
        `eps` [B, latent] optionally replaces the N(0,1) draw of `rsample`."""
        self._sync_flat()
        if not self._flat.is_cuda:
            raise RuntimeError('PSVAEModel runs on the GPU only: move it with .cuda() (no CPU implementation exists)')
        nb = self._native(batch)
        if eps is None:
            eps = self.draw_eps(nb.sizes['B'], self._first_graph(nb))
        eps = eps.to(self._flat.device, torch.float32).contiguous()
        alpha = float(self.config['alpha'])
        out, self._last_total = _StepFn.apply(self._trigger, self, nb, eps, (1.0 - alpha, alpha, float(self.cal_beta())))
        pred_loss, losses = out[0], (out[1], out[2], out[3], out[4])
        if return_accu:
            return pred_loss, (losses, self._accuracies(nb))
        return pred_loss, losses

    def _first_graph(self, nb):
        """Global index of a batch's first graph: what its loader recorded, else `shard_rank` equal shards in."""
        return nb.graph_base if nb.graph_base is not None else self.shard_rank * nb.sizes['B']

    def draw_eps(self, rows, first_row=0):
        """The N(0,1) draw of `rsample` (`src/modules/vae_piece_decoder.py:61`) for graphs `first_row ..
        first_row + rows` of the global batch, as a counter-based function of (`eps_seed`, `eps_draws`, graph,
        column): ranks that hold shards of one global batch draw what a single process would draw for all of it.
        Every call advances `eps_draws` (one draw per forward pass)."""
        dev = self._flat.device if self._flat is not None else next(self.parameters()).device
        out = torch.empty(rows, self._dims.latent, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().psvae_normal_rows(self.eps_seed, self.eps_draws, int(first_row), rows, self._dims.latent,
                                                out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), 'psvae_normal_rows')
        self.eps_draws += 1
        return out

    def _ws_view(self, name, rows, cols, ld=None):
        ws, nb = self._last_ws
        L = _lib.lib()
        off = L.psvae_workspace_offset(C.byref(self._dims), C.byref(nb.struct()), name.encode())
        if off < 0:
            raise KeyError(name)
        ld = cols if ld is None else ld
        flat = ws.view(torch.float32)
        return flat[off:off + rows * ld].view(rows, ld)[:, :cols]

    def _accuracies(self, nb):
        """`src/modules/vae_piece_decoder.py:120-124`."""
        B, L, S, V2 = nb.sizes['B'], nb.sizes['L'], nb.sizes['S'], self._dims.num_piece
        T = L - 1
        logits = self._ws_view('logits', B * T, V2, (V2 + 3) // 4 * 4)
        off = nb.offsets[_lib.BLOB_FIELDS.index('pieces')]
        gold = nb.blob[off:off + B * L].view(B, L)[:, 1:].reshape(-1).long()
        not_pad = gold != self.tokenizer.pad_idx()
        piece_accu = ((logits.argmax(-1) == gold) & not_pad).sum().item() / not_pad.sum().item()
        pair_logits = self._ws_view('pair_logits', S, 4)
        goff = nb.offsets[_lib.BLOB_FIELDS.index('pair_gold')]
        golden = nb.blob[goff:goff + S].long()
        edge_accu = (pair_logits.argmax(-1) == golden).sum().item() / max(S, 1)
        return piece_accu, edge_accu

    def unit_grad(self):
        """The scalar 1.0 on the model's device, for `total.backward(model.unit_grad())`: autograd then launches no
        fill kernel for the implicit gradient and the step's backward recognises the tensor and uses a constant
        upstream-gradient vector instead of copying the scalar into one (two tiny dependent launches per step).
        The tensor must not be modified."""
        dev = self._flat.device if self._flat is not None else next(self.parameters()).device
        if self._unit is None or self._unit[0].device != dev:
            g6 = torch.zeros(6, dtype=torch.float32, device=dev)
            g6[5] = 1.0
            self._unit = (torch.ones((), dtype=torch.float32, device=dev), g6)
        return self._unit[0]

    def _status_words(self, device):
        if self._status is None or self._status.device != device:
            self._status = torch.zeros(8, dtype=torch.int32, device=device)
        return self._status

    def recurrence_gave_up(self, clear=False):
        """True if any step since the model was built (or since the last call with `clear`) left invalid results
        behind (status word 0 of `psvae_batch.status`).  No kernel of the current library sets it: a multi-CU GRU
        recurrence whose workgroups stop waiting for each other is redone on the single-CU kernels inside the same
        step (`recurrence_redone`).  Synchronises the device."""
        if self._status is None:
            return False
        torch.cuda.synchronize(self._status.device)
        gave_up = bool(self._status[0].item() != 0)
        if clear:
            self._status.zero_()
        return gave_up

    def recurrence_redone(self):
        """Number of GRU recurrences (forward or backward) since the model was built that were redone on the single-CU
        kernels because a workgroup of the multi-CU recurrence stopped waiting for its partners (bounded waits: a GPU
        that cannot keep the group resident must not hang the process).  Results are valid either way; a non-zero
        count means such steps took seconds.  Synchronises the device."""
        if self._status is None:
            return 0
        torch.cuda.synchronize(self._status.device)
        return int(self._status[1].item())

    def activation(self, name, rows, cols, ld=None):
        """View of a saved activation of the most recent forward (tests, debugging)."""
        return self._ws_view(name, rows, cols, ld)

    def cal_beta(self):
        step, cfg = self.global_step, self.config
        if step < cfg['kl_warmup']:
            beta = 0
        else:
            beta = min(cfg['max_beta'], ((step - cfg['kl_warmup']) // cfg['kl_anneal_iter'] + 1) * cfg['step_beta'])
        return beta + cfg['beta']

    def weighted_loss(self, pred_loss, rec_loss, kl_loss):
        alpha = self.config['alpha']
        return alpha * rec_loss + (1 - alpha) * pred_loss + self.cal_beta() * kl_loss

    def training_step(self, batch, batch_idx=0, eps=None):
        """`ps_vae_model.py:76-88`: forward + weighted loss.  The weighted sum is formed inside the
        fused step (same value as `weighted_loss(pred, rec, kl)` at the current `global_step`)."""
        self.forward(batch, eps=eps)
        return self._last_total

    def train_step(self, batch, optimizer=None, eps=None):
        """One optimiser step as ONE library call (`psvae_train_step`): `training_step`, `loss.backward()`, gradient
        clipping and `optimizer.step()` of the reference's loop (`ps_vae_model.py:76-88`, `src/train.py:47`,
        `ps_vae_model.py:123-127`) with no Python or autograd between them - the same kernels in the same order, so the
        same numbers, at two thirds of the host time.  Returns the six losses `[pred, rec, piece, edge, kl, total]` (a
        device tensor, no synchronisation).  Without `optimizer` only forward and backward run and the gradients are
        left in the flat buffer (`flat_buffers()`, the parameters' `.grad`): the data-parallel form - reduce them, then
        `optimizer.step()`.  Needs the gradient buffer empty (`zero_grad()`, which `optimizer.step()` implies)."""
        self._sync_flat()
        if not self._flat.is_cuda:
            raise RuntimeError('PSVAEModel runs on the GPU only: move it with .cuda() (no CPU implementation exists)')
        if not self._grads_fresh:
            raise RuntimeError('train_step writes the gradients of one step: call zero_grad() (or optimizer.step()) first; '
                               'use training_step + backward to accumulate gradients over several batches')
        nb = self._native(batch)
        dev = self._flat.device
        draw = None
        if eps is None:      # the library draws it itself, beside the step's first kernels (the same numbers as draw_eps)
            eps = torch.empty(nb.sizes['B'], self._dims.latent, dtype=torch.float32, device=dev)
            draw = _lib.EpsDraw(self.eps_seed, self.eps_draws, int(self._first_graph(nb)))
            self.eps_draws += 1
        eps = eps.to(dev, torch.float32).contiguous()
        L = _lib.lib()
        bs = nb.struct(eps, self._status_words(dev))
        key = tuple(nb.sizes[k] for k in ('B', 'n', 'E', 'E_in', 'S', 'L'))
        nbytes = self._ws_bytes.get(key)
        if nbytes is None:
            nbytes = self._ws_bytes[key] = L.psvae_workspace_bytes(C.byref(self._dims), C.byref(bs))
            if nbytes == 0:
                raise RuntimeError('psvae_workspace_bytes rejected the model/batch sizes')
        if self._ws_keep is None or self._ws_keep.numel() < nbytes or self._ws_keep.device != dev:
            self._ws_keep = torch.empty(nbytes, dtype=torch.uint8, device=dev)     # kept: every step reuses it
        ws = self._ws_keep
        losses = torch.empty(6, dtype=torch.float32, device=dev)
        alpha = float(self.config['alpha'])
        w3 = (C.c_float * 3)(1.0 - alpha, alpha, float(self.cal_beta()))
        self.unit_grad()
        ev = self._early_events
        adam = optimizer.fused_struct() if optimizer is not None else None
        _lib.check(L.psvae_train_step(C.byref(self._dims), C.byref(self._params_struct), C.byref(bs), ws.data_ptr(), nbytes,
                                      C.byref(w3), losses.data_ptr(), self._unit[1].data_ptr(), C.byref(self._grads_struct),
                                      C.byref(adam) if adam is not None else None, C.byref(draw) if draw is not None else None,
                                      ev[0].cuda_event if ev else None, ev[1].cuda_event if ev else None,
                                      torch.cuda.current_stream(dev).cuda_stream), 'psvae_train_step')
        self._last_ws = (ws, nb)
        self._last_total = losses[5]
        self.last_backward_signalled = ev is not None
        self._grads_fresh = optimizer is not None      # (the optimiser consumed them: the next step overwrites)
        return losses

    def validation_step(self, batch, batch_idx=0, eps=None):
        """`ps_vae_model.py:90-96`: weighted loss and the two accuracies of one batch."""
        pred_loss, (losses, accus) = self.forward(batch, return_accu=True, eps=eps)
        return {'val_loss': self.weighted_loss(pred_loss, losses[0], losses[-1]).detach(),
                'piece accu': accus[0], 'edge accu': accus[1]}

    def test_step(self, batch, batch_idx=0, eps=None):
        """`ps_vae_model.py:118-121`."""
        pred_loss, losses = self.forward(batch, eps=eps)
        return {'test_loss': self.weighted_loss(pred_loss, losses[0], losses[-1]).detach()}

    def configure_optimizers(self):
        from .optim import FusedClipAdam
        return FusedClipAdam(self, lr=self.config['lr'])

    # ------------------------------------------------------------------ interface
    def encode(self, batch):
        """Deterministic encoder output: (z_mean, z_log_var), each [B, latent]."""
        self._sync_flat()
        if not self._flat.is_cuda:
            raise RuntimeError('PSVAEModel runs on the GPU only (no CPU implementation exists)')
        L = _lib.lib()
        nb = self._native(batch)
        bs = nb.struct()
        nbytes = L.psvae_workspace_bytes(C.byref(self._dims), C.byref(bs))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self._flat.device)
        B, Z = nb.sizes['B'], self._dims.latent
        mean = torch.empty(B, Z, dtype=torch.float32, device=self._flat.device)
        lvar = torch.empty_like(mean)
        _lib.check(L.psvae_encode(C.byref(self._dims), C.byref(self._params_struct), C.byref(bs), ws.data_ptr(), nbytes,
                                  mean.data_ptr(), lvar.data_ptr(), torch.cuda.current_stream().cuda_stream), 'psvae_encode')
        return mean, lvar

    def get_z(self, batch, eps=None):
        """`ps_vae_model.py:160-170`: a sample z = mean + exp(log_var / 2) * eps."""
        with torch.no_grad():
            mean, lvar = self.encode(batch)
            if eps is None:
                nb = self._native(batch)
                eps = self.draw_eps(mean.shape[0], self._first_graph(nb))
            return mean + torch.exp(lvar / 2) * eps.to(mean.device)

    def sample_z(self, n, device=None):
        return torch.randn(n, self.decoder.latent_dim, device=device if device is not None else self._flat.device)

    def predict_props(self, z):
        props = self.predictor(z)
        idx = self.config['selected_properties']
        if props.dim() == 1:
            out = torch.zeros(3, device=props.device)
            out[idx] = props
            return out
        out = torch.zeros(props.shape[0], 3, device=props.device)
        out[:, idx] = props
        return out

    def get_from_batch(self, batch, key):
        if key == 'batch_size':
            return batch['x'].shape[0]
        if key == 'props':
            return batch['props']
        raise KeyError(key)
