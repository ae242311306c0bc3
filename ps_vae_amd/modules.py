"""nn.Module surface of the reference model with HIP kernels underneath.

Constructor signatures, method names, argument meaning and state-dict keys
follow `src/modules/common_nn.py`, `encoder.py`, `predictor.py` and
`vae_piece_decoder.py` of the reference.  torch layers (`nn.Linear`, `nn.GRU`,
`nn.Embedding`) are used only as parameter containers so that the state-dict
layout is the reference's; their `forward` is never called.
"""
import torch
import torch.nn as nn

from . import ops


class MLP(nn.Module):
    """`src/modules/common_nn.py:8-23`: Linear, act, ..., Linear (no final activation)."""

    def __init__(self, dim_in, dim_hidden, dim_out, act_func, num_layers):
        super().__init__()
        assert num_layers > 0
        if act_func is not nn.ReLU:
            raise ValueError('the HIP kernels fuse ReLU; other activations are not implemented')
        if num_layers == 1:
            self.seq = nn.Linear(dim_in, dim_out)
        else:
            seq = [nn.Linear(dim_in, dim_hidden), act_func()]
            for _ in range(num_layers - 2):
                seq += [nn.Linear(dim_hidden, dim_hidden), act_func()]
            seq.append(nn.Linear(dim_hidden, dim_out))
            self.seq = nn.Sequential(*seq)

    def forward(self, x):
        if isinstance(self.seq, nn.Linear):
            return ops.linear(x, self.seq.weight, self.seq.bias)
        layers = [m for m in self.seq if isinstance(m, nn.Linear)]
        for i, lin in enumerate(layers):
            x = ops.linear(x, lin.weight, lin.bias, relu=i + 1 < len(layers))
        return x


class GINEConv(nn.Module):
    """Parameter container with the state-dict layout of torch_geometric's
    GINEConv as the reference builds it (`src/modules/encoder.py:48`):
    `nn` (the MLP) and the non-trainable buffer `eps`."""

    def __init__(self, nn_module, eps=0.0):
        super().__init__()
        self.nn = nn_module
        self.register_buffer('eps', torch.Tensor([eps]))


class Encoder(nn.Module):
    """`src/modules/encoder.py:39-80`."""

    def __init__(self, dim_in, num_edge_type, dim_hidden, dim_out, t=4):
        super().__init__()
        if num_edge_type != 4:
            raise ValueError('the HIP kernels are built for the 4 bond types of the reference vocabulary')
        self.num_edge_type = num_edge_type
        self.t = t
        self.node_trans = nn.Linear(dim_in, dim_hidden)
        self.edge_trans = nn.Linear(num_edge_type, dim_hidden)
        self.conv = GINEConv(MLP(dim_hidden, dim_hidden, dim_hidden, nn.ReLU, 2))
        self.linear = nn.Linear(dim_hidden * self.t, dim_out)

    def embed_node(self, x, edge_index, edge_attr):
        """x [rows, dim_in]; edge_index [2, E]; edge_attr [E, 4] one-hot.  Returns (x_t, all_x)."""
        h = ops.linear(x.float(), self.node_trans.weight, self.node_trans.bias)
        etype = edge_attr.argmax(dim=-1) if edge_attr.numel() else edge_attr.new_zeros(0)
        csr = ops.Csr(edge_index, etype, h.shape[0])
        ope = 1.0 + float(self.conv.eps)
        lin1, lin2 = self.conv.nn.seq[0], self.conv.nn.seq[2]
        all_x = []
        for _ in range(self.t):
            a = ops.GineAggregateFn.apply(h, self.edge_trans.weight, self.edge_trans.bias, csr, ope)
            h = ops.linear(ops.linear(a, lin1.weight, lin1.bias, relu=True), lin2.weight, lin2.bias)
            all_x.append(h)
        return h, torch.cat(all_x, dim=-1)

    def embed_graph(self, all_x, graph_ids, node_mask=None):
        """Sum-pool per graph then `linear` (rows of one graph must be contiguous,
        as `PSVAEModel.forward` builds them)."""
        n_graph = int(graph_ids[-1]) + 1
        if node_mask is not None:
            graph_ids, all_x = graph_ids[~node_mask], all_x[~node_mask]
        counts = torch.bincount(graph_ids, minlength=n_graph)
        off = torch.zeros(n_graph + 1, dtype=torch.int32, device=all_x.device)
        off[1:] = torch.cumsum(counts, 0).to(torch.int32)
        pooled = ops.SegmentPoolFn.apply(all_x, off, graph_ids.to(torch.int32).contiguous())
        return ops.linear(pooled, self.linear.weight, self.linear.bias)

    def forward(self, batch, return_x=False):
        x, all_x = self.embed_node(batch.x, batch.edge_index, batch.edge_attr)
        graph_ids = batch.batch
        n_graph = batch.num_graphs
        counts = torch.bincount(graph_ids, minlength=n_graph)
        off = torch.zeros(n_graph + 1, dtype=torch.int32, device=all_x.device)
        off[1:] = torch.cumsum(counts, 0).to(torch.int32)
        pooled = ops.SegmentPoolFn.apply(all_x, off, graph_ids.to(torch.int32).contiguous())
        res = ops.linear(pooled, self.linear.weight, self.linear.bias)
        if return_x:
            return res, x
        return res


class Predictor(nn.Module):
    """`src/modules/predictor.py:6-21`."""

    def __init__(self, dim_feature, dim_hidden, num_property):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(dim_feature, dim_hidden), nn.ReLU(),
                                 nn.Linear(dim_hidden, dim_hidden), nn.ReLU())
        self.output = nn.Linear(dim_hidden, num_property)

    def forward(self, x):
        h = ops.linear(x, self.mlp[0].weight, self.mlp[0].bias, relu=True)
        h = ops.linear(h, self.mlp[2].weight, self.mlp[2].bias, relu=True)
        return ops.linear(h, self.output.weight, self.output.bias)


class VAEPieceDecoder(nn.Module):
    """`src/modules/vae_piece_decoder.py:16-126` (training surface).  The
    autoregressive `inference` methods (RDKit assembly) are not part of the
    training step and are not provided."""

    def __init__(self, atom_embedding_dim, piece_embedding_dim, max_pos, pos_embedding_dim, piece_hidden_dim,
                 node_hidden_dim, num_edge_type, cond_dim, latent_dim, tokenizer, t=4):
        super().__init__()
        self.tokenizer = tokenizer
        self.atom_embedding = nn.Embedding(tokenizer.num_atom_type(), atom_embedding_dim)
        self.piece_embedding = nn.Embedding(tokenizer.num_piece_type(), piece_embedding_dim)
        self.pos_embedding = nn.Embedding(max_pos, pos_embedding_dim)
        self.latent_to_rnn_hidden = nn.Linear(latent_dim, piece_hidden_dim)
        self.rnn = nn.GRU(piece_embedding_dim, piece_hidden_dim, batch_first=True)
        self.to_vocab = nn.Linear(piece_hidden_dim, tokenizer.num_piece_type())
        node_dim = atom_embedding_dim + piece_embedding_dim + pos_embedding_dim
        self.graph_embedding = Encoder(node_dim, num_edge_type, node_hidden_dim, dim_out=1, t=t)
        mlp_in = node_hidden_dim * 2 + latent_dim
        self.edge_predictor = nn.Sequential(
            MLP(dim_in=mlp_in, dim_hidden=mlp_in // 2, dim_out=mlp_in, act_func=nn.ReLU, num_layers=3),
            nn.Linear(mlp_in, num_edge_type))
        self.latent_dim = latent_dim
        self.W_mean = nn.Linear(cond_dim, latent_dim)
        self.W_log_var = nn.Linear(cond_dim, latent_dim)

    def rsample(self, conds, eps=None):
        """`:56-63`.  `eps` (optional) replaces the N(0,1) draw."""
        batch_size = conds.shape[0]
        z_mean = ops.linear(conds, self.W_mean.weight, self.W_mean.bias)
        z_log_var = -torch.abs(ops.linear(conds, self.W_log_var.weight, self.W_log_var.bias))
        kl_loss = -0.5 * torch.sum(1.0 + z_log_var - z_mean * z_mean - torch.exp(z_log_var)) / batch_size
        epsilon = torch.randn_like(z_mean) if eps is None else eps
        return z_mean + torch.exp(z_log_var / 2) * epsilon, kl_loss

    def embed_atom(self, atom_ids, piece_ids, pos_ids):
        """`:65-69`: three table lookups, concatenated."""
        return torch.cat([self.atom_embedding.weight[atom_ids], self.piece_embedding.weight[piece_ids],
                          self.pos_embedding.weight[pos_ids]], dim=-1)

    def forward(self, x, x_pieces, x_pos, edge_index, edge_attr, pieces, conds, edge_select, golden_edge,
                return_accu=False, eps=None):
        """`:71-126`: returns `(z, (rec, piece, edge, kl))`, with `return_accu` `(z, ((...), (piece_accu,
        edge_accu)))`.  Runs the fused step in decoder-only mode; inside `PSVAEModel.forward` the same
        kernels run together with the encoder in one call.  `eps` optionally replaces the N(0,1) draw."""
        from .decoder_step import decoder_forward
        return decoder_forward(self, x, x_pieces, x_pos, edge_index, edge_attr, pieces, conds, edge_select, golden_edge,
                               return_accu=return_accu, eps=eps)
