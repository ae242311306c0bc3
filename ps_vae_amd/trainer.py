"""What PyTorch Lightning drives for the reference (`src/train.py:32-58`): the epoch loop with
validation, checkpointing on the monitored value, and the KL-warm-up-aware early stopping
(`src/utils/nn_utils.py:126-136`), plus checkpoint files in Lightning's layout
(`state_dict`, `epoch`, `global_step`, `optimizer_states`, `hyper_parameters`) that hold tensors
and plain containers only, so `torch.load(..., weights_only=True)` reads them."""
import math
import os

import torch

from .data import to_device


class VAEEarlyStopping:
    """`VAEEarlyStopping` (`src/utils/nn_utils.py:126-136`) on top of Lightning's `EarlyStopping(monitor,
    patience, mode='min', min_delta=0)`: stop when the monitored value has not improved for `patience`
    validations, but keep resetting the record while `global_step < finish_anneal_step` (the loss is still
    being re-weighted by the KL schedule)."""

    def __init__(self, finish_anneal_step, patience=3, min_delta=0.0):
        self.finish_anneal_step, self.patience, self.min_delta = finish_anneal_step, patience, min_delta
        self.best_score, self.wait_count = math.inf, 0

    def on_validation_end(self, monitored, global_step):
        """Returns True when training should stop."""
        stop = False
        if monitored < self.best_score - self.min_delta:
            self.best_score, self.wait_count = monitored, 0
        else:
            self.wait_count += 1
            stop = self.wait_count >= self.patience
        if global_step < self.finish_anneal_step:
            self.best_score, self.wait_count = math.inf, 0
        return stop


def finish_anneal_step(config):
    """`src/train.py:37`: the step at which beta reaches its maximum."""
    return config['kl_warmup'] + (config['max_beta'] // config['step_beta'] - 1) * config['kl_anneal_iter']


def _plain(obj):
    """Config values as plain Python containers (what a weights-only load accepts)."""
    if isinstance(obj, dict):
        return {str(k): _plain(v) for k, v in obj.items() if _is_plain(v)}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj if _is_plain(v)]
    return obj


def _is_plain(v):
    return isinstance(v, (bool, int, float, str, type(None), dict, list, tuple))


def save_checkpoint(path, model, optimizer, epoch):
    opt = optimizer.state_dict()
    ckpt = {
        'epoch': int(epoch), 'global_step': int(model.global_step),
        'state_dict': {k: v.detach().cpu() for k, v in model.state_dict().items()},
        # Adam's moments per parameter name (as torch's own optimisers key their state per parameter): the file does
        # not depend on the order of this version's flat buffers
        'optimizer_states': [{'step': int(opt['step']), 'epoch': int(opt['epoch']),
                              'exp_avg': None if opt['exp_avg'] is None else {k: v.detach().cpu().clone() for k, v in opt['exp_avg'].items()},
                              'exp_avg_sq': None if opt['exp_avg_sq'] is None else {k: v.detach().cpu().clone() for k, v in opt['exp_avg_sq'].items()}}],
        'hyper_parameters': {'config': _plain(model.config)},
        # where the counter-based epsilon draw of rsample stands: a resumed run goes on with fresh noise instead of
        # replaying the first steps' draws
        'eps_state': {'seed': int(model.eps_seed), 'draws': int(model.eps_draws)},
    }
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(ckpt, path)


def load_checkpoint(path, model, optimizer=None):
    """Restores a checkpoint written by `save_checkpoint`, or the `state_dict` of a reference (Lightning)
    checkpoint.  Only a weights-only load is attempted: a reference `.ckpt` also pickles its tokenizer and
    config objects, which such a load refuses — export its `state_dict` with the reference code base first."""
    ckpt = torch.load(path, map_location='cpu', weights_only=True)
    model.load_state_dict(ckpt['state_dict'])
    model.global_step = int(ckpt.get('global_step', 0))
    if 'eps_state' in ckpt:
        model.eps_seed, model.eps_draws = int(ckpt['eps_state']['seed']), int(ckpt['eps_state']['draws'])
    if optimizer is not None and ckpt.get('optimizer_states'):
        optimizer.load_state_dict(ckpt['optimizer_states'][0])     # (refuses moments it cannot map by name)
    return int(ckpt.get('epoch', 0))


def validate(model, loader, device, reducer=None):
    """Mean of `validation_step` over a loader (`ps_vae_model.py:90-96`).  With a `reducer` every rank passes
    its own shard of the validation set and all ranks get the mean over all shards' batches."""
    keys = ('val_loss', 'piece accu', 'edge accu')
    tot = torch.zeros(len(keys) + 1, dtype=torch.float64)
    with torch.no_grad():
        for batch in loader:
            out = model.validation_step(to_device(batch, device))
            for i, k in enumerate(keys):
                tot[i] += float(out[k])
            tot[-1] += 1
    if reducer is not None and reducer.active:
        tot = tot.to(device)
        reducer._all_reduce(tot)
        tot = tot.cpu()
    n = max(float(tot[-1]), 1.0)
    return {k: float(tot[i]) / n for i, k in enumerate(keys)}


def fit(model, train_loader, valid_loader, max_epochs, save_dir=None, patience=3, device='cuda', log=None, reducer=None):
    """Epoch loop of `trainer.fit` as the reference configures it: Adam with lr / (epoch + 1), clip 10 (both in
    `configure_optimizers()`'s fused optimiser), validation after every epoch, best-`val_loss` checkpoint,
    KL-aware early stopping.  Returns the per-epoch validation records.

    Data-parallel (no reference counterpart; its Lightning 'dp' mode mis-splits the batch dict, `src/train.py:49-50`):
    pass `parallel.make_reducer(model)` (or a `GradReducer`) and give every rank loaders over its own shard of
    the graphs, with the same number of batches on every rank.  Each step then divides the local loss sums by the global counts and sums
    the flat gradient over ranks, which is the step of the undivided batch; validation records are means over
    all ranks, so every rank takes the same early-stopping decision, and rank 0 alone writes the checkpoint."""
    opt = model.configure_optimizers()
    stopper = VAEEarlyStopping(finish_anneal_step(model.config), patience=patience)
    history, best = [], math.inf
    dp = reducer is not None and reducer.active
    rank = torch.distributed.get_rank(reducer.group) if dp else 0
    writer = rank == 0
    if dp:
        model.shard_rank = rank      # batches without a recorded position (validation) draw epsilon rows of their own shard
    for epoch in range(max_epochs):
        opt.set_epoch(epoch)
        # a new order (and new negative pairs) every epoch, as DataLoader(shuffle=True) and the reference's
        # collate give: loaders that key their order on an epoch number are told it
        for obj in (train_loader, getattr(train_loader, 'sampler', None), getattr(train_loader, 'batch_sampler', None)):
            if hasattr(obj, 'set_epoch'):
                obj.set_epoch(epoch)
        for batch in train_loader:
            batch = to_device(batch, device)
            if dp:
                if '_native' not in batch:
                    raise ValueError('data-parallel fit needs batches from BPEMolDataset, PackedMolDataset '
                                     'or DeviceLoader (their loss denominators are exchanged between ranks)')
                nb = batch['_native']
                # global counts from every rank's OWN counts (kept apart, so a resident batch that comes round
                # again next epoch is reduced from the same numbers)
                reducer.reduce_denoms(nb.local_denoms, out=nb.denoms)
                if nb.graph_base is None:       # shard position inside the global batch, for the epsilon draw
                    nb.graph_base = rank * nb.sizes['B']
            opt.zero_grad()
            # training_step, backward (and, alone, clip + Adam) as one library call: same kernels, less host time
            model.train_step(batch, None if dp else opt)
            if dp:
                if hasattr(reducer, 'reduce'):
                    reducer.reduce()
                else:
                    reducer.reduce_grads(model.flat_buffers()[1])
                opt.step()
            model.global_step += 1
        # (checked here, where the host synchronises anyway; no kernel of the current library reports invalid results:
        #  a multi-CU recurrence that times out is redone on the single-CU kernels inside the step)
        if model.recurrence_gave_up():
            raise RuntimeError(f'epoch {epoch}: a training step reported invalid results; parameters must not be kept')
        redone = model.recurrence_redone()
        if redone and log:
            log({'epoch': epoch, 'gru_recurrences_redone_on_one_cu': redone,
                 'hint': 'the GPU could not keep the multi-CU recurrence resident (shared with another process?); '
                         "set option gru_multi_cu = 0 (PSVAE_GRU_MULTI_CU=0) to avoid the seconds-long waits"})
        rec = validate(model, valid_loader, device, reducer)
        rec.update(epoch=epoch, global_step=model.global_step)
        history.append(rec)
        if log:
            log(rec)
        if save_dir is not None and rec['val_loss'] < best:
            best = rec['val_loss']
            if writer:
                save_checkpoint(os.path.join(save_dir, 'best.ckpt'), model, opt, epoch)
        if stopper.on_validation_end(rec['val_loss'], model.global_step):
            break
    return history
